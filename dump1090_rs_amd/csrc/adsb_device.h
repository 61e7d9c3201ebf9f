// adsb_device.h -- data that crosses the kernel/host line inside libadsb_hip.so.
//
// Device pipeline for one call over C chunks (chunk = one MagnitudeBuffer's worth of
// samples, 131072, reference src/lib.rs:22):
//
//   scan    IQ (or u16 magnitudes) -> per-tile magnitudes in LDS -> preamble/SNR/quiet
//           gates -> 5 trial phases sliced -> DF + CRC-24 residual per trial ->
//             * self-validating trials (clean DF11 / DF17 / DF18)  -> hit list,
//               and their address is OR-ed into the 2^24-bit address bitmap
//             * address/parity trials (DF 0,4,5,16,20,21,24..31)   -> AP list
//   match   AP list x bitmap -> hit list      (bitmap is now complete for the call)
//   records hit list -> {chunk, j, try_phase, 14 message bytes, 33-sample power}
//
// The host then replays the records in (chunk, j, try_phase) order through the
// reference's score/filter logic.  Trials the device drops can neither add to the
// filter nor score >= 0, so the replay is exact ("Why the split is exact").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adsb {

constexpr int kChunkSamples = 131072;  // MODES_MAG_BUF_SAMPLES, src/lib.rs:22
constexpr int kLead = 326;             // TRAILING_SAMPLES, src/lib.rs:24
constexpr int kMagDataLen = kLead + kChunkSamples;
constexpr int kReach = 290;            // furthest sample a preamble at j touches: j+290

// 64-bit list entry: crc24 | tp_idx<<24 | j<<27 | chunk<<44   (tp_idx = try_phase-4)
__host__ __device__ inline uint64_t pack_entry(uint32_t crc, uint32_t tp_idx, uint32_t j,
                                               uint64_t chunk)
{
    return (uint64_t)(crc & 0xFFFFFFu) | ((uint64_t)tp_idx << 24) | ((uint64_t)j << 27) |
           (chunk << 44);
}
__host__ __device__ inline uint32_t entry_crc(uint64_t e) { return (uint32_t)e & 0xFFFFFFu; }
__host__ __device__ inline uint32_t entry_tp(uint64_t e) { return (uint32_t)(e >> 24) & 7u; }
__host__ __device__ inline uint32_t entry_j(uint64_t e) { return (uint32_t)(e >> 27) & 0x1FFFFu; }
__host__ __device__ inline uint64_t entry_chunk(uint64_t e) { return e >> 44; }
constexpr uint64_t kMaxChunks = 1ull << 20;

// One trial message handed to the host replay (32 bytes).
struct TrialRecord {
    uint64_t power;     // sum of the 33 squared magnitudes from j+19 (demod_2400.rs:191-196)
    uint32_t chunk;
    uint32_t j_tp;      // j | try_phase << 24
    uint8_t msg[14];
    uint16_t pad;
};
static_assert(sizeof(TrialRecord) == 32, "TrialRecord layout");

// Device counters block (one per context).
struct Counters {
    uint32_t n_hits;       // entries in the hit list
    uint32_t n_ap;         // entries in the AP list
    uint32_t n_cand;       // candidates (diagnostic)
    uint32_t overflow;     // bit0: hit list full, bit1: AP list full
    uint32_t pad[4];
};

struct ScanParams {
    const void *src;        // IQ as {re,im} int16 pairs, or u16 magnitudes (from_mag)
    uint64_t n_samples;     // IQ: total samples in the call.  from_mag: `length` of the buffer
    uint32_t n_chunks;
    uint32_t *bitmap;       // 2^24 bits
    uint64_t *hits;
    uint32_t hits_cap;
    uint64_t *ap;
    uint32_t ap_cap;
    Counters *ctr;
};

// launches (adsb_kernels.hip); all asynchronous on `stream`
int launch_to_mag(const void *d_iq, uint32_t n, uint16_t *d_data, void *stream);
int launch_scan(const ScanParams &p, bool from_mag, void *stream);
int launch_mag_digest(uint32_t first_bits, uint32_t count, unsigned long long *d_out, void *stream);
int launch_match(const ScanParams &p, void *stream);
int launch_records(const ScanParams &p, bool from_mag, TrialRecord *d_rec, void *stream);

}  // namespace adsb
