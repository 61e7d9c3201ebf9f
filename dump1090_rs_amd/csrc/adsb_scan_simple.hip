// adsb_scan_simple.hip -- the reference-shaped scan kernel.
//
// One workgroup = up to 4096 preamble positions j of one chunk: magnitudes into LDS,
// then every j through the gates (src/demod_2400.rs:121-146) exactly as the reference
// walks them, then every trial phase sliced bit by bit (:158-182) and scored as far as
// the device goes (DF class + CRC residual, src/mode_s/mod.rs:41-135).  It is slow
// (every thread walks 112 bits serially) and serves as the chunk-by-chunk fallback for
// input so dense that the fast scan's lists overflow: its lists hold the worst case of a
// chunk (every position sliced).  Both input kinds: IQ, or a caller-supplied MagnitudeBuffer.
#include "adsb_dev_common.h"

namespace adsb {

namespace {

constexpr int kTile = 4096;
constexpr int kPad = 2;  // 326 + 2 = 4 * 82: slot 0 sits on a 16-byte IQ boundary
constexpr int kSlots = kTile + kPad + kReach;  // 4388
constexpr int kTilesPerChunk = kChunkSamples / kTile;
static_assert(kSlots % 4 == 0, "slots must be whole 4-sample groups");
static_assert((kLead + kPad) % 4 == 0, "tile origin must be 16-byte aligned in IQ space");

// fill smag[0..kSlots) with data[jbase - kPad ...] of `chunk` (len = samples in the chunk)
template <bool FROM_MAG>
__device__ __forceinline__ void load_tile(const ScanParams &p, uint64_t chunk, int jbase, int len,
                                          uint16_t *smag)
{
    const void *src = p.src;
    const int d0 = jbase - kPad;
    if (FROM_MAG) {
        // src = MagnitudeBuffer.data (kMagDataLen u16), used as handed in
        const uint16_t *data = (const uint16_t *)src;
        for (int i = threadIdx.x; i < kSlots; i += blockDim.x) {
            const int d = d0 + i;
            smag[i] = (d >= 0 && d < kMagDataLen) ? data[d] : (uint16_t)0;
        }
    } else {
        const uint32_t *iq = (const uint32_t *)src + chunk * (uint64_t)kChunkSamples;
        const int k0 = d0 - kLead;  // IQ sample index of slot 0 (multiple of 4, may be < 0)
        if (p.carry == nullptr) {   // the reference's semantics: nothing before the buffer
            for (int g = threadIdx.x; g < kSlots / 4; g += blockDim.x)
                *(uint2 *)(smag + 4 * g) = mag4(iq, k0 + 4 * g, len);
        } else {                    // carry-over mode (adsb_device.h): the lead-in holds what preceded
            const bool lead = chunk > 0 || p.lead_from_src;
            for (int i = threadIdx.x; i < kSlots; i += blockDim.x) {
                const int k = k0 + i;
                uint32_t w = 0;
                if (k >= 0)
                    w = k < len ? iq[k] : 0u;
                else if (k >= -kLead)
                    w = lead ? *(iq + k) : p.carry[k + kCarrySamples];
                smag[i] = (uint16_t)mag_of_dword(w);
            }
        }
    }
}

template <bool FROM_MAG>
__device__ __forceinline__ void scan_simple_tile(const ScanParams &p, uint32_t chunk, int jbase,
                                                 int jn, int len, uint16_t *smag, uint32_t *scrc,
                                                 uint16_t *scand, uint32_t *sncand)
{
    scrc[threadIdx.x] = crc_table_entry(threadIdx.x);
    if (threadIdx.x == 0) *sncand = 0;
    load_tile<FROM_MAG>(p, chunk, jbase, len, smag);
    __syncthreads();

    // --- preamble / SNR / quiet gates for every j of the tile (demod_2400.rs:121-146)
    for (int jj = threadIdx.x; jj < jn; jj += blockDim.x) {
        if (preamble_gates(smag + kPad + jj)) {
            const uint32_t slot = atomicAdd(sncand, 1u);
            scand[slot] = (uint16_t)jj;
        }
    }
    __syncthreads();
    const int ncand = (int)*sncand;
    if (threadIdx.x == 0 && ncand) atomicAdd(&p.ctr->n_cand_simple, (uint32_t)ncand);

    // --- five trial phases per candidate (demod_2400.rs:158-184): slice, DF, CRC
    const int ntrial = ncand * 5;
    for (int base = 0; base < ntrial; base += blockDim.x) {
        const int t = base + threadIdx.x;
        bool is_hit = false, is_ap = false;
        uint64_t entry = 0;
        if (t < ntrial) {
            const int jj = scand[t / 5];
            const uint32_t code = 10u + (uint32_t)(t % 5);  // value = residual itself
            uint32_t w[4];
            slice_message(smag + kPad + jj, 4 + t % 5, w);
            const uint32_t df = w[0] >> 27;  // mode_s/mod.rs:41
            const bool nonzero = (w[0] | w[1] | w[2] | w[3]) != 0;  // :51
            if (nonzero) {
                const uint32_t j = (uint32_t)(jbase + jj);
                const uint32_t addr = w[0] & 0xFFFFFFu;  // message bits 9..32
                if (df == 11) {  // :73-90
                    const uint32_t c = modes_checksum(w, 7, scrc);
                    if ((c & 0xFFFF80u) == 0) {
                        is_hit = true;
                        entry = pack_entry(c, code, j, chunk);
                        if ((c & 0x7Fu) == 0) bitmap_set(p.bitmap, p.bitmap_lg, addr);  // the replay adds it
                    }
                } else if (df == 17 || df == 18) {  // :91-109
                    const uint32_t c = modes_checksum(w, 14, scrc);
                    if (c == 0) {
                        is_hit = true;
                        entry = pack_entry(c, code, j, chunk);
                        // DF18 adds addr | 1<<25, which no 24-bit test can match
                        if (df == 17) bitmap_set(p.bitmap, p.bitmap_lg, addr);
                    }
                } else if (df == 0 || df == 4 || df == 5) {  // :56-72
                    is_ap = true;
                    entry = pack_entry(modes_checksum(w, 7, scrc), code, j, chunk);
                } else if (df == 16 || df == 20 || df == 21 || df >= 24) {  // :110-135
                    is_ap = true;
                    entry = pack_entry(modes_checksum(w, 14, scrc), code, j, chunk);
                }
            }
        }
        wave_append(is_hit, entry, p.hits, p.hits_cap, &p.ctr->n_hits, &p.ctr->overflow, 1u);
        wave_append(is_ap, entry, p.dap, p.dap_cap, &p.ctr->n_dap, &p.ctr->overflow, 8u);
    }
}

// regular grid: block -> (chunk, 4096-tile)
template <bool FROM_MAG>
__global__ __launch_bounds__(256) void k_scan_simple(ScanParams p)
{
    __shared__ __attribute__((aligned(16))) uint16_t smag[kSlots];
    __shared__ uint32_t scrc[256];
    __shared__ uint16_t scand[kTile];
    __shared__ uint32_t sncand;

    const uint32_t chunk = blockIdx.x / kTilesPerChunk;
    const int jbase = (int)(blockIdx.x % kTilesPerChunk) * kTile;
    const int len = FROM_MAG ? (int)p.n_samples : chunk_len(p.n_samples, chunk);
    if (jbase >= len) return;
    scan_simple_tile<FROM_MAG>(p, chunk, jbase, min(kTile, len - jbase), len, smag, scrc, scand,
                               &sncand);
}

inline int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
// hipGetLastError is sticky across unrelated calls (the caller's too): start every launch clean
inline void hip_clear() { (void)hipGetLastError(); }

}  // namespace

int launch_scan_simple(const ScanParams &p, bool from_mag, void *stream)
{
    hip_clear();
    const uint32_t blocks = p.n_chunks * kTilesPerChunk;
    if (blocks == 0) return 0;
    if (from_mag)
        hipLaunchKernelGGL(k_scan_simple<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_scan_simple<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

}  // namespace adsb
