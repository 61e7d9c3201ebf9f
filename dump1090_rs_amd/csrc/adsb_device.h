// adsb_device.h -- data that crosses the kernel/host line inside libadsb_hip.so.
//
// Device pipeline for one call over C chunks (chunk = one MagnitudeBuffer's worth of
// samples, 131072, reference src/lib.rs:22):
//
//   scan    IQ -> per-tile magnitudes in LDS -> sign planes -> preamble pattern
//           (bit-parallel) -> SNR/quiet gates -> 5 trial phases per candidate: DF + CRC-24
//           syndrome straight from the sign planes ->
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

#include <cstdlib>

#include "adsb_record.h"

// Measurement switches (tools/*.sh: ADSB_DEBUG_STOP, ADSB_STAGGER, ADSB_SCAN_BLOCKS_PER_CU,
// ADSB_STREAM_PRIO, ADSB_NO_EXT_EVENTS, ADSB_ONE_SCAN_STREAM, ADSB_DONE_FENCE, ADSB_TIMELINE) exist
// only in a library built with -DADSB_TUNING (-DADSB_KERNEL_ACCT implies it): the release build
// reads nothing from the environment, so no stray variable can change what it computes.
#if defined(ADSB_KERNEL_ACCT) && !defined(ADSB_TUNING)
#define ADSB_TUNING 1
#endif
#ifdef ADSB_TUNING
#define ADSB_STOP_AT(p, n) ((p).debug_stop == (n))
#else
#define ADSB_STOP_AT(p, n) false
#endif

namespace adsb {

inline const char *tuning_env(const char *name)
{
#ifdef ADSB_TUNING
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

constexpr int kChunkSamples = 131072;  // MODES_MAG_BUF_SAMPLES, src/lib.rs:22
constexpr int kLead = 326;             // TRAILING_SAMPLES, src/lib.rs:24
constexpr int kMagDataLen = kLead + kChunkSamples;
constexpr int kReach = 290;            // furthest sample a preamble at j touches: j+290

// 64-bit list entry: value24 | code<<24 | j<<28 | chunk<<45
//   code 0..4   short message (56 bits), try_phase = 4 + code, value = H' (see below)
//   code 5..9   long message (112 bits), try_phase = 4 + code - 5, value = H'
//   code 10..14 try_phase = 4 + code - 10, value = the CRC residual itself
// H' = x^51 * H is the CRC syndrome of the fast scan: the residual itself for a 56-bit
// message, and the residual is x^56 * H' in GF(2)[x]/(0x1FFF409) for a 112-bit one -- the
// match kernel applies that multiplier (adsb_tables.h explains the factorisation).
__host__ __device__ inline uint64_t pack_entry(uint32_t value, uint32_t code, uint32_t j,
                                               uint64_t chunk)
{
    return (uint64_t)(value & 0xFFFFFFu) | ((uint64_t)code << 24) | ((uint64_t)j << 28) |
           (chunk << 45);
}
__host__ __device__ inline uint32_t entry_value(uint64_t e) { return (uint32_t)e & 0xFFFFFFu; }
__host__ __device__ inline uint32_t entry_code(uint64_t e) { return (uint32_t)(e >> 24) & 15u; }
__host__ __device__ inline uint32_t entry_tp(uint64_t e) { return 4u + entry_code(e) % 5u; }
__host__ __device__ inline uint32_t entry_j(uint64_t e) { return (uint32_t)(e >> 28) & 0x1FFFFu; }
__host__ __device__ inline uint64_t entry_chunk(uint64_t e) { return e >> 45; }
constexpr uint64_t kMaxChunks = 1ull << 19;  // per device pass (256 GiB of IQ)


// The AP list of the fast scan is split into kApWaveSegs equal segments: every wave of
// every persistent workgroup owns one outright, so appending needs no atomic and no shared
// counter at all (a returning global atomic costs a workgroup 1-2 us per tile, one hot
// counter saturates near 90 atomics/us on this chip, and even an LDS counter is a round
// trip per trial pass).  The simple kernel appends to a second list, `dap`, through one
// shared counter: it is the slow path anyway.
constexpr int kHitFieldWords = 8;  // ScanParams::hit_fields: f0..f4, flag, 2 spare
constexpr int kNewAddrCap = 16;
constexpr int kFusedMaxTiles = 16 * 18;  // the largest one-launch pass: 16 buffers of 17 tiles (18 with -DADSB_TILE=7284)
constexpr int kApSegments = 1280;
constexpr int kApWaveSegs = 4 * kApSegments;  // one per wave of a persistent workgroup

// Device counters block (one per context).
struct Counters {
    uint32_t n_hits;       // entries in the hit list
    uint32_t overflow;     // bit0: hit list, bit1: an AP segment, bit3: dap list
    uint32_t scan_blocks_done;  // one-launch pass: scan workgroups that have finished (the last one runs the tail)
    uint32_t n_dap;        // entries in the dap list
    uint32_t n_cand_simple;  // candidates seen by the simple kernel (diagnostic)
    uint32_t blocks_done;    // records kernel: blocks that have finished (last one publishes)
    uint32_t rec_sum[2];     // records kernel: 64-bit sum of every u64 word of the records it wrote (8-byte aligned)
    uint32_t learned_new;    // one-launch pass: how often a trial of this pass set an address bit that was clear before
    uint32_t n_rec;          // one-launch pass: records its workgroups have written in place (k_scan_fast: emit_records)
    uint32_t new_addr[kNewAddrCap];  // ... and the first of those addresses
    uint32_t unordered;      // one-launch pass: a workgroup gave up waiting for the tiles before its own (bounded wait)
    uint32_t t_start[2];     // one-launch pass: the 100 MHz wall clock when its first workgroup started
    uint32_t bitmap_ready;   // one-launch pass behind an icao_flush: its first workgroup has cleared the (folded) bitmap
    uint32_t n_fresh;        // shard phase 1 (ScanParams::fresh): distinct addresses this scan's trials can add ...
    uint32_t fresh_sum;      // ... and the sum of those addresses (the host checks the list it reads against it)
    uint32_t tile_done[kFusedMaxTiles];  // one-launch pass: tile t's address bits and list entries are published
    uint32_t seg_ap[kApWaveSegs];    // entries in each wave's AP segment
    uint32_t seg_cand[kApSegments];  // candidates seen by each fast workgroup (diagnostic)
};

// What the host needs after a pass, gathered by the records kernel (the last one to run)
// so that one small copy brings it back.
struct Summary {
    uint32_t n_hits, overflow;
    uint32_t rec_sum_lo;    // 64-bit sum of every u64 word of the n_hits records (low half): the records and
                            // this summary reach host memory as separate posted writes -- a record that has
                            // not landed (or is torn) when the host reads it makes the sums disagree
    uint32_t n_dap;
    uint32_t n_ap_total;    // all AP entries (fast segments + dap)
    uint32_t n_cand_total;  // all candidates
    uint32_t rec_sum_hi;
    uint32_t seq;           // the pass's sequence number (never 0): lets the host check it reads its own pass
    uint32_t ticks;         // one-launch pass: its duration on the device's 100 MHz wall clock, first workgroup's
                            // entry to the summary (written before seq): the launch needs no timing events
    uint32_t check;         // summary_check() of the nine words above: the ten words are separate posted writes,
                            // and a host that polls `seq` must not pair it with a neighbour that has not landed
};

// The summary's own checksum: each word rotated by its index (a swap of two words changes it), folded.
__host__ __device__ inline uint32_t summary_check(const uint32_t w[9])
{
    uint32_t x = 0xA5D5B17Cu;
    for (int k = 0; k < 9; k++) x ^= (w[k] << (k + 1)) | (w[k] >> (31 - k));
    return x;
}

// GF(2) tables, 256 u32 each (adsb_tables.h): F'0 F'1 F'2 | X56_0..2
constexpr int kTabF = 0, kTabX56 = 3, kTabCount = 6;
// after them in the same buffer: R16 (16 u32, adsb_tables.h) and the field table (300 u32)
constexpr int kTabR16Off = kTabCount * 256, kTabFieldOff = kTabR16Off + 16, kTabBitsOff = kTabFieldOff + 300,
              kTabWords = kTabBitsOff + 168;  // + per-bit residual constants (build_bit_residuals)

// Device-side scoring of a pass whose hits are in (buffer, j, try_phase) order: the sequential part
// of demodulate2400 (src/mode_s/mod.rs:34-139 scores read AND write the ICAO filter,
// src/demod_2400.rs:184-207 keeps the best of the five phases) done in parallel.  What makes that
// possible: a trial's score depends on the filter only through "is value v in the filter when this
// trial is scored", and v is in it then iff it was in it when the pass began (the exact bitmap) or an
// EARLIER trial of the pass added it -- the first index at which a clean DF11 (IID 0) / DF17 with
// address v occurs, found with one atomic-min hash table over the adders.  (DF18 adds addr | 1 << 25,
// which no 24-bit test value equals.)  The only thing this cannot reproduce is the reference's
// behaviour once the 4096-slot table fills up; the host checks the count and replays such a pass
// itself from the records, which are always there.
struct ScoreState {
    uint32_t n;            // hits of this pass (0 when it is not to be scored: overflow, too many)
    uint32_t blocks_done;  // k_emit
    uint32_t scored;       // 1: k_score / k_emit handle this pass (a pass without a single hit included)
    uint32_t reserved;
    unsigned long long msg_sum;  // 64-bit sum of every u64 word of the messages written
};
struct ScoreSummary {      // in mapped host memory
    uint32_t n_msgs, n_adds;
    uint32_t msg_sum_lo, msg_sum_hi;
    uint32_t scored;       // 1: messages / adds are the pass's result; 0: the host replays the records
    uint32_t pad[2];
    uint32_t seq;
};
struct ScoreDev {
    uint32_t cap;                  // hits a pass may have to be scored here
    uint32_t *si;                  // per hit: value24 | kind << 24
    TrialRecord *rec;              // per hit: the record (device copy)
    unsigned long long *pos;       // per hit: buffer << 24 | j (what groups the five trial phases of a position)
    uint32_t *flag;                // per hit: bit 0 emit, bit 1 add; bits 8.. : score + 3
    unsigned long long *hash;      // adders: (value << 32 | first index), ~0 = empty
    uint32_t hash_mask;
    uint32_t *slot;                // per hit: the hash slot its key sits in (adders), else 0xFFFFFFFF
    uint32_t *blk;                 // per k_score block: emits, adds
    uint32_t *exact;               // 2^24 bits: the filter as it stands before this pass
    const uint32_t *earlier;       // a shard of an adsb_multi: 2^24 bits, the addresses the shards BEFORE this one (lower
                                   // buffer ranges of the same capture) add to the filter -- in it for every trial of this
                                   // shard, whichever adds them first over there; null everywhere else
    uint32_t *exact_retired;       // after an icao_flush: the bitmap the passes before it used, for k_emit to
                                   // clear (the next flush switches back to it), else null
    ScoreState *state;
    void *out_msgs;                // adsb_msg[cap], mapped host memory
    uint32_t *out_adds;            // mapped host memory: values handed to icao_filter_add, in order
    ScoreSummary *summary;         // mapped host memory
    uint32_t seq;
};
constexpr int kScoreBlocks = 256;  // k_score / k_emit grid: block b owns a contiguous run of hits
enum ScoreKind : uint32_t { kSkOther = 0, kSkApShort, kSkApLong, kSkDf11Iid0, kSkDf11, kSkDf17, kSkDf18, kSkNone };

struct ScanParams {
    const void *src;        // IQ as {re,im} int16 pairs, or u16 magnitudes (from_mag)
    uint64_t n_samples;     // IQ: total samples in the call.  from_mag: `length` of the buffer
    uint32_t n_chunks;
    uint32_t *bitmap;       // 2^bitmap_lg bits + the 4096-bit summary
    uint32_t bitmap_lg;     // 24: bit a is address a; kSmallBitmapLg: folded (contexts for passes of a few buffers)
    uint32_t bitmap_fresh;  // an icao_flush precedes this pass and `bitmap` is the next one in the rotation, NOT yet
                            // clean: a one-launch pass clears it itself -- its first workgroup does, the others wait for
                            // Counters::bitmap_ready before they touch it (folded bitmaps only: 64 KB); every pass that
                            // used it has been collected (one bitmap more than passes in flight), so nobody is waited for
    uint64_t *hits;
    uint32_t hits_cap;
    uint64_t *ap;           // wave segment g at ap + g * seg_cap (only the segments a context's largest pass can use are allocated)
    uint32_t ap_cap;        // entries allocated
    uint32_t seg_cap;       // entries per wave segment
    uint64_t *dap;          // AP entries of the simple kernel
    uint32_t dap_cap;
    const uint32_t *tables; // kTabCount x 256
    Counters *ctr;
    uint32_t *clean_bitmap; // a retired bitmap for the records kernel to clear (after an icao_flush), or null
    Summary *summary;       // in mapped host memory
    uint32_t seq;           // this pass's sequence number
    void *ev_start, *ev_stop;  // HIP events stamped by the scan launch itself (or null)
    uint32_t stagger_ticks; // fast scan: start offset between the workgroups of a CU (clock64 ticks)
    int debug_stop;         // profiling only (ADSB_DEBUG_STOP): leave the fast scan after phase N
    unsigned long long *timeline;  // profiling only (ADSB_TIMELINE): per-phase clock stamps, or null
    uint32_t keep_counters; // records kernel: leave the counters and lists as they are (first phase of a shard)
    // first phase of a shard of an adsb_multi: every address a trial of this scan can add to the filter is appended
    // here (mapped host memory), once -- fresh_seen, 2^24 bits of the shard's own, all clear when the scan starts, says
    // which have been.  That list is all the exchange needs: no records kernel and no per-record work on the host in
    // the first phase.  (Not "whose bit in `bitmap` was clear": the scans of consecutive captures overlap on two
    // streams, and the later one may set an address's bit first -- the earlier capture's list would then lack an address
    // its own second phase on the OTHER devices needs; found by the soak, tests/fuzz_gpu.py --multi.)  Null elsewhere.
    uint32_t *fresh;
    uint32_t fresh_cap;
    uint32_t *fresh_seen;
    // carry-over mode (opt-in, not the reference's semantics): the 326-sample lead-in of a
    // buffer holds the samples that preceded it.  Buffers after the first take them from src
    // itself; the first takes them from `carry` (kCarrySamples IQ samples, the end of the
    // previous call), or from src[-kCarrySamples..] when lead_from_src is set.
    const uint32_t *carry;
    uint32_t lead_from_src;
    // device-side ordering of the hit list (k_order_prefix / k_records; null: the host sorts).
    // Whoever finds a hit puts it straight into its buffer's bucket, in the sub-bucket of its tile --
    // order_tmp[chunk * kOrderBucket + tile * kTileBucket + place] -- and the hit list proper is written by k_records,
    // each buffer's bucket sorted, at the buckets' exclusive prefix (order_base, one per buffer).  order_cnt: one count
    // per TILE (17 n_chunks), all zero between passes; order_base: n_chunks + 1; order_tmp: n_chunks * kOrderBucket.
    uint32_t *order_cnt, *order_base;
    uint64_t *order_tmp;
    // self-test only (adsb_selftest_stage_lists): every position that passes the gates is also
    // appended here as chunk << 32 | j; null in every production pass
    uint64_t *cand_out;
    uint32_t *cand_count;
    uint32_t cand_cap;
    // device-side scoring (k_score / k_emit; score.si null: the host replays the records)
    ScoreDev score;
    // What the scan already knows about a self-validating hit, handed to the record builder: per slot of the
    // hit list (p.hits index, or buffer * kOrderBucket + place in the bucket) eight words -- the five bit-class
    // fields of the trial (message bit 5k + r = bit k of field r: all 112 sliced bits) and a flag that says
    // they are there.  Hits the match finds (address/parity trials) clear the flag: their message is sliced
    // from the samples as before.  Null: nobody writes or reads it.
    uint32_t *hit_fields;
    // one-launch pass (k_scan_fast<.., FUSED>: scan + match + records in one kernel, for passes of a few
    // buffers): where its records go (mapped host memory), else null
    TrialRecord *fused_rec;
    // ... a host-pointer call copies its samples into pinned memory WHILE the launch is on its way: how many
    // samples of src are there so far (a word in mapped host memory the copying thread advances; null: all)
    const unsigned long long *src_ready;
    uint32_t src_host;      // one-launch pass: `src` is host memory read in place over the link (its tiles trickle in)
    uint32_t order_polls;   // ... how often a workgroup polls for the tiles before its own before it gives up
                            // (200, ~0.2 ms; the self-test hook sets 0: every tile gives up, the second look decides)
};



// The address bitmap: 2^24 bits, followed by a 4096-bit summary (bit a & 4095 is set when any
// address a is): the match kernel tests the summary from LDS and goes to the big bitmap only
// for the few per cent of residuals that pass it.
constexpr uint32_t kBitmapWords = 1u << 19, kCoarseWords = 128;
constexpr uint32_t kBitmapAllocWords = kBitmapWords + kCoarseWords;
// A context for passes of a few buffers keeps its supersets FOLDED: 2^19 bits (64 KB) addressed by a ^ (a >> 19)
// instead of 2^24 (2 MiB) addressed by a.  The bitmap only has to be a superset of the filter (DESIGN.md section 3:
// a false positive costs one more trial record, which the ordered replay scores against the real filter), a receiver
// hears a few hundred aircraft, and such a context used to carry eleven 2 MiB bitmaps -- 23 of its 33 MB -- and clear
// one behind every icao_flush.  bitmap_lg = log2 of the bits: 24 (exact) or kSmallBitmapLg.
constexpr uint32_t kFullBitmapLg = 24, kSmallBitmapLg = 19;
__host__ __device__ inline uint32_t bitmap_words(uint32_t lg) { return 1u << (lg - 5u); }
__host__ __device__ inline uint32_t bitmap_alloc_words(uint32_t lg) { return bitmap_words(lg) + kCoarseWords; }
__host__ __device__ inline uint32_t bitmap_index(uint32_t a, uint32_t lg)
{
    return lg >= kFullBitmapLg ? a : ((a ^ (a >> lg)) & ((1u << lg) - 1u));
}

// hits one buffer's bucket holds on a dense stream (device-side ordering): ~20x a busy airspace;
// a fuller one is an overflow like any other list's (the pass is redone buffer by buffer)
constexpr uint32_t kOrderBucket = 1024;
// ... cut into one sub-bucket per tile of the buffer (17 of them): a tile has ONE writer in the scan, so its staged
// hits go to places 0 .. n-1 of its sub-bucket and its count is a plain store -- no returning atomic, no wait, no
// barrier at the end of a tile (round 5: the two dependent atomics there, behind an s_waitcnt that also covered the
// next tile's prefetch, were the dense stream's 13 us; profiles/r5_dense_acct.txt).  order_cnt has one count per tile.
constexpr uint32_t kTileBucket = 56;   // (x 18 tiles <= kOrderBucket, should the tile ever shrink: adsb_scan_geometry.h)
constexpr int kCarrySamples = 328;  // kLead rounded up to whole 16-byte loads

// launches; all asynchronous on `stream`, return a hipError_t as int
int launch_to_mag(const void *d_iq, uint32_t n, uint16_t *d_data, void *stream);
// zero a counters block and, when `bitmap` is non-null, clear an address bitmap (bit 0 stays
// set); only needed once per context: afterwards every pass cleans up for the next one
int launch_reset(Counters *ctr, uint32_t *bitmap, uint32_t bitmap_lg, void *stream);
int launch_scan(const ScanParams &p, bool from_mag, void *stream);   // fast (IQ) or simple (mag)
// the whole pass in one launch (p.fused_rec set): one workgroup per tile, the last one to finish matches
// what the pass learned late, builds the records and publishes the summary; for passes of a few buffers
int launch_pass_fused(const ScanParams &p, bool from_mag, void *stream);
int scan_resident_blocks();  // workgroups of the fast scan's persistent grid (<= kApSegments)
int launch_scan_simple(const ScanParams &p, bool from_mag, void *stream);  // reference-shaped path
int launch_match(const ScanParams &p, void *stream);
int launch_records(const ScanParams &p, bool from_mag, TrialRecord *d_rec, void *stream);
// sort the hit list by (buffer, j, try_phase) on the device, so that the records come out in the
// order the host replays them in (src/demod_2400.rs:121,158: ascending j, then try_phase)
int launch_order_hits(const ScanParams &p, void *stream);
// first phase of a shard that lists its fresh addresses (ScanParams::fresh): the summary alone -- n_hits, overflow,
// Summary::rec_sum_lo = sum of the fresh addresses, Summary::n_dap = how many -- behind the scan on its stream
int launch_shard_summary(const ScanParams &p, void *stream);
// score the (ordered) hits of the pass on the device: messages, filter additions and a summary
// into mapped host memory (p.score)
int launch_score(const ScanParams &p, void *stream);
// OR a list of 24-bit addresses into a bitmap (addresses learned by other shards)
int launch_set_addresses(const uint32_t *d_addrs, uint32_t n, uint32_t *bitmap, uint32_t bitmap_lg, void *stream);
// next[i] = sample (n - kCarrySamples + i) of the stream: from d_src, or from `prev` where the
// call was shorter than the carry
int launch_update_carry(const uint32_t *prev, const void *d_src, uint64_t n_samples, uint32_t *next, void *stream);
int launch_mag_digest(uint32_t first_bits, uint32_t count, unsigned long long *d_out, void *stream);

}  // namespace adsb
