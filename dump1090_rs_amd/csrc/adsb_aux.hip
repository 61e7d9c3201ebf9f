// adsb_aux.hip -- the small kernels around the scan: to_mag alone, the address/parity
// match, the record builder and the magnitude self-test digest.
#include <algorithm>

#include "../../include/adsb_hip.h"
#include "adsb_dev_common.h"
#include "adsb_scan_geometry.h"
#include "adsb_tail_dev.h"

namespace adsb {

namespace {

// ---------------------------------------------------------------------------
// to_mag alone (adsb_to_mag).  data[0..326) = 0, data[326+k] = mag(iq[k]), rest 0
// (src/lib.rs:36-50, src/utils.rs:43-58).
// ---------------------------------------------------------------------------

// Eight outputs per thread: one 16-byte store (the output may be pinned host memory, written over the link:
// 2-byte stores would be 128 bytes per wave-instruction there); the eight samples behind them through a
// buffer resource over the n input samples (a dword each, 4-byte aligned: the 326-sample lead-in shifts the
// output by 6 of 8; out of range -- lead-in, tail -- reads as zero IQ, whose magnitude is zero).
__global__ __launch_bounds__(256) void k_to_mag(const uint32_t *__restrict__ iq, uint32_t n,
                                                uint16_t *__restrict__ data)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t i0 = 8u * t;                       // first output of this thread
    if (i0 >= (uint32_t)kMagDataLen) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)iq, 0, (int)(n * 4u), 0x00020000);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int off = ((int)i0 + i - kLead) * 4;          // negative in the lead-in: out of range, zero
        asm volatile("" : "+v"(off));                 // (not folded into the immediate: adsb_tail_dev.h, records)
        w[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
    }
    uint32_t pk[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pk[i] = mag2(w[2 * i], w[2 * i + 1]);
    if (i0 + 8u <= (uint32_t)kMagDataLen) {
        *(uint4 *)(data + i0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    } else {  // the last, partial group (131398 = 8 * 16424 + 6)
        for (uint32_t i = 0; i0 + i < (uint32_t)kMagDataLen; i++) data[i0 + i] = (uint16_t)(pk[i >> 1] >> (16 * (i & 1)));
    }
}

// ---------------------------------------------------------------------------
// reset: one launch per pass instead of a string of memsets.  Zeroes the counters and,
// after an icao_flush, the 2 MiB address bitmap; address 0 always tests true
// (src/icao_filter.rs:71-80: an empty slot equals 0), so bit 0 starts set.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reset(Counters *ctr, uint32_t *bitmap, uint32_t lg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t kCtrDwords = sizeof(Counters) / 4;
    if (i < kCtrDwords) ((uint32_t *)ctr)[i] = 0;
    if (bitmap) bitmap_clear(bitmap, lg, i, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------
// match.  An address/parity trial can only score >= 0 if its CRC residual is in
// the filter when it is scored (mode_s/mod.rs:71,115,130); the bitmap now holds
// every address the filter can contain at any point of this call (plus 0, which
// icao_filter_test always accepts, icao_filter.rs:71-80).  Entries from the fast
// scan carry H' = x^51*H: the residual itself for 56-bit trials, x^56*H' for 112-bit
// ones (adsb_tables.h).
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_match(ScanParams p)
{
    if (p.order_cnt) TAIL_PRIO();  // dense streams only: elsewhere the scan is what bounds the step
    // the x^56 multiplier table (3 KB) and the bitmap's 4096-bit summary in LDS: an entry costs
    // one coalesced 8-byte load and LDS lookups; only the few per cent of residuals whose low 12
    // bits are taken by some address go on to the 2 MiB bitmap
    __shared__ uint32_t stab[3 * 256];
    __shared__ uint32_t coarse[kCoarseWords];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) stab[i] = p.tables[kTabX56 * 256 + i];
    if (threadIdx.x < kCoarseWords) coarse[threadIdx.x] = p.bitmap[bitmap_words(p.bitmap_lg) + threadIdx.x];
    __syncthreads();
    const uint32_t seg_cap = p.seg_cap;
    // work units: pairs of wave segments of the fast scan's list (a few hundred entries per
    // segment), then the dap list of the simple kernel, which all the blocks past the segments
    // share.  Four entries per thread per trip -- two from each segment of the pair -- so that
    // their list loads, then their bitmap loads, are in flight together (the chain entry ->
    // residual -> bitmap word is all latency) and the usual pair is a single trip.
    const bool is_dap = blockIdx.x >= (uint32_t)kApWaveSegs / 2;
    uint32_t n[2];
    const uint64_t *ap[2];
    uint32_t first = 0, stride = 2 * blockDim.x;
    if (is_dap) {  // both halves walk the same list, interleaved
        n[0] = n[1] = min(p.ctr->n_dap, p.dap_cap);
        ap[0] = ap[1] = p.dap;
        first = (blockIdx.x - kApWaveSegs / 2) * 4 * blockDim.x;
        stride = (gridDim.x - kApWaveSegs / 2) * 4 * blockDim.x;
    } else {
        for (int h = 0; h < 2; h++) {
            const uint32_t sg = 2 * blockIdx.x + h;
            n[h] = min(p.ctr->seg_ap[sg], seg_cap);
            ap[h] = p.ap + (uint64_t)sg * seg_cap;
        }
    }
    const uint32_t nmax = max(n[0], n[1]);
    for (uint32_t i0 = first + threadIdx.x; i0 < nmax; i0 += stride) {
        uint64_t e[4];
        uint32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // segment pair: k = 0,1 from the first, 2,3 from the second; dap: four consecutive strides
            const int h = is_dap ? 0 : k >> 1;
            const uint32_t i = i0 + (is_dap ? (uint32_t)k : (uint32_t)(k & 1)) * blockDim.x;
            e[k] = i < n[h] ? ap[h][i] : ~0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t code = entry_code(e[k]);
            c[k] = entry_value(e[k]);
            if (code >= 5 && code < 10) c[k] = gf_apply(stab, c[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e[k] == ~0ull || !((coarse[(c[k] & 4095u) >> 5] >> (c[k] & 31)) & 1u)) continue;
            const uint32_t at = bitmap_index(c[k], p.bitmap_lg);
            if ((p.bitmap[at >> 5] >> (at & 31)) & 1u) {  // rare: one atomic each
                const uint32_t idx = atomicAdd(&p.ctr->n_hits, 1u);
                if (p.order_cnt) {  // dense stream: into the buffer's bucket, its tile's part of it (adsb_device.h: order_tmp)
                    const uint32_t ch = (uint32_t)entry_chunk(e[k]), tl = entry_j(e[k]) / (uint32_t)fastgeo::kTile;
                    const uint32_t at = atomicAdd(&p.order_cnt[ch * fastgeo::kTilesPerChunk + tl], 1u);
                    if (at < kTileBucket) {
                        const size_t place = (size_t)ch * kOrderBucket + tl * kTileBucket + at;
                        p.order_tmp[place] = e[k];
                        // (an address/parity hit: no fields from the scan, the record builder slices it)
                        if (p.hit_fields) p.hit_fields[place * kHitFieldWords + 5] = 0u;
                    } else {
                        atomicOr(&p.ctr->overflow, 1u);
                    }
                } else if (idx < p.hits_cap) {
                    p.hits[idx] = e[k];
                    if (p.hit_fields) p.hit_fields[(size_t)idx * kHitFieldWords + 5] = 0u;
                } else {
                    atomicOr(&p.ctr->overflow, 1u);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// order.  The hit list is filled in whatever order workgroups finish; the host replay needs
// (buffer, j, try_phase) order (demodulate2400 walks j upwards and tries phases 4..8 at each,
// src/demod_2400.rs:121,158).  Sorting it here is a counting sort by buffer followed by a rank
// sort inside each buffer's handful of hits: a busy airspace leaves tens of hits per buffer, and
// keys are unique (a trial is either self-validating or address/parity, never both).
//   producers       whoever finds a hit puts it into its buffer's bucket (order_tmp, order_cnt)
//   k_order_prefix  one workgroup: exclusive prefix of the counts (order_base)
//   k_records       takes whole buckets: a workgroup sorts a buffer's bucket in LDS (rank sort) and
//                   writes its records at the bucket's place; the count goes back to zero, which
//                   is how the next pass must find it.  (The sorted hit list itself is never stored.)
// A pass whose lists overflowed is redone by the host anyway: its counts are only zeroed.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_order_prefix(ScanParams p)
{
    TAIL_PRIO();
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t T = fastgeo::kTilesPerChunk;
    if (p.ctr->overflow) {  // (uniform) nothing will be ordered: leave the counts clean
        for (uint32_t c = tid; c < p.n_chunks * T; c += 1024) p.order_cnt[c] = 0;
        return;
    }
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 <= p.n_chunks; c0 += 1024) {
        const uint32_t c = c0 + tid;
        uint32_t v = 0;   // the buffer's hits = its tiles' counts (adsb_device.h: kTileBucket)
        if (c < p.n_chunks)
#pragma unroll
            for (uint32_t t = 0; t < T; t++) v += min(p.order_cnt[c * T + t], kTileBucket);
        part[tid] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t add = tid >= off ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (c <= p.n_chunks) p.order_base[c] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------
// device-side scoring (adsb_device.h: ScoreDev).  first index at which an address is added: an
// open-addressing table of (value << 32 | index), atomic-min per key.
// ---------------------------------------------------------------------------

// index of the first adder of v in this pass, or 0xFFFFFFFF
__device__ __forceinline__ uint32_t score_hash_first(const ScoreDev &sd, uint32_t v)
{
    uint32_t h = score_hash_slot(v) & sd.hash_mask;
    for (;;) {
        const unsigned long long cur = __hip_atomic_load(&sd.hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == ~0ull) return 0xFFFFFFFFu;
        if ((uint32_t)(cur >> 32) == v) return (uint32_t)cur;
        h = (h + 1u) & sd.hash_mask;
    }
}

// "v is in the filter when trial i is scored" (src/icao_filter.rs:65-97; address 0 always is)
__device__ __forceinline__ bool score_in_filter(const ScoreDev &sd, uint32_t v, uint32_t i)
{
    if (v == 0) return true;
    if ((sd.exact[v >> 5] >> (v & 31)) & 1u) return true;
    if (sd.earlier && ((sd.earlier[v >> 5] >> (v & 31)) & 1u)) return true;
    return score_hash_first(sd, v) < i;
}

// src/mode_s/mod.rs:56-135 for trial i; *adds: the value this trial hands to icao_filter_add (or 0)
__device__ __forceinline__ int score_trial(const ScoreDev &sd, uint32_t i, uint32_t *adds)
{
    const uint32_t w = sd.si[i], v = w & 0xFFFFFFu, kind = w >> 24;
    *adds = 0;
    switch (kind) {
    case kSkApShort: return score_in_filter(sd, v, i) ? 1000 : -1;
    case kSkApLong: return score_in_filter(sd, v, i) ? 1000 : -2;
    case kSkDf11: return score_in_filter(sd, v, i) ? 1000 : -1;
    case kSkDf11Iid0:
        if (score_in_filter(sd, v, i)) return 1600;
        *adds = v;
        return 750;
    case kSkDf17:
        if (score_in_filter(sd, v, i)) return 1800;
        *adds = v;
        return 1400;
    case kSkDf18:
        if (score_in_filter(sd, v, i)) return 1800;
        *adds = v | (1u << 25);                       // ICAO_FILTER_ADSB_NT, src/icao_filter.rs:6
        return 1400;
    case kSkNone: return -3;                          // the reference's None: never taken
    default: return -2;
    }
}


// k_score: one thread per hit.  Its own score, whether it is the one its (buffer, j) emits
// (src/demod_2400.rs:149-207: strictly greater wins, from -2; emitted when >= 0) and what it adds.
__global__ __launch_bounds__(256) void k_score(ScanParams p)
{
    TAIL_PRIO();
    const ScoreDev &sd = p.score;
    const uint32_t n = sd.state->n;
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t first = blockIdx.x * per, last = min(n, first + per);
    uint32_t emits = 0, addc = 0;
    for (uint32_t i = first + threadIdx.x; i < last; i += blockDim.x) {
        const uint64_t pos = sd.pos[i];
        uint32_t g0 = i;
        while (g0 > 0 && i - g0 < 8 && sd.pos[g0 - 1] == pos) g0--;
        int best = -2, mine = -2;
        uint32_t win = 0xFFFFFFFFu, my_add = 0;
        for (uint32_t k = g0; k < n && k < g0 + 16 && sd.pos[k] == pos; k++) {
            uint32_t a;
            const int s = score_trial(sd, k, &a);
            if (k == i) {
                mine = s;
                my_add = a;
            }
            if (s > best) {
                best = s;
                win = k;
            }
        }
        const bool emit = win == i && best >= 0;
        sd.flag[i] = (emit ? 1u : 0u) | (my_add ? 2u : 0u) | ((uint32_t)(mine + 3) << 8);
        emits += emit;
        addc += my_add != 0;
    }
    __shared__ uint32_t tot[2];
    if (threadIdx.x < 2) tot[threadIdx.x] = 0;
    __syncthreads();
    if (emits) atomicAdd(&tot[0], emits);
    if (addc) atomicAdd(&tot[1], addc);
    __syncthreads();
    if (threadIdx.x < 2) sd.blk[2 * blockIdx.x + threadIdx.x] = tot[threadIdx.x];
}

// k_emit: the messages and the additions in order (block b writes behind what blocks < b write),
// the additions committed to the exact bitmap, the hash table left empty, the summary last.
__global__ __launch_bounds__(256) void k_emit(ScanParams p)
{
    TAIL_PRIO();
    const ScoreDev &sd = p.score;
    const uint32_t n = sd.state->n;
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t first = blockIdx.x * per, last = min(n, first + per);
    static_assert(kScoreBlocks == 256, "the block-offset reduction below takes one earlier block per thread");
    __shared__ uint32_t base[2], scan[2][256];
    __shared__ unsigned long long stage[5 * 256];  // this round's messages, 40 bytes each
    __shared__ uint32_t wtot[2][4];
    {
        // what the blocks before this one write: all threads fetch, one reduction (kScoreBlocks == blockDim)
        const uint32_t k = threadIdx.x;
        scan[0][k] = k < blockIdx.x ? sd.blk[2 * k] : 0u;
        scan[1][k] = k < blockIdx.x ? sd.blk[2 * k + 1] : 0u;
        __syncthreads();
        for (uint32_t off = 128; off > 0; off >>= 1) {
            if (k < off) {
                scan[0][k] += scan[0][k + off];
                scan[1][k] += scan[1][k + off];
            }
            __syncthreads();
        }
        if (k < 2) base[k] = scan[k][0];
        __syncthreads();
    }
    // an icao_flush preceded this pass: it scores against the other (clean) bitmap; the one the passes
    // before it used is cleared here, for the flush after this one
    if (sd.exact_retired) {
        uint4 *w = (uint4 *)sd.exact_retired;
        for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < kBitmapAllocWords / 4; k += gridDim.x * blockDim.x)
            w[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    adsb_msg *out = (adsb_msg *)sd.out_msgs;
    unsigned long long my_sum = 0;
    for (uint32_t i0 = first; i0 < last; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t f = i < last ? sd.flag[i] : 0u;
        {
            // inclusive scan of the two flags over the block: ballots inside a wave, four wave totals
            const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
            const unsigned long long m0 = __ballot(f & 1u), m1 = __ballot((f >> 1) & 1u);
            const unsigned long long below = ln == 63u ? ~0ull : ((1ull << (ln + 1u)) - 1ull);
            const uint32_t i0 = (uint32_t)__popcll(m0 & below), i1 = (uint32_t)__popcll(m1 & below);
            if (ln == 0) {
                wtot[0][wv] = (uint32_t)__popcll(m0);
                wtot[1][wv] = (uint32_t)__popcll(m1);
            }
            __syncthreads();
            uint32_t b0w = 0, b1w = 0;
            for (uint32_t k = 0; k < wv; k++) {
                b0w += wtot[0][k];
                b1w += wtot[1][k];
            }
            scan[0][threadIdx.x] = b0w + i0;
            scan[1][threadIdx.x] = b1w + i1;
            __syncthreads();
        }
        if (i < last) {
            const TrialRecord r = sd.rec[i];
            const uint32_t w = sd.si[i], v = w & 0xFFFFFFu, kind = w >> 24;
            if (f & 1u) {
                adsb_msg m;
                for (int k = 0; k < 14; k++) m.msg[k] = r.msg[k];
                m.len = (r.msg[0] & 0x80) ? ADSB_MODES_LONG_MSG_BYTES : ADSB_MODES_SHORT_MSG_BYTES;
                m.try_phase = (uint8_t)(r.j_tp >> 24);
                m.score = (int32_t)(f >> 8) - 3;
                m.j = r.j_tp & 0xFFFFFFu;
                m.chunk = r.chunk;
                // demod_2400.rs:191-198: the same three divisions, in this order
                const double signal_power = (double)(r.power & ((1ull << 40) - 1)) / 65535.0 / 65535.0;
                m.signal_level = signal_power / 33.0;
                // staged: the block's messages of this round are contiguous in the output, so they
                // leave as consecutive 8-byte words from consecutive lanes (separate 8-byte writes
                // to host memory from one lane each cost ~30 ns apiece)
                const unsigned long long *mw = (const unsigned long long *)&m;
                unsigned long long *sw = stage + 5u * (scan[0][threadIdx.x] - 1u);
#pragma unroll
                for (int k = 0; k < 5; k++) sw[k] = mw[k];
            }
            if (f & 2u) {
                const uint32_t val = kind == kSkDf18 ? (v | (1u << 25)) : v;
                host_store32(sd.out_adds + base[1] + scan[1][threadIdx.x] - 1u, val);
                if (kind != kSkDf18) atomicOr(&sd.exact[v >> 5], 1u << (v & 31));  // visible to later passes only
            }
            // leave the hash table empty for the next pass: every adder resets the slot its key sits in
            // (remembered at insertion; probes only happen in k_score, which has finished)
            const uint32_t hs = sd.slot[i];
            if (hs != 0xFFFFFFFFu) sd.hash[hs] = ~0ull;
        }
        __syncthreads();
        {
            const uint32_t words = 5u * scan[0][255];
            unsigned long long *ow = (unsigned long long *)(out + base[0]);
            for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) {
                const unsigned long long v = stage[w];
                __hip_atomic_store(ow + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                my_sum += v;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            base[0] += scan[0][255];
            base[1] += scan[1][255];
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_sum += __shfl_down(my_sum, off);
    if ((threadIdx.x & 63) == 0 && my_sum) atomicAdd(&sd.state->msg_sum, my_sum);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ bool is_last;
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&sd.state->blocks_done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    // the last block: totals and summary, then the state and the hash table back to empty
    __shared__ uint32_t tot[2][4];
    {
        uint32_t nm = threadIdx.x < gridDim.x ? sd.blk[2 * threadIdx.x] : 0u;   // kScoreBlocks == blockDim.x
        uint32_t na = threadIdx.x < gridDim.x ? sd.blk[2 * threadIdx.x + 1] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nm += __shfl_down(nm, off);
            na += __shfl_down(na, off);
        }
        if ((threadIdx.x & 63) == 0) {
            tot[0][threadIdx.x >> 6] = nm;
            tot[1][threadIdx.x >> 6] = na;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nm = tot[0][0] + tot[0][1] + tot[0][2] + tot[0][3];
        const uint32_t na = tot[1][0] + tot[1][1] + tot[1][2] + tot[1][3];
        const unsigned long long ms = atomicAdd(&sd.state->msg_sum, 0ull);
        uint32_t *sm = (uint32_t *)sd.summary;
        const uint32_t vals[8] = {nm, na, (uint32_t)ms, (uint32_t)(ms >> 32), sd.state->scored, 0u, 0u, sd.seq};
#pragma unroll
        for (int k = 0; k < 8; k++) host_store32(sm + k, vals[k]);
        sd.state->n = 0;
        sd.state->scored = 0;
        sd.state->blocks_done = 0;
        sd.state->msg_sum = 0;
    }
}


// ---------------------------------------------------------------------------
// records (adsb_tail_dev.h: records_block).  One wave per hit: lanes are message bits.  The 291-sample
// window behind j is rebuilt from IQ (rare path: a handful of hits per chunk, so magnitudes are not
// kept in HBM), the 112 bits of the trial phase are sliced (demod_2400.rs:158-182) and the 33-sample
// power summed (:191-196).
// ---------------------------------------------------------------------------
template <bool FROM_MAG, bool BUCKETS>
__global__ __launch_bounds__(256) void k_records(ScanParams p, TrialRecord *rec)
{
    if (p.order_cnt) TAIL_PRIO();  // dense streams only: elsewhere the scan is what bounds the step
    // (dynamic LDS, only asked for by device-ordered launches: with 8 KB more a block of a sparse
    // stream's launch held up the next scan's workgroups on its CU -- sparse step +3.6 %)
    extern __shared__ uint64_t sorted[];
    records_block<FROM_MAG, BUCKETS>(p, rec, blockIdx.x, gridDim.x, sorted, true);
}

// carry-over mode: the last kCarrySamples samples of the stream so far
__global__ __launch_bounds__(kCarrySamples) void k_update_carry(const uint32_t *__restrict__ prev,
                                                               const uint32_t *__restrict__ src, long long n,
                                                               uint32_t *__restrict__ next)
{
    const long long i = threadIdx.x, idx = n - kCarrySamples + i;
    next[i] = idx >= 0 ? src[idx] : prev[i + n];
}

// first phase of a shard that listed its fresh addresses as it scanned (ScanParams::fresh): only the summary is
// left to write.  Same ten words, same order as the records kernel's (seq last); rec_sum_lo carries the sum of the
// listed addresses, n_dap their count.
__global__ __launch_bounds__(64) void k_shard_summary(ScanParams p)
{
    if (threadIdx.x != 0) return;
    uint32_t *sm = (uint32_t *)p.summary;
    const uint32_t vals[9] = {p.ctr->n_hits, p.ctr->overflow, p.ctr->fresh_sum, p.ctr->n_fresh, 0u, 0u, 0u, p.seq, 0u};
    host_store32(sm + 9, summary_check(vals));
    host_store32(sm + 8, 0u);
#pragma unroll
    for (int k = 0; k < 8; k++) host_store32(sm + k, vals[k]);
}

// addresses learned elsewhere (other shards of the same capture) join the superset
__global__ __launch_bounds__(256) void k_set_addresses(const uint32_t *__restrict__ addrs, uint32_t n,
                                                       uint32_t *bitmap, uint32_t lg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bitmap_set(bitmap, lg, addrs[i] & 0xFFFFFFu);
}

// ---------------------------------------------------------------------------
// self-test: digest of the magnitude tail over consecutive f32 bit patterns of
// X = im^2 + rn(re^2) (an integer-valued float in [0, 2^31]).  Lets a test sweep every
// representable X against the CPU pipeline, which proves the folded constant and
// the device sqrt exactly.  out[0] += sum of outputs, out[1] ^= order-free hash.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mag_digest(uint32_t first_bits, uint32_t count,
                                                    unsigned long long *out)
{
    unsigned long long sum = 0, h = 0;
    // pairs (i, i + half) so that both lanes of the packed pipeline are exercised
    const uint32_t half = (count + 1) / 2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < half; i += gridDim.x * blockDim.x) {
        const uint32_t b0 = first_bits + i, b1 = first_bits + i + half;
        const f32x2 x = {__uint_as_float(b0), __uint_as_float(i + half < count ? b1 : 0u)};
        const uint32_t pk = mag_tail2(x);
        const uint32_t u0 = pk & 0xFFFFu, u1 = pk >> 16;
        sum += u0;
        h ^= ((unsigned long long)u0 + 1ull) * (2ull * b0 + 1ull);
        if (i + half < count) {
            sum += u1;
            h ^= ((unsigned long long)u1 + 1ull) * (2ull * b1 + 1ull);
        }
    }
    atomicAdd(&out[0], sum);
    atomicXor(&out[1], h);
}

inline int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
// hipGetLastError is sticky across unrelated calls (the caller's too): start every launch clean
inline void hip_clear() { (void)hipGetLastError(); }

}  // namespace

int launch_to_mag(const void *d_iq, uint32_t n, uint16_t *d_data, void *stream)
{
    hip_clear();
    const int blocks = ((kMagDataLen + 7) / 8 + 255) / 256;
    hipLaunchKernelGGL(k_to_mag, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)d_iq, n, d_data);
    return hip_ok(hipGetLastError());
}

int launch_reset(Counters *ctr, uint32_t *bitmap, uint32_t bitmap_lg, void *stream)
{
    hip_clear();
    const uint32_t ctr_blocks = (uint32_t)((sizeof(Counters) / 4 + 255) / 256);
    const uint32_t blocks = bitmap ? std::max(ctr_blocks, std::min(512u, bitmap_alloc_words(bitmap_lg) / 4 / 256 + 1u)) : ctr_blocks;
    hipLaunchKernelGGL(k_reset, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ctr, bitmap, bitmap_lg);
    return hip_ok(hipGetLastError());
}

int launch_match(const ScanParams &p, void *stream)
{
    hip_clear();
    // one block per two wave segments of the fast scan's AP list, 64 for the dap list; the fill
    // counts live on the device
    const uint32_t blocks = kApWaveSegs / 2 + 64;  // 64 blocks share the dap list
    hipLaunchKernelGGL(k_match, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_records(const ScanParams &p, bool from_mag, TrialRecord *d_rec, void *stream)
{
    hip_clear();
    // Contiguous runs of hits per block (the count lives on the device); on sparse input most
    // blocks find nothing and leave at once.  Measured on dense input (17 000 hits per pass, beside
    // a scan): the kernel's duration does not depend on the block count between one per CU and
    // one per buffer (the scan's four workgroups leave a SIMD 96 registers per lane, one wave of
    // this kernel, whatever the grid), and four or eight blocks per buffer are two and four
    // times slower (more rounds of the fixed per-block latencies).  What it does depend on is the
    // instruction count per hit: see the hit loop.
    // (every block ends with one atomic on the same counter, ~90 of them a microsecond on this chip: a pass of
    // 4096 buffers with a block per buffer spent 126 us in a kernel that writes a few hundred records --
    // profiles/r5_multi_overhead.txt; beyond 1024 blocks the runs per block simply get longer)
    uint32_t blocks = p.n_chunks + 8u;
    if (blocks > 1024) blocks = 1024;
    if (from_mag)  // (one caller-supplied buffer: never device-ordered)
        hipLaunchKernelGGL((k_records<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, d_rec);
    else if (p.order_cnt)  // device-ordered: dynamic LDS for the bucket being sorted
        hipLaunchKernelGGL((k_records<false, true>), dim3(blocks), dim3(256), kOrderBucket * (sizeof(uint64_t) + sizeof(uint16_t)),
                           (hipStream_t)stream, p, d_rec);
    else
        hipLaunchKernelGGL((k_records<false, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, d_rec);
    return hip_ok(hipGetLastError());
}

int launch_order_hits(const ScanParams &p, void *stream)
{
    hip_clear();
    if (!p.order_cnt || !p.order_base || !p.order_tmp) return 0;
    hipLaunchKernelGGL(k_order_prefix, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_score(const ScanParams &p, void *stream)
{
    hip_clear();
    if (!p.score.si) return 0;
    hipLaunchKernelGGL(k_score, dim3(kScoreBlocks), dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(k_emit, dim3(kScoreBlocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_set_addresses(const uint32_t *d_addrs, uint32_t n, uint32_t *bitmap, uint32_t bitmap_lg, void *stream)
{
    hip_clear();
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_set_addresses, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_addrs, n,
                       bitmap, bitmap_lg);
    return hip_ok(hipGetLastError());
}

int launch_shard_summary(const ScanParams &p, void *stream)
{
    hip_clear();
    hipLaunchKernelGGL(k_shard_summary, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_update_carry(const uint32_t *prev, const void *d_src, uint64_t n_samples, uint32_t *next, void *stream)
{
    hip_clear();
    hipLaunchKernelGGL(k_update_carry, dim3(1), dim3(kCarrySamples), 0, (hipStream_t)stream, prev,
                       (const uint32_t *)d_src, (long long)n_samples, next);
    return hip_ok(hipGetLastError());
}

int launch_mag_digest(uint32_t first_bits, uint32_t count, unsigned long long *d_out, void *stream)
{
    hip_clear();
    hipLaunchKernelGGL(k_mag_digest, dim3(1024), dim3(256), 0, (hipStream_t)stream, first_bits, count,
                       d_out);
    return hip_ok(hipGetLastError());
}

}  // namespace adsb
