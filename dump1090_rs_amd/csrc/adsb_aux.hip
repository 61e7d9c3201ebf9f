// adsb_aux.hip -- the small kernels around the scan: to_mag alone, the address/parity
// match, the record builder and the magnitude self-test digest.
#include "../../include/adsb_hip.h"
#include "adsb_dev_common.h"

namespace adsb {

namespace {

// ---------------------------------------------------------------------------
// to_mag alone (adsb_to_mag).  data[0..326) = 0, data[326+k] = mag(iq[k]), rest 0
// (src/lib.rs:36-50, src/utils.rs:43-58).
// ---------------------------------------------------------------------------
// The tail kernels are a dependent chain of small launches that run beside a scan which keeps every
// SIMD's vector pipe busy: their waves ask for issue priority, so the chain costs the scan the cycles
// it needs instead of waiting for the cycles the scan leaves (-DADSB_TAIL_PRIO=0: measurement).
#ifndef ADSB_TAIL_PRIO
#define ADSB_TAIL_PRIO 3
#endif
#define TAIL_PRIO() __builtin_amdgcn_s_setprio(ADSB_TAIL_PRIO)

__global__ __launch_bounds__(256) void k_to_mag(const uint32_t *__restrict__ iq, uint32_t n,
                                                uint16_t *__restrict__ data)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)kMagDataLen) return;
    uint32_t v = 0;
    if (i >= (uint32_t)kLead && i - kLead < n) v = mag_of_dword(iq[i - kLead]);
    data[i] = (uint16_t)v;
}

// ---------------------------------------------------------------------------
// reset: one launch per pass instead of a string of memsets.  Zeroes the counters and,
// after an icao_flush, the 2 MiB address bitmap; address 0 always tests true
// (src/icao_filter.rs:71-80: an empty slot equals 0), so bit 0 starts set.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reset(Counters *ctr, uint32_t *bitmap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t kCtrDwords = sizeof(Counters) / 4;
    if (i < kCtrDwords) ((uint32_t *)ctr)[i] = 0;
    if (bitmap) bitmap_clear(bitmap, i, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------
// match.  An address/parity trial can only score >= 0 if its CRC residual is in
// the filter when it is scored (mode_s/mod.rs:71,115,130); the bitmap now holds
// every address the filter can contain at any point of this call (plus 0, which
// icao_filter_test always accepts, icao_filter.rs:71-80).  Entries from the fast
// scan carry H' = x^51*H: the residual itself for 56-bit trials, x^56*H' for 112-bit
// ones (adsb_tables.h).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gf_apply(const uint32_t *tab3, uint32_t h)
{
    return tab3[h & 255u] ^ tab3[256 + ((h >> 8) & 255u)] ^ tab3[512 + (h >> 16)];
}

__global__ __launch_bounds__(256) void k_match(ScanParams p)
{
    if (p.order_cnt) TAIL_PRIO();  // dense streams only: elsewhere the scan is what bounds the step
    // the x^56 multiplier table (3 KB) and the bitmap's 4096-bit summary in LDS: an entry costs
    // one coalesced 8-byte load and LDS lookups; only the few per cent of residuals whose low 12
    // bits are taken by some address go on to the 2 MiB bitmap
    __shared__ uint32_t stab[3 * 256];
    __shared__ uint32_t coarse[kCoarseWords];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) stab[i] = p.tables[kTabX56 * 256 + i];
    if (threadIdx.x < kCoarseWords) coarse[threadIdx.x] = p.bitmap[kBitmapWords + threadIdx.x];
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
            if ((p.bitmap[c[k] >> 5] >> (c[k] & 31)) & 1u) {  // rare: one atomic each
                const uint32_t idx = atomicAdd(&p.ctr->n_hits, 1u);
                if (p.order_cnt) {  // dense stream: into the buffer's bucket (adsb_device.h: order_tmp)
                    const uint32_t ch = (uint32_t)entry_chunk(e[k]);
                    const uint32_t at = atomicAdd(&p.order_cnt[ch], 1u);
                    if (at < kOrderBucket) p.order_tmp[(size_t)ch * kOrderBucket + at] = e[k];
                    else atomicOr(&p.ctr->overflow, 1u);
                } else if (idx < p.hits_cap) {
                    p.hits[idx] = e[k];
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
__device__ __forceinline__ uint64_t order_key(uint64_t e)
{
    return ((e >> 28) << 3) | (uint64_t)(entry_code(e) % 5u);  // (chunk, j) | try_phase - 4
}

__global__ __launch_bounds__(1024) void k_order_prefix(ScanParams p)
{
    TAIL_PRIO();
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    if (p.ctr->overflow) {  // (uniform) nothing will be ordered: leave the counts clean
        for (uint32_t c = tid; c <= p.n_chunks; c += 1024) p.order_cnt[c] = 0;
        return;
    }
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 <= p.n_chunks; c0 += 1024) {
        const uint32_t c = c0 + tid;
        const uint32_t v = c < p.n_chunks ? p.order_cnt[c] : 0u;
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
// records.  One wave per hit: lanes are message bits.  The 291-sample window behind
// j is rebuilt from IQ (rare path: a handful of hits per chunk, so magnitudes are not
// kept in HBM), the 112 bits of the trial phase are sliced (demod_2400.rs:158-182)
// and the 33-sample power summed (:191-196).
// ---------------------------------------------------------------------------
// Stores into mapped host memory: system scope, i.e. written through the caches, so they
// are in host memory when the kernel has drained -- no cache flush needed afterwards.
__device__ __forceinline__ void host_store32(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 16 bytes straight to (mapped host) memory at system scope
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void host_store128(void *p, u32x4_t v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// src/icao_filter.rs:19-43 (u64 intermediates, & 4095), so that the host replay does not hash
__device__ __forceinline__ uint32_t icao_hash_dev(uint32_t a)
{
    unsigned long long h = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        h += (a >> (8 * k)) & 0xFFu;
        h += h << 10;
        h ^= h >> 6;
    }
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return (uint32_t)h & 4095u;
}

// ---------------------------------------------------------------------------
// device-side scoring (adsb_device.h: ScoreDev).  first index at which an address is added: an
// open-addressing table of (value << 32 | index), atomic-min per key.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t score_hash_slot(uint32_t v) { return (v * 2654435761u) >> 8; }

__device__ __forceinline__ uint32_t score_hash_insert(const ScoreDev &sd, uint32_t v, uint32_t idx)
{
    const unsigned long long mine = (unsigned long long)v << 32 | idx;
    uint32_t h = score_hash_slot(v) & sd.hash_mask;
    for (;;) {
        unsigned long long cur = atomicCAS(&sd.hash[h], ~0ull, mine);
        if (cur == ~0ull) return h;                    // claimed an empty slot
        if ((uint32_t)(cur >> 32) == v) {              // the key's slot: keep the smallest index
            atomicMin(&sd.hash[h], mine);
            return h;
        }
        h = (h + 1u) & sd.hash_mask;
    }
}

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

__device__ __forceinline__ uint64_t score_pos(const TrialRecord &r) { return (uint64_t)r.chunk << 24 | (r.j_tp & 0xFFFFFFu); }

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

constexpr int kRecWindow = 296;   // magnitudes a trial can touch: data[j+19 .. j+290], rounded up
constexpr int kRecRow = 320;      // a window's row in LDS: five magnitudes per lane, stored unguarded
constexpr int kRecBatch = 64;     // records a block stages before writing them out together
#ifndef ADSB_REC_GROUP
#define ADSB_REC_GROUP 4
#endif
constexpr int kRecGroup = ADSB_REC_GROUP;  // hits a wave works on at once

template <bool FROM_MAG, bool BUCKETS>
__global__ __launch_bounds__(256) void k_records(ScanParams p, TrialRecord *rec)
{
    if (p.order_cnt) TAIL_PRIO();  // dense streams only: elsewhere the scan is what bounds the step
    const uint32_t n = min(p.ctr->n_hits, p.hits_cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Housekeeping so that no pass needs a reset launch: after an icao_flush retired a
    // bitmap, clear it here (address 0 always tests true, src/icao_filter.rs:71-80: bit 0
    // starts set); it comes back into use two flushes later.  This pass's own counters are
    // zeroed at the very end, by the last block to finish.
    if (p.clean_bitmap) bitmap_clear(p.clean_bitmap, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    // A block owns a contiguous run of hits, so its records leave as one contiguous burst of
    // 16-byte stores (mapped host memory sits behind PCIe: thousands of separate 8-byte writes
    // cost ~6 ns each, wide neighbouring ones combine).  Per hit, a wave: the window of
    // magnitudes behind j is rebuilt from IQ into LDS with coalesced loads (rare path: a handful
    // of hits per chunk, so magnitudes are never kept in HBM), lanes are message bits
    // (demod_2400.rs:158-182), the 33-sample power is summed (:191-196).
    __shared__ uint16_t win[4][kRecGroup][kRecRow];
    __shared__ alignas(16) TrialRecord stage[kRecBatch];
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t first = blockIdx.x * per, last = min(n, first + per);
    unsigned long long my_sum = 0;  // of the u64 words this thread sent to the host
    const bool overflowed = p.ctr->overflow != 0;                             // (uniform) the host redoes the pass
    const bool do_score = p.score.si && n <= p.score.cap && !overflowed;      // (uniform) k_score follows
    // The hits come in runs.  Host-ordered passes: one run, this block's share of the hit list as it
    // was filled.  Device-ordered passes (dense streams): one run per buffer this block takes -- the
    // buffer's bucket, sorted here by (j, try_phase) with a rank sort in LDS, its records written at the
    // bucket's place order_base[buffer]: the sorted hit list as such is never stored.
    // (two instantiations: the host-ordered one keeps its plain loads from the hit list and none of the
    // run bookkeeping -- as one kernel the sparse stream's step was 1.7 % longer)
    constexpr bool buckets = BUCKETS;
    // (dynamic LDS, only asked for by device-ordered launches: with 8 KB more a block of a sparse
    // stream's launch held up the next scan's workgroups on its CU -- sparse step +3.6 %)
    extern __shared__ uint64_t sorted[];
    uint32_t next_chunk = blockIdx.x;
    bool flat_done = false;
    for (;;) {
    const uint64_t *src;
    uint32_t src_off, run_first, run_last;
    if constexpr (buckets) {
        if (overflowed || next_chunk >= p.n_chunks) break;
        const uint32_t c = next_chunk;
        next_chunk += gridDim.x;
        const uint32_t bn = min(p.order_cnt[c], kOrderBucket), lo = p.order_base[c];
        const uint64_t *bucket = p.order_tmp + (size_t)c * kOrderBucket;
        __syncthreads();  // (the previous run is through with `sorted`, everyone has read this count)
        if (threadIdx.x == 0) p.order_cnt[c] = 0;
        uint64_t mine[kOrderBucket / 256];
        uint32_t rank[kOrderBucket / 256];
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++) {
            const uint32_t i = threadIdx.x + 256u * k;
            mine[k] = i < bn ? bucket[i] : 0ull;
            if (i < bn) sorted[i] = mine[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++) {
            const uint32_t i = threadIdx.x + 256u * k;
            rank[k] = 0;
            if (i < bn) {
                const uint64_t key = order_key(mine[k]);
                for (uint32_t q = 0; q < bn; q++) rank[k] += order_key(sorted[q]) < key;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++)
            if (threadIdx.x + 256u * k < bn) sorted[rank[k]] = mine[k];
        __syncthreads();
        src = sorted;
        src_off = lo;
        run_first = lo;
        run_last = lo + bn;
    } else {
        if (flat_done) break;
        flat_done = true;
        src = p.hits;
        src_off = 0;
        run_first = first;
        run_last = last;
    }
    for (uint32_t b0 = run_first; b0 < run_last; b0 += kRecBatch) {
        const uint32_t cnt = min((uint32_t)kRecBatch, run_last - b0);
        // A wave takes kRecGroup hits at a time and lane h writes the record of the group's hit h.
        // Beside a scan only one wave of this kernel fits a SIMD (the scan leaves 96 registers), so the
        // group is the wave's only source of independent work: its hits' memory latencies (entry, IQ
        // window, residual constants, hash insertion) and LDS round trips overlap.  On its own that
        // changed nothing measurable; what did (dense stream: step -7 %) is the instruction count per
        // hit -- range-checked buffer loads instead of four compares and branches per IQ word, two
        // magnitudes per pass of the packed arithmetic, a slicer without the five-way branch on the
        // phase: 1375 -> 906 vector and 274 -> 82 branch instructions per group of four.
        for (uint32_t g0 = wave * kRecGroup; g0 < cnt; g0 += 4 * kRecGroup) {
            const uint32_t ng = min((uint32_t)kRecGroup, cnt - g0);
            uint64_t e[kRecGroup];
#pragma unroll
            for (int h = 0; h < kRecGroup; h++) {
                // (every lane reads the same entry; made wave-uniform for the buffer resource below)
                const uint64_t v = src[b0 - src_off + g0 + min((uint32_t)h, ng - 1u)];  // (past ng: a repeat, unused)
                e[h] = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32 |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            }
            // win[h][k] = data[j_h + 19 + k]
            if (FROM_MAG) {
#pragma unroll
                for (int h = 0; h < kRecGroup; h++)
                    for (int k = lane; k < kRecWindow; k += 64) {
                        const int d = (int)entry_j(e[h]) + 19 + k;
                        win[wave][h][k] = d < kMagDataLen ? ((const uint16_t *)p.src)[d] : (uint16_t)0;
                    }
            } else {
                // The IQ behind the window through a buffer resource that spans exactly the samples this
                // buffer may see (as the scan's tile loads, adsb_scan_fast.hip: load_tile_iq): the range
                // check returns zero before the start, past the ragged end of a short last buffer, and
                // -- carry-over mode -- the resource starts kCarrySamples early when those samples are
                // in src.  No branch per load, and all loads of the group are in flight together.
                constexpr int kQ = kRecRow / 64;  // 5 per lane (the row is a little longer than the window)
                static_assert(kRecRow % 64 == 0 && kRecRow >= kRecWindow, "whole lanes");
                uint32_t w[kRecGroup][kQ + 1];
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) {
                    const uint64_t chunk = entry_chunk(e[h]);
                    const int len = chunk_len(p.n_samples, chunk);
                    const uint32_t *iq = (const uint32_t *)p.src + chunk * (uint64_t)kChunkSamples;
                    const bool lead = p.carry != nullptr && (chunk > 0 || p.lead_from_src);
                    const int shift = lead ? kCarrySamples : 0;
                    const __amdgpu_buffer_rsrc_t rsrc =
                        __builtin_amdgcn_make_buffer_rsrc((void *)(iq - shift), 0, (len + shift) * 4, 0x00020000);
                    const int s0 = (int)entry_j(e[h]) + 19 - kLead + shift + lane;  // IQ sample behind data[j+19+lane]
                    // (each offset is made opaque: left to itself the compiler folds the "+ 256 q5" into the
                    // instruction's immediate offset, and the hardware adds that to the register offset
                    // without wrapping at 32 bits -- a negative register offset plus a positive immediate
                    // is then out of range although their sum is not.  Seen: zeros for samples 0 and 1.)
                    int off[kQ];
#pragma unroll
                    for (int q5 = 0; q5 < kQ; q5++) {
                        off[q5] = (s0 + 64 * q5) * 4;
                        asm volatile("" : "+v"(off[q5]));
                        w[h][q5] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[q5], 0, 0);
                    }
                    w[h][kQ] = 0;
                    if (p.carry != nullptr && !lead) {
                        // first buffer of a call: the samples before it are the end of the previous call
                        // (one resource's zero is the other's sample: the two loads OR together)
                        const __amdgpu_buffer_rsrc_t crsrc =
                            __builtin_amdgcn_make_buffer_rsrc((void *)p.carry, 0, kCarrySamples * 4, 0x00020000);
#pragma unroll
                        for (int q5 = 0; q5 < kQ; q5++) {
                            int coff = off[q5] + kCarrySamples * 4;
                            asm volatile("" : "+v"(coff));
                            w[h][q5] |= __builtin_amdgcn_raw_buffer_load_b32(crsrc, coff, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < kRecGroup; h++)
#pragma unroll
                    for (int q5 = 0; q5 < kQ; q5 += 2) {  // two magnitudes per pass of the packed arithmetic
                        const uint32_t m = mag2(w[h][q5], w[h][q5 + 1]);
                        win[wave][h][lane + 64 * q5] = (uint16_t)m;
                        if (q5 + 1 < kQ) win[wave][h][lane + 64 * (q5 + 1)] = (uint16_t)(m >> 16);
                    }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned long long half[kRecGroup][2];
            uint32_t crc[kRecGroup];
            unsigned long long pw[kRecGroup];
            const uint32_t *tb = p.tables + kTabBitsOff;
#pragma unroll
            for (int h = 0; h < kRecGroup; h++) {
                const uint32_t tp = entry_tp(e[h]);
                bool mybit[2];
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const int nbit = lane + 64 * hh;
                    bool bit = false;
                    if (nbit < 112) {
                        const uint32_t pos = tp + 12u * (uint32_t)nbit;  // relative to 5 * (j + 19); < 1341
                        const uint32_t sidx = __umul24(pos, 13108u) >> 16;  // pos / 5
                        bit = slice_value_any(&win[wave][h][sidx], (int)(pos - 5u * sidx)) > 0;
                    }
                    mybit[hh] = bit;
                    // lane n holds message bit n; the message is MSB-first
                    half[h][hh] = __brevll(__ballot(bit));
                }
                // the CRC residual, so that the host replay does not have to walk the bytes: XOR over
                // the set bits n of x^(bits-1-n) mod g (adsb_tables.h: build_bit_residuals), bits =
                // 112 when DF >= 16 (message bit 0 set), else 56
                const bool lng = (half[h][0] >> 63) != 0;
                uint32_t c = 0;
                if (mybit[0] && (lng || lane < 56)) c = tb[(lng ? 0 : 112) + lane];
                if (mybit[1] && lng) c ^= tb[64 + lane];
                crc[h] = c;
                unsigned long long m = 0;
                if (lane < 33) m = win[wave][h][lane];
                pw[h] = m * m;
            }
            // (butterflies: every lane ends up with every hit's totals, lane h keeps those of hit h)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) {
                    crc[h] ^= __shfl_xor(crc[h], off);
                    pw[h] += __shfl_xor(pw[h], off);
                }
            if ((uint32_t)lane < ng) {
                uint64_t me = e[0];
                unsigned long long h0 = half[0][0], h1 = half[0][1], mpw = pw[0];
                uint32_t mcrc = crc[0];
#pragma unroll
                for (int h = 1; h < kRecGroup; h++)
                    if (lane == h) {
                        me = e[h];
                        h0 = half[h][0];
                        h1 = half[h][1];
                        mpw = pw[h];
                        mcrc = crc[h];
                    }
                const uint32_t q = g0 + (uint32_t)lane;
                const uint32_t j = entry_j(me), tp = entry_tp(me);
                const bool lng = (h0 >> 63) != 0;
                TrialRecord r;
                r.power = mpw | ((unsigned long long)mcrc << 40);  // pw < 2^38
                r.chunk = (uint32_t)entry_chunk(me);
                r.j_tp = j | (tp << 24);
#pragma unroll
                for (int k = 0; k < 8; k++) r.msg[k] = (uint8_t)(h0 >> (56 - 8 * k));
#pragma unroll
                for (int k = 0; k < 6; k++) r.msg[8 + k] = (uint8_t)(h1 >> (56 - 8 * k));
                // `power` carries the residual (bit 0); bits 4..15: icao_hash of what this DF will ask
                // the filter about (bit 1) -- the residual for the address/parity DFs, else the address
                const uint32_t df = (uint32_t)(h0 >> 59);
                const bool ap = ((0xFF310031u >> df) & 1u) != 0;
                const uint32_t addr = (uint32_t)(h0 >> 32) & 0xFFFFFFu;
                r.pad = (uint16_t)(3u | (icao_hash_dev(ap ? mcrc : addr) << 4));
                if (do_score) {
                    // for k_score: what this trial asks the filter about, and what it may add
                    // (src/mode_s/mod.rs:56-135); clean DF11 (IID 0) / DF17 register as adders
                    const bool zero = (h0 | h1) == 0;
                    const bool d11 = df == 11u, d17 = df == 17u, d18 = df == 18u;
                    const bool clean11 = d11 && (mcrc & 0xFFFF80u) == 0, iid0 = (mcrc & 0x7Fu) == 0;
                    const bool clean17 = (d17 || d18) && mcrc == 0;
                    uint32_t kind = kSkOther;
                    if (zero) kind = kSkNone;
                    else if (ap) kind = lng ? kSkApLong : kSkApShort;
                    else if (clean11) kind = iid0 ? kSkDf11Iid0 : kSkDf11;
                    else if (clean17) kind = d17 ? kSkDf17 : kSkDf18;
                    const uint32_t v = ap ? mcrc : addr;
                    p.score.si[b0 + q] = v | (kind << 24);
                    p.score.pos[b0 + q] = score_pos(r);
                    p.score.slot[b0 + q] = (kind == kSkDf11Iid0 || kind == kSkDf17) ? score_hash_insert(p.score, v, b0 + q)
                                                                                   : 0xFFFFFFFFu;
                }
                stage[q] = r;
            }
            __builtin_amdgcn_wave_barrier();  // win is rewritten for the wave's next group
        }
        __syncthreads();
        if (threadIdx.x < 2 * cnt) {
            // a pass that k_score takes over keeps its records in HBM: the host only wants them when
            // it cannot use the device's result, and fetches them then (adsb_collect.cpp: finish_pass)
            if (do_score)
                ((u32x4_t *)(p.score.rec + b0))[threadIdx.x] = ((const u32x4_t *)stage)[threadIdx.x];
            else
                host_store128((char *)(rec + b0) + 16 * threadIdx.x, ((const u32x4_t *)stage)[threadIdx.x]);
            // (its own read of the two words: the 128-bit value above is only ever an asm operand)
            const unsigned long long *sw = (const unsigned long long *)stage + 2 * threadIdx.x;
            my_sum += sw[0] + sw[1];
        }
        __syncthreads();  // stage is refilled by the next batch
    }
    }  // runs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's records have left
    // the pass's record checksum (the host recomputes it over what it finds in its memory)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) my_sum += __shfl_down(my_sum, off);
        if (lane == 0 && my_sum) atomicAdd((unsigned long long *)p.ctr->rec_sum, my_sum);
        // performed (device scope) before this block reports itself done below; no cache flush: a
        // __threadfence() here would write back the XCD's L2 under the running scan, once per block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // The last block to finish totals the counters into the summary for the host -- like the
    // records it goes straight into mapped host memory with write-through stores, so the
    // completion event behind this kernel needs no system-scope cache flush -- and then
    // zeroes this pass's counters block, which the same slot's next pass starts from.
    __shared__ bool is_last;
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&p.ctr->blocks_done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    {
        // (every thread's loads are issued together: as a loop of one wave this total was most of the
        // kernel's duration beside a scan -- a hundred dependent round trips to a contended L2)
        static_assert(kApWaveSegs % 256 == 0 && kApSegments % 256 == 0, "unrolled below");
        uint32_t ap = 0, cand = 0;
#pragma unroll
        for (int i = 0; i < kApWaveSegs / 256; i++) ap += p.ctr->seg_ap[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < kApSegments / 256; i++) cand += p.ctr->seg_cand[i * 256 + threadIdx.x];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ap += __shfl_down(ap, off);
            cand += __shfl_down(cand, off);
        }
        __shared__ uint32_t tot[2][4];
        if (lane == 0) {
            tot[0][wave] = ap;
            tot[1][wave] = cand;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            ap = tot[0][0] + tot[0][1] + tot[0][2] + tot[0][3];
            cand = tot[1][0] + tot[1][1] + tot[1][2] + tot[1][3];
            uint32_t *sm = (uint32_t *)p.summary;
            if (p.score.si) {
                const uint32_t nh = p.ctr->n_hits;
                const bool ok = !p.ctr->overflow && nh <= p.score.cap;
                p.score.state->n = ok ? nh : 0u;
                p.score.state->scored = ok ? 1u : 0u;
            }
            const unsigned long long rs = atomicAdd((unsigned long long *)p.ctr->rec_sum, 0ull);
            const uint32_t vals[8] = {p.ctr->n_hits, p.ctr->overflow, (uint32_t)rs, p.ctr->n_dap,
                                      ap + p.ctr->n_dap, cand + p.ctr->n_cand_simple, (uint32_t)(rs >> 32), p.seq};
#pragma unroll
            for (int k = 0; k < 8; k++) host_store32(sm + k, vals[k]);
        }
    }
    __syncthreads();
    if (p.keep_counters) {  // first phase of a shard: the match still has to see the lists
        if (threadIdx.x == 0) {
            p.ctr->blocks_done = 0;
            p.ctr->rec_sum[0] = p.ctr->rec_sum[1] = 0;  // the second phase's records kernel starts its own sum
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < sizeof(Counters) / 4; i += blockDim.x) ((uint32_t *)p.ctr)[i] = 0;
}

// carry-over mode: the last kCarrySamples samples of the stream so far
__global__ __launch_bounds__(kCarrySamples) void k_update_carry(const uint32_t *__restrict__ prev,
                                                               const uint32_t *__restrict__ src, long long n,
                                                               uint32_t *__restrict__ next)
{
    const long long i = threadIdx.x, idx = n - kCarrySamples + i;
    next[i] = idx >= 0 ? src[idx] : prev[i + n];
}

// addresses learned elsewhere (other shards of the same capture) join the superset
__global__ __launch_bounds__(256) void k_set_addresses(const uint32_t *__restrict__ addrs, uint32_t n,
                                                       uint32_t *bitmap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bitmap_set(bitmap, addrs[i] & 0xFFFFFFu);
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
    const int blocks = (kMagDataLen + 255) / 256;
    hipLaunchKernelGGL(k_to_mag, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)d_iq, n, d_data);
    return hip_ok(hipGetLastError());
}

int launch_reset(Counters *ctr, uint32_t *bitmap, void *stream)
{
    hip_clear();
    const uint32_t blocks = bitmap ? 512u : (uint32_t)((sizeof(Counters) / 4 + 255) / 256);
    hipLaunchKernelGGL(k_reset, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ctr, bitmap);
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
    uint32_t blocks = p.n_chunks + 8u;
    if (blocks > 4096) blocks = 4096;
    if (from_mag)  // (one caller-supplied buffer: never device-ordered)
        hipLaunchKernelGGL((k_records<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, d_rec);
    else if (p.order_cnt)  // device-ordered: dynamic LDS for the bucket being sorted
        hipLaunchKernelGGL((k_records<false, true>), dim3(blocks), dim3(256), kOrderBucket * sizeof(uint64_t), (hipStream_t)stream, p, d_rec);
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

int launch_set_addresses(const uint32_t *d_addrs, uint32_t n, uint32_t *bitmap, void *stream)
{
    hip_clear();
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_set_addresses, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_addrs, n,
                       bitmap);
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
