// adsb_tail_dev.h -- device code of the tail of a pass (match helpers, the record builder) shared by the
// tail kernels (adsb_aux.hip) and the one-launch small pass (adsb_scan_fast.hip: k_scan_fast<.., FUSED>),
// whose last workgroup to finish runs the tail itself.
#pragma once
#include "../../include/adsb_hip.h"
#include "adsb_dev_common.h"
#include "adsb_scan_geometry.h"

namespace adsb {
namespace {

// The tail kernels are a dependent chain of small launches that run beside a scan which keeps every
// SIMD's vector pipe busy: their waves ask for issue priority, so the chain costs the scan the cycles
// it needs instead of waiting for the cycles the scan leaves (-DADSB_TAIL_PRIO=0: measurement).
#ifndef ADSB_TAIL_PRIO
#define ADSB_TAIL_PRIO 3
#endif
#define TAIL_PRIO() __builtin_amdgcn_s_setprio(ADSB_TAIL_PRIO)

__device__ __forceinline__ uint32_t gf_apply(const uint32_t *tab3, uint32_t h)
{
    return tab3[h & 255u] ^ tab3[256 + ((h >> 8) & 255u)] ^ tab3[512 + (h >> 16)];
}

__device__ __forceinline__ uint64_t order_key(uint64_t e)
{
    return ((e >> 28) << 3) | (uint64_t)(entry_code(e) % 5u);  // (chunk, j) | try_phase - 4
}

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

__device__ __forceinline__ uint64_t score_pos(const TrialRecord &r) { return (uint64_t)r.chunk << 24 | (r.j_tp & 0xFFFFFFu); }

// What one workgroup of a one-launch pass writes for another one to read (hits, their bit fields, list entries,
// fill counts), and how that one reads it: written THROUGH to the memory side and read from there (agent scope),
// access by access.  The alternative -- plain accesses with a release fence on one side and an acquire fence on
// the other -- writes back / invalidates the XCD's whole L2, and such a fence does not come back before every
// read in flight through that L2 has, including the 512 KB another pass beside this one is pulling over the link
// (tools/pcie_read_probe.hip pass: one fence per workgroup, four passes side by side: +2.5 us per pass; the
// one-buffer ring: kernels of 56 us instead of 28).  With these, "release" is s_waitcnt vmcnt(0) -- the
// write-through stores are acknowledged from the memory side -- and "acquire" is nothing.
template <typename T>
struct as_is {
    typedef T type;
};
template <bool SHARED, typename T>
__device__ __forceinline__ void st_shared(T *at, typename as_is<T>::type v)
{
    if constexpr (SHARED) __hip_atomic_store(at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *at = v;
}
template <bool SHARED, typename T>
__device__ __forceinline__ T ld_shared(const T *at)
{
    if constexpr (SHARED) return __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *at;
}

constexpr int kRecWindow = 296;   // magnitudes a trial can touch: data[j+19 .. j+290], rounded up
constexpr int kRecRow = 320;      // a window's row in LDS: five magnitudes per lane, stored unguarded
constexpr int kRecBatch = 64;     // records a block stages before writing them out together
#ifndef ADSB_REC_GROUP
#define ADSB_REC_GROUP 4
#endif
constexpr int kRecGroup = ADSB_REC_GROUP;  // hits a wave works on at once

// The record builder as block `bid` of `nblk` blocks of 256 threads: k_records is a grid of them, the
// one-launch small pass runs it as one block (its last workgroup to finish).  `sorted`: LDS for one
// bucket (device-ordered launches only).  `clean`: this call also does its share of clearing a retired
// bitmap (the one-launch pass has all its workgroups do that before they scan).
// SINGLE: the caller is the only block (the one-launch pass): no global atomics for the checksum or the block
// count, totals and zeroing over the `used_blocks` scan workgroups' counters only, no wait for the records'
// write acknowledgements before the summary (the host checks the checksum of what it finds and retries),
// and the per-bit residual constants from `tb_lds` (LDS) instead of memory.
template <bool FROM_MAG, bool BUCKETS, bool SINGLE = false>
__device__ __forceinline__ void records_block(const ScanParams &p, TrialRecord *rec, const uint32_t bid, const uint32_t nblk,
                                              uint64_t *sorted, const bool clean, const uint32_t used_blocks = 0,
                                              const uint32_t *tb_lds = nullptr)
{
    // (SINGLE: other workgroups of this very launch, and this one's second look, counted hits in with
    // atomics a moment ago: read past any line this CU's cache may still hold)
    uint32_t n_hits_now = SINGLE ? __hip_atomic_load(&p.ctr->n_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.ctr->n_hits;
    // SINGLE: the pass's workgroups have written `placed` records in place already (k_scan_fast: emit_records);
    // the hit list holds only what the second look found (usually nothing), its records go behind those
    uint32_t placed = 0;
    if constexpr (SINGLE) {
        placed = ld_shared<true>(&p.ctr->n_rec);
        if (placed > p.hits_cap || n_hits_now > p.hits_cap - placed) {   // (emit_records has flagged its own overflow)
            if (threadIdx.x == 0) atomicOr(&p.ctr->overflow, 1u);
            placed = min(placed, p.hits_cap);
            n_hits_now = 0;
        }
        rec += placed;
    }
    const uint32_t n = min(n_hits_now, p.hits_cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Housekeeping so that no pass needs a reset launch: after an icao_flush retired a
    // bitmap, clear it here (address 0 always tests true, src/icao_filter.rs:71-80: bit 0
    // starts set); it comes back into use two flushes later.  This pass's own counters are
    // zeroed at the very end, by the last block to finish.
    if (clean && p.clean_bitmap) bitmap_clear(p.clean_bitmap, p.bitmap_lg, bid * blockDim.x + threadIdx.x, nblk * blockDim.x);
    // A block owns a contiguous run of hits, so its records leave as one contiguous burst of
    // 16-byte stores (mapped host memory sits behind PCIe: thousands of separate 8-byte writes
    // cost ~6 ns each, wide neighbouring ones combine).  Per hit, a wave: the window of
    // magnitudes behind j is rebuilt from IQ into LDS with coalesced loads (rare path: a handful
    // of hits per chunk, so magnitudes are never kept in HBM), lanes are message bits
    // (demod_2400.rs:158-182), the 33-sample power is summed (:191-196).
    __shared__ uint16_t win[4][kRecGroup][kRecRow];
    __shared__ alignas(16) TrialRecord stage[kRecBatch];
    const uint32_t per = (n + nblk - 1) / nblk;
    const uint32_t first = bid * per, last = min(n, first + per);
    unsigned long long my_sum = 0;  // of the u64 words this thread sent to the host
    const bool overflowed = ld_shared<SINGLE>(&p.ctr->overflow) != 0;         // (uniform) the host redoes the pass
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
    uint32_t next_chunk = bid;
    bool flat_done = false;
    for (;;) {
    const uint64_t *src;
    uint32_t src_off, run_first, run_last;
    size_t field_base = 0;   // buckets: the bucket's first slot in ScanParams::hit_fields
    if constexpr (buckets) {
        if (overflowed || next_chunk >= p.n_chunks) break;
        const uint32_t c = next_chunk;
        next_chunk += nblk;
        // the buffer's bucket is seventeen sub-buckets, one per tile (adsb_device.h: kTileBucket): their counts, the
        // places their entries take in the compacted bucket
        constexpr uint32_t T = fastgeo::kTilesPerChunk;
        static_assert(T * kTileBucket <= kOrderBucket && kOrderBucket % 256 == 0, "sub-buckets fit the bucket");
        __shared__ uint32_t tcnt[T], tpre[T + 1];
        const uint32_t lo = p.order_base[c];
        const uint64_t *bucket = p.order_tmp + (size_t)c * kOrderBucket;
        __syncthreads();  // (the previous run is through with `sorted` and the counts)
        if (threadIdx.x < T) tcnt[threadIdx.x] = min(p.order_cnt[c * T + threadIdx.x], kTileBucket);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t at = 0;
            for (uint32_t t = 0; t < T; t++) {
                tpre[t] = at;
                at += tcnt[t];
            }
            tpre[T] = at;
        }
        if (threadIdx.x < T && !p.keep_counters) p.order_cnt[c * T + threadIdx.x] = 0;   // (a shard's first phase: its second still appends)
        __syncthreads();
        const uint32_t bn = tpre[T];
        uint64_t mine[kOrderBucket / 256];
        uint32_t rank[kOrderBucket / 256];
        bool have[kOrderBucket / 256];
        // (behind the sorted entries: where each one sits in the bucket -- its slot in ScanParams::hit_fields)
        uint16_t *const sorted_at = reinterpret_cast<uint16_t *>(sorted + kOrderBucket);
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++) {
            const uint32_t sl = threadIdx.x + 256u * k;                 // place in the bucket: tile * kTileBucket + index
            const uint32_t t = sl / kTileBucket, i = sl - t * kTileBucket;
            have[k] = t < T && i < tcnt[t];
            mine[k] = have[k] ? bucket[sl] : 0ull;
            if (have[k]) sorted[tpre[t] + i] = mine[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++) {
            rank[k] = 0;
            if (have[k]) {
                const uint64_t key = order_key(mine[k]);
                for (uint32_t q = 0; q < bn; q++) rank[k] += order_key(sorted[q]) < key;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (int)kOrderBucket / 256; k++)
            if (have[k]) {
                sorted[rank[k]] = mine[k];
                sorted_at[rank[k]] = (uint16_t)(threadIdx.x + 256u * k);
            }
        __syncthreads();
        src = sorted;
        src_off = lo;
        field_base = (size_t)c * kOrderBucket;
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
                // (past ng: a repeat, unused.  SINGLE: the hit list was appended to by this very launch)
                const uint64_t *at = &src[b0 - src_off + g0 + min((uint32_t)h, ng - 1u)];
                const uint64_t v = SINGLE ? __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *at;
                e[h] = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32 |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            }
            unsigned long long half[kRecGroup][2];
            uint32_t crc[kRecGroup];
            unsigned long long pw[kRecGroup];
            // What the scan knew about a self-validating hit when it found it -- all 112 sliced bits, as the
            // five bit-class fields, and the residual (the entry's value: 0 for a clean DF17 / DF18, the IID
            // bits for a clean DF11) -- came along in hit_fields: such a hit needs no window, no slicer and no
            // CRC here, only the 33-sample power (demod_2400.rs:191-196).  A group with a hit the match found
            // (address/parity trial: flag clear) takes the whole road for all four.
            bool from_fields = p.hit_fields != nullptr;
            size_t place[kRecGroup];
            if (from_fields) {
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) {
                    const uint32_t q = b0 - src_off + g0 + min((uint32_t)h, ng - 1u);
                    if constexpr (buckets) place[h] = field_base + reinterpret_cast<const uint16_t *>(sorted + kOrderBucket)[q];
                    else place[h] = q;
                }
                uint32_t all = 1u;
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) all &= ld_shared<SINGLE>(&p.hit_fields[place[h] * kHitFieldWords + 5]);
                from_fields = __builtin_amdgcn_readfirstlane((int)all) != 0;
            }
            if (from_fields) {
                const uint32_t k0 = __umul24((uint32_t)lane, 13108u) >> 16, r0 = (uint32_t)lane - 5u * k0;          // lane / 5, % 5
                const uint32_t k1 = __umul24((uint32_t)lane + 64u, 13108u) >> 16, r1 = (uint32_t)lane + 64u - 5u * k1;
                uint32_t fw0[kRecGroup], fw1[kRecGroup], w[kRecGroup];
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) {
                    fw0[h] = ld_shared<SINGLE>(&p.hit_fields[place[h] * kHitFieldWords + r0]);
                    fw1[h] = ld_shared<SINGLE>(&p.hit_fields[place[h] * kHitFieldWords + r1]);
                    if constexpr (FROM_MAG) {   // caller-supplied magnitudes: data[j + 19 + lane] as it is
                        const int d = (int)entry_j(e[h]) + 19 + lane;
                        w[h] = d < kMagDataLen ? ((const uint16_t *)p.src)[d] : 0u;
                        continue;
                    }
                    // the IQ behind data[j + 19 + lane] (the range-checked resource of the full path below)
                    const uint64_t chunk = entry_chunk(e[h]);
                    const int len = chunk_len(p.n_samples, chunk);
                    const uint32_t *iq = (const uint32_t *)p.src + chunk * (uint64_t)kChunkSamples;
                    const bool lead = p.carry != nullptr && (chunk > 0 || p.lead_from_src);
                    const int shift = lead ? kCarrySamples : 0;
                    const __amdgpu_buffer_rsrc_t rsrc =
                        __builtin_amdgcn_make_buffer_rsrc((void *)(iq - shift), 0, (len + shift) * 4, 0x00020000);
                    int off = ((int)entry_j(e[h]) + 19 - kLead + shift + lane) * 4;
                    asm volatile("" : "+v"(off));
                    w[h] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
                    if (p.carry != nullptr && !lead) {
                        const __amdgpu_buffer_rsrc_t crsrc =
                            __builtin_amdgcn_make_buffer_rsrc((void *)p.carry, 0, kCarrySamples * 4, 0x00020000);
                        int coff = off + kCarrySamples * 4;
                        asm volatile("" : "+v"(coff));
                        w[h] |= __builtin_amdgcn_raw_buffer_load_b32(crsrc, coff, 0, 0);
                    }
                }
#pragma unroll
                for (int h = 0; h < kRecGroup; h += 2) {  // two magnitudes per pass of the packed arithmetic
                    const uint32_t m2 = FROM_MAG ? (w[h] & 0xFFFFu) | (h + 1 < kRecGroup ? w[h + 1] << 16 : 0u)
                                                 : mag2(w[h], h + 1 < kRecGroup ? w[h + 1] : 0u);
                    const unsigned long long ma = lane < 33 ? (m2 & 0xFFFFu) : 0u, mb = lane < 33 ? (m2 >> 16) : 0u;
                    pw[h] = ma * ma;
                    if (h + 1 < kRecGroup) pw[h + 1] = mb * mb;
                }
#pragma unroll
                for (int h = 0; h < kRecGroup; h++) {
                    half[h][0] = __brevll(__ballot(((fw0[h] >> k0) & 1u) != 0));
                    half[h][1] = __brevll(__ballot(lane < 48 && ((fw1[h] >> k1) & 1u) != 0));
                    crc[h] = entry_value(e[h]);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                    for (int h = 0; h < kRecGroup; h++) pw[h] += __shfl_xor(pw[h], off);
            } else {
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
            const uint32_t *tb = SINGLE ? tb_lds : p.tables + kTabBitsOff;
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
            }  // the whole road
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
    if constexpr (SINGLE) {
        // ---- the one-launch pass: this block is the whole tail
        __shared__ unsigned long long wsum[4];
        __shared__ uint32_t stot[2][4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) my_sum += __shfl_down(my_sum, off);
        uint32_t ap = 0, cand = 0;
        for (uint32_t i = threadIdx.x; i < 4u * used_blocks; i += blockDim.x) ap += ld_shared<true>(&p.ctr->seg_ap[i]);
        for (uint32_t i = threadIdx.x; i < used_blocks; i += blockDim.x) cand += ld_shared<true>(&p.ctr->seg_cand[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ap += __shfl_down(ap, off);
            cand += __shfl_down(cand, off);
        }
        if (lane == 0) {
            wsum[wave] = my_sum;
            stot[0][wave] = ap;
            stot[1][wave] = cand;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long rs = wsum[0] + wsum[1] + wsum[2] + wsum[3] +
                                          atomicAdd((unsigned long long *)p.ctr->rec_sum, 0ull);   // (+ the records built in place)
            ap = stot[0][0] + stot[0][1] + stot[0][2] + stot[0][3];
            cand = stot[1][0] + stot[1][1] + stot[1][2] + stot[1][3];
            uint32_t *sm = (uint32_t *)p.summary;
            const uint32_t ovf = __hip_atomic_load(&p.ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // how long the launch took, on the device's own clock (its first workgroup's entry to here)
            const unsigned long long t0 = (unsigned long long)__hip_atomic_load(&p.ctr->t_start[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32 |
                                          __hip_atomic_load(&p.ctr->t_start[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t vals[9] = {placed + n_hits_now, ovf, (uint32_t)rs, p.ctr->n_dap,
                                      ap + p.ctr->n_dap, cand + p.ctr->n_cand_simple, (uint32_t)(rs >> 32), p.seq,
                                      (uint32_t)((unsigned long long)wall_clock64() - t0)};
            host_store32(sm + 9, summary_check(vals));
            host_store32(sm + 8, vals[8]);
#pragma unroll
            for (int k = 0; k < 8; k++) host_store32(sm + k, vals[k]);  // seq last
        }
        __syncthreads();
        // the counters this pass touched back to zero (the rest of the block never left it)
        constexpr uint32_t kHead = offsetof(Counters, seg_ap) / 4;
        for (uint32_t i = threadIdx.x; i < kHead; i += blockDim.x) ((uint32_t *)p.ctr)[i] = 0;
        for (uint32_t i = threadIdx.x; i < 4u * used_blocks; i += blockDim.x) p.ctr->seg_ap[i] = 0;
        for (uint32_t i = threadIdx.x; i < used_blocks; i += blockDim.x) p.ctr->seg_cand[i] = 0;
        return;
    }
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
    if (threadIdx.x == 0) is_last = atomicAdd(&p.ctr->blocks_done, 1u) == nblk - 1;
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
            const uint32_t vals[9] = {p.ctr->n_hits, p.ctr->overflow, (uint32_t)rs, p.ctr->n_dap,
                                      ap + p.ctr->n_dap, cand + p.ctr->n_cand_simple, (uint32_t)(rs >> 32), p.seq, 0u};
            host_store32(sm + 9, summary_check(vals));
            host_store32(sm + 8, 0u);
#pragma unroll
            for (int k = 0; k < 8; k++) host_store32(sm + k, vals[k]);  // seq last
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

}  // namespace
}  // namespace adsb
