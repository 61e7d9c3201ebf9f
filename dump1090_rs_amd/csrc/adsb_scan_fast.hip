// adsb_scan_fast.hip -- the gfx950 scan kernel: IQ in, trial syndromes out, magnitudes
// never leave the CU.
//
// One workgroup (256 threads = 4 wave64) owns a tile of 7712 preamble positions j of
// one chunk; the 8004 magnitudes those positions can touch (j .. j+290) live in LDS.
// The reference walks j serially and slices 5 x 112 bits per surviving j with a small
// state machine (src/demod_2400.rs:121-207).  Here the same decisions are taken densely
// and bit-parallel:
//
//  P1 magnitudes   dwordx4 IQ loads (4 samples / lane, aligned, coalesced) -> exact f32
//                  magnitude (src/utils.rs:47-55) -> u16 in LDS.
//  P2 sign planes  every decision the reference can ever take on this tile is the sign of
//                  a short integer correlation of neighbouring magnitudes:
//                    slicer phase ph at sample s (demod_2400.rs:72-83)   5 kinds
//                    m[s] > m[s+1] (check_preamble :221-317)              1 kind
//                  (check_preamble's "<" is taken as "<=", the complement of ">", and made
//                  strict again in P4).  All six are taken for every sample.  A lane walks samples 12 apart
//                  (bit n and bit n+5 of a message are 12 samples apart), four
//                  neighbouring residues at a time so the first differences are shared,
//                  and shifts each sign into an accumulator with one v_alignbit -- no
//                  compare, no cross-lane traffic.  The accumulators are stored as bytes
//                  of bit planes: plane (kind, s mod 12), bit s div 12.
//  P3 preamble     check_preamble's five patterns are AND/OR of the ">" planes and their
//                  complements at fixed offsets: one lane evaluates 32 positions j per instruction.
//  P4 gates        the ~4.5 % of positions that match a pattern get the value tests
//                  (high/SNR/quiet, :129-146) from LDS magnitudes, one lane each, and the
//                  strict form of the "<" tests of the branch they matched (equal neighbours:
//                  the reference's own test sequence decides).
//  P5 trials       for the ~1 % that survive, each (j, try_phase) is one lane: the five
//                  bit classes n mod 5 of the message are five 23-bit fields cut out of the
//                  sign planes with two dword loads and a funnel shift; DF and the CRC-24
//                  syndrome come from table lookups on the fields (adsb_tables.h); the
//                  message bytes are never assembled here.
//
// The kernel is VALU-issue bound (tools/valu_rate.hip: ~4.2 cycles per wave64 VOP3 /
// mad / cvt / compare, ~2.7 for plain VOP2 add/and/shift), so the code below is written
// to the instruction: 24-bit multiplies with magic constants instead of divisions,
// shifts and masks instead of bit-field extracts, u16 LDS reads instead of unpacking.
//
// Nothing in the kernel has a capacity that input density could exceed: each wave keeps
// its matches and candidates in its own small LDS regions and drains them in rounds
// (P3..P5 below), so exactness never depends on how dense the signal is.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "adsb_dev_common.h"
#include "adsb_scan_geometry.h"
#include "adsb_tail_dev.h"

namespace adsb {

namespace {

using namespace fastgeo;

#ifndef ADSB_PRIO_LATE
#define ADSB_PRIO_LATE 1   // wave priority during P3..P5 (0 = leave it alone)
#endif
#ifndef ADSB_PRIO_LATE_DENSE
#define ADSB_PRIO_LATE_DENSE 0  // ... also on dense streams (measured round 3: see DESIGN.md)
#endif
#ifndef ADSB_GATE_ASM
#define ADSB_GATE_ASM 1    // P4: the 19 magnitudes of a match as opaque zero-extended LDS reads (0 = C++ u16 loads)
#endif
#ifndef ADSB_SCAN_THREADS
#define ADSB_SCAN_THREADS 256
#endif
constexpr int kThreads = ADSB_SCAN_THREADS;   // 256 (512 was measured: 7 % slower)
#ifndef ADSB_SCAN_OCC
#define ADSB_SCAN_OCC 4
#endif
#ifndef ADSB_SCAN_RES
#define ADSB_SCAN_RES 4
#endif
constexpr int kWavesPerSimd = kThreads == 512 ? 8 : ADSB_SCAN_OCC;  // = workgroups per CU
constexpr int kResPerItem = kThreads == 512 ? 2 : ADSB_SCAN_RES;  // residues one P2 lane walks
constexpr int kAllocSlots = 96 * kPlaneBytes + 16;  // 8080 magnitudes P2 may read
constexpr int kPlaneGT = 60;                  // planes 0..59: slicer sign, kind*12 + residue
constexpr int kPlanes = 84;                   // 60..83: GT ("m[s] > m[s+1]") residues 0..23
                                              //   (residue r+12 = residue r advanced one bit)
constexpr int kItems2 = (12 / kResPerItem) * kPlaneBytes;  // P2 items: (residue group, plane byte)
constexpr int kItems3 = 12 * (kPlaneBytes / 4);  // 252 P3 items: (residue, plane dword)
static_assert(kItems3 <= 256, "one P3 item per thread");
static_assert(kAllocSlots <= 8192, "slots fit 13 bits");
constexpr int kWaves = kThreads / 64;
constexpr int kPatPerWave = 256;              // a wave's pattern matches (one round)
constexpr int kRoundBits = 4;                 // plane bits per round when they do not fit: 64 x 4 <= 256
constexpr int kCandPerWave = 128;             // a wave's candidates waiting for the trial stage
static_assert(64 * kRoundBits <= kPatPerWave, "wave-private regions");
constexpr int kHitCap = 32;                   // staged hits per tile (more go straight to HBM)

// LDS accesses wider than their address is aligned are legal on gfx950 but replayed at 64
// cycles (SQ_LDS_UNALIGNED_STALL); with unaligned-access-mode on (the default) the
// compiler merges neighbouring u16 / u32 LDS reads into exactly those.  The scan kernel is
// compiled with the mode off: merges only happen where alignment is known.
#if defined(__HIP_DEVICE_COMPILE__)
#define ADSB_NO_UNALIGNED __attribute__((target("no-unaligned-access-mode")))
#else
#define ADSB_NO_UNALIGNED
#endif

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for the next tile's IQ prefetch at every phase boundary; here only LDS
// traffic (lgkmcnt) is drained before s_barrier, global loads stay in flight.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Order this wave's own LDS traffic: writes before, reads after.  LDS operations of one
// wave complete in order, so draining lgkmcnt is all it takes; "memory" keeps the
// compiler from moving LDS accesses across.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);  // ({hi,lo} >> sh)[31:0], sh in 0..31
}

// shift the sign bit of v into acc from the right
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, int v)
{
    return alignbit(acc, (uint32_t)v, 31);
}

// inclusive prefix sum across the 64 lanes of a wave, in registers (DPP row shifts and
// row broadcasts; lanes with no source add 0)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
    return (uint32_t)v;
}

// base + rank of this lane among the set bits of a wave mask (the base rides in mbcnt's addend)
__device__ __forceinline__ uint32_t mask_rank(unsigned long long m, uint32_t base = 0u)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}

__device__ __forceinline__ uint32_t lowmask(int n)  // n low bits set, n clamped to 0..32
{
    return n <= 0 ? 0u : (n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u);
}


struct alignas(16) FastLds {
    uint16_t mag[kAllocSlots];         // P1..P4
    uint32_t plane[kPlanes * kPlaneDw];
    uint32_t tab[3 * 256];             // F'0 F'1 F'2 (adsb_tables.h)
    uint32_t r16[16];                  // x^24..x^27 reduction
    uint32_t field[300];               // field addressing (adsb_tables.h: build_field_table)
    uint16_t pat[kWaves * kPatPerWave];    // per wave: slot | branch (0..4) << 13
    uint16_t cand[kWaves * kCandPerWave];  // per wave: slot (cand_entry() expands it for the trial stage)
    uint64_t hit[kHitCap];
    uint32_t nhit[2], hit_base;  // staged-hit count of a tile, double-buffered by tile parity
};

// P4, one pattern match: high / base_signal / base_noise of the branch that matched first
// (demod_2400.rs:227-317), the 3.5 dB test (:129) and the quiet samples (:135-146).
// Branch-free; returns 1 when the position goes on to be sliced.
__device__ __forceinline__ uint32_t gate_eval(const uint16_t *mag, uint32_t ent)
{
    // one u16 LDS read per magnitude (no unpacking on the VALU).  The kernel is compiled
    // without unaligned-access-mode (ADSB_NO_UNALIGNED below), or these would be merged into
    // 8/16-byte reads at a 2-byte aligned address, which the LDS replays at 64 cycles each.
    const uint16_t *pm = mag + (ent & 0x1FFFu);
    const uint32_t br = (ent >> 13) & 7u;  // which branch's pattern matched, with "<=" for "<"
#if ADSB_GATE_ASM
    // The reads as the instructions themselves, results as plain 32-bit values: left to the compiler
    // the u16 loads become "any-extending" ones whose users are SDWA forms (4.2 cycles where the
    // plain add / sub takes 2.7) plus six v_and 0xffff in front of the max3 chain.  One block, one
    // wait: nothing else of this wave's is in flight here (the pattern entry was waited for).
    int p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, q14, q15, q16, q17, q18;
    {
        const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t *)pm;
        asm volatile(
            "ds_read_u16 %0, %18\n\tds_read_u16 %1, %18 offset:2\n\tds_read_u16 %2, %18 offset:4\n\t"
            "ds_read_u16 %3, %18 offset:6\n\tds_read_u16 %4, %18 offset:8\n\tds_read_u16 %5, %18 offset:10\n\t"
            "ds_read_u16 %6, %18 offset:12\n\tds_read_u16 %7, %18 offset:14\n\tds_read_u16 %8, %18 offset:16\n\t"
            "ds_read_u16 %9, %18 offset:18\n\tds_read_u16 %10, %18 offset:20\n\tds_read_u16 %11, %18 offset:22\n\t"
            "ds_read_u16 %12, %18 offset:24\n\tds_read_u16 %13, %18 offset:28\n\tds_read_u16 %14, %18 offset:30\n\t"
            "ds_read_u16 %15, %18 offset:32\n\tds_read_u16 %16, %18 offset:34\n\tds_read_u16 %17, %18 offset:36\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7), "=&v"(p8),
              "=&v"(p9), "=&v"(p10), "=&v"(p11), "=&v"(p12), "=&v"(q14), "=&v"(q15), "=&v"(q16), "=&v"(q17), "=&v"(q18)
            : "v"(a)
            : "memory");
    }
#else
    const int p0 = pm[0];
    const int p1 = pm[1], p2 = pm[2], p3 = pm[3], p4 = pm[4], p5 = pm[5], p6 = pm[6], p7 = pm[7],
              p8 = pm[8], p9 = pm[9], p10 = pm[10], p11 = pm[11], p12 = pm[12];
    const int q14 = pm[14], q15 = pm[15], q16 = pm[16], q17 = pm[17], q18 = pm[18];
#endif
    // high / base_signal / base_noise of the five branches (:227-317), written around what they
    // share: with X = p3+p9 (branches 1-3) or p4+p10 (branches 4, 5)
    //   high  = (p1 + p12 + X + [1]p11 + [3](p4+p10) + [5]p2) / 4
    //   sig   = X*not[3] + p1*not[5] + p12*not[1]
    //   noise = p6 + p7 + [1,2,4]p5 + [2,4,5]p8
    // five 0 / -1 masks per branch, 5 bits each in one constant: A=[1] B=[3] C=[5] G=[1,2,4] H=[2,4,5]
    constexpr uint32_t kMasksPacked = 9u | (24u << 5) | (2u << 10) | (24u << 15) | (20u << 20);
    const int mk = (int)(kMasksPacked >> (5u * br));
#define MASK(bit) __builtin_amdgcn_sbfe(mk, (bit), 1)
    const int mA = MASK(0), mB = MASK(1), mC = MASK(2), mG = MASK(3), mH = MASK(4);
#undef MASK
    const int s39 = p3 + p9, s410 = p4 + p10;
    const int X = br >= 3u ? s410 : s39;
    const int high = (p1 + p12 + X + (p11 & mA) + (s410 & mB) + (p2 & mC)) >> 2;
    const int sig = (X & ~mB) + (p1 & ~mC) + (p12 & ~mA);
    const int noise = p6 + p7 + (p5 & mG) + (p8 & mH);
    const int loud = max(max(max(p5, p6), max(p7, p8)), max(max(q14, q15), max(max(q16, q17), q18)));
    uint32_t pass = (uint32_t)(2 * sig >= 3 * noise) & (uint32_t)(loud < high);  // :129, :135-146
    // The pattern stage has no "<" plane: it took p[o] <= p[o+1] for the four "<" of the branch
    // (:221 and the branch's own three).  Equal neighbours are rare; when one of those four
    // pairs is equal the reference may have taken a later branch or none, so that position is
    // decided by the reference's own sequence of tests (preamble_gates, adsb_dev_common.h).
    const int dx = br >= 3u ? p4 - p3 : p3 - p2;      // branches 4, 5: p3 < p4;  1-3: p2 < p3
    const int dy = br >= 3u ? p10 - p9 : p9 - p8;     //               p9 < p10;      p8 < p9
    const int dz = br == 0u ? p11 - p10 : p12 - p11;  // branch 1: p10 < p11;  others: p11 < p12
    if (min(min(p1 - p0, dx), min(dy, dz)) <= 0) pass = (uint32_t)preamble_gates(pm);
    return pass;
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // v_bitop3_b32: a ^ b ^ c in one op
}

// candidate entry: slot | slot/12 << 13 | slot%12 << 23
__device__ __forceinline__ uint32_t cand_entry(uint32_t slot)
{
    const uint32_t qs = (slot * 10923u) >> 17;  // slot / 12 (slot < 16384)
    return slot | (qs << 13) | ((slot - 12u * qs) << 23);
}

// P5, one trial.  Message bit n = 5k + r of trial phase tp sits at 5x-oversampled position
// 5*(slot+19) + tp + 12n, i.e. sample slot + 19 + (tp+12r)/5 + 12k with slicer phase
// (tp+12r) % 5: field r is 23 consecutive bits of one sign plane; which plane and where
// comes from s.field.  Branch-free so that two trials per lane interleave.
struct Trial {
    uint32_t f[5];   // the five bit classes n mod 5 (bit k = message bit 5k + r)
    uint32_t h;      // x^51 * H reduced: short messages' CRC residual as is (adsb_tables.h)
    uint32_t code;   // try_phase - 4, + 5 for 112-bit messages
    uint32_t cslot;
    bool is_ap, is_hit, learn;  // address/parity trial; self-validating hit; hit that adds its address
};

__device__ __forceinline__ void trial_eval(const FastLds &s, uint32_t ce, uint32_t tpi, Trial &o)
{
    const uint32_t qs = (ce >> 13) & 0x3FFu, rs = ce >> 23;
    o.cslot = ce & 0x1FFFu;
    const uint32_t *ft = s.field + __umul24(tpi, 60u) + rs;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t fe = ft[r * 12];       // LDS address of the plane row | bit offset << 16 (P0)
        const uint32_t qq = qs + (fe >> 16);  // plane bit of message bit r
        // 4-byte aligned only: becomes one ds_read2_b32 (not an 8-byte read off its alignment,
        // which is replayed at 64 cycles -- ADSB_NO_UNALIGNED)
        typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
        lds_u32 pl = (lds_u32)(uintptr_t)((fe & 0xFFFFu) + ((qq >> 3) & 0x7Cu));
        const uint32_t lo = pl[0], hi = pl[1];
        o.f[r] = alignbit(hi, lo, qq);  // the shift is qq mod 32
    }
    const uint32_t *f = o.f;
    // mod.rs:41: DF = message bits 0..4 = bit 0 of the five fields
    const uint32_t df = ((f[0] & 1u) << 4) | ((f[1] & 1u) << 3) | ((f[2] & 1u) << 2) | ((f[3] & 1u) << 1) | (f[4] & 1u);
    const uint32_t lng = f[0] & 1u;  // DF >= 16: 112 bits
    // 112 bits: n <= 111 -> k <= 22 for r < 2, k <= 21 otherwise; 56 bits: n <= 55 -> k <= 11
    // for r = 0, k <= 10 otherwise.  (The reference's all-zero-message test, mod.rs:51, is left
    // to the host replay: an all-zero trial is DF 0 with residual 0, goes out as an address/
    // parity entry, matches address 0 and is dropped there -- it cannot arise in bulk, zero
    // samples match no preamble.)
    const uint32_t mk0 = lng ? 0x7FFFFFu : 0xFFFu, mk1 = lng ? 0x7FFFFFu : 0x7FFu, mk2 = lng ? 0x3FFFFFu : 0x7FFu;
    const uint32_t fm[5] = {f[0] & mk0, f[1] & mk1, f[2] & mk2, f[3] & mk2, f[4] & mk2};
    // sum_r x^(4-r) * F'(f_r) as a 28-bit polynomial, reduced once (adsb_tables.h)
    const uint32_t *tF = s.tab;
    uint32_t g[5];
#pragma unroll
    for (int r = 0; r < 5; r++)  // one byte-select-and-shift per index, one three-way XOR per field
        g[r] = xor3(tF[fm[r] & 0xFFu], tF[256 + ((fm[r] >> 8) & 0xFFu)], tF[512 + ((fm[r] >> 16) & 0xFFu)]);
    const uint32_t hp = xor3(g[0] << 4, g[1] << 3, xor3(g[2] << 2, g[3] << 1, g[4]));
    const uint32_t h = (hp & 0xFFFFFFu) ^ s.r16[hp >> 24];
    o.h = h;
    o.code = tpi + 5u * lng;
    // DF classes as bit sets indexed by DF (mod.rs:56-135).  Comparisons, so that the class
    // logic lives in lane masks on the scalar unit rather than in VALU arithmetic.
    const bool ap = ((0xFF310031u >> df) & 1u) != 0;        // 0,4,5,16,20,21,24..31: address/parity
    const bool d1718 = ((0x00060000u >> df) & 1u) != 0;     // clean iff residual == 0
    const bool d11 = df == 11u;                             // clean iff residual & 0xFFFF80 == 0
    const bool z = h == 0, z11 = (h & 0xFFFF80u) == 0;
    o.is_ap = ap;
    o.is_hit = (d1718 && z) || (d11 && z11);
    // DF17 and DF11 with IID 0 add their address; DF18 adds addr | 1 << 25, never matched
    o.learn = z && (d11 || df == 17u);
}

__device__ __forceinline__ uint32_t trial_addr(const Trial &t)  // message bits 8..31
{
    uint32_t addr = 0;
#pragma unroll
    for (int n = 8; n < 32; n++) addr |= ((t.f[n % 5] >> (n / 5)) & 1u) << (31 - n);
    return addr;
}

// One-launch pass: the record of a hit is built where the hit is found, by the wave that found it, from what the
// tile has in LDS -- the trial's five bit fields are all 112 sliced bits, the 33 magnitudes behind j + 19 are in
// s.mag (demod_2400.rs:158-182, 191-196) -- and goes straight into the pass's records in mapped host memory.  No
// hit list, no second pass over the samples: the record builder of the three-launch passes (k_records) reads the
// IQ behind every hit again, which for a slot read in place is a round trip over the link per hit, queued behind
// the 512 KB the passes beside this one are pulling through it.
// `m`: the lanes that hold a hit (a ballot); f / cslot / entry / residual: that lane's trial.
// One hit (wave-uniform arguments: its five fields, its LDS slot, its entry, its residual), built by the whole wave.
__device__ __forceinline__ void emit_record(const ScanParams &p, const FastLds &s, const uint32_t (&ff)[5], uint32_t cs, uint64_t me,
                                            uint32_t mcrc, int lane)
{
    const uint32_t k0 = __umul24((uint32_t)lane, 13108u) >> 16, r0 = (uint32_t)lane - 5u * k0;          // lane / 5, % 5
    const uint32_t k1 = __umul24((uint32_t)lane + 64u, 13108u) >> 16, r1 = (uint32_t)lane + 64u - 5u * k1;
    // lane n: message bits n and n + 64 (bit 5k + r of the message is bit k of field r)
    const uint32_t w0 = r0 == 0 ? ff[0] : r0 == 1 ? ff[1] : r0 == 2 ? ff[2] : r0 == 3 ? ff[3] : ff[4];
    const uint32_t w1 = r1 == 0 ? ff[0] : r1 == 1 ? ff[1] : r1 == 2 ? ff[2] : r1 == 3 ? ff[3] : ff[4];
    const unsigned long long h0 = __brevll(__ballot(((w0 >> k0) & 1u) != 0));
    const unsigned long long h1 = __brevll(__ballot(lane < 48 && ((w1 >> k1) & 1u) != 0));
    unsigned long long pw = lane < 33 ? (unsigned long long)s.mag[cs + 19u + (uint32_t)lane] : 0ull;
    pw *= pw;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pw += __shfl_xor(pw, off);
    if (lane == 0) {
        const uint32_t j = entry_j(me), tp = entry_tp(me);
        TrialRecord r;
        r.power = pw | ((unsigned long long)mcrc << 40);  // pw < 2^38
        r.chunk = (uint32_t)entry_chunk(me);
        r.j_tp = j | (tp << 24);
#pragma unroll
        for (int k = 0; k < 8; k++) r.msg[k] = (uint8_t)(h0 >> (56 - 8 * k));
#pragma unroll
        for (int k = 0; k < 6; k++) r.msg[8 + k] = (uint8_t)(h1 >> (56 - 8 * k));
        // (as k_records: what this DF will ask the filter about, hashed for the host replay)
        const uint32_t df = (uint32_t)(h0 >> 59);
        const bool ap = ((0xFF310031u >> df) & 1u) != 0;
        const uint32_t addr = (uint32_t)(h0 >> 32) & 0xFFFFFFu;
        r.pad = (uint16_t)(3u | (icao_hash_dev(ap ? mcrc : addr) << 4));
        const uint32_t idx = atomicAdd(&p.ctr->n_rec, 1u);
        if (idx < p.hits_cap) {
            u32x4_t q[2];
            __builtin_memcpy(q, &r, sizeof(r));
            host_store128((char *)(p.fused_rec + idx), q[0]);
            host_store128((char *)(p.fused_rec + idx) + 16, q[1]);
            unsigned long long w[4];
            __builtin_memcpy(w, &r, sizeof(r));
            atomicAdd((unsigned long long *)p.ctr->rec_sum, w[0] + w[1] + w[2] + w[3]);
        } else {
            atomicOr(&p.ctr->overflow, 1u);
        }
    }
}

// `m`: the lanes that hold a hit (a ballot); f / cslot / entry / residual: that lane's trial.
__device__ __forceinline__ void emit_records(const ScanParams &p, const FastLds &s, unsigned long long m, const uint32_t (&f)[5],
                                             uint32_t cslot, uint64_t entry, uint32_t residual, int lane)
{
    while (m) {
        const int from = (int)__builtin_ctzll(m);   // (wave-uniform: m is)
        m &= m - 1ull;
        uint32_t ff[5];
#pragma unroll
        for (int r = 0; r < 5; r++) ff[r] = (uint32_t)__builtin_amdgcn_readlane((int)f[r], from);
        const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cslot, from);
        const uint32_t mcrc = (uint32_t)__builtin_amdgcn_readlane((int)residual, from);
        const uint64_t me = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(entry >> 32), from) << 32 |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)entry, from);
        emit_record(p, s, ff, cs, me, mcrc, lane);
    }
}

// a self-validating trial: staged in LDS, flushed to the hit list at the end of the tile
template <bool SHARED>
__device__ __forceinline__ void put_hit_fields(const ScanParams &p, size_t place, const uint32_t (&f)[5])
{
    uint32_t *w = p.hit_fields + place * kHitFieldWords;
#pragma unroll
    for (int r = 0; r < 5; r++) st_shared<SHARED>(&w[r], f[r]);
    st_shared<SHARED>(&w[5], 1u);
}

// LDS only the instantiations that hand hit fields over have (FIELDS: dense streams, whose record builder is
// what the scan's vector pipes share their cycles with, and one-launch passes).  The sparse stream's
// instantiation stays at 102 registers and 31.2 KB: two registers more and a fifth workgroup no longer
// fits a CU while consecutive launches overlap (measured: +3 % on the pipelined step).
// A one-launch pass (FUSED) stages kFusedExtraHits more per tile: its workgroups build the records of what is
// staged themselves (emit_record), what is not goes through the hit list to ONE workgroup at the end -- a buffer
// packed with frames (~94 hits a tile) took 0.22 ms that way.
constexpr int kFusedExtraHits = 96;
template <bool FIELDS, bool FUSED = false>
struct HitFieldLds {
    uint32_t f[kHitCap][5];   // the five bit-class fields of each staged hit (ScanParams::hit_fields)
};
template <>
struct HitFieldLds<true, true> {
    uint32_t f[kHitCap][5];
    uint64_t xhit[kFusedExtraHits];      // staged hits kHitCap .. kHitCap + kFusedExtraHits - 1 of the tile
    uint32_t xf[kFusedExtraHits][5];
};
template <bool FUSED>
struct HitFieldLds<false, FUSED> {
};

template <bool FUSED, bool FIELDS>
__device__ __forceinline__ void stage_hit(const ScanParams &p, FastLds &s, HitFieldLds<FIELDS, FUSED> &hf, bool is_hit, uint64_t entry,
                                          int lane, uint32_t par, const uint32_t (&f)[5])
{
    const unsigned long long mh = __ballot(is_hit);
    if (!mh) return;
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&s.nhit[par], (uint32_t)__popcll(mh));
    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + mask_rank(mh);
    if (is_hit) {
        if (at < (uint32_t)kHitCap) {
            s.hit[at] = entry;
            if constexpr (FIELDS) {
#pragma unroll
                for (int r = 0; r < 5; r++) hf.f[at][r] = f[r];
            }
        } else if (FUSED && FIELDS && at < (uint32_t)(kHitCap + kFusedExtraHits)) {
            if constexpr (FUSED && FIELDS) {
                hf.xhit[at - kHitCap] = entry;
#pragma unroll
                for (int r = 0; r < 5; r++) hf.xf[at - kHitCap][r] = f[r];
            }
        } else {  // more hits in one tile than the staging holds: one by one
            const uint32_t gi = atomicAdd(&p.ctr->n_hits, 1u);
            if (p.order_cnt) {  // dense stream: into the buffer's bucket, its tile's part of it (adsb_device.h: order_tmp)
                const uint32_t c = (uint32_t)entry_chunk(entry), tl = entry_j(entry) / (uint32_t)kTile;
                const uint32_t k = atomicAdd(&p.order_cnt[c * kTilesPerChunk + tl], 1u);
                if (k < kTileBucket) {
                    const size_t at = (size_t)c * kOrderBucket + tl * kTileBucket + k;
                    p.order_tmp[at] = entry;
                    if constexpr (FIELDS) put_hit_fields<FUSED>(p, at, f);
                } else {
                    atomicOr(&p.ctr->overflow, 1u);
                }
            } else if (gi < p.hits_cap) {
                st_shared<FUSED>(&p.hits[gi], entry);
                if constexpr (FIELDS) put_hit_fields<FUSED>(p, gi, f);
            } else {
                atomicOr(&p.ctr->overflow, 1u);
            }
        }
    }
}

// IQ of one tile, as each thread holds it between the load and the magnitude pass:
// 8 aligned dwordx4 = 32 samples per thread, 8080 per workgroup.
constexpr int kLoadsPerThread = (kAllocSlots / 4 + kThreads - 1) / kThreads;  // 8
#ifndef ADSB_TRICKLE
#define ADSB_TRICKLE 3
#endif
constexpr int kTrickle = ADSB_TRICKLE;   // loads in flight per thread while a tile is read in place from host memory

struct TileRef {
    uint32_t chunk;
    int tile, len, jbase;
};

template <bool FROM_MAG>
__device__ __forceinline__ TileRef tile_ref(const ScanParams &p, uint32_t t)
{
    TileRef r;
    r.chunk = t / kTilesPerChunk;
    r.tile = (int)(t % kTilesPerChunk);
    // a caller-supplied MagnitudeBuffer is one buffer: n_samples is its `length`
    r.len = FROM_MAG ? (int)p.n_samples : chunk_len(p.n_samples, r.chunk);
    r.jbase = r.tile * kTile;
    return r;
}

// The tile's IQ through a buffer resource that spans exactly this chunk's samples: the
// hardware range check returns zero for every dword outside [0, len) -- the 326-sample
// lead-in before the chunk (negative offsets wrap to huge unsigned ones), the zero tail
// and the ragged end of a short last chunk -- so the eight dwordx4 loads are issued
// back to back with no branch and no wait between them.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool FROM_MAG>
__device__ __forceinline__ void load_tile_iq(const ScanParams &p, const TileRef &r, int tid,
                                             uint4 (&pre)[kLoadsPerThread])
{
    if (FROM_MAG) {
        // caller-supplied magnitudes (adsb_demodulate2400): MagnitudeBuffer.data as handed in,
        // lead-in included; 4 u16 per load, zero outside [0, kMagDataLen) by the range check
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)p.src, 0, kMagDataLen * 2, 0x00020000);
        const int d0 = r.jbase - kPad;  // data index of slot 0
#pragma unroll
        for (int i = 0; i < kLoadsPerThread; i++) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            // (opaque: a "+ 2048 i" folded into the instruction's immediate offset is added to a
            // negative register offset without wrapping, i.e. out of range -- adsb_aux.hip: k_records)
            int off = (d0 + 4 * (tid + i * kThreads)) * 2;
            asm volatile("" : "+v"(off));
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
            pre[i] = make_uint4(v.x, v.y, 0u, 0u);
        }
        return;
    }
    const uint32_t *iq = (const uint32_t *)p.src + r.chunk * (uint64_t)kChunkSamples;
    // carry-over mode: the resource starts kCarrySamples before the buffer when those samples
    // exist in src, so the lead-in is simply in range (the reference's mode: it is not)
    const bool lead = p.carry != nullptr && (r.chunk > 0 || p.lead_from_src);
    const int shift = lead ? kCarrySamples : 0;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(iq - shift), 0, (r.len + shift) * 4, 0x00020000);
    // (offsets opaque to the compiler, as in the branch above: a constant folded into the
    // instruction's immediate offset would break the range check for the lanes before sample 0)
    const int k0 = r.jbase - kPad - kLead + shift;  // IQ sample index of slot 0 (multiple of 4)
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; i++) {
        int off = (k0 + 4 * (tid + i * kThreads)) * 4;
        asm volatile("" : "+v"(off));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        pre[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
    if (p.carry != nullptr && !lead && r.tile == 0) {
        // first buffer of a call: its lead-in is the end of the previous call (out of range of
        // one resource = zero from it, so the two loads just OR together)
        const __amdgpu_buffer_rsrc_t crsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)p.carry, 0, kCarrySamples * 4, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(crsrc, (k0 + kCarrySamples + 4 * tid) * 4, 0, 0);
        pre[0].x |= v.x;
        pre[0].y |= v.y;
        pre[0].z |= v.z;
        pre[0].w |= v.w;
    }
}

// Load i of the eight alone (a tile read in place from host memory, no carry-over: k_scan_fast trickles those).
__device__ __forceinline__ void load_tile_iq_one(const ScanParams &p, const TileRef &r, int tid, uint4 (&pre)[kLoadsPerThread], int i)
{
    const uint32_t *iq = (const uint32_t *)p.src + r.chunk * (uint64_t)kChunkSamples;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)iq, 0, r.len * 4, 0x00020000);
    const int k0 = r.jbase - kPad - kLead;
    int off = (k0 + 4 * (tid + i * kThreads)) * 4;
    asm volatile("" : "+v"(off));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    pre[i] = make_uint4(v.x, v.y, v.z, v.w);
}

// A lane writes its own matches (bits of m; the branch 0..4 of each from the three code planes)
// into the wave's pattern region from index `at` on: slot | branch << 13.
__device__ __forceinline__ void compact_matches(uint32_t m, uint32_t code0, uint32_t code1, uint32_t code2,
                                                uint32_t slot0, uint16_t *wpat, uint32_t at)
{
    while (m) {
        const uint32_t bit = (uint32_t)__ffs(m) - 1u;
        m &= m - 1;
        // three single-bit extracts and two shift-ors (left to itself the compiler shifts and
        // masks each plane separately: eight ops)
        uint32_t k = __builtin_amdgcn_ubfe(code0, bit, 1u);
        k |= __builtin_amdgcn_ubfe(code1, bit, 1u) << 1;
        k |= __builtin_amdgcn_ubfe(code2, bit, 1u) << 2;
        wpat[at++] = (uint16_t)((slot0 + 12u * bit) | (k << 13));
    }
}

// One 64-lane pass of the gates: `ent` is this lane's pattern match (valid lanes only count),
// passing positions are appended to the wave's candidate region.
template <bool SELFTEST>
__device__ __forceinline__ void gate_pass(const ScanParams &p, const FastLds &s, uint32_t ent, bool valid,
                                          uint16_t *wcand, uint32_t &ncand_w, int jbase, uint32_t chunk)
{
    const bool pass = (gate_eval(s.mag, ent) & (uint32_t)valid) != 0;
    const unsigned long long mask = __ballot(pass);
    if (mask) {
        if (pass) wcand[mask_rank(mask, ncand_w)] = (uint16_t)(ent & 0x1FFFu);
        ncand_w += (uint32_t)__popcll(mask);
    }
    if (SELFTEST && p.cand_out && valid) {
        // self-test instantiation only (adsb_selftest_stage_lists / _gate_stages): every pattern match
        // that is a preamble by the reference's own sequence of tests goes out as
        // chunk << 32 | production gate verdict << 30 | stage (1..3, preamble_stage) << 28 | j
        const int stage = preamble_stage(s.mag + (ent & 0x1FFFu));
        if (stage | (int)pass) {
            const uint32_t at = atomicAdd(p.cand_count, 1u);
            if (at < p.cand_cap)
                p.cand_out[at] = (uint64_t)chunk << 32 | (uint64_t)(pass ? 1u : 0u) << 30 | (uint64_t)stage << 28 |
                                 (uint32_t)(jbase - kPad + (int)(ent & 0x1FFFu));
        }
    }
}

// One 64-lane pass of the trials: lane = (candidate entry ce, try_phase 4 + tpi).
template <bool FUSED, bool FIELDS>
__device__ __forceinline__ void trial_pass(const ScanParams &p, FastLds &s, HitFieldLds<FIELDS, FUSED> &hf, uint32_t ce, uint32_t tpi,
                                           bool live, int jbase, uint32_t chunk, uint64_t *seg, uint32_t seg_cap,
                                           uint32_t &ap_count, int lane, uint32_t par)
{
    Trial tr;
    trial_eval(s, ce, tpi, tr);
    const bool is_ap = live && tr.is_ap, is_hit = live && tr.is_hit, learn = live && tr.learn;
    // entry = value24 | code << 24 | j << 28 | chunk << 45   (adsb_device.h)
    const uint32_t j = (uint32_t)(jbase - kPad) + tr.cslot;
    const uint64_t entry = ((uint64_t)((j >> 4) | (chunk << 13)) << 32) | (tr.h | (tr.code << 24) | (j << 28));
    // AP entries: straight into this wave's own segment of the list (no atomic, no shared
    // counter: the fill count is a wave-uniform register)
    const unsigned long long ma = __ballot(is_ap);
    if (ma) {
        const uint32_t mine = mask_rank(ma, ap_count);
        if (is_ap && mine < seg_cap) st_shared<FUSED>(&seg[mine], entry);   // (the last workgroup may look at it again)
        ap_count += (uint32_t)__popcll(ma);
    }
    if (__ballot(is_hit)) stage_hit<FUSED, FIELDS>(p, s, hf, is_hit, entry, lane, par, tr.f);  // rare
    if (__ballot(learn)) {  // rare: the host replay will add this address to the filter
        // (one-launch pass: an address bit that was clear until now means trials this pass has already
        // matched may have missed it -- its last workgroup then matches the lists once more)
        if constexpr (FUSED) {
            if (learn) {
                const uint32_t addr = trial_addr(tr);
                if (bitmap_set(p.bitmap, p.bitmap_lg, addr)) {
                    const uint32_t k = atomicAdd(&p.ctr->learned_new, 1u);
                    if (k < (uint32_t)kNewAddrCap) st_shared<true>(&p.ctr->new_addr[k], addr);
                }
            }
        } else {
            if (learn) {
                const uint32_t addr = trial_addr(tr);
                const bool bit_was_clear = bitmap_set(p.bitmap, p.bitmap_lg, addr);
                // (a shard of an adsb_multi lists the addresses its trials can add, each once: ScanParams::fresh)
                if (p.fresh) {
                    const uint32_t bit = 1u << (addr & 31u);
#ifdef ADSB_FRESH_BY_BITMAP   // the first version, kept to show the soak finds its hole (adsb_device.h: ScanParams::fresh)
                    if (bit_was_clear) {
#else
                    (void)bit_was_clear;
                    if ((atomicOr(&p.fresh_seen[addr >> 5], bit) & bit) == 0) {
#endif
                        const uint32_t k = atomicAdd(&p.ctr->n_fresh, 1u);
                        atomicAdd(&p.ctr->fresh_sum, addr);
                        if (k < p.fresh_cap) p.fresh[k] = addr;
                    }
                }
            }
        }
    }
}

// t5 = 5 c + tpi for t5 < 3277 (24-bit multiply, not the slow 32-bit one)
__device__ __forceinline__ void split5(uint32_t t5, uint32_t &c, uint32_t &tpi)
{
    c = __umul24(t5, 13108u) >> 16;
    tpi = t5 - __umul24(5u, c);
}

// profiling aids, compiled in with -DADSB_KERNEL_ACCT only (they cost registers):
// wave 0 of a few workgroups stamps the shader clock at phase boundaries
#ifdef ADSB_KERNEL_ACCT
#define STAMP(slot)                                                                          \
    do {                                                                                     \
        if (p.timeline && !ADSB_STOP_AT(p, 100) && tid == 0 && (blockIdx.x & 127) == 0 && iter < 8)                   \
            p.timeline[((blockIdx.x >> 7) * 8 + iter) * 8 + (slot)] = (unsigned long long)clock64(); \
    } while (0)

// (ADSB_DEBUG_STOP=100 ADSB_TIMELINE=2): every wave totals the clocks it spends in each
// phase and waiting at each workgroup barrier
#define ACCT(k)                                                   \
    do {                                                          \
        if (acct) {                                               \
            const unsigned long long now_ = clock64();            \
            acc_t[k] += now_ - acc_last;                          \
            acc_last = now_;                                      \
        }                                                         \
    } while (0)
// (ADSB_TIMELINE=3): where a one-launch pass spends its time, on the 100 MHz wall clock: entry, tables in
// LDS, tiles done, own match done, counted in (from here on: the last workgroup), second look done, end
#define FSTAMP(k)                                                                                  \
    do {                                                                                           \
        if (FUSED && p.timeline && tid == 0) {                                                      \
            p.timeline[448 + (k)] = (unsigned long long)wall_clock64();                            \
            if ((k) < 8) p.timeline[1024 + ((p.seq & 15u) * 32u + min(blockIdx.x, 31u)) * 8u + (k)] = (unsigned long long)wall_clock64(); \
        }                                                                                          \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define ACCT(k) do {} while (0)
#define FSTAMP(k) do {} while (0)
#endif

// Persistent: the grid is what is resident at once and each workgroup walks tiles
// t = block, block + grid, ...  The IQ of the next tile is loaded into registers right
// after the magnitudes of the current one are in LDS, so HBM latency hides behind P2..P5.
// LDS only the one-launch instantiation has: the x^56 multiplier (adsb_tables.h) for its own match
template <bool FUSED>
struct FusedLds {
    uint32_t x56[3 * 256];
    uint32_t bits[168];   // per-bit residual constants (adsb_tables.h: build_bit_residuals) for the record builder
    uint32_t is_last;
};
template <>
struct FusedLds<false> {
};

// One address/parity entry against the bitmap (k_match's test: adsb_aux.hip); a match goes to the hit
// list and the entry is marked (code 15) so that a second look does not report it twice.
__device__ __forceinline__ bool fused_match_entry(const ScanParams &p, const uint32_t *x56, uint64_t *slot, uint64_t e)
{
    const uint32_t code = entry_code(e);
    if (code == 15u) return false;
    uint32_t c = entry_value(e);
    if (code >= 5u && code < 10u) c = gf_apply(x56, c);
    // (agent scope: bits other workgroups of this launch have set, not a line this CU's cache holds)
    const uint32_t at = bitmap_index(c, p.bitmap_lg);
    const uint32_t w = __hip_atomic_load(&p.bitmap[at >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((w >> (at & 31)) & 1u) {
        const uint32_t idx = atomicAdd(&p.ctr->n_hits, 1u);
        if (idx < p.hits_cap) {
            st_shared<true>(&p.hits[idx], e);
            if (p.hit_fields) st_shared<true>(&p.hit_fields[(size_t)idx * kHitFieldWords + 5], 0u);
        } else {
            atomicOr(&p.ctr->overflow, 1u);
        }
        st_shared<true>(slot, e | (15ull << 24));
        return true;
    }
    return false;
}

template <bool FROM_MAG, bool SELFTEST = false, bool FUSED = false, bool FIELDS = false>
__global__ __launch_bounds__(kThreads, FUSED ? 2 : kWavesPerSimd) ADSB_NO_UNALIGNED void k_scan_fast(ScanParams p)
{
    __shared__ FastLds s;
    __shared__ FusedLds<FUSED> fs;
    __shared__ HitFieldLds<FIELDS, FUSED> hf;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t n_tiles = p.n_chunks * kTilesPerChunk;
    FSTAMP(0);
    if constexpr (FUSED) {
        if (blockIdx.x == 0 && tid == 0) {
            const unsigned long long t0 = (unsigned long long)wall_clock64();
            st_shared<true>(&p.ctr->t_start[0], (uint32_t)t0);
            st_shared<true>(&p.ctr->t_start[1], (uint32_t)(t0 >> 32));
        }
        for (int i = tid; i < 3 * 256; i += kThreads) fs.x56[i] = p.tables[kTabX56 * 256 + i];
        if (tid < 168) fs.bits[tid] = p.tables[kTabBitsOff + tid];
        // an icao_flush retired a bitmap: every workgroup clears its share (k_records does it for the
        // passes of three launches)
        if (p.clean_bitmap) bitmap_clear(p.clean_bitmap, p.bitmap_lg, blockIdx.x * kThreads + tid, gridDim.x * kThreads);
        // ... or (a context for passes of a few buffers: folded bitmaps) the pass starts on the NEXT bitmap of the
        // rotation and clears it itself: the first workgroup does -- 64 KB written through, acknowledged, then the
        // flag -- and every other one checks the flag behind its first tile's loads, long before it first sets or
        // tests a bit (bitmap_wait below).  Nobody else is using that bitmap: there is one more than passes in flight.
        if (p.bitmap_fresh && blockIdx.x == 0) {
            const uint32_t words = bitmap_alloc_words(p.bitmap_lg), bits = bitmap_words(p.bitmap_lg);
            for (uint32_t v = (uint32_t)tid; v < words; v += kThreads)
                st_shared<true>(&p.bitmap[v], (v == 0u || v == bits) ? 1u : 0u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_shared<true>(&p.ctr->bitmap_ready, 1u);
        }
    }

    // ---------------------------------------------------------------- P0 once per workgroup
    for (int i = tid; i < 3 * 256; i += kThreads) s.tab[i] = p.tables[kTabF * 256 + i];
    for (int i = tid; i < 316; i += kThreads) {
        const uint32_t v = p.tables[kTabR16Off + i];
        if (i < 16)
            s.r16[i] = v;
        else  // plane row byte offset -> its LDS address, so that the trial stage adds nothing
            s.field[i - 16] = v + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)s.plane;
    }
    if (tid < kPlanes) s.plane[tid * kPlaneDw + kPlaneDw - 1] = 0;  // read slack
    if (tid < 2) s.nhit[tid] = 0;

    const uint32_t seg_cap = p.seg_cap;
    const uint32_t my_seg = blockIdx.x * kWaves + (uint32_t)(tid >> 6);
    uint64_t *const seg = p.ap + (uint64_t)my_seg * seg_cap;  // this wave's own AP segment
    uint32_t ap_count = 0, cand_count = 0;  // wave-uniform running totals of this wave

    uint4 pre[kLoadsPerThread];
    // Which tiles this workgroup walks.  Blocks b, b + 8, b + 16, ... run on the same XCD (observed
    // placement, used for speed only), so each XCD gets one contiguous eighth of the tiles and its
    // blocks walk it side by side: the 368 samples two neighbouring tiles share are then read from
    // HBM once and found in that XCD's L2 by the neighbour.  Any other grid: plain round robin.
    uint32_t t_first = blockIdx.x, t_end = n_tiles, t_stride = gridDim.x;
    // (a one-launch pass: tile = block -- its workgroups publish and wait for each other in tile order by their
    // block index, and a pass of a few buffers has nothing to gain from the placement)
    if (!FUSED && (gridDim.x & 7u) == 0 && n_tiles >= gridDim.x) {
        const uint32_t x = blockIdx.x & 7u;
        t_first = ((x * n_tiles) >> 3) + (blockIdx.x >> 3);
        t_end = ((x + 1u) * n_tiles) >> 3;
        t_stride = gridDim.x >> 3;
    }
    if constexpr (FUSED && !FROM_MAG) {
        // The samples may still be on their way into the pinned buffer (adsb_demod_iq copies them there while this
        // launch travels to the device): wait until the host has copied what this tile reads.  One thread polls
        // the host's word over the link (~1.5 us a poll; the whole copy is ~10 us); bounded -- a host that never
        // finishes is reported as an overflow, and the pass is redone once the call has all its samples.
        if (p.src_ready != nullptr && t_first < t_end) {
            const TileRef r0 = tile_ref<FROM_MAG>(p, t_first);
            const long long last = min((long long)r0.len, (long long)r0.jbase - kPad - kLead + kAllocSlots);
            const unsigned long long need = (unsigned long long)r0.chunk * kChunkSamples + (unsigned long long)max(last, 0ll);
            if (tid == 0) {
                uint32_t polls = 0;
                while (__hip_atomic_load(p.src_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
                    if (++polls > 40000u) {   // tens of milliseconds
                        atomicOr(&p.ctr->overflow, 32u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        }
    }
    // A tile read in place from host memory keeps kTrickle of its eight loads in flight, not all of them: every
    // request waits its turn in the same L2 queues as everything else on the device, and four passes side by
    // side with 512 KB each outstanding put ~36 us of link time in front of any other miss -- the tables of a
    // pass that is just starting, the address bits and list entries of one that is matching (seen: 17 us for
    // the tables instead of 2).  Two loads per thread in flight already fill the link (tools/pcie_read_probe.hip); three measured best.
    bool trickle = false;
    if constexpr (FUSED && !FROM_MAG) trickle = p.src_host != 0u && p.carry == nullptr;
    if (t_first < t_end) {
        if (trickle) {
#pragma unroll
            for (int i = 0; i < kTrickle; i++) load_tile_iq_one(p, tile_ref<FROM_MAG>(p, t_first), tid, pre, i);
        } else {
            load_tile_iq<FROM_MAG>(p, tile_ref<FROM_MAG>(p, t_first), tid, pre);
        }
    }

    // Workgroups that share a CU start a fraction of a tile period apart, so that the
    // VALU-dense phases of one overlap the latency-bound phases of the others instead of
    // all of them marching through the same phase together.
#ifdef ADSB_TUNING
    if (p.stagger_ticks) {
        const uint32_t k = (blockIdx.x * 4u) / gridDim.x;  // 0..3: which quarter of the grid
        const unsigned long long until = clock64() + (unsigned long long)k * p.stagger_ticks;
        while ((unsigned long long)clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
#endif

#ifdef ADSB_KERNEL_ACCT
    const bool acct = ADSB_STOP_AT(p, 100) && p.timeline != nullptr;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc_last = acct ? clock64() : 0;
#endif

    const bool late_prio = ADSB_PRIO_LATE != 0 && (ADSB_PRIO_LATE_DENSE != 0 || p.order_cnt == nullptr);
    FSTAMP(1);
    uint32_t iter = 0;
    for (uint32_t t = t_first; t < t_end; t += t_stride, iter++) {
    const TileRef cur = tile_ref<FROM_MAG>(p, t);
    STAMP(0);
    const uint32_t chunk = cur.chunk;
    const int len = cur.len, jbase = cur.jbase;
    const int jn = min(kTile, len - jbase);  // <= 0 for tiles past the end of a short chunk

    const uint32_t par = iter & 1u;  // which copy of the tile counters this tile uses

    // ---------------------------------------------------------------- P1 magnitudes
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; i++) {
        if constexpr (FUSED && !FROM_MAG) {
            if (trickle && i + kTrickle < kLoadsPerThread) {   // load i has arrived: the next one may go
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTrickle - 1) : "memory");
                load_tile_iq_one(p, cur, tid, pre, i + kTrickle);
            }
        }
        const int g = tid + i * kThreads;
        if (g < kAllocSlots / 4) *(uint2 *)(s.mag + 4 * g) = FROM_MAG ? make_uint2(pre[i].x, pre[i].y) : mag4_of(pre[i]);
    }
    if (t + t_stride < t_end) load_tile_iq<FROM_MAG>(p, tile_ref<FROM_MAG>(p, t + t_stride), tid, pre);
    ACCT(0);
    if constexpr (FUSED) {
        // (bitmap_wait) behind an icao_flush the first workgroup clears the pass's bitmap: it has, by the time this
        // workgroup's first tile has arrived -- one look, bounded like the other waits of a one-launch pass (a first
        // workgroup that has not been given a CU yet: the pass is reported as overflowed and redone)
        if (p.bitmap_fresh && iter == 0 && blockIdx.x != 0 && tid == 0) {
            uint32_t polls = 0;
            while (ld_shared<true>(&p.ctr->bitmap_ready) == 0u) {
                if (++polls > 20000u) {
                    atomicOr(&p.ctr->overflow, 64u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    lds_barrier();
    // every thread is past the previous tile's epilogue: its counters can be zeroed for the next
    // tile (this tile counts in the other copy), so the tile needs no barrier at its end
    if (tid == 0) s.nhit[par ^ 1u] = 0;
    ACCT(1);
    STAMP(1);
    if (jn <= 0 || ADSB_STOP_AT(p, 1)) {
        lds_barrier();
        continue;
    }

    // ---------------------------------------------------------------- P2 sign planes
    // item = (g, kw): residues 4g..4g+3, plane bits k = 8kw..8kw+7, i.e. samples
    // 12k + 4g + {0..3} (+3 of look-ahead).  Bit k of plane (kind, r) is the sign taken
    // at sample 12k + r.  Walking k downwards leaves bit (k & 7) of the byte = k.
    for (int item = tid; item < kItems2; item += kThreads) {
        constexpr int R = kResPerItem, G = 12 / R;
        const int g = item % G, kw = item / G;
        const uint16_t *base = s.mag + 96 * kw + R * g;  // 4-byte aligned (R even)
        uint32_t acc[6][R];
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
            for (int r = 0; r < R; r++) acc[q][r] = 0;
#pragma unroll
        for (int kk = 8; kk >= 0; --kk) {
            // m[0 .. R+2]: the R samples of this lane and three of look-ahead.  Explicit
            // 8-byte reads (the address is 8-byte aligned, no more): left to itself the
            // compiler merges dword reads into one 16-byte read, and an LDS access off its
            // natural alignment is replayed at 64 cycles (SQ_LDS_UNALIGNED_STALL).
            int m[R + 4];
            if constexpr (R == 4) {
                const uint2 lo = *(const uint2 *)(base + 12 * kk);
                const uint2 hi = *(const uint2 *)(base + 12 * kk + 4);
                m[0] = (int)(lo.x & 0xFFFFu);
                m[1] = (int)(lo.x >> 16);
                m[2] = (int)(lo.y & 0xFFFFu);
                m[3] = (int)(lo.y >> 16);
                m[4] = (int)(hi.x & 0xFFFFu);
                m[5] = (int)(hi.x >> 16);
                m[6] = (int)(hi.y & 0xFFFFu);
                m[7] = (int)(hi.y >> 16);
            } else {  // R == 2: 4-byte aligned, three separate dword reads
                typedef const volatile __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
                lds_u32_ptr src = (lds_u32_ptr)(base + 12 * kk);
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const uint32_t w = src[d];
                    m[2 * d] = (int)(w & 0xFFFFu);
                    m[2 * d + 1] = (int)(w >> 16);
                }
            }
            int e[R + 2];  // first differences m[s+1] - m[s]
#pragma unroll
            for (int i = 0; i < R + 2; i++) e[i] = m[i + 1] - m[i];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int ea = e[r], eb = e[r + 1], ec = e[r + 2];
                if (kk < 8) {
                    // slicer value D(ph) at this sample (demod_2400.rs:72-83), negated so that
                    // "D > 0" is the sign bit: with a = m0-m1 = -e0, b = m1-m2 = -e1, c = m2-m3:
                    //   D0 = 5a+2b  D1 = 4a+3b  D2 = 3a+4b  D3 = 2a+5b  D4 = a+6b+c
                    const int n0 = __mul24(ea, 5) + (eb + eb);
                    const int u = eb - ea;
                    const int n1 = n0 + u, n2 = n1 + u, n3 = n2 + u;
                    const int n4 = n3 + u + ec;  // a + 6b + c = (2a + 5b) + (b - a) + c: one add3
                    acc[0][r] = push_sign(acc[0][r], n0);
                    acc[1][r] = push_sign(acc[1][r], n1);
                    acc[2][r] = push_sign(acc[2][r], n2);
                    acc[3][r] = push_sign(acc[3][r], n3);
                    acc[4][r] = push_sign(acc[4][r], n4);
                }
                // kk == 8 is one plane bit beyond the byte, for GT only: it completes the "advanced
                // by one bit" copies that P3 addresses as residues 12..23.  There is no "<" plane:
                // P3 works with "<=" (the complement of ">") and the gates re-check strictness.
                acc[5][r] = push_sign(acc[5][r], ea);   // GT: m[s] > m[s+1]
            }
        }
        uint8_t *pb = (uint8_t *)s.plane;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int res = R * g + r;
#pragma unroll
            for (int q = 0; q < 5; q++) pb[(q * 12 + res) * (kPlaneDw * 4) + kw] = (uint8_t)acc[q][r];
            // 9 bits: k = 8kw .. 8kw+8.  Residue res holds bits 0..7, residue res+12 (the
            // same plane advanced one bit) holds bits 1..8.
            pb[(kPlaneGT + res) * (kPlaneDw * 4) + kw] = (uint8_t)acc[5][r];
            pb[(kPlaneGT + 12 + res) * (kPlaneDw * 4) + kw] = (uint8_t)(acc[5][r] >> 1);
        }
    }
    ACCT(2);
    lds_barrier();
    ACCT(3);
    STAMP(2);
    if (ADSB_STOP_AT(p, 2)) continue;
    // The wave-private stages are chains of LDS round trips with few instructions between them:
    // at a raised issue priority they get through their dependent steps without queueing behind the
    // other workgroups' P1 / P2 on the same SIMD, which have instructions to spare for every slot
    // those chains leave (measured: pipelined -2 %, a launch on its own 105 -> 100 us; levels 1, 2
    // and 3 alike).  Not on dense streams: there the tail kernels beside the scan are the ones that
    // must not wait (adsb_aux.hip: TAIL_PRIO), and the step got 3 % longer.
    if (late_prio) __builtin_amdgcn_s_setprio(ADSB_PRIO_LATE);

    // ================================================================ P3..P5, wave-private
    // From here to the end of the tile every wave works alone on the positions of its own
    // P3 items: matches, candidates and trials stay in the wave's own LDS regions, so there
    // is no workgroup barrier and no shared counter between the stages, the waves of a
    // workgroup drift apart, and their latency-bound stages overlap the VALU-dense ones of
    // the others.  Nothing here can overflow: a wave with more matches than its region
    // holds takes them in rounds of a few plane bits, and candidates are flushed through
    // the trial stage whenever their region fills.
    {
    const int wave = tid >> 6;
    uint16_t *const wpat = s.pat + wave * kPatPerWave;
    uint16_t *const wcand = s.cand + wave * kCandPerWave;

    // ---------------------------------------------------------------- P3 preamble patterns
    // item = (res, w): the 32 positions with slot = 12*(32w + bit) + res.
    // (with 512 threads an item is half a dword, so that all eight waves own positions)
    constexpr int kHalves = kThreads / 256;
    uint32_t b[5] = {0u, 0u, 0u, 0u, 0u};
    const int ptid = tid % 256, phalf = tid / 256;
    const int pres = ptid % 12, pw = ptid / 12;
    if (ptid < kItems3) {
        const int res = pres, w = pw;
        const uint32_t *GT = s.plane + (kPlaneGT + res) * kPlaneDw + w;
#define GTO(o) GT[(o) * kPlaneDw]     // p[o] > p[o+1]
#define LTO(o) (~GT[(o) * kPlaneDw])  // p[o] <= p[o+1]: a superset of the reference's "<"; the
                                      // gates test the strict form of the branch they are handed
        // positions that are real j of this tile: kPad <= slot < kPad + jn
        // (x + 11) / 12 with a 24-bit multiply (x < 16384), not the 32-bit mul_hi the compiler would use
        const int kmin = (int)(__umul24((uint32_t)(kPad - res + 11), 10923u) >> 17),
                  kmax = (int)(__umul24((uint32_t)(kPad + jn - res + 11), 10923u) >> 17);
        uint32_t ok = lowmask(kmax - 32 * w) & ~lowmask(kmin - 32 * w);
        if (kHalves == 2) ok &= phalf ? 0xFFFF0000u : 0x0000FFFFu;
        ok &= LTO(0) & GTO(12);                               // demod_2400.rs:221
        const uint32_t A = GTO(1) & LTO(2);                   // p1>p2 p2<p3
        const uint32_t C = LTO(8) & GTO(9);                   // p8<p9 p9>p10
        const uint32_t E = GTO(4) & LTO(9) & GTO(10) & LTO(11);
        const uint32_t b1 = ok & A & GTO(3) & C & LTO(10);                    // :227
        const uint32_t b2 = ok & A & GTO(3) & C & LTO(11) & ~b1;              // :242
        const uint32_t b3 = ok & A & GTO(4) & LTO(8) & GTO(10) & LTO(11) & ~(b1 | b2);  // :262
        const uint32_t b4 = ok & GTO(1) & LTO(3) & E & ~(b1 | b2 | b3);       // :280
        const uint32_t b5 = ok & GTO(2) & LTO(3) & E & ~(b1 | b2 | b3 | b4);  // :300
#undef LTO
#undef GTO
        b[0] = b1;
        b[1] = b2;
        b[2] = b3;
        b[3] = b4;
        b[4] = b5;
    }
    const uint32_t slot0 = (uint32_t)(12 * 32 * pw + pres);
    const uint32_t any_all = b[0] | b[1] | b[2] | b[3] | b[4];
    // which branch matched, as three planes of a 3-bit code (0..4), so that compaction is one
    // loop over the union instead of one per branch
    const uint32_t code0 = b[1] | b[3], code1 = b[2] | b[3], code2 = b[4];
    const uint32_t cnt_all = (uint32_t)__popc(any_all);
    const uint32_t incl_all = wave_inclusive_scan(cnt_all);
    const uint32_t total_all = (uint32_t)__builtin_amdgcn_readlane((int)incl_all, 63);
    // all matches in one round when they fit the wave's region (the normal case: ~90 of
    // 256), else rounds of kRoundBits plane bits: at most 64 lanes x kRoundBits matches each
    const int nrounds = total_all <= (uint32_t)kPatPerWave ? 1 : 32 / kRoundBits;
    uint32_t ncand_w = 0;  // candidates waiting in wcand (wave-uniform)
    if (ADSB_STOP_AT(p, 3)) goto tile_end;  // profiling: patterns only

    // One loop, one copy of each stage: take the next round of matches when the previous one
    // is used up, run one 64-lane pass of the gates, and run the trials whenever the
    // candidate region could not take another pass's worth (or nothing else is left).
    int round = 0;
    uint32_t npat_w = 0, base = 0;
    bool in_round = false;
    for (;;) {
        if (round < nrounds) {
            if (!in_round) {
                // ---- compaction of this round's matches into wpat: exclusive scan of the lane
                // counts (DPP, no LDS traffic), then every lane writes its own
                const uint32_t rmask = nrounds == 1 ? 0xFFFFFFFFu
                                                    : (((1u << kRoundBits) - 1u) << (round * kRoundBits));
                uint32_t cnt = cnt_all, incl = incl_all;
                npat_w = total_all;
                if (nrounds != 1) {
                    cnt = (uint32_t)__popc(any_all & rmask);
                    incl = wave_inclusive_scan(cnt);
                    npat_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                if (npat_w == 0) {
                    round++;
                    continue;
                }
                compact_matches(any_all & rmask, code0, code1, code2, slot0, wpat, incl - cnt);
                wave_lds_fence();
                in_round = true;
                base = 0;
                if (ADSB_STOP_AT(p, 6)) {  // profiling: patterns + compaction, no gates
                    in_round = false;
                    round++;
                    continue;
                }
            }

            // ------------------------------------------------------------ P4 value gates
            // one lane per pattern match (gate_pass)
            {
                const uint32_t idx = base + (uint32_t)lane;
                gate_pass<SELFTEST>(p, s, wpat[min(idx, npat_w - 1u)], idx < npat_w, wcand, ncand_w, jbase, chunk);
            }
            base += 64;
            if (base >= npat_w) {
                in_round = false;
                round++;
            }
            // room for another pass of the gates and more of them to come: not yet
            if (round < nrounds && ncand_w + 64 <= (uint32_t)kCandPerWave) continue;
        }
        if (ncand_w == 0) {
            if (round >= nrounds) break;
            continue;
        }
        wave_lds_fence();
        if (ADSB_STOP_AT(p, 4)) {  // profiling: gates only
            ncand_w = 0;
            if (round >= nrounds) break;
            continue;
        }

        // ---------------------------------------------------------------- P5 trials
        // lane = (candidate, try_phase) (trial_pass)
        {
            const uint32_t ntrial = ncand_w * 5u;
            cand_count += ncand_w;
            ncand_w = 0;
            for (uint32_t tb = 0; tb < ntrial; tb += 64) {
                const uint32_t t5 = tb + (uint32_t)lane;
                uint32_t c, tpi;
                split5(min(t5, ntrial - 1u), c, tpi);
                trial_pass<FUSED, FIELDS>(p, s, hf, cand_entry(wcand[c]), tpi, t5 < ntrial, jbase, chunk, seg, seg_cap, ap_count, lane, par);
            }
            wave_lds_fence();  // wcand is reused by the next passes of the gates
        }
        if (round >= nrounds) break;
    }
    }
tile_end:
    if (late_prio) __builtin_amdgcn_s_setprio(0);
    ACCT(4);
    lds_barrier();
    ACCT(5);
    STAMP(5);
    if (ADSB_STOP_AT(p, 5)) {
        lds_barrier();
        continue;
    }

    // ---------------------------------------------------------------- tile epilogue
    // (the AP fill counts are registers; they are written back when the workgroup retires)
    const uint32_t nhit = min(s.nhit[par], (uint32_t)kHitCap);
    if constexpr (FUSED && FIELDS) {
        // one-launch pass: the staged hits' records are built here and now, a wave a hit (emit_record); only what
        // did not fit the staging went to the hit list (stage_hit), for the record builder at the end
        const uint32_t nstaged = min(s.nhit[par], (uint32_t)(kHitCap + kFusedExtraHits));
        for (uint32_t i = (uint32_t)(tid >> 6); i < nstaged; i += (uint32_t)kWaves) {
            const bool extra = i >= (uint32_t)kHitCap;   // (wave-uniform)
            const uint64_t me = extra ? hf.xhit[i - kHitCap] : s.hit[i];
            uint32_t ff[5];
#pragma unroll
            for (int r = 0; r < 5; r++) ff[r] = extra ? hf.xf[i - kHitCap][r] : hf.f[i][r];
            emit_record(p, s, ff, (uint32_t)((int)entry_j(me) - (jbase - kPad)), me, entry_value(me), lane);
        }
    } else
    if (nhit) {  // sparse streams: a handful per buffer; dense ones: most tiles
        // Dense stream: the hits of a tile go into the tile's own part of its buffer's bucket.  This workgroup is that
        // part's only writer during the scan, so unless the staging overflowed (more than kHitCap hits in one tile:
        // the rest went in one by one, stage_hit) they take places 0 .. nhit - 1 and the count is a plain store:
        // nothing to wait for, no barrier.  Sparse stream (one flat list): a place from the list's counter.
        const bool dense = p.order_cnt != nullptr;
        const uint32_t tile_g = chunk * (uint32_t)kTilesPerChunk + (uint32_t)cur.tile;
        const bool alone = dense && s.nhit[par] <= (uint32_t)kHitCap;   // (uniform)
        uint64_t *const dst = dense ? p.order_tmp + (size_t)chunk * kOrderBucket + (size_t)cur.tile * kTileBucket : p.hits;
        const uint32_t dst_cap = dense ? kTileBucket : p.hits_cap;
        uint32_t hit_base = 0;
        if (alone) {
            if (tid == 0) {
                atomicAdd(&p.ctr->n_hits, nhit);   // (no value taken: fire and forget)
                p.order_cnt[tile_g] = nhit;
            }
        } else {
            if (tid == 0) {
                const uint32_t at = atomicAdd(&p.ctr->n_hits, nhit);
                s.hit_base = dense ? atomicAdd(&p.order_cnt[tile_g], nhit) : at;
            }
            lds_barrier();
            hit_base = s.hit_base;
        }
        if (hit_base + nhit > dst_cap) {
            if (tid == 0) atomicOr(&p.ctr->overflow, 1u);
        } else {
            for (uint32_t i = tid; i < nhit; i += kThreads) {
                st_shared<FUSED>(&dst[hit_base + i], s.hit[i]);
                if constexpr (FIELDS)
                    put_hit_fields<FUSED>(p, (size_t)(dst - (dense ? p.order_tmp : p.hits)) + hit_base + i, hf.f[i]);
            }
        }
    }
    ACCT(6);
    STAMP(6);
    }  // tile loop
#ifdef ADSB_KERNEL_ACCT
    if (acct && lane == 0)
        for (int k = 0; k < 8; k++) p.timeline[((size_t)blockIdx.x * kWaves + (tid >> 6)) * 8 + k] = acc_t[k];
#endif
    // candidate counts were kept per wave (diagnostic): lane 0 of each wave adds its own
    if (lane == 0 && cand_count) atomicAdd(&p.ctr->seg_cand[blockIdx.x], cand_count);
    if (lane == 0) {
        if (ap_count > seg_cap) atomicOr(&p.ctr->overflow, 2u);
        st_shared<FUSED>(&p.ctr->seg_ap[my_seg], min(ap_count, seg_cap));
    }
    if constexpr (FUSED) {
        // ============================================================ the tail of a one-launch pass
        // (a) Publish this tile -- its address bits, list entries and counts -- and wait, for a bounded time,
        // until every tile BEFORE it in the buffer order has done the same: a trial must see what earlier
        // positions taught the filter (src/mode_s/mod.rs:71,115,130 read what :83,99 wrote), and workgroups
        // finish in any order.  Workgroups only ever wait for tiles in front of them and publish before they
        // wait, so the chain cannot close on itself; the wait is bounded anyway (a workgroup in front may
        // not have been given a CU yet): whoever gives up says so, and then the last workgroup looks at
        // every list once more (c).
        FSTAMP(2);
        static_assert(kFusedMaxTiles >= 16 * kTilesPerChunk, "tile flags of the largest one-launch pass");
        // (release: everything this workgroup wrote for others went through to the memory side -- st_shared,
        // atomics -- and is acknowledged from there; no cache write-back: adsb_tail_dev.h, st_shared)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&p.ctr->tile_done[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64) {
            bool ordered = true;
            for (uint32_t spins = 0;; spins++) {
                bool all = true;
                for (uint32_t k = (uint32_t)lane; k < blockIdx.x; k += 64u)
                    all = all && __hip_atomic_load(&p.ctr->tile_done[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (__all(all)) break;
                if (spins >= p.order_polls) {  // 200: ~0.2 ms
                    ordered = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0 && !ordered) atomicOr(&p.ctr->unordered, 1u);
        }
        __syncthreads();
        // Each wave matches its own address/parity entries against the bitmap as it stands now: it holds
        // every address of the passes before this one (stream order; the host redoes a pass whose
        // predecessor on another scan stream turns out to have learned one: adsb_collect.cpp), what the
        // tiles before this one learned, and whatever else this pass has learned so far (harmless: the host
        // replay scores in order).
        {
            // (a match's record is built here and now: this workgroup's tile is still in LDS -- emit_records)
            const uint32_t n_mine = min(ap_count, seg_cap);
            const int slot0 = tile_ref<FROM_MAG>(p, blockIdx.x).jbase - kPad;   // data index of LDS slot 0 (one tile per workgroup)
            for (uint32_t i0 = 0; i0 < n_mine; i0 += 64u) {
                const uint32_t i = i0 + (uint32_t)lane;
                const uint64_t e = i < n_mine ? ld_shared<true>(&seg[i]) : (15ull << 24);
                const uint32_t code = entry_code(e);
                uint32_t c = entry_value(e);
                if (code >= 5u && code < 10u) c = gf_apply(fs.x56, c);
                // (agent scope: bits other workgroups of this launch have set, not a line this CU's cache holds)
                const uint32_t at = bitmap_index(c, p.bitmap_lg);
                const bool hit = code != 15u && ((ld_shared<true>(&p.bitmap[at >> 5]) >> (at & 31u)) & 1u) != 0u;
                const unsigned long long mm = __ballot(hit);
                if (mm) {
                    const uint32_t cs = hit ? (uint32_t)((int)entry_j(e) - slot0) : (uint32_t)kPad;
                    Trial tr;
                    trial_eval(s, cand_entry(cs), code % 5u, tr);
                    emit_records(p, s, mm, tr.f, cs, e, c, lane);
                    if (hit) st_shared<true>(&seg[i], e | (15ull << 24));   // the second look must not report it again
                }
            }
        }
        // (b) The last workgroup to get here runs the rest alone: release what the match wrote (hits, marks;
        // usually nothing -- the tile itself was published above), count this workgroup in, acquire what
        // the others wrote.
        FSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // what the match wrote (hits, marks) has arrived
        __syncthreads();
        if (tid == 0) fs.is_last = atomicAdd(&p.ctr->scan_blocks_done, 1u) == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (!fs.is_last) return;
        // (acquire: nothing -- from here on what the other workgroups wrote is read with ld_shared)
        FSTAMP(4);
        // (c) The fallback: some workgroup matched without having seen all the tiles before it, and an
        // address bit that was clear when the pass began was set on the way -- its entries may have missed
        // it.  Once more over every segment, marked entries skipped.
        const uint32_t n_new = __hip_atomic_load(&p.ctr->learned_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_new && __hip_atomic_load(&p.ctr->unordered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const uint32_t nseg = gridDim.x * (uint32_t)kWaves;
            // every segment's fill count into LDS first (one round trip for all of them), then each wave
            // takes its segments nine at a time, their entries in flight together; with a handful of new
            // addresses an entry is compared with those directly (no trip to the bitmap)
            uint32_t *const cnt = s.plane;
            static_assert(sizeof(s.plane) / 4 >= 4 * 16 * kTilesPerChunk, "fill counts of the largest one-launch pass");
            for (uint32_t g = (uint32_t)tid; g < nseg; g += kThreads) cnt[g] = min(ld_shared<true>(&p.ctr->seg_ap[g]), seg_cap);
            uint32_t fresh[kNewAddrCap];
            const bool by_list = n_new <= (uint32_t)kNewAddrCap;
#pragma unroll
            for (int k = 0; k < kNewAddrCap; k++)
                fresh[k] = by_list && (uint32_t)k < n_new ? ld_shared<true>(&p.ctr->new_addr[k]) : 0xFFFFFFFFu;
            lds_barrier();
            FSTAMP(8);
            constexpr uint32_t U = 9;
            for (uint32_t g0 = (uint32_t)(tid >> 6) * U; g0 < nseg; g0 += kWaves * U) {
                uint32_t n[U], nmax = 0;
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    n[u] = g0 + u < nseg ? cnt[g0 + u] : 0u;
                    nmax = max(nmax, n[u]);
                }
                for (uint32_t i = (uint32_t)lane; i < nmax; i += 64u) {
                    uint64_t e[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) e[u] = i < n[u] ? ld_shared<true>(&p.ap[(uint64_t)(g0 + u) * seg_cap + i]) : (15ull << 24);
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        if (!by_list) {
                            fused_match_entry(p, fs.x56, &p.ap[(uint64_t)(g0 + u) * seg_cap + i], e[u]);
                            continue;
                        }
                        const uint32_t code = entry_code(e[u]);
                        uint32_t c = entry_value(e[u]);
                        if (code >= 5u && code < 10u) c = gf_apply(fs.x56, c);
                        bool hit = false;
#pragma unroll
                        for (int k = 0; k < kNewAddrCap; k++) hit = hit || c == fresh[k];
                        if (hit && code != 15u) {
                            const uint32_t idx = atomicAdd(&p.ctr->n_hits, 1u);
                            if (idx < p.hits_cap) {
                                st_shared<true>(&p.hits[idx], e[u]);
                                if (p.hit_fields) st_shared<true>(&p.hit_fields[(size_t)idx * kHitFieldWords + 5], 0u);
                            } else {
                                atomicOr(&p.ctr->overflow, 1u);
                            }
                        }
                    }
                }
            }
            FSTAMP(7);
            // (this workgroup's own appends: written through, and read past this CU's cache by the record
            // builder -- no cache write-back needed, only their completion)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        FSTAMP(5);
        // (d) records, checksum, summary into mapped host memory; the counters back to zero
        // (the records built in place are there already; what is left is what the second look found, if it ran)
        records_block<FROM_MAG, false, true>(p, p.fused_rec, 0u, 1u, nullptr, false, gridDim.x, fs.bits);
        FSTAMP(6);
    }
}

inline int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
// hipGetLastError is sticky across unrelated calls (the caller's too): start every launch clean
inline void hip_clear() { (void)hipGetLastError(); }

}  // namespace

// persistent grid = what is resident at once (occupancy API x CUs) on the CURRENT device, found once per device
int scan_resident_blocks()
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cached[kMaxDevices];   // (zero-initialised: 0 = not asked yet; two threads asking at once
                                                   // compute the same number)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (const int r = cached[dev].load(std::memory_order_relaxed)) return r;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_fast<false>, kThreads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0) {
        per_cu = 2;
        cus = 256;
    }
    int r = per_cu * cus;
    if (const char *e = tuning_env("ADSB_SCAN_BLOCKS_PER_CU")) r = std::atoi(e) * cus;
    if (r > kApSegments) r = kApSegments;  // four private AP segments (one per wave) each
    if (r < 1) r = 1;
    if (tuning_env("ADSB_TIMELINE"))
        std::fprintf(stderr, "k_scan_fast: device %d: occupancy %d blocks/CU x %d CUs\n", dev, per_cu, cus);
    cached[dev].store(r, std::memory_order_relaxed);
    return r;
}

int launch_pass_fused(const ScanParams &p, bool from_mag, void *stream)
{
    hip_clear();
    const uint32_t tiles = p.n_chunks * kTilesPerChunk;  // one workgroup per tile: a pass of a few buffers
    if (tiles == 0 || !p.fused_rec) return (int)hipErrorInvalidValue;
    // (one-launch passes stage their hits with their bit fields: p.hit_fields is set for them, for the hits of
    // a tile that did not fit the staging)
    if (!p.hit_fields) return (int)hipErrorInvalidValue;
    if (from_mag)
        hipLaunchKernelGGL((k_scan_fast<true, false, true, true>), dim3(tiles), dim3(kThreads), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_scan_fast<false, false, true, true>), dim3(tiles), dim3(kThreads), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_scan(const ScanParams &p, bool from_mag, void *stream)
{
    hip_clear();
    const uint32_t tiles = p.n_chunks * kTilesPerChunk;
    if (tiles == 0) return 0;
    const int resident = scan_resident_blocks();
    const uint32_t blocks = tiles < (uint32_t)resident ? tiles : (uint32_t)resident;
    // With events, the launch itself carries them (hipExtLaunchKernelGGL): the dispatch
    // packet's own begin/end timestamps, no barrier packets in the stream around it.
    if (p.cand_out)  // the self-test's instantiation: also writes the gate-stage position list
        hipLaunchKernelGGL((k_scan_fast<false, true>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p);
    else if (from_mag)  // adsb_demodulate2400: one caller-supplied MagnitudeBuffer
        hipLaunchKernelGGL(k_scan_fast<true>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p);
    else if (p.hit_fields && p.ev_start && p.ev_stop)  // dense stream: the scan hands its hits' bit fields to the record builder
        hipExtLaunchKernelGGL((k_scan_fast<false, false, false, true>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,
                              (hipEvent_t)p.ev_start, (hipEvent_t)p.ev_stop, 0, p);
    else if (p.hit_fields)
        hipLaunchKernelGGL((k_scan_fast<false, false, false, true>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p);
    else if (p.ev_start && p.ev_stop)
        hipExtLaunchKernelGGL(k_scan_fast<false>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,
                              (hipEvent_t)p.ev_start, (hipEvent_t)p.ev_stop, 0, p);
    else
        hipLaunchKernelGGL(k_scan_fast<false>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

}  // namespace adsb
