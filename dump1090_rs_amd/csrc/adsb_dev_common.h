// adsb_dev_common.h -- device helpers shared by the gfx950 kernels.
//
// What they compute is fixed by the reference (rsadsb/dump1090_rs v0.8.1):
//   magnitude      src/utils.rs:43-58
//   preamble+gates src/demod_2400.rs:127-146, 215-321
//   bit slicing    src/demod_2400.rs:7-84, 158-182
//   DF / CRC-24    src/mode_s/mod.rs:41-47, src/crc.rs:263-282
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the magnitude pipeline
// must keep its one rounded multiply and two fused multiply-adds exactly).
#pragma once
#include <hip/hip_runtime.h>

#include "adsb_device.h"

namespace adsb {

// ---------------------------------------------------------------------------
// magnitude: src/utils.rs:47-55
//   fi = im/2^15, fq = re/2^15, mag = sqrt(fma(fi,fi,rn(fq*fq))),
//   out = sat_u16(trunc(fma(mag, 65535, 0.5)))
// Scaling by 2^-15 is exact and commutes with every rounding here (no value is
// subnormal or overflows: X = rn(im^2 + rn(re^2)) is 0 or in [1, 2^31]), so the
// two divisions fold into the last constant: 65535 * 2^-15 is a 16-bit value.
//
// sqrt must be the correctly rounded one (IEEE), as Rust's f32::sqrt is.  The raw v_sqrt_f32
// (1 ulp) is not good enough: it flips the u16 result for about one sample in 10^5.  The root
// is built from ONE transcendental instead: with y = v_rsq_f32(x) (1 ulp),
//     s = x * y          within 2 ulp of the root
//     h = y / 2
//     d = fma(-s, s, x)  the remainder x - s^2, rounded once
//     r = fma(d, h, s)   the correctly rounded root
// (Markstein's final correction step).  That r is the IEEE root for EVERY f32 x in [1, 2^31] --
// a superset of what im^2 + rn(re^2) can be -- is not taken on trust: tools/sqrt_markstein.hip
// compares it with the neighbour-test construction LLVM uses for sqrtf over all 260 046 849
// values (0 differences), and tests/test_gpu_parity.py sweeps the same range through mag_tail2
// against the CPU.  For x = 0: y = inf, s = NaN, and the NaN runs through to the float->u32
// conversion, which turns it into 0 (V_CVT_U32_F32: "NaN is converted to 0") -- the magnitude of
// a zero sample; the conversion is written as the instruction itself because a C++ cast of NaN
// to an integer is undefined.
//
// Everything is done on pairs so the multiplies and fused multiply-adds are
// v_pk_mul_f32 / v_pk_fma_f32 (two samples per instruction) and the saturating cast +
// pack is one v_cvt_pk_u16_u32: 8 VALU instructions per sample from the IQ dword to the packed
// u16 (the neighbour-test form took 12.5).
// ---------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}

// truncating, saturating float -> u32 with NaN -> 0 (the instruction's own definition)
__device__ __forceinline__ uint32_t cvt_u32_sat(float v)
{
    uint32_t u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(v));
    return u;
}

// two X = im^2 + rn(re^2) -> two magnitudes packed as u16 (low half = first)
__device__ __forceinline__ uint32_t mag_tail2(f32x2 x)
{
    const f32x2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const f32x2 s = x * y;
    const f32x2 hk = {0.5f, 0.5f};
    const f32x2 h = y * hk;
    const f32x2 d = pk_fma(-s, s, x);
    const f32x2 r = pk_fma(d, h, s);  // == sqrt_rn(x) for x in [1, 2^31]; NaN for x = 0
    const f32x2 c = {65535.0f / 32768.0f, 65535.0f / 32768.0f}, half = {0.5f, 0.5f};
    const f32x2 o = pk_fma(r, c, half);
    // o >= 0.5 or NaN; the u32 conversion truncates, the pack saturates at 65535 (Rust `as u16`)
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(o.x), cvt_u32_sat(o.y)));
}

// two IQ dwords {re (low half), im (high half)} -> two magnitudes packed as u16
__device__ __forceinline__ uint32_t mag2(uint32_t w0, uint32_t w1)
{
    const f32x2 fq = {(float)(int16_t)(w0 & 0xFFFFu), (float)(int16_t)(w1 & 0xFFFFu)};
    const f32x2 fi = {(float)((int32_t)w0 >> 16), (float)((int32_t)w1 >> 16)};
    const f32x2 t = fq * fq;               // the separately rounded square (utils.rs:53)
    return mag_tail2(pk_fma(fi, fi, t));   // fi.mul_add(fi, fq*fq)
}

__device__ __forceinline__ uint32_t mag_of_dword(uint32_t w) { return mag2(w, 0u) & 0xFFFFu; }

// Four consecutive IQ samples starting at sample k of a chunk of `len` samples ->
// four magnitudes packed as u16 pairs.  k is a multiple of 4 (16-byte aligned
// load) and may be negative (lead-in) or run past len (zero tail / short chunk).
__device__ __forceinline__ uint4 load_iq4(const uint32_t *__restrict__ iq, int k, int len)
{
    uint4 v = {0u, 0u, 0u, 0u};  // mag(0,0) = 0, so zero IQ stands for "no sample"
    if (k >= 0 && k + 3 < len) {
        v = *(const uint4 *)(iq + k);
    } else if (k >= 0 && k < len) {  // ragged end of a short last chunk
        v.x = iq[k];
        if (k + 1 < len) v.y = iq[k + 1];
        if (k + 2 < len) v.z = iq[k + 2];
    }
    return v;
}

__device__ __forceinline__ uint2 mag4_of(uint4 v)
{
    uint2 pk;
    pk.x = mag2(v.x, v.y);
    pk.y = mag2(v.z, v.w);
    return pk;
}

__device__ __forceinline__ uint2 mag4(const uint32_t *__restrict__ iq, int k, int len)
{
    return mag4_of(load_iq4(iq, k, len));
}

// ---------------------------------------------------------------------------
// CRC-24, generator 0xFFF409 (src/crc.rs).  Table entry i = i<<16 through 8
// MSB-first steps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t crc_table_entry(uint32_t i)
{
    uint32_t c = i << 16;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c & 0x800000u) ? ((c << 1) ^ 0xFFF409u) : (c << 1);
    return c & 0xFFFFFFu;
}

// message held MSB-first in 4 words: w[0] bits 31..0 = message bits 0..31, ...
__device__ __forceinline__ uint32_t msg_byte(const uint32_t w[4], int i)
{
    return (w[i >> 2] >> (24 - 8 * (i & 3))) & 0xFFu;
}

// src/crc.rs:263-282 over nbytes (7 or 14)
__device__ __forceinline__ uint32_t modes_checksum(const uint32_t w[4], int nbytes,
                                                   const uint32_t *tab)
{
    uint32_t rem = 0;
    for (int i = 0; i < nbytes - 3; i++)
        rem = ((rem << 8) ^ tab[msg_byte(w, i) ^ ((rem >> 16) & 0xFFu)]) & 0xFFFFFFu;
    rem ^= (msg_byte(w, nbytes - 3) << 16) | (msg_byte(w, nbytes - 2) << 8) | msg_byte(w, nbytes - 1);
    return rem;
}

// ---------------------------------------------------------------------------
// preamble + gates: src/demod_2400.rs:127-146, 215-321.  p = &data[j].
// Returns true when j goes on to be sliced.
// ---------------------------------------------------------------------------
template <typename Ptr>
__device__ __forceinline__ bool preamble_gates(Ptr p)
{
    const int p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6],
              p7 = p[7], p8 = p[8], p9 = p[9], p10 = p[10], p11 = p[11], p12 = p[12],
              p13 = p[13];
    if (!(p0 < p1 && p12 > p13)) return false;  // :221
    int high;
    unsigned sig, noise;
    if (p1 > p2 && p2 < p3 && p3 > p4 && p8 < p9 && p9 > p10 && p10 < p11) {         // :227
        high = (p1 + p3 + p9 + p11 + p12) / 4;
        sig = p1 + p3 + p9;
        noise = p5 + p6 + p7;
    } else if (p1 > p2 && p2 < p3 && p3 > p4 && p8 < p9 && p9 > p10 && p11 < p12) {  // :242
        high = (p1 + p3 + p9 + p12) / 4;
        sig = p1 + p3 + p9 + p12;
        noise = p5 + p6 + p7 + p8;
    } else if (p1 > p2 && p2 < p3 && p4 > p5 && p8 < p9 && p10 > p11 && p11 < p12) { // :262
        high = (p1 + p3 + p4 + p9 + p10 + p12) / 4;
        sig = p1 + p12;
        noise = p6 + p7;
    } else if (p1 > p2 && p3 < p4 && p4 > p5 && p9 < p10 && p10 > p11 && p11 < p12) { // :280
        high = (p1 + p4 + p10 + p12) / 4;
        sig = p1 + p4 + p10 + p12;
        noise = p5 + p6 + p7 + p8;
    } else if (p2 > p3 && p3 < p4 && p4 > p5 && p9 < p10 && p10 > p11 && p11 < p12) { // :300
        high = (p1 + p2 + p4 + p10 + p12) / 4;
        sig = p4 + p10 + p12;
        noise = p6 + p7 + p8;
    } else {
        return false;
    }
    if (sig * 2 < 3 * noise) return false;  // :129
    const int p14 = p[14], p15 = p[15], p16 = p[16], p17 = p[17], p18 = p[18];
    if (p5 >= high || p6 >= high || p7 >= high || p8 >= high || p14 >= high || p15 >= high ||
        p16 >= high || p17 >= high || p18 >= high)
        return false;  // :135-146
    return true;
}

// The same sequence stage by stage, for the self-test's position lists (adsb_selftest_gate_stages):
// 0 = check_preamble returns None, 1 = Some, 2 = and the 3.5 dB test (:129), 3 = and the quiet
// samples (:135-146), i.e. the position is sliced.
template <typename Ptr>
__device__ __forceinline__ int preamble_stage(Ptr p)
{
    const int p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6],
              p7 = p[7], p8 = p[8], p9 = p[9], p10 = p[10], p11 = p[11], p12 = p[12],
              p13 = p[13];
    if (!(p0 < p1 && p12 > p13)) return 0;
    int high;
    unsigned sig, noise;
    if (p1 > p2 && p2 < p3 && p3 > p4 && p8 < p9 && p9 > p10 && p10 < p11) {
        high = (p1 + p3 + p9 + p11 + p12) / 4, sig = p1 + p3 + p9, noise = p5 + p6 + p7;
    } else if (p1 > p2 && p2 < p3 && p3 > p4 && p8 < p9 && p9 > p10 && p11 < p12) {
        high = (p1 + p3 + p9 + p12) / 4, sig = p1 + p3 + p9 + p12, noise = p5 + p6 + p7 + p8;
    } else if (p1 > p2 && p2 < p3 && p4 > p5 && p8 < p9 && p10 > p11 && p11 < p12) {
        high = (p1 + p3 + p4 + p9 + p10 + p12) / 4, sig = p1 + p12, noise = p6 + p7;
    } else if (p1 > p2 && p3 < p4 && p4 > p5 && p9 < p10 && p10 > p11 && p11 < p12) {
        high = (p1 + p4 + p10 + p12) / 4, sig = p1 + p4 + p10 + p12, noise = p5 + p6 + p7 + p8;
    } else if (p2 > p3 && p3 < p4 && p4 > p5 && p9 < p10 && p10 > p11 && p11 < p12) {
        high = (p1 + p2 + p4 + p10 + p12) / 4, sig = p4 + p10 + p12, noise = p6 + p7 + p8;
    } else {
        return 0;
    }
    if (sig * 2 < 3 * noise) return 1;
    const int p14 = p[14], p15 = p[15], p16 = p[16], p17 = p[17], p18 = p[18];
    if (p5 >= high || p6 >= high || p7 >= high || p8 >= high || p14 >= high || p15 >= high ||
        p16 >= high || p17 >= high || p18 >= high)
        return 2;
    return 3;
}

// ---------------------------------------------------------------------------
// bit slicer: src/demod_2400.rs:72-83 (+ the Phase walk :22-70 in closed form).
// Bit n of trial phase tp at preamble j sits at 5x-oversampled position
// 5*(j+19) + tp + 12*n; sample = pos/5, phase = pos%5.  m = &data[j] here.
// ---------------------------------------------------------------------------
template <typename Ptr>
__device__ __forceinline__ int slice_value(Ptr m, int phase)
{
    const int m0 = m[0], m1 = m[1], m2 = m[2];
    switch (phase) {
    case 0: return 5 * m0 - 3 * m1 - 2 * m2;
    case 1: return 4 * m0 - m1 - 3 * m2;
    case 2: return 3 * m0 + m1 - 4 * m2;
    case 3: return 2 * m0 + 3 * m1 - 5 * m2;
    default: return m0 + 5 * m1 - 5 * m2 - (int)m[3];
    }
}

// The same for a phase that differs from lane to lane, without the five-way branch: the
// coefficients of phases 0..3 are linear in the phase, (5 - ph, 2 ph - 3, -(2 + ph)), and
// phase 4 is that line's (1, 5, -6) plus (0, 0, 1, -1).  Reads m[3] whatever the phase.
template <typename Ptr>
__device__ __forceinline__ int slice_value_any(Ptr m, int phase)
{
    const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
    const int v = (5 - phase) * m0 + (2 * phase - 3) * m1 - (2 + phase) * m2;
    return phase == 4 ? v + m2 - m3 : v;
}

template <typename Ptr>
__device__ __forceinline__ void slice_message(Ptr at_j, int tp, uint32_t w[4])
{
    w[0] = w[1] = w[2] = w[3] = 0;
    int pos = 5 * 19 + tp;
    for (int n = 0; n < 112; n++, pos += 12) {
        const int s = pos / 5, ph = pos - 5 * s;
        if (slice_value(at_j + s, ph) > 0) w[n >> 5] |= 0x80000000u >> (n & 31);
    }
}

// append `e` to a global list for the lanes with `has`; one atomic per wave.
// Must be reached by whole waves.
__device__ __forceinline__ void wave_append(bool has, uint64_t e, uint64_t *list, uint32_t cap,
                                            uint32_t *count, uint32_t *overflow, uint32_t ovf_bit)
{
    const unsigned long long mask = __ballot(has);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (has) {
        const uint32_t idx = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (idx < cap)
            list[idx] = e;
        else
            atomicOr(overflow, ovf_bit);
    }
}

// clear a bitmap (2^lg bits) and its summary; address 0 always tests true (src/icao_filter.rs:71-80: an
// empty slot equals 0), so bit 0 starts set in both.  Call with all threads of the grid.
__device__ __forceinline__ void bitmap_clear(uint32_t *bitmap, uint32_t lg, uint32_t gi, uint32_t gn)
{
    const uint32_t words = bitmap_words(lg);
    for (uint32_t v = gi; v < (words + kCoarseWords) / 4; v += gn)
        ((uint4 *)bitmap)[v] = make_uint4((v == 0 || v == words / 4) ? 1u : 0u, 0u, 0u, 0u);
}

// returns true when the address bit was clear before
__device__ __forceinline__ bool bitmap_set(uint32_t *bitmap, uint32_t lg, uint32_t addr)
{
    const uint32_t at = bitmap_index(addr, lg);
    const uint32_t old = atomicOr(&bitmap[at >> 5], 1u << (at & 31));
    atomicOr(&bitmap[bitmap_words(lg) + ((addr & 4095u) >> 5)], 1u << (addr & 31));  // the summary
    return ((old >> (at & 31)) & 1u) == 0;
}

// samples in `chunk` of a call over n_samples (the last chunk may be short)
__device__ __forceinline__ int chunk_len(uint64_t n_samples, uint64_t chunk)
{
    const uint64_t remaining = n_samples - chunk * (uint64_t)kChunkSamples;
    return remaining < (uint64_t)kChunkSamples ? (int)remaining : kChunkSamples;
}

}  // namespace adsb
