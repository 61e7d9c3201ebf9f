// adsb_tables.h -- GF(2) lookup tables for the Mode-S CRC-24 as the scan kernel computes it.
//
// The reference computes  residual = crc24(first n-3 bytes) XOR last 3 bytes
// (src/crc.rs:263-282), which is M(x) mod g(x) for the whole `bits`-bit message
// M(x) = sum_n bit_n * x^(bits-1-n), g(x) = x^24 + 0xFFF409.
//
// The scan kernel never assembles the message: it pulls, for each of the five bit
// classes r = n mod 5, one field f_r whose bit k is message bit n = 5k + r (those bits
// are 12 samples apart, which is how the sign planes are laid out).  With
// y = x^-5 mod g:
//     M(x) = sum_r x^(bits-1-r) * F(f_r),      F(f) = sum_k f[k] * y^k
//          = x^(bits-5) * H,                   H = sum_r x^(4-r) * F(f_r)   (Horner in x)
// The tables hold F' = x^51 * F (three 256-entry lookups, one per byte of the field), so
// H' = x^51 * H is the residual itself for 56-bit messages (DF11's test and the short
// address/parity trials need nothing more) and x^56 * H' for 112-bit ones; that constant
// multiplier is three more lookups, applied only in the match kernel.  A clean DF17/18 is
// simply H' == 0 (x is invertible mod g).
#pragma once
#include <cstdint>
#include <vector>

namespace adsb {

constexpr uint32_t kCrcPoly = 0xFFF409u;  // src/crc.rs: g(x) minus the x^24 term

// multiply by x in GF(2)[x]/g
inline uint32_t gf_mulx(uint32_t a)
{
    const uint32_t hi = a & 0x800000u;
    a = (a << 1) & 0xFFFFFFu;
    return hi ? a ^ kCrcPoly : a;
}
// multiply by x^-1 (g has constant term 1, so x is invertible)
inline uint32_t gf_divx(uint32_t a)
{
    return (a & 1u) ? ((a ^ kCrcPoly) >> 1) | 0x800000u : a >> 1;
}

// tables[t][v]: t = kTabF+b:   x^51 * sum of y^(8b+i) over set bits i of v
//               t = kTabX56+b: (v << 8b) * x^56
inline std::vector<uint32_t> build_gf_tables()
{
    std::vector<uint32_t> t(6 * 256, 0);
    // powers of y = x^-5
    uint32_t ypow[24];
    uint32_t p = 1;
    for (int e = 0; e < 51; e++) p = gf_mulx(p);  // x^51 folded into F
    for (int k = 0; k < 24; k++) {
        ypow[k] = p;
        for (int s = 0; s < 5; s++) p = gf_divx(p);
    }
    // x^56 * x^i for i = 0..23
    uint32_t x56[24];
    p = 1;
    for (int e = 0; e < 56; e++) p = gf_mulx(p);
    for (int i = 0; i < 24; i++, p = gf_mulx(p)) x56[i] = p;
    for (int b = 0; b < 3; b++)
        for (uint32_t v = 0; v < 256; v++) {
            uint32_t f = 0, a = 0;
            for (int i = 0; i < 8; i++)
                if (v & (1u << i)) {
                    f ^= ypow[8 * b + i];
                    a ^= x56[8 * b + i];
                }
            t[(0 + b) * 256 + v] = f;
            t[(3 + b) * 256 + v] = a;
        }
    return t;
}

// Reduction of x^24..x^27: R16[v] = sum of x^(24+i) over set bits i of v (16 entries).
inline std::vector<uint32_t> build_r16()
{
    uint32_t xp[4];
    uint32_t p = 1;
    for (int e = 0; e < 24; e++) p = gf_mulx(p);
    for (int i = 0; i < 4; i++, p = gf_mulx(p)) xp[i] = p;
    std::vector<uint32_t> r(16, 0);
    for (uint32_t v = 0; v < 16; v++)
        for (int i = 0; i < 4; i++)
            if (v & (1u << i)) r[v] ^= xp[i];
    return r;
}

// Per-bit residual constants for the records kernel: the residual of a message is the XOR of
// x^(bits-1-n) mod g over its set bits n (src/crc.rs:263-282 is M(x) mod g, see above).
// [0..112): x^(111-n) for 112-bit messages; [112..168): x^(55-n) for 56-bit ones.
inline std::vector<uint32_t> build_bit_residuals()
{
    std::vector<uint32_t> t(168, 0);
    uint32_t p = 1;
    for (int e = 0; e < 112; e++, p = gf_mulx(p)) {
        t[111 - e] = p;
        if (e < 56) t[112 + 55 - e] = p;
    }
    return t;
}

// Field addressing of the fast scan's trial phase.  Message bit n = 5k + r of trial phase
// tp = 4 + tpi at a preamble whose LDS slot is 12*qs + rs sits in sign plane
// (ph, res), bit qs + carry + k, with
//     P = tp + 12r,  a = P / 5,  ph = P % 5,  res' = rs + 19 + a,  carry = res' / 12,  res = res' % 12
// (reference src/demod_2400.rs:158-182 in closed form).  Entry [tpi][r][rs] =
// byte offset of plane row (ph*12 + res) | carry << 16; plane rows are plane_bytes long.
inline std::vector<uint32_t> build_field_table(uint32_t plane_bytes)
{
    std::vector<uint32_t> t(5 * 5 * 12, 0);
    for (int tpi = 0; tpi < 5; tpi++)
        for (int r = 0; r < 5; r++)
            for (int rs = 0; rs < 12; rs++) {
                const int P = 4 + tpi + 12 * r, a = P / 5, ph = P % 5;
                const int rp = rs + 19 + a, carry = rp / 12, res = rp % 12;
                t[(tpi * 5 + r) * 12 + rs] = (uint32_t)((ph * 12 + res) * plane_bytes) | ((uint32_t)carry << 16);
            }
    return t;
}

}  // namespace adsb
