// adsb_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the demod_2400 hot path.
//
// What the kernels compute is fixed by the reference (rsadsb/dump1090_rs v0.8.1):
//   magnitude      src/utils.rs:43-58
//   preamble+gates src/demod_2400.rs:127-146, 215-321
//   bit slicing    src/demod_2400.rs:7-84, 158-182
//   DF / CRC-24    src/mode_s/mod.rs:41-47, src/crc.rs:263-282
// How they compute it is not: see DESIGN.md for the tile/LDS layout.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the magnitude pipeline
// must keep its one rounded multiply and two fused multiply-adds exactly).
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
// sqrt must be the correctly rounded one (IEEE), as Rust's f32::sqrt is.  HIP's
// __fsqrt_rn is the raw v_sqrt_f32 (1 ulp) -- not good enough: it flips the u16
// result for about one sample in 10^5.  sqrt_rn below is v_sqrt_f32 plus the
// neighbour test LLVM uses for IEEE sqrtf, without the subnormal scaling and
// class checks x never needs (x is 0 or in [1, 2^31]): the correctly rounded root is
// the candidate s, or its lower neighbour if s_dn*s >= x, or its upper one if
// s_up*s < x (the products stand for the squared midpoints).  For x = 0 both
// residuals are NaN / 0 and s = 0 stays.  tests/test_gpu_parity.py sweeps every
// f32 x in the range against the CPU.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __fmaf_rn(-s_dn, s, x);
    const float r_up = __fmaf_rn(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}

__device__ __forceinline__ uint32_t mag_from_x(float x)
{
    float m = sqrt_rn(x);
    float o = __fmaf_rn(m, 65535.0f / 32768.0f, 0.5f);
    o = fminf(o, 65535.0f);  // Rust `as u16` saturates; o >= 0.5 always
    return (uint32_t)o;      // truncates
}

__device__ __forceinline__ uint32_t mag_u16(int re, int im)
{
    float fq = (float)re, fi = (float)im;
    float t = __fmul_rn(fq, fq);       // the separately rounded square (utils.rs:53)
    float x = __fmaf_rn(fi, fi, t);    // fi.mul_add(fi, fq*fq)
    return mag_from_x(x);
}

// one dword = one IQ sample in memory order {re (low half), im (high half)}
__device__ __forceinline__ uint32_t mag_of_dword(uint32_t w)
{
    return mag_u16((int)(int16_t)(w & 0xFFFFu), (int)(int16_t)(w >> 16));
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

// ---------------------------------------------------------------------------
// K0: to_mag alone (adsb_to_mag).  data[0..326) = 0, data[326+k] = mag(iq[k]),
// rest 0 (src/lib.rs:36-50).
// ---------------------------------------------------------------------------
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
// K1: scan.  One workgroup = one tile of kTile preamble positions j of one chunk.
// LDS slot i holds data[d0 + i], d0 = tile*kTile - kPad; kPad = 2 makes the IQ
// address of slot 0 a multiple of 4 samples (326 + 2 = 4*82) so every global load
// is an aligned dwordx4.
// ---------------------------------------------------------------------------
constexpr int kTile = 4096;
constexpr int kPad = 2;
constexpr int kSlots = kTile + kPad + kReach;  // 4388, a multiple of 4
constexpr int kTilesPerChunk = kChunkSamples / kTile;
static_assert(kSlots % 4 == 0, "slots must be whole 4-sample groups");
static_assert((kLead + kPad) % 4 == 0, "tile origin must be 16-byte aligned in IQ space");

// fill smag[0..kSlots) for (chunk, tile).  len = samples in this chunk.
template <bool FROM_MAG>
__device__ __forceinline__ void load_tile(const void *src, uint64_t chunk, int tile, int len,
                                          uint16_t *smag)
{
    const int d0 = tile * kTile - kPad;
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
        for (int g = threadIdx.x; g < kSlots / 4; g += blockDim.x) {
            const int k = k0 + 4 * g;
            uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
            if (k >= 0 && k + 3 < len) {
                const uint4 v = *(const uint4 *)(iq + k);
                m0 = mag_of_dword(v.x);
                m1 = mag_of_dword(v.y);
                m2 = mag_of_dword(v.z);
                m3 = mag_of_dword(v.w);
            } else if (k >= 0 && k < len) {  // ragged end of a short last chunk
                m0 = mag_of_dword(iq[k]);
                if (k + 1 < len) m1 = mag_of_dword(iq[k + 1]);
                if (k + 2 < len) m2 = mag_of_dword(iq[k + 2]);
            }
            uint2 pk;
            pk.x = m0 | (m1 << 16);
            pk.y = m2 | (m3 << 16);
            *(uint2 *)(smag + 4 * g) = pk;
        }
    }
}

// append `e` to a global list for the lanes with `has`; one atomic per wave
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

template <bool FROM_MAG>
__global__ __launch_bounds__(256) void k_scan(ScanParams p)
{
    __shared__ __attribute__((aligned(16))) uint16_t smag[kSlots];
    __shared__ uint32_t scrc[256];
    __shared__ uint16_t scand[kTile];
    __shared__ uint32_t sncand;

    const uint32_t chunk = blockIdx.x / kTilesPerChunk;
    const int tile = blockIdx.x % kTilesPerChunk;
    // samples in this chunk (the last one may be short)
    const uint64_t remaining = FROM_MAG ? p.n_samples : p.n_samples - (uint64_t)chunk * kChunkSamples;
    const int len = remaining < (uint64_t)kChunkSamples ? (int)remaining : kChunkSamples;
    const int jbase = tile * kTile;
    if (jbase >= len) return;

    scrc[threadIdx.x] = crc_table_entry(threadIdx.x);
    if (threadIdx.x == 0) sncand = 0;
    load_tile<FROM_MAG>(p.src, chunk, tile, len, smag);
    __syncthreads();

    // --- preamble / SNR / quiet gates for every j of the tile (demod_2400.rs:121-146)
    const int jn = min(kTile, len - jbase);
    for (int jj = threadIdx.x; jj < jn; jj += blockDim.x) {
        if (preamble_gates(smag + kPad + jj)) {
            const uint32_t slot = atomicAdd(&sncand, 1u);
            scand[slot] = (uint16_t)jj;
        }
    }
    __syncthreads();
    const int ncand = (int)sncand;
    if (threadIdx.x == 0 && ncand) atomicAdd(&p.ctr->n_cand, (uint32_t)ncand);

    // --- five trial phases per candidate (demod_2400.rs:158-184): slice, DF, CRC
    const int ntrial = ncand * 5;
    for (int base = 0; base < ntrial; base += blockDim.x) {
        const int t = base + threadIdx.x;
        bool is_hit = false, is_ap = false;
        uint64_t entry = 0;
        if (t < ntrial) {
            const int jj = scand[t / 5];
            const int tpi = t % 5;
            uint32_t w[4];
            slice_message(smag + kPad + jj, 4 + tpi, w);
            const uint32_t df = w[0] >> 27;  // mode_s/mod.rs:41
            const bool nonzero = (w[0] | w[1] | w[2] | w[3]) != 0;  // :51 (w[3] low 16 bits are 0)
            if (nonzero) {
                const uint32_t j = (uint32_t)(jbase + jj);
                if (df == 11) {  // :73-90
                    const uint32_t c = modes_checksum(w, 7, scrc);
                    if ((c & 0xFFFF80u) == 0) {
                        is_hit = true;
                        entry = pack_entry(c, tpi, j, chunk);
                        if ((c & 0x7Fu) == 0) {  // iid 0: the replay will add this address
                            const uint32_t addr = ((w[0] & 0xFFFFFFu));
                            atomicOr(&p.bitmap[addr >> 5], 1u << (addr & 31));
                        }
                    }
                } else if (df == 17 || df == 18) {  // :91-109
                    const uint32_t c = modes_checksum(w, 14, scrc);
                    if (c == 0) {
                        is_hit = true;
                        entry = pack_entry(c, tpi, j, chunk);
                        if (df == 17) {  // DF18 adds addr | 1<<25, which no 24-bit test matches
                            const uint32_t addr = ((w[0] & 0xFFFFFFu));
                            atomicOr(&p.bitmap[addr >> 5], 1u << (addr & 31));
                        }
                    }
                } else if (df == 0 || df == 4 || df == 5) {  // :56-72
                    is_ap = true;
                    entry = pack_entry(modes_checksum(w, 7, scrc), tpi, j, chunk);
                } else if (df == 16 || df == 20 || df == 21 || df >= 24) {  // :110-135
                    is_ap = true;
                    entry = pack_entry(modes_checksum(w, 14, scrc), tpi, j, chunk);
                }
            }
        }
        wave_append(is_hit, entry, p.hits, p.hits_cap, &p.ctr->n_hits, &p.ctr->overflow, 1u);
        wave_append(is_ap, entry, p.ap, p.ap_cap, &p.ctr->n_ap, &p.ctr->overflow, 2u);
    }
}

// ---------------------------------------------------------------------------
// K2: match.  An address/parity trial can only score >= 0 if its CRC residual is in
// the filter when it is scored (mode_s/mod.rs:71,115,130); the bitmap now holds
// every address the filter can contain at any point of this call (plus 0, which
// icao_filter_test always accepts, icao_filter.rs:71-80).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match(ScanParams p)
{
    const uint32_t n = min(p.ctr->n_ap, p.ap_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;; i += gridDim.x * blockDim.x) {
        // whole waves stay in the loop together so wave_append's ballot is uniform
        const uint32_t wave_first = i - (threadIdx.x & 63);
        if (wave_first >= n) break;
        bool has = false;
        uint64_t e = 0;
        if (i < n) {
            e = p.ap[i];
            const uint32_t c = entry_crc(e);
            has = (p.bitmap[c >> 5] >> (c & 31)) & 1u;
        }
        wave_append(has, e, p.hits, p.hits_cap, &p.ctr->n_hits, &p.ctr->overflow, 1u);
    }
}

// ---------------------------------------------------------------------------
// K3: records.  One thread per hit: rebuild the 291-sample window behind j, slice the
// 112 bits of the trial phase and sum the 33-sample power (demod_2400.rs:191-196).
// Rare path (a handful per chunk), so it recomputes magnitudes instead of keeping
// them in HBM.
// ---------------------------------------------------------------------------
template <bool FROM_MAG>
struct WindowReader {
    const void *src;
    uint64_t chunk;
    int len;
    int j;
    __device__ uint32_t operator[](int off) const
    {
        const int d = j + off;  // index into MagnitudeBuffer.data
        if (FROM_MAG) return ((const uint16_t *)src)[d];
        const int k = d - kLead;
        if (k < 0 || k >= len) return 0;
        return mag_of_dword(((const uint32_t *)src)[chunk * (uint64_t)kChunkSamples + k]);
    }
    __device__ WindowReader operator+(int off) const
    {
        WindowReader r = *this;
        r.j += off;
        return r;
    }
};

template <bool FROM_MAG>
__global__ __launch_bounds__(64) void k_records(ScanParams p, TrialRecord *rec)
{
    const uint32_t n = min(p.ctr->n_hits, p.hits_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t e = p.hits[i];
        const uint64_t chunk = entry_chunk(e);
        const uint32_t j = entry_j(e), tpi = entry_tp(e);
        const uint64_t remaining = FROM_MAG ? p.n_samples : p.n_samples - chunk * kChunkSamples;
        WindowReader<FROM_MAG> win{p.src, chunk,
                                   remaining < (uint64_t)kChunkSamples ? (int)remaining : kChunkSamples,
                                   (int)j};
        uint32_t w[4];
        slice_message(win, 4 + (int)tpi, w);
        uint64_t power = 0;
        for (int k = 0; k < 33; k++) {
            const uint64_t m = win[19 + k];
            power += m * m;
        }
        TrialRecord r;
        r.power = power;
        r.chunk = (uint32_t)chunk;
        r.j_tp = j | ((4 + tpi) << 24);
#pragma unroll
        for (int b = 0; b < 14; b++) r.msg[b] = (uint8_t)msg_byte(w, b);
        r.pad = 0;
        rec[i] = r;
    }
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
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t bits = first_bits + i;
        const uint32_t u = mag_from_x(__uint_as_float(bits));
        sum += u;
        h ^= ((unsigned long long)u + 1ull) * (2ull * bits + 1ull);
    }
    atomicAdd(&out[0], sum);
    atomicXor(&out[1], h);
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
static inline int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

int launch_to_mag(const void *d_iq, uint32_t n, uint16_t *d_data, void *stream)
{
    const int blocks = (kMagDataLen + 255) / 256;
    hipLaunchKernelGGL(k_to_mag, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)d_iq, n, d_data);
    return hip_ok(hipGetLastError());
}

int launch_mag_digest(uint32_t first_bits, uint32_t count, unsigned long long *d_out, void *stream)
{
    hipLaunchKernelGGL(k_mag_digest, dim3(1024), dim3(256), 0, (hipStream_t)stream, first_bits, count,
                       d_out);
    return hip_ok(hipGetLastError());
}

int launch_scan(const ScanParams &p, bool from_mag, void *stream)
{
    const uint32_t blocks = p.n_chunks * kTilesPerChunk;
    if (blocks == 0) return 0;
    if (from_mag)
        hipLaunchKernelGGL(k_scan<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_scan<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_match(const ScanParams &p, void *stream)
{
    // grid-stride over a count only the device knows; sized for the usual ~2 % of samples
    uint64_t guess = p.n_samples / 32 + 1;
    uint32_t blocks = (uint32_t)((guess + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_match, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_ok(hipGetLastError());
}

int launch_records(const ScanParams &p, bool from_mag, TrialRecord *d_rec, void *stream)
{
    uint32_t blocks = p.n_chunks * 4 + 4;
    if (blocks > 4096) blocks = 4096;
    if (from_mag)
        hipLaunchKernelGGL(k_records<true>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, p, d_rec);
    else
        hipLaunchKernelGGL(k_records<false>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, p, d_rec);
    return hip_ok(hipGetLastError());
}

}  // namespace adsb
