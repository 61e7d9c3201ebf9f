// Probe (gfx950): what v_ashr_pk_i8_i32 writes, with and without op_sel:[0,0,0,1], and the
// sign-bytes -> plane-byte step with v_dot4_i32_i8 (weights -1,-2,-4,-8 / -16,-32,-64,-128).
// hipcc --offload-arch=gfx950 -O3 tools/ashr_pk_probe.hip -o /tmp/ashr_pk_probe && /tmp/ashr_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const int *in, uint32_t *out)
{
    const int a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
    uint32_t lo = 0xAAAAAAAAu, hi = 0xAAAAAAAAu, both = 0xAAAAAAAAu;
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 31" : "+v"(lo) : "v"(a), "v"(b));
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 31 op_sel:[0,0,0,1]" : "+v"(hi) : "v"(a), "v"(b));
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 31\n\tv_ashr_pk_i8_i32 %0, %2, %1, 31 op_sel:[0,0,0,1]" : "+v"(both) : "v"(a), "v"(b));
    uint32_t sh4 = 0;
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 4" : "+v"(sh4) : "v"(a), "v"(b));
    out[8 * threadIdx.x + 0] = lo;
    out[8 * threadIdx.x + 1] = hi;
    out[8 * threadIdx.x + 2] = both;
    out[8 * threadIdx.x + 3] = sh4;
    // plane byte from 8 signs: bits 0..3 from `both` = signs (a, b, b, a), bits 4..7 the same again
    const int w0 = (int)0xF8FCFEFFu;  // bytes -1, -2, -4, -8
    const int w1 = (int)0x80C0E0F0u;  // bytes -16, -32, -64, -128
    int acc = __builtin_amdgcn_sdot4((int)both, w0, 0, false);
    acc = __builtin_amdgcn_sdot4((int)both, w1, acc, false);
    out[8 * threadIdx.x + 4] = (uint32_t)acc;
}
int main()
{
    const int vals[] = {0, 1, -1, 5, -5, 127, 128, -128, -129, 2047, -2048, 1 << 20, -(1 << 20), 0x7FFFFFFF, (int)0x80000000, 100, -100, 16, -16, 15, -15, 17};
    const int nv = sizeof(vals) / sizeof(vals[0]);
    int in[2 * 64]; for (int i = 0; i < 64; i++) { in[2 * i] = vals[i % nv]; in[2 * i + 1] = vals[(i * 7 + 3) % nv]; }
    int *di; uint32_t *dout; (void)hipMalloc(&di, sizeof(in)); (void)hipMalloc(&dout, 64 * 8 * 4);
    (void)hipMemcpy(di, in, sizeof(in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    uint32_t out[64 * 8]; (void)hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 24; i++) {
        const int a = in[2 * i], b = in[2 * i + 1];
        const uint32_t sa = a < 0 ? 0xFFu : 0u, sb = b < 0 ? 0xFFu : 0u;
        const uint32_t want_lo = 0xAAAA0000u | sb << 8 | sa, want_hi = 0x0000AAAAu | sb << 24 | sa << 16;
        const uint32_t want_both = sa << 24 | sb << 16 | sb << 8 | sa;
        const uint32_t bits = (a < 0) | (b < 0) << 1 | (b < 0) << 2 | (a < 0) << 3;
        const uint32_t want_byte = bits | bits << 4;
        printf("a=%11d b=%11d  lo=%08x hi=%08x both=%08x sh4=%08x dot=%08x  %s%s%s%s\n", a, b, out[8 * i], out[8 * i + 1], out[8 * i + 2], out[8 * i + 3], out[8 * i + 4],
               out[8 * i] == want_lo ? "" : "LO? ", out[8 * i + 1] == want_hi ? "" : "HI? ", out[8 * i + 2] == want_both ? "" : "BOTH? ",
               (out[8 * i + 4] & 0xFF) == want_byte ? "" : "DOT? ");
        bad += out[8 * i] != want_lo || out[8 * i + 1] != want_hi || out[8 * i + 2] != want_both || (out[8 * i + 4] & 0xFF) != want_byte;
    }
    printf("%s\n", bad ? "SEMANTICS DIFFER FROM THE ASSUMPTION" : "as assumed: D.half = {sat_i8(S1>>S2), sat_i8(S0>>S2)}, op_sel[3] picks the half, the other half is kept");
    return 0;
}
