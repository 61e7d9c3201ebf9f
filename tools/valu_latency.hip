// Issue rate of ONE wave per SIMD as a function of independent chains (dependent-op latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
template <int OP, int CH>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[CH]; float f[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 7 + i; f[i] = (float)a[i]; }
    uint32_t b = seed * 3 + 1; float fb = 1.0001f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 3) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(f[i]));
            if (OP == 4) asm volatile("v_mad_i32_i24 %0, %0, 5, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CH; i++) r ^= a[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP, int CH> void run(const char *name, uint32_t *d, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // one wave per SIMD per block-per-CU: cycles per instruction per wave
    double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * CH);
    printf("%-14s chains %d  waves/SIMD %d  cycles/instr/wave(@2.4GHz) %6.2f\n", name, CH, blocks_per_cu, cyc);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0,1>("v_alignbit", d, 1); run<0,2>("v_alignbit", d, 1); run<0,4>("v_alignbit", d, 1); run<0,8>("v_alignbit", d, 1);
    run<1,1>("v_add_u32", d, 1); run<1,2>("v_add_u32", d, 1); run<1,4>("v_add_u32", d, 1); run<1,8>("v_add_u32", d, 1);
    run<2,1>("v_fma_f32", d, 1); run<2,4>("v_fma_f32", d, 1); run<2,8>("v_fma_f32", d, 1);
    run<3,1>("v_sqrt_f32", d, 1); run<3,4>("v_sqrt_f32", d, 1); run<3,8>("v_sqrt_f32", d, 1);
    run<4,1>("v_mad_i32_i24", d, 1); run<4,8>("v_mad_i32_i24", d, 1);
    run<0,1>("v_alignbit", d, 4); run<0,8>("v_alignbit", d, 4); run<1,8>("v_add_u32", d, 2); run<1,8>("v_add_u32", d, 4);
    return 0;
}
