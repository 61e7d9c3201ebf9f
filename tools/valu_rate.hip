// Micro-benchmark: issue rate of the VALU ops the scan kernel leans on (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[CHAINS];
    float f[CHAINS];
    double d2[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i] = seed + threadIdx.x * 7 + i; f[i] = (float)a[i]; d2[i] = (double)a[i]; }
    uint32_t b = seed * 3 + 1;
    float fb = 1.0001f;
    double db = 1.0001;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_lshrrev_b32_e32 %0, 16, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_ashrrev_i32_e32 %0, 31, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_mul_u32_u24_e32 %0, 5, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_mul_i32_i24_e32 %0, 5, %0" : "+v"(a[i]));
            if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, 5, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_mad_i32_i24 %0, %0, 5, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
            if (OP == 18) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 19) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b));
            if (OP == 22) asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
            if (OP == 23) asm volatile("v_cmp_gt_i32_e64 s[10:11], %0, %1" :: "v"(a[i]), "v"(b) : "s10","s11");
            if (OP == 24) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) :: "vcc");
            if (OP == 25) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b));
            if (OP == 26) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 27) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 28) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 29) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 30) asm volatile("v_fma_f32 %0, -%0, %1, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 31) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
            if (OP == 32) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 33) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 34) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(f[i]));
            if (OP == 35) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(f[i]));
            if (OP == 36) asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if (OP == 37) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[i]) : "v"(a[i]));
            if (OP == 38) asm volatile("v_cvt_u32_f32_e32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            if (OP == 39) asm volatile("v_cvt_pk_u16_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 40) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
            if (OP == 41) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d2[i]) : "v"(db));
            if (OP == 42) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d2[i]) : "v"(db));
            if (OP == 43) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 44) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 45) asm volatile("v_pk_mad_i16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 46) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 47) asm volatile("v_mad_u32_u16 %0, %0, %1, %1 op_sel:[1,0,0,0]" : "+v"(a[i]) : "v"(b));
            if (OP == 48) asm volatile("v_ashr_pk_i8_i32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 49) asm volatile("v_dot4_i32_i8 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 50) asm volatile("v_dot4c_i32_i8_e32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 51) asm volatile("v_dot2_i32_i16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 52) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 53) asm volatile("v_sad_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 54) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[i]) : "v"(b));
            if (OP == 55) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 56) asm volatile("v_ffbl_b32_e32 %0, %0" : "+v"(a[i]));
            if (OP == 57) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (OP == 58) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[i]));
            if (OP == 59) asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a[i]) : "v"(b));
            if (OP == 60) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d2[i]) : "v"(db));
            if (OP == 61) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 62) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 63) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d2[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> void run(const char *name, uint32_t *d)
{
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU -> 8 waves/SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITERS * CHAINS;       // wave-instructions
    double per_simd_cycle = winstr / (ms * 1e-3) / (256.0 * 4) / 2.4e9;
    printf("%-20s %8.3f ms  %6.2f T lane-ops/s  cycles/instr/SIMD@2.4GHz %5.2f\n", name, ms,
           winstr * 64 / ms / 1e9, 1.0 / per_simd_cycle);
}

int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32_e32", d);
    run<1>("v_sub_u32_e32", d);
    run<2>("v_and_b32_e32", d);
    run<3>("v_or_b32_e32", d);
    run<4>("v_xor_b32_e32", d);
    run<5>("v_lshlrev_b32_e32", d);
    run<6>("v_lshrrev_b32_e32", d);
    run<7>("v_ashrrev_i32_e32", d);
    run<8>("v_mul_u32_u24_e32", d);
    run<9>("v_mul_i32_i24_e32", d);
    run<10>("v_mad_u32_u24", d);
    run<11>("v_mad_i32_i24", d);
    run<12>("v_add3_u32", d);
    run<13>("v_lshl_add_u32", d);
    run<14>("v_lshl_or_b32", d);
    run<15>("v_and_or_b32", d);
    run<16>("v_alignbit_b32", d);
    run<17>("v_bfe_u32", d);
    run<18>("v_bfi_b32", d);
    run<19>("v_perm_b32", d);
    run<20>("v_cndmask_e32_vcc", d);
    run<21>("v_cndmask_e64_s", d);
    run<22>("v_cmp_gt_i32_e32", d);
    run<23>("v_cmp_gt_i32_e64", d);
    run<24>("v_addc_co_u32", d);
    run<25>("v_mov_b32_e32", d);
    run<26>("v_add_f32_e32", d);
    run<27>("v_mul_f32_e32", d);
    run<28>("v_fmac_f32_e32", d);
    run<29>("v_fma_f32", d);
    run<30>("v_fma_f32_neg", d);
    run<31>("v_min_f32_e32", d);
    run<32>("v_max_i32_e32", d);
    run<33>("v_max3_i32", d);
    run<34>("v_sqrt_f32_e32", d);
    run<35>("v_rsq_f32_e32", d);
    run<36>("v_cvt_f32_i32_e32", d);
    run<37>("v_cvt_f32_i32_sdwa", d);
    run<38>("v_cvt_u32_f32_e32", d);
    run<39>("v_cvt_pk_u16_u32", d);
    run<40>("v_sub_u32_sdwa", d);
    run<41>("v_pk_fma_f32", d);
    run<42>("v_pk_mul_f32", d);
    run<43>("v_pk_add_u16", d);
    run<44>("v_pk_sub_i16", d);
    run<45>("v_pk_mad_i16", d);
    run<46>("v_pk_max_i16", d);
    run<47>("v_mad_u32_u16", d);
    run<48>("v_ashr_pk_i8_i32", d);
    run<49>("v_dot4_i32_i8", d);
    run<50>("v_dot4c_i32_i8", d);
    run<51>("v_dot2_i32_i16", d);
    run<52>("v_cvt_pk_i16_i32", d);
    run<53>("v_sad_u16", d);
    run<54>("v_bitop3_b32", d);
    run<55>("v_mbcnt_lo", d);
    run<56>("v_ffbl_b32", d);
    run<57>("v_add_u32_dpp", d);
    run<58>("v_bfe_i32", d);
    run<59>("v_lshrrev_sdwa", d);
    run<60>("v_pk_add_f32", d);
    run<61>("v_mul_lo_u32", d);
    run<62>("v_min3_i32", d);
    run<63>("v_or3_b32", d);
    return 0;
}
