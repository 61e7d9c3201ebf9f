// Can the two-neighbour fix-up behind v_sqrt_f32 (adsb_dev_common.h: mag_tail2) be replaced by a
// Markstein-style final correction  root = fma(x - s*s, 0.5/sqrt(x), s)  ?  Over every f32 bit pattern
// X in {0} U [1, 2^31] (a superset of what im^2 + rn(re^2) can be) count where each variant's
//   (a) root differs from the proven correctly rounded one,  (b) final u16 magnitude differs.
//   V2: s = v_sqrt(x),      h = 0.5 * v_rsq(x)
//   V3: s = x * v_rsq(x),   h = 0.5 * v_rsq(x)        (one transcendental)
//   V4: as V3, correction applied twice
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sqrt_markstein.hip -o /tmp/sqrt_markstein
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ uint32_t out16(float root)
{
    const float o = __fmaf_rn(root, 65535.0f / 32768.0f, 0.5f);
    const uint32_t u = (uint32_t)o;   // NaN -> 0
    return u > 65535u ? 65535u : u;
}
__device__ float exact_root(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t sb = __float_as_uint(s);
    const float sdn = __uint_as_float(sb - 1u), sup = __uint_as_float(sb + 1u);
    const float qdn = __fmaf_rn(sdn, s, -x), qup = __fmaf_rn(sup, s, -x);
    return __uint_as_float(sb - 1u + (__float_as_uint(qup) >> 31) + (__float_as_uint(qdn) >> 31));
}
__global__ void k(uint32_t first, unsigned long long count, unsigned long long *out)
{
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + (uint32_t)i);
        const float want = exact_root(x);
        const uint32_t w16 = out16(want);
        const float y = __builtin_amdgcn_rsqf(x), h = 0.5f * y;
        {
            const float s = __builtin_amdgcn_sqrtf(x);
            const float r = __fmaf_rn(__fmaf_rn(-s, s, x), h, s);
            c[0] += __float_as_uint(r) != __float_as_uint(want);
            c[1] += out16(r) != w16;
        }
        {
            const float s = x * y;
            const float r = __fmaf_rn(__fmaf_rn(-s, s, x), h, s);
            c[2] += __float_as_uint(r) != __float_as_uint(want);
            c[3] += out16(r) != w16;
            const float r2 = __fmaf_rn(__fmaf_rn(-r, r, x), h, r);
            c[4] += __float_as_uint(r2) != __float_as_uint(want);
            c[5] += out16(r2) != w16;
        }
        {   // no correction at all, for scale
            const float s = __builtin_amdgcn_sqrtf(x);
            c[6] += __float_as_uint(s) != __float_as_uint(want);
            c[7] += out16(s) != w16;
        }
    }
    for (int j = 0; j < 8; j++) atomicAdd(&out[j], c[j]);
}
int main()
{
    unsigned long long *d, h[8];
    (void)hipMalloc(&d, 64);
    const uint32_t lo = 0x3F800000u, hi = 0x4F000000u;  // 1.0f .. 2^31
    const char *name[4] = {"V2 sqrt+rsq", "V3 x*rsq", "V4 x*rsq twice", "raw v_sqrt"};
    for (int part = 0; part < 2; part++) {
        (void)hipMemset(d, 0, 64);
        if (part == 0) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 0u, 1ull, d);  // X = 0
        else hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, lo, (unsigned long long)(hi - lo) + 1ull, d);
        (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf(part == 0 ? "X = 0:\n" : "X in [1, 2^31], %llu values:\n", (unsigned long long)(hi - lo) + 1ull);
        for (int v = 0; v < 4; v++) printf("  %-16s root differs %12llu   u16 differs %12llu\n", name[v], h[2 * v], h[2 * v + 1]);
    }
    return 0;
}
