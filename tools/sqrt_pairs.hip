// Over every (|re|, |im|) in [0, 32768]^2 -- i.e. every X = rn(im^2 + rn(re^2)) the IQ path can
// produce -- how often does each half of the sqrt fix-up of adsb_dev_common.h matter?
//   down: the correctly rounded root is s - 1 ulp      up: it is s + 1 ulp
//   and for each, whether leaving that half out changes the final u16 magnitude.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sqrt_pairs.hip -o /tmp/sqrt_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ uint32_t out16(float root)
{
    const float o = __fmaf_rn(root, 65535.0f / 32768.0f, 0.5f);
    const uint32_t u = (uint32_t)o;
    return u > 65535u ? 65535u : u;
}
__global__ void k(unsigned long long *out)
{
    unsigned long long dn = 0, up = 0, dn_flips = 0, up_flips = 0, raw_flips = 0;
    const uint32_t q = blockIdx.x;  // |re|
    const float fq = (float)q, t = fq * fq;
    for (uint32_t i = threadIdx.x; i <= 32768u; i += blockDim.x) {
        const float fi = (float)i;
        const float x = __fmaf_rn(fi, fi, t);
        const float s = __builtin_amdgcn_sqrtf(x);
        const uint32_t sb = __float_as_uint(s);
        const float sdn = __uint_as_float(sb - 1u), sup = __uint_as_float(sb + 1u);
        const float qdn = __fmaf_rn(sdn, s, -x), qup = __fmaf_rn(sup, s, -x);
        const uint32_t d = (__float_as_uint(qdn) >> 31) ^ 1u, u = __float_as_uint(qup) >> 31;
        const uint32_t full = sb - d + u;  // == dnb + sign(qup) + sign(qdn)
        const uint32_t want = out16(__uint_as_float(full));
        dn += d & (x != 0.0f);
        up += u;
        dn_flips += out16(__uint_as_float(sb + u)) != want;
        up_flips += out16(__uint_as_float(sb - d)) != want;
        raw_flips += out16(s) != want;
    }
    atomicAdd(&out[0], dn); atomicAdd(&out[1], up); atomicAdd(&out[2], dn_flips);
    atomicAdd(&out[3], up_flips); atomicAdd(&out[4], raw_flips);
}
int main()
{
    unsigned long long *d, h[5] = {0, 0, 0, 0, 0};
    (void)hipMalloc(&d, 40); (void)hipMemset(d, 0, 40);
    hipLaunchKernelGGL(k, dim3(32769), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("pairs %llu: root is s-1ulp for %llu, s+1ulp for %llu; u16 changes without the down test: %llu, "
           "without the up test: %llu, with neither: %llu\n", 32769ull * 32769ull, h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
