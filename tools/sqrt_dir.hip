// How does v_sqrt_f32 err on X in [1, 2^31]?  Counts where the correctly rounded root is
// s-1ulp / s / s+1ulp (fix-up test of adsb_dev_common.h).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t first, uint32_t count, unsigned long long *out)
{
    unsigned long long dn = 0, up = 0, same = 0, both = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        float x = __uint_as_float(first + i);
        float s = __builtin_amdgcn_sqrtf(x);
        float sdn = __uint_as_float(__float_as_uint(s) - 1u), sup = __uint_as_float(__float_as_uint(s) + 1u);
        float qdn = __fmaf_rn(sdn, s, -x), qup = __fmaf_rn(sup, s, -x);
        bool d = !(__float_as_uint(qdn) >> 31), u = __float_as_uint(qup) >> 31;
        dn += d; up += u; both += (d && u); same += (!d && !u);
    }
    atomicAdd(&out[0], dn); atomicAdd(&out[1], up); atomicAdd(&out[2], same); atomicAdd(&out[3], both);
}
int main()
{
    unsigned long long *d, h[4] = {0, 0, 0, 0};
    (void)hipMalloc(&d, 32); (void)hipMemset(d, 0, 32);
    const uint32_t lo = 0x3F800000u, hi = 0x4F000000u;
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, lo, hi - lo + 1, d);
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("down %llu  up %llu  same %llu  both %llu  of %u\n", h[0], h[1], h[2], h[3], hi - lo + 1);
    return 0;
}
