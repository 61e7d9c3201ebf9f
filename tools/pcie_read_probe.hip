// Micro-benchmark: how fast a kernel reads pinned host memory in place over the link (gfx950), by
// allocation kind, cache-policy bits of the load, loads in flight per thread and grid size -- the one-launch
// pass reads ring slots this way (csrc/adsb_scan_fast.hip: load_tile_iq), the copy engine moves ~52 GB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_read_probe.hip -o /tmp/pcie_read_probe && /tmp/pcie_read_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

// each workgroup takes 4 KB * DEPTH at a time (256 threads x DEPTH loads of 16 bytes, all issued before the
// first is consumed), grid-stride over the buffer
template <int AUX, int DEPTH>
__global__ __launch_bounds__(256) void k_read(const void *src, uint32_t bytes, uint32_t *out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
    const uint32_t piece = 4096u * DEPTH;
    uint32_t acc = 0;
    for (uint32_t base = blockIdx.x * piece; base < bytes; base += gridDim.x * piece) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            int off = (int)(base + 4096u * i + 16u * threadIdx.x);
            asm volatile("" : "+v"(off));
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX);
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int AUX, int DEPTH>
static double run(const void *dsrc, uint32_t bytes, int grid, uint32_t *dout)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    k_read<AUX, DEPTH><<<grid, 256>>>(dsrc, bytes, dout);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        k_read<AUX, DEPTH><<<grid, 256>>>(dsrc, bytes, dout);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return bytes / (best * 1e-3) / 1e9;
}

// A pass-shaped kernel: 17 workgroups read 512 KB in place (all loads issued at once), then "compute" for
// spin_us, then (optionally) write a line of results back to host memory; launched round robin on n_streams
// streams over 8 distinct pinned buffers: what the one-buffer ring does, without the demodulator.
__global__ __launch_bounds__(256) void k_pass(const void *src, uint32_t bytes, uint32_t spin_ticks, uint32_t *host_out, uint32_t *out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes & 0x0FFFFFFFu, 0x00020000);
    u32x4 v[8];
    const uint32_t base = blockIdx.x * 32768u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int off = (int)(base + 4096u * i + 16u * threadIdx.x);
        asm volatile("" : "+v"(off));
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    // what a one-launch pass does between its stages: a store other workgroups read, then a fence
    const uint32_t fence_kind = bytes >> 28;   // (smuggled in the top bits of `bytes`)
    if (fence_kind) {
        out[64 + blockIdx.x * 64 + (threadIdx.x & 63)] = acc;
        if (fence_kind == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (fence_kind == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (fence_kind == 3) __threadfence();
        if (fence_kind == 4) __threadfence_system();
        acc ^= out[64 + ((blockIdx.x + 1) % gridDim.x) * 64 + (threadIdx.x & 63)];
    }
    if (host_out && threadIdx.x == 0) __hip_atomic_store(&host_out[blockIdx.x * 16], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (acc == 0x12345678u) out[0] = acc;
}

static void pass_shaped(uint32_t *dout)
{
    constexpr int kBuf = 8;
    void *h[kBuf], *d[kBuf];
    uint32_t *hout, *dhout;
    CHECK(hipHostMalloc((void **)&hout, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&dhout, hout, 0));
    for (int k = 0; k < kBuf; k++) {
        CHECK(hipHostMalloc(&h[k], 17 * 32768, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h[k], k + 1, 17 * 32768);
        CHECK(hipHostGetDevicePointer(&d[k], h[k], 0));
    }
    hipStream_t q[4];
    for (auto &s : q) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::printf("pass-shaped kernels (17 workgroups x 32 KB in place, then a spin), 4000 launches round robin:\n");
    for (int spin_us : {0, 20})
        for (int ns : {1, 4})
            for (int wr = 0; wr < 5; wr++) {
                for (int warm = 0; warm < 2; warm++) {
                    CHECK(hipDeviceSynchronize());
                    hipEvent_t a, b;
                    CHECK(hipEventCreate(&a));
                    CHECK(hipEventCreate(&b));
                    CHECK(hipEventRecord(a, q[0]));
                    const int n = 4000;
                    for (int i = 0; i < n; i++)
                        k_pass<<<17, 256, 0, q[i % ns]>>>(d[i % kBuf], 17 * 32768 + ((uint32_t)wr << 28), (uint32_t)spin_us * 100u, nullptr, dout);
                    for (int k = 0; k < ns; k++) CHECK(hipStreamSynchronize(q[k]));
                    CHECK(hipEventRecord(b, q[0]));
                    CHECK(hipEventSynchronize(b));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, a, b));
                    if (warm)
                        std::printf("  spin %2d us, %d stream(s)%s: %6.2f us per launch = %5.1f GB/s\n", spin_us, ns,
                                    wr == 0 ? ", no fence" : wr == 1 ? ", release fence (agent)" : wr == 2 ? ", acquire fence (agent)" : wr == 3 ? ", __threadfence()" : ", __threadfence_system()", ms * 1e3 / n, 17 * 32768.0 * n / (ms * 1e-3) / 1e9);
                    CHECK(hipEventDestroy(a));
                    CHECK(hipEventDestroy(b));
                }
            }
}

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "pass")) {
        uint32_t *o;
        CHECK(hipMalloc(&o, 1 << 20));
        pass_shaped(o);
        return 0;
    }

    const uint32_t bytes = 64u << 20;
    uint32_t *dout;
    CHECK(hipMalloc(&dout, 64));
    struct Kind { const char *name; unsigned flags; bool reg; } kinds[] = {
        {"hipHostMalloc mapped|coherent (the ring's)", hipHostMallocMapped | hipHostMallocCoherent, false},
        {"hipHostMalloc mapped|non-coherent", hipHostMallocMapped | hipHostMallocNonCoherent, false},
        {"hipHostMalloc default", hipHostMallocDefault, false},
        {"malloc + hipHostRegister mapped", hipHostRegisterMapped, true},
    };
    void *dev;
    CHECK(hipMalloc(&dev, bytes));
    for (const Kind &kd : kinds) {
        void *h = nullptr, *d = nullptr;
        if (kd.reg) {
            if (posix_memalign(&h, 4096, bytes)) return 1;
            std::memset(h, 1, bytes);
            CHECK(hipHostRegister(h, bytes, kd.flags));
        } else {
            CHECK(hipHostMalloc(&h, bytes, kd.flags));
            std::memset(h, 1, bytes);
        }
        CHECK(hipHostGetDevicePointer(&d, h, 0));
        std::printf("%s\n", kd.name);
        // copy engine for reference
        {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            CHECK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            CHECK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, 0));
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::printf("  hipMemcpyAsync H2D:                         %6.1f GB/s\n", bytes / (ms * 1e-3) / 1e9);
        }
        for (int grid : {17, 68, 272, 1024}) {
            std::printf("  grid %4d: aux 0 depth 1/2/4/8: %5.1f %5.1f %5.1f %5.1f | depth 8, aux sc0 / nt / sc0+nt / sc1 / sc0+sc1 / sc0+sc1+nt: "
                        "%5.1f %5.1f %5.1f %5.1f %5.1f %5.1f GB/s\n",
                        grid, run<0, 1>(d, bytes, grid, dout), run<0, 2>(d, bytes, grid, dout), run<0, 4>(d, bytes, grid, dout),
                        run<0, 8>(d, bytes, grid, dout), run<1, 8>(d, bytes, grid, dout), run<2, 8>(d, bytes, grid, dout),
                        run<3, 8>(d, bytes, grid, dout), run<16, 8>(d, bytes, grid, dout), run<17, 8>(d, bytes, grid, dout),
                        run<19, 8>(d, bytes, grid, dout));
        }
        if (kd.reg) {
            CHECK(hipHostUnregister(h));
            std::free(h);
        } else {
            CHECK(hipHostFree(h));
        }
    }
    return 0;
}
