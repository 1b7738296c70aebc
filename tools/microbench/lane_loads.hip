// lane_loads.hip — what does a dependent random read cost when ONE lane fetches the whole granule (B bytes as B/16 dwordx4
// loads of one line), as the search kernel's one-chain-per-lane form does?  Rate of granules/s for B = 16, 32, 64, 128, and
// for B = 64 with K extra load instructions per step that only 4 lanes execute (the kernel's rare states).
// Build: hipcc --offload-arch=gfx950 -O3 -o lane_loads lane_loads.hip ; run: ./lane_loads [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; return x * 0x2545f4914f6cdd1dull; }

template <int B, int EXTRA>
__global__ void __launch_bounds__(256) chase(const uint8_t *buf, uint64_t nGran, uint32_t steps, unsigned long long *sink) {
    constexpr int PER = B / 16;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = mix(0x9e3779b97f4a7c15ull * (tid + 1));
    unsigned long long acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint64_t g = mix(x) % nGran;
        const uint8_t *p = buf + g * B;
        ulonglong2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = *reinterpret_cast<const ulonglong2 *>(p + 16 * i);
        uint32_t f = 0;
        if (EXTRA) {                      // K more load instructions, each executed by 4 lanes only, other random lines
#pragma unroll
            for (int k = 0; k < EXTRA; k++)
                if ((threadIdx.x & 63) >> 2 == (uint32_t)k) {
                    const uint64_t g2 = mix(x + k + 1) % nGran;
                    f += (uint32_t)*reinterpret_cast<const uint64_t *>(buf + g2 * B);
                }
        }
#pragma unroll
        for (int i = 0; i < PER; i++) f += (uint32_t)__popcll(v[i].x) + (uint32_t)__popcll(v[i].y);
        x = x * 6364136223846793005ull + 1442695040888963407ull + (f & 3u);
        acc += f;
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
__global__ void fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = mix(i + 1);
}
template <int B, int EXTRA>
void run(const uint8_t *buf, uint64_t bytes, int blocks, uint32_t steps, unsigned long long *sink, const char *tag) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase<B, EXTRA>), dim3(blocks), dim3(256), 0, 0, buf, bytes / B, 4u, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chase<B, EXTRA>), dim3(blocks), dim3(256), 0, 0, buf, bytes / B, steps, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * 256 * steps;
    printf("%-34s blocks/CU %d  %7.2f G granules/s\n", tag, blocks / 256, loads / (ms * 1e-3) / 1e9);
    fflush(stdout);
}
int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 128 * 128;
    uint8_t *buf; unsigned long long *sink;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
    CK(hipDeviceSynchronize());
    printf("# buffer %.1f GiB; dependent random reads, ONE lane per chain, 128 steps\n", gib);
    for (int bpc : {4, 8}) {
        const int blocks = 256 * bpc;
        run<16, 0>(buf, bytes, blocks, 128, sink, "16 B  (1 load)");
        run<32, 0>(buf, bytes, blocks, 128, sink, "32 B  (2 loads)");
        run<64, 0>(buf, bytes, blocks, 128, sink, "64 B  (4 loads)");
        run<128, 0>(buf, bytes, blocks, 128, sink, "128 B (8 loads)");
        run<64, 4>(buf, bytes, blocks, 128, sink, "64 B + 4 four-lane loads");
        run<64, 12>(buf, bytes, blocks, 128, sink, "64 B + 12 four-lane loads");
    }
    return 0;
}
