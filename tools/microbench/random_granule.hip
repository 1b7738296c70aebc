// random_granule.hip — does the rate of dependent random reads (requests/s) hold when the granule shrinks from the 128-byte
// side to 64 or 32 bytes?  (VERDICT r1 #3b: a load-time re-layout of the index to 64-byte half-sides pays only if it does.)
// Chains of 2 lanes; a chain reads one aligned granule of B bytes per step (each lane B/2 bytes as 16-byte loads; B = 32:
// 16 bytes per lane) and derives the next address from what it read.  Also: B = 128 + a second, cached read (the
// super-block base a u32-occ layout needs).
// Build: hipcc --offload-arch=gfx950 -O3 -o random_granule random_granule.hip ; run: ./random_granule [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; return x * 0x2545f4914f6cdd1dull; }

template <int B, bool SUPER>
__global__ void __launch_bounds__(256) chase(const uint8_t *buf, uint64_t nGran, const uint64_t *super, uint32_t steps, unsigned long long *sink) {
    constexpr int PER = B / 32;                  // 16-byte chunks per lane (2 lanes per chain)
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid >> 1, sub = tid & 1;
    uint64_t x = mix(0x9e3779b97f4a7c15ull * (grp + 1));
    unsigned long long acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint64_t g = mix(x) % nGran;
        const uint8_t *p = buf + g * B + 16 * PER * sub;
        ulonglong2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = *reinterpret_cast<const ulonglong2 *>(p + 16 * i);
        uint32_t f = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) f += (uint32_t)__popcll(v[i].x) + (uint32_t)__popcll(v[i].y);
        if (SUPER) f += (uint32_t)super[(g >> 16) & 1023];
        f += __shfl_xor(f, 1, 64);
        x = x * 6364136223846793005ull + 1442695040888963407ull + (f & 3u);
        acc += f;
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
__global__ void fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = mix(i + 1);
}
template <int B, bool SUPER>
void run(const uint8_t *buf, uint64_t bytes, const uint64_t *super, int blocks, uint32_t steps, unsigned long long *sink, const char *tag) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase<B, SUPER>), dim3(blocks), dim3(256), 0, 0, buf, bytes / B, super, 4u, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chase<B, SUPER>), dim3(blocks), dim3(256), 0, 0, buf, bytes / B, super, steps, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * 128 * steps;
    printf("%-28s blocks/CU %d  %7.2f G requests/s  %8.1f GB/s\n", tag, blocks / 256, loads / (ms * 1e-3) / 1e9, loads * B / (ms * 1e-3) / 1e9);
    fflush(stdout);
}
int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 128 * 128;
    uint8_t *buf; unsigned long long *sink; uint64_t *super;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 8)); CK(hipMalloc((void **)&super, 8192));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
    hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, 0, super, 1024);
    CK(hipDeviceSynchronize());
    printf("# buffer %.1f GiB; dependent random reads, 2 lanes per chain, 256 steps\n", gib);
    for (int bpc : {4, 8, 12, 16}) {
        const int blocks = 256 * bpc;
        run<128, false>(buf, bytes, super, blocks, 256, sink, "128 B granule");
        run<64, false>(buf, bytes, super, blocks, 256, sink, "64 B granule");
        run<32, false>(buf, bytes, super, blocks, 256, sink, "32 B granule");
        run<64, true>(buf, bytes, super, blocks, 256, sink, "64 B + cached super-block");
    }
    return 0;
}
