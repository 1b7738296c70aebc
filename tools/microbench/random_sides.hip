// random_sides.hip — roofline denominator experiments for the FM-index walk:
// dependent chains of random 128-byte "side" reads over a buffer far larger than
// the 256 MiB Infinity Cache.  Each chain's next address depends on the data it
// just loaded (as an LF step does).  Variants: G lanes cooperate on one 128-byte
// side (G = 8,4,2,1: 16,32,64,128 bytes per lane), C independent chains per
// lane group (ILP), blocks per CU (occupancy).
// Build: hipcc --offload-arch=gfx950 -O3 -o random_sides random_sides.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    return x * 0x2545f4914f6cdd1dull;
}

template <int G, int C>
__global__ void __launch_bounds__(256) chase(const uint8_t *buf, uint64_t nSides, uint32_t steps, unsigned long long *sink) {
    constexpr int PER = 8 / G;                   // 16-byte chunks per lane
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    uint64_t x[C];
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = mix(0x9e3779b97f4a7c15ull * (grp * C + c + 1));
    unsigned long long acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        ulonglong2 v[C][PER];
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint64_t side = mix(x[c]) % nSides;
            const uint8_t *p = buf + side * 128 + 16 * PER * sub;
#pragma unroll
            for (int i = 0; i < PER; i++) v[c][i] = *reinterpret_cast<const ulonglong2 *>(p + 16 * i);
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t f = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) f += (uint32_t)__popcll(v[c][i].x) + (uint32_t)__popcll(v[c][i].y);
#pragma unroll
            for (int m = 1; m < G; m <<= 1) f += __shfl_xor(f, m, 64);
            x[c] = x[c] * 6364136223846793005ull + 1442695040888963407ull + (f & 3u);
            acc += f;
        }
    }
    if (acc == 0x1234567u) sink[0] = acc;
}

template <int G, int C>
double run(const uint8_t *buf, uint64_t nSides, int blocks, uint32_t steps, unsigned long long *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase<G, C>), dim3(blocks), dim3(256), 0, 0, buf, nSides, 4u, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chase<G, C>), dim3(blocks), dim3(256), 0, 0, buf, nSides, steps, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * 256 / G * C * steps;
    return loads * 128.0 / (ms * 1e-3) / 1e9;
}

__global__ void fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = mix(i + 1);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const uint32_t steps = argc > 2 ? atoi(argv[2]) : 256;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 128 * 128;
    uint8_t *buf; unsigned long long *sink;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const uint64_t nSides = bytes / 128;
    printf("# buffer %.2f GiB, %d CUs, steps %u; GB/s of 128-byte random dependent reads\n", gib, cus, steps);
    printf("%-6s %-3s %-4s %10s\n", "G", "C", "bpc", "GB/s");
    for (int bpc : {2, 4, 6, 8}) {
        const int blocks = cus * bpc;
#define R(G, C) printf("%-6d %-3d %-4d %10.1f\n", G, C, bpc, run<G, C>(buf, nSides, blocks, steps, sink)); fflush(stdout);
        R(8, 1) R(8, 2) R(8, 4)
        R(4, 1) R(4, 2) R(4, 4)
        R(2, 1) R(2, 2)
        R(1, 1) R(1, 2)
    }
    return 0;
}
