// d2h_overlap.hip — what a device-to-host result copy does to kernels that run beside it (profiles/r02d: the runtime
// serves hipMemcpyAsync D2H with a blit KERNEL, and write-heavy kernels of the next batch ran 10-50x slower while it
// was in flight).  Victims: a streaming writer (like the prefix-sum write pass) and a dependent random 128-byte reader
// with scattered 32-byte stores (like the search kernel).  Egress variants: hipMemcpyAsync, and an own copy kernel that
// stores into the pinned buffer directly with B blocks (throttled: PCIe needs few stores in flight).
// Build: hipcc -O2 --offload-arch=gfx950 -o d2h_overlap d2h_overlap.hip ; run: ./d2h_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4 v, uint4 *p) { v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p)); }
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_writer(uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) k_chaser(const uint4 *src, size_t nLines, uint4 *hits, size_t nHits, int steps, unsigned long long *sink) {
    const unsigned g = (blockIdx.x * blockDim.x + threadIdx.x) >> 1, sub = threadIdx.x & 1;
    unsigned long long x = 0x9e3779b97f4a7c15ull * (g + 1), acc = 0;
    for (int s = 0; s < steps; s++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const size_t line = (x * 0x2545f4914f6cdd1dull) % nLines;
        uint4 a = src[line * 8 + sub * 4], b = src[line * 8 + sub * 4 + 1], c = src[line * 8 + sub * 4 + 2], d = src[line * 8 + sub * 4 + 3];
        unsigned f = a.x ^ b.y ^ c.z ^ d.w;
        f += __shfl_xor(f, 1, 64);
        x += f & 1u; acc += f;
        if ((s & 7) == 0 && sub == 0) {                        // a hit record every 8 steps
            const size_t h = (x >> 7) % nHits;
            nt_store(make_uint4(f, 1, 2, 3), &hits[2 * h]);
            nt_store(make_uint4(f, 4, 5, 6), &hits[2 * h + 1]);
        }
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
// own egress: B blocks stream `n16` 16-byte words from HBM into pinned host memory
__global__ void __launch_bounds__(256) k_egress(const uint4 *src, uint4 *hostDst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        nt_store(src[i], &hostDst[i]);
}

int main() {
    OK(hipSetDevice(0));
    const size_t copyBytes = 400u << 20, victimBytes = 2048ull << 20, poolBytes = 4096ull << 20, hitBytes = 2048ull << 20;
    uint4 *dCopy, *dVictim, *dPool, *dHits, *hPinned; unsigned long long *sink;
    OK(hipMalloc((void **)&dCopy, copyBytes)); OK(hipMalloc((void **)&dVictim, victimBytes)); OK(hipMalloc((void **)&dPool, poolBytes));
    OK(hipMalloc((void **)&dHits, hitBytes)); OK(hipMalloc((void **)&sink, 8));
    OK(hipHostMalloc((void **)&hPinned, copyBytes, hipHostMallocDefault));
    OK(hipMemset(dCopy, 1, copyBytes)); OK(hipMemset(dPool, 3, poolBytes));
    hipStream_t sK, sC; OK(hipStreamCreateWithFlags(&sK, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
    hipEvent_t a, b, c, d, dep; OK(hipEventCreate(&dep)); OK(hipEventCreate(&a)); OK(hipEventCreate(&b)); OK(hipEventCreate(&c)); OK(hipEventCreate(&d));
    auto victim = [&](int which) {
        if (which == 0) hipLaunchKernelGGL(k_writer, dim3(2048), dim3(256), 0, sK, dVictim, victimBytes / 16);
        else hipLaunchKernelGGL(k_chaser, dim3(2048), dim3(256), 0, sK, dPool, poolBytes / 128, dHits, hitBytes / 32, 400, sink);
    };
    // egress variant e: 0 none, 1 hipMemcpyAsync, >= 2: own kernel with e blocks
    auto run = [&](int which, int e, const char *tag) {
        float best = 1e9f, bestC = 0;
        for (int rep = 0; rep < 3; rep++) {
            OK(hipDeviceSynchronize());
            if (e == -1) {                                     // the copy waits (on the device) for a kernel of the other stream
                hipLaunchKernelGGL(k_writer, dim3(256), dim3(256), 0, sK, dVictim, (size_t)(64u << 20) / 16);
                OK(hipEventRecord(dep, sK));
                OK(hipStreamWaitEvent(sC, dep, 0));
            }
            OK(hipEventRecord(c, sC));
            if (e == -2) OK(hipMemcpyAsync(hPinned, dCopy, 64, hipMemcpyDeviceToHost, sC));      // a tiny copy in front of the big one
            if (e == -3) {                                     // the big one in four pieces
                for (int q = 0; q < 4; q++) OK(hipMemcpyAsync((char *)hPinned + q * (copyBytes / 4), (char *)dCopy + q * (copyBytes / 4), copyBytes / 4, hipMemcpyDeviceToHost, sC));
            }
            if (e == 1 || e == -1 || e == -2) OK(hipMemcpyAsync(hPinned, dCopy, copyBytes, hipMemcpyDeviceToHost, sC));
            else if (e >= 2) hipLaunchKernelGGL(k_egress, dim3(e), dim3(256), 0, sC, dCopy, hPinned, copyBytes / 16);
            OK(hipEventRecord(d, sC));
            OK(hipEventRecord(a, sK));
            victim(which);
            if (which == 0) { victim(which); victim(which); victim(which); }       // ~4 x 2 GiB of writes: longer than the copy
            OK(hipEventRecord(b, sK));
            OK(hipDeviceSynchronize());
            float ms, msC; OK(hipEventElapsedTime(&ms, a, b)); OK(hipEventElapsedTime(&msC, c, d));
            if (ms < best) { best = ms; bestC = msC; }
        }
        std::printf("victim %-7s egress %-22s victim %.2f ms, egress %.2f ms (%.1f GB/s)\n", which ? "chaser" : "writer", tag, best, bestC,
                    e ? copyBytes / (bestC * 1e-3) / 1e9 : 0.0);
        std::fflush(stdout);
    };
    for (int which = 0; which < 2; which++) {
        run(which, 0, "none");
        run(which, 1, "hipMemcpyAsync");
        run(which, -1, "memcpy after WaitEvent");
        run(which, -2, "tiny + big memcpy");
        run(which, -3, "4 memcpys");
        run(which, 4, "own kernel, 4 blocks");
        run(which, 16, "own kernel, 16 blocks");
        run(which, 64, "own kernel, 64 blocks");
        run(which, 512, "own kernel, 512 blocks");
    }
    return 0;
}
