// fetch_calib.hip — what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for THIS path's access pattern?  (MI355X_MICROARCH.md: the
// counters are calibrated for wide coalesced streaming reads only — FETCH_SIZE = 1/2 of the bytes there —, "calibrate on a known
// byte count in your own access pattern before trusting an absolute".)  Kernels with a known number of requests and bytes:
//   calib_rand<16> / <8>   every lane loads `steps` independent pseudo-random aligned 16- / 8-byte granules of a buffer far larger
//                          than L2 + Infinity Cache (the search kernel's plane / sample / wide-ftab requests)
//   calib_rand_store16     ... stores 16 bytes to them (the hit records)
//   calib_stream16         a coalesced streaming read of the whole buffer, 16 bytes per lane (the guide's reference case)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/pmc_calib.py divides the counters by the
// request counts printed here.  Build: hipcc --offload-arch=gfx950 -O3 -o bin/fetch_calib fetch_calib.hip ; run: bin/fetch_calib [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; return x ^ (x >> 31); }

template <int B>
__global__ void __launch_bounds__(256) calib_rand(const uint8_t *buf, uint64_t nGran, uint32_t steps, unsigned long long *sink) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint64_t g = mix(tid * 1315423911ull + s * 0x9e3779b97f4a7c15ull + 1) % nGran;
        if (B == 16) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(buf + g * 16); acc += v.x ^ v.y; }
        else acc += *reinterpret_cast<const unsigned long long *>(buf + g * 8);
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) calib_rand_store16(uint8_t *buf, uint64_t nGran, uint32_t steps) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (uint32_t s = 0; s < steps; s++) {
        const uint64_t g = mix(tid * 1315423911ull + s * 0x9e3779b97f4a7c15ull + 7) % nGran;
        ulonglong2 v; v.x = tid; v.y = s;
        *reinterpret_cast<ulonglong2 *>(buf + g * 16) = v;
    }
}
__global__ void __launch_bounds__(256) calib_stream16(const uint8_t *buf, uint64_t n16, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(buf + i * 16);
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
__global__ void fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = mix(i + 1);
}
int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 4096 * 4096;
    uint8_t *buf; unsigned long long *sink;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 8));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
    CK(hipDeviceSynchronize());
    const int blocks = 256 * 16;
    const uint32_t steps = 256;
    const uint64_t req = (uint64_t)blocks * 256 * steps;
    hipLaunchKernelGGL(calib_rand<16>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, steps, sink);
    CK(hipDeviceSynchronize());
    printf("calib_rand<16> requests %llu bytes_requested %llu\n", (unsigned long long)req, (unsigned long long)req * 16);
    hipLaunchKernelGGL(calib_rand<8>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 8, steps, sink);
    CK(hipDeviceSynchronize());
    printf("calib_rand<8> requests %llu bytes_requested %llu\n", (unsigned long long)req, (unsigned long long)req * 8);
    hipLaunchKernelGGL(calib_stream16, dim3(256 * 32), dim3(256), 0, 0, buf, bytes / 16, sink);
    CK(hipDeviceSynchronize());
    printf("calib_stream16 requests %llu bytes_requested %llu\n", (unsigned long long)(bytes / 16), (unsigned long long)bytes);
    hipLaunchKernelGGL(calib_rand_store16, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, steps);
    CK(hipDeviceSynchronize());
    printf("calib_rand_store16 requests %llu bytes_requested %llu\n", (unsigned long long)req, (unsigned long long)req * 16);
    return 0;
}
