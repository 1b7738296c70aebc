// cf_prims.hpp — the data-parallel primitives of the index builder (cf_build.hip), written for gfx950:
//
//   block_exclusive_sum256   prefix sum over a 256-thread block: inside a wavefront by 64-lane shuffles (six steps),
//                            across the four wavefronts through 16 bytes of LDS
//   device_scan<T, INCL>     prefix sum of an array: reduce-then-scan in three launches (tile totals, scan of the totals
//                            by one block, the tiles again + their offset) — the input is read twice, every output
//                            written once, all coalesced; no block waits for another (cf_scan.hpp does the same for the
//                            batch plan's fused scans)
//   sort_pairs               stable LSD radix sort of (key, value) pairs over a bit range: rocPRIM's device radix sort,
//                            called directly (AMD's own primitive library; rounds 1-5 went through the hipCUB layer)
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

namespace cfamd {

// exclusive prefix sum of v over the block's 256 threads; total = the block's sum, valid in every thread.  lds4: four words.
// Two barriers; safe to call again right away with the same lds4 (the second barrier protects the reads).
__device__ __forceinline__ void block_exclusive_sum256(uint32_t v, uint32_t &excl, uint32_t &total, uint32_t *lds4) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint32_t s = lds4[i]; all += s; if (i < w) before += s; }
    __syncthreads();
    excl = before + inc - v;
    total = all;
}

constexpr int kPrimBlock = 256, kPrimPer = 8, kPrimTile = kPrimBlock * kPrimPer;

template <typename T>
__global__ void __launch_bounds__(kPrimBlock) kp_tile_sums(const T *in, uint64_t n, T *tileSum) {
    __shared__ T lds[kPrimBlock / 64];
    const uint64_t first = (uint64_t)blockIdx.x * kPrimTile;
    T a = 0;
#pragma unroll
    for (int i = 0; i < kPrimPer; i++) {                           // item i * 256 + thread: coalesced
        const uint64_t idx = first + (uint64_t)i * kPrimBlock + threadIdx.x;
        if (idx < n) a += in[idx];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { T s = 0; for (int i = 0; i < kPrimBlock / 64; i++) s += lds[i]; tileSum[blockIdx.x] = s; }
}

// exclusive scan of the tile totals in place, by one block: chunks of 256 tiles with a running carry
template <typename T>
__global__ void __launch_bounds__(kPrimBlock) kp_scan_tiles(T *tileSum, uint32_t nTiles) {
    __shared__ T lds[kPrimBlock / 64];
    __shared__ T carryS;
    if (threadIdx.x == 0) carryS = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nTiles; base += kPrimBlock) {
        const uint32_t i = base + threadIdx.x;
        const T a = i < nTiles ? tileSum[i] : (T)0;
        T inc = a;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const T t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        T before = carryS;
        for (int k = 0; k < w; k++) before += lds[k];
        if (i < nTiles) tileSum[i] = before + inc - a;
        __syncthreads();
        if (threadIdx.x == kPrimBlock - 1) carryS = before + inc;
        __syncthreads();
    }
}

template <typename T, bool INCLUSIVE>
__global__ void __launch_bounds__(kPrimBlock) kp_scan_write(const T *in, uint64_t n, const T *tileOff, T *out) {
    __shared__ T lds[kPrimBlock / 64];
    // thread t owns the kPrimPer consecutive items first + t * kPrimPer .. (a thread-contiguous tile, so the running sum is a
    // register loop); loads and stores of a wavefront still fall into one 2 KB / 4 KB span
    const uint64_t first = (uint64_t)blockIdx.x * kPrimTile + (uint64_t)threadIdx.x * kPrimPer;
    T x[kPrimPer];
    T a = 0;
#pragma unroll
    for (int i = 0; i < kPrimPer; i++) { x[i] = first + i < n ? in[first + i] : (T)0; a += x[i]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const T t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    T off = tileOff[blockIdx.x] + inc - a;
    for (int k = 0; k < w; k++) off += lds[k];
#pragma unroll
    for (int i = 0; i < kPrimPer; i++) {
        if (first + i < n) out[first + i] = INCLUSIVE ? off + x[i] : off;
        off += x[i];
    }
}

inline uint32_t prim_tiles_for(uint64_t n) { return (uint32_t)((n + kPrimTile - 1) / kPrimTile); }
// bytes of scratch device_scan<T> wants for n items
template <typename T>
inline size_t device_scan_bytes(uint64_t n) { return (size_t)(prim_tiles_for(n) + 1) * sizeof(T); }

// out[i] = sum of in[0 .. i) (exclusive) or in[0 .. i] (inclusive), n items, in != out.  scratch: device_scan_bytes<T>(n).
template <typename T, bool INCLUSIVE>
inline hipError_t device_scan(void *scratch, const T *in, T *out, uint64_t n, hipStream_t st = 0) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = prim_tiles_for(n);
    T *tileSum = reinterpret_cast<T *>(scratch);
    hipLaunchKernelGGL((kp_tile_sums<T>), dim3(tiles), dim3(kPrimBlock), 0, st, in, n, tileSum);
    hipLaunchKernelGGL((kp_scan_tiles<T>), dim3(1), dim3(kPrimBlock), 0, st, tileSum, tiles);
    hipLaunchKernelGGL((kp_scan_write<T, INCLUSIVE>), dim3(tiles), dim3(kPrimBlock), 0, st, in, n, (const T *)tileSum, out);
    return hipGetLastError();
}

// stable sort of n (key, value) pairs by key bits [beginBit, endBit); tmp == nullptr: only the scratch size is returned in bytes
template <typename K, typename V>
inline hipError_t sort_pairs(void *tmp, size_t &bytes, const K *keysIn, K *keysOut, const V *valsIn, V *valsOut, size_t n, int beginBit,
                             int endBit, hipStream_t st = 0) {
    return rocprim::radix_sort_pairs(tmp, bytes, keysIn, keysOut, valsIn, valsOut, n, (unsigned)beginBit, (unsigned)endBit, st);
}

}  // namespace cfamd
