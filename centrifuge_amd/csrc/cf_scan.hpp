// cf_scan.hpp — the exclusive prefix sums of a batch (word offsets of the reads, work-list slots and hit-list bases,
// row bases of the queries, first printed row of the queries), as three small launches each:
//
//   k_scan_sums   one block per tile of 4096 items: the tile's total(s)
//   k_scan_tiles  one block: exclusive scan of the tile totals (a few thousand numbers)
//   k_scan_write  one block per tile: the items again, block-level scan + the tile's offset -> outputs
//
// i.e. reduce-then-scan: the input is read twice and every output written once, all of it coalesced 64-byte-per-lane
// vector loads, no spinning on other blocks' progress.  (hipcub's decoupled look-back scans of these 10 M-element
// arrays measured 0.34-1.7 ms each inside the pipelined batch loop, profiles/r02b; five of them per batch.)
//
// Every scan here maps a u32 input element to its summand(s) on the fly, so the scan inputs (flags, doubled capacities,
// word counts) are never materialised:
//   MODE_WORDS   v = ceil(in / 32)                     read length -> packed words        (woff)
//   MODE_PLAIN   v = in                                rows per query, printed rows       (qBase, rowFirst)
//   MODE_HITS    v = 2 * in, c = in != 0               hit capacity per strand -> hit-list base, work-list slot
// n items give n + 1 outputs (the last one is the total).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cfamd {

enum : int { SCAN_WORDS = 0, SCAN_PLAIN = 1, SCAN_HITS = 2 };
constexpr int kScanBlock = 256, kScanPer = 16, kScanTile = kScanBlock * kScanPer;

template <int MODE>
__device__ __forceinline__ uint64_t scan_value(uint32_t x) {
    return MODE == SCAN_WORDS ? (((uint64_t)x + 31) >> 5) : MODE == SCAN_HITS ? 2ull * x : (uint64_t)x;
}

// this thread's 16 items (zeros past n), as four 16-byte loads
__device__ __forceinline__ void scan_load(const uint32_t *in, uint64_t n, uint64_t first, uint32_t (&x)[kScanPer]) {
    if (first + kScanPer <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + first);
#pragma unroll
        for (int i = 0; i < kScanPer / 4; i++) { const uint4 v = p[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
    } else {
#pragma unroll
        for (int i = 0; i < kScanPer; i++) x[i] = first + i < n ? in[first + i] : 0u;
    }
}

// block-wide sums of (a, c); valid in every thread
__device__ __forceinline__ void scan_block_reduce(uint64_t &a, uint32_t &c, uint64_t *ldsA, uint32_t *ldsC) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m, 64); c += __shfl_xor(c, m, 64); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ldsA[w] = a; ldsC[w] = c; }
    __syncthreads();
    a = 0; c = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; i++) { a += ldsA[i]; c += ldsC[i]; }
}

template <int MODE>
__global__ void __launch_bounds__(kScanBlock) k_scan_sums(const uint32_t *in, uint64_t n, uint64_t *tileA, uint32_t *tileC) {
    __shared__ uint64_t ldsA[kScanBlock / 64];
    __shared__ uint32_t ldsC[kScanBlock / 64];
    uint32_t x[kScanPer];
    scan_load(in, n, (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer, x);
    uint64_t a = 0;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; i++) { a += scan_value<MODE>(x[i]); if (MODE == SCAN_HITS) c += x[i] != 0; }
    scan_block_reduce(a, c, ldsA, ldsC);
    if (threadIdx.x == 0) { tileA[blockIdx.x] = a; if (MODE == SCAN_HITS) tileC[blockIdx.x] = c; }
}

// exclusive scan of the tile totals in place, by one block: chunks of kScanBlock tiles with a running carry
__global__ void __launch_bounds__(kScanBlock) k_scan_tiles(uint64_t *tileA, uint32_t *tileC, uint32_t nTiles, bool withC) {
    __shared__ uint64_t sA[kScanBlock];
    __shared__ uint32_t sC[kScanBlock];
    uint64_t carryA = 0;
    uint32_t carryC = 0;
    for (uint32_t base = 0; base < nTiles; base += kScanBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t a = i < nTiles ? tileA[i] : 0;
        const uint32_t c = (withC && i < nTiles) ? tileC[i] : 0;
        sA[threadIdx.x] = a; sC[threadIdx.x] = c;
        __syncthreads();
        // Hillis-Steele inclusive scan over the chunk (256 numbers: 8 rounds)
        for (int d = 1; d < kScanBlock; d <<= 1) {
            const uint64_t ta = threadIdx.x >= (unsigned)d ? sA[threadIdx.x - d] : 0;
            const uint32_t tc = threadIdx.x >= (unsigned)d ? sC[threadIdx.x - d] : 0;
            __syncthreads();
            sA[threadIdx.x] += ta; sC[threadIdx.x] += tc;
            __syncthreads();
        }
        if (i < nTiles) { tileA[i] = carryA + sA[threadIdx.x] - a; if (withC) tileC[i] = carryC + sC[threadIdx.x] - c; }
        carryA += sA[kScanBlock - 1]; carryC += sC[kScanBlock - 1];
        __syncthreads();
    }
}

// outputs i = 0 .. n (n + 1 of them): outA[i] = sum of the values before item i, outC likewise for the counts
template <int MODE>
__global__ void __launch_bounds__(kScanBlock) k_scan_write(const uint32_t *in, uint64_t n, const uint64_t *tileA, const uint32_t *tileC,
                                                            uint64_t *outA, uint32_t *outC) {
    __shared__ uint64_t ldsA[kScanBlock / 64];
    __shared__ uint32_t ldsC[kScanBlock / 64];
    const uint64_t first = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
    uint32_t x[kScanPer];
    scan_load(in, n, first, x);
    uint64_t a = 0;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; i++) { a += scan_value<MODE>(x[i]); if (MODE == SCAN_HITS) c += x[i] != 0; }
    // exclusive scan of the thread totals over the block: inside the wave by shuffles, across waves through LDS
    uint64_t ia = a;
    uint32_t ic = c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t ta = __shfl_up(ia, d, 64);
        const uint32_t tc = __shfl_up(ic, d, 64);
        if (lane >= d) { ia += ta; ic += tc; }
    }
    if (lane == 63) { ldsA[w] = ia; ldsC[w] = ic; }
    __syncthreads();
    uint64_t offA = tileA[blockIdx.x] + ia - a;
    uint32_t offC = (MODE == SCAN_HITS ? tileC[blockIdx.x] : 0u) + ic - c;
    for (int i = 0; i < w; i++) { offA += ldsA[i]; offC += ldsC[i]; }
#pragma unroll
    for (int i = 0; i < kScanPer; i++) {
        const uint64_t idx = first + i;
        if (idx <= n) { outA[idx] = offA; if (MODE == SCAN_HITS) outC[idx] = offC; }
        offA += scan_value<MODE>(x[i]);
        if (MODE == SCAN_HITS) offC += x[i] != 0;
    }
}

inline uint32_t scan_tiles_for(uint64_t n) { return (uint32_t)((n + 1 + kScanTile - 1) / kScanTile); }

// enqueue the scan of n items -> n + 1 outputs; tileA / tileC hold scan_tiles_for(n) entries
template <int MODE>
inline void scan_enqueue(const uint32_t *in, uint64_t n, uint64_t *outA, uint32_t *outC, uint64_t *tileA, uint32_t *tileC, hipStream_t st) {
    const uint32_t tiles = scan_tiles_for(n);
    hipLaunchKernelGGL((k_scan_sums<MODE>), dim3(tiles), dim3(kScanBlock), 0, st, in, n, tileA, tileC);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kScanBlock), 0, st, tileA, tileC, tiles, MODE == SCAN_HITS);
    hipLaunchKernelGGL((k_scan_write<MODE>), dim3(tiles), dim3(kScanBlock), 0, st, in, n, (const uint64_t *)tileA, (const uint32_t *)tileC, outA, outC);
}

}  // namespace cfamd
