// cf_platform.hpp — the few wavefront primitives the kernels are written against.
//
// Product build (hipcc, gfx950): real CDNA4 intrinsics, 64-lane wavefronts.
// CF_HOST_EMU build (g++, tests/emu only): a one-lane "wavefront" so the kernel
// bodies can be stepped through on a CPU in the unit tests — or, with CF_EMU_WAVE64
// as well, a 64-lane one (fibers that meet at the cross-lane primitives).  The
// emulation is never part of libcentrifuge_amd.so.
#pragma once
#include <cstdint>

#ifdef CF_HOST_EMU
#include <cstring>
#define CF_DEV inline
#define CF_GLOBAL inline
namespace cfamd {
struct EmuCtx { uint32_t tid = 0, nthreads = 1; };
extern thread_local EmuCtx g_emu;
inline uint32_t cf_global_thread() { return g_emu.tid; }
inline uint32_t cf_global_threads() { return g_emu.nthreads; }
#ifdef CF_EMU_WAVE64
// The 64-LANE harness (round 6; tests/emu/emu.cpp emu_run_wave): a wavefront is 64 fibers, one per lane, each running the kernel
// body on a stack of its own; a cross-lane primitive is a RENDEZVOUS — the lane leaves its operand and yields, and when every
// lane that has not returned from the body has arrived (at the same primitive: anything else is a divergent collective and
// aborts the test) the results are formed as the hardware forms them (ballot over the live lanes, shuffles between them,
// readfirstlane = the lowest live lane) and the lanes go on.  Between two rendezvous the lanes run one after the other, so
// atomics and the LDS need no locking — and a body that counts on its lanes running in lockstep BETWEEN cross-lane calls (it must
// not) fails here.  Outside emu_run_wave (the per-thread bodies the harness calls in plain loops) the primitives are those of
// a one-lane wavefront, as in the CF_WAVE == 1 build.
#define CF_WAVE 64
enum : int { EMU_OP_BALLOT = 1, EMU_OP_SHFL = 2, EMU_OP_FIRST = 3, EMU_OP_SWAP1 = 4, EMU_OP_FENCE = 5 };   // (SWAP1: the DPP exchange of a lane pair — a rendezvous of the two lanes only, legal inside divergent code as on the hardware)
int emu_wave_lane();                                             // -1 outside a wavefront
uint64_t emu_collective(int op, uint64_t v, int src);
inline uint32_t cf_lane() { const int l = emu_wave_lane(); return l < 0 ? 0u : (uint32_t)l; }
inline uint32_t cf_local_thread() { return cf_lane(); }
inline uint32_t cf_block_threads() { return emu_wave_lane() < 0 ? 1u : (uint32_t)CF_WAVE; }
#else
#define CF_WAVE 1
inline uint32_t cf_lane() { return 0; }
inline uint32_t cf_local_thread() { return 0; }
inline uint32_t cf_block_threads() { return 1; }
#endif
inline int cf_ctz32(uint32_t x) { return __builtin_ctz(x); }
inline int cf_ctz64(uint64_t x) { return __builtin_ctzll(x); }
#ifdef CF_EMU_WAVE64
// (a same-wavefront LDS hand-off between lanes — one lane's stores read by its neighbour behind the fence — rests on the lanes'
// lockstep; the fibers get it back here: a lane at a fence lets every other lane reach its next fence or cross-lane primitive first)
inline void cf_compiler_fence() { asm volatile("" ::: "memory"); if (emu_wave_lane() >= 0) (void)emu_collective(EMU_OP_FENCE, 0, 0); }
#else
inline void cf_compiler_fence() { asm volatile("" ::: "memory"); }
#endif
#ifdef CF_EMU_WAVE64
inline uint32_t cf_swap1(uint32_t v) { return emu_wave_lane() < 0 ? v : (uint32_t)emu_collective(EMU_OP_SWAP1, v, emu_wave_lane() ^ 1); }
#else
inline uint32_t cf_swap1(uint32_t v) { return v; }          // never reached with one-lane chains
#endif
inline int cf_popc32(uint32_t x) { return __builtin_popcount(x); }
inline uint32_t cf_brev32(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (x & 1u); x >>= 1; } return r; }
inline uint64_t cf_brev64(uint64_t x) { uint64_t r = 0; for (int i = 0; i < 64; i++) { r = (r << 1) | (x & 1ull); x >>= 1; } return r; }
#ifdef CF_EMU_WAVE64
inline uint64_t cf_ballot(bool p) { return emu_wave_lane() < 0 ? (p ? 1ull : 0ull) : emu_collective(EMU_OP_BALLOT, p ? 1 : 0, 0); }
inline uint32_t cf_first_lane_u32(uint32_t v) { return emu_wave_lane() < 0 ? v : (uint32_t)emu_collective(EMU_OP_FIRST, v, 0); }
template <typename T> inline T cf_shfl(T v, int src) { static_assert(sizeof(T) <= 8, "shuffles move up to 64 bits"); return emu_wave_lane() < 0 ? v : (T)emu_collective(EMU_OP_SHFL, (uint64_t)v, src & (CF_WAVE - 1)); }
template <typename T> inline T cf_shfl_xor(T v, int m) { return emu_wave_lane() < 0 ? v : (T)emu_collective(EMU_OP_SHFL, (uint64_t)v, (emu_wave_lane() ^ m) & (CF_WAVE - 1)); }
#else
inline uint64_t cf_ballot(bool p) { return p ? 1ull : 0ull; }
inline uint32_t cf_first_lane_u32(uint32_t v) { return v; }
template <typename T> inline T cf_shfl(T v, int) { return v; }
template <typename T> inline T cf_shfl_xor(T v, int) { return v; }
#endif
inline int cf_popc64(uint64_t x) { return __builtin_popcountll(x); }
inline uint32_t cf_atomic_add(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline unsigned long long cf_atomic_add(unsigned long long *p, unsigned long long v) { auto o = *p; *p = o + v; return o; }
inline uint32_t cf_atomic_cas(uint32_t *p, uint32_t expect, uint32_t v) { uint32_t o = *p; if (o == expect) *p = v; return o; }
inline void cf_atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
inline void cf_atomic_or64(uint64_t *p, uint64_t v) { *p |= v; }
inline void cf_atomic_max(uint32_t *p, uint32_t v) { if (v > *p) *p = v; }
struct u64x2 { uint64_t x, y; };
inline u64x2 cf_load16(const uint8_t *p) { u64x2 v; std::memcpy(&v, p, 16); return v; }
inline uint64_t cf_load8(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline void cf_wait_vmem() {}
inline void cf_block_sync() {}                       // a one-thread block
inline void cf_store16_stream(void *p, uint64_t a, uint64_t b) { uint64_t v[2] = {a, b}; std::memcpy(p, v, 16); }
}  // namespace cfamd
#else
#include <hip/hip_runtime.h>
#define CF_DEV __device__ __forceinline__
#define CF_GLOBAL __global__
#define CF_WAVE 64
namespace cfamd {
CF_DEV uint32_t cf_lane() { return __lane_id(); }
CF_DEV uint32_t cf_global_thread() { return blockIdx.x * blockDim.x + threadIdx.x; }
CF_DEV uint32_t cf_global_threads() { return gridDim.x * blockDim.x; }
CF_DEV uint32_t cf_local_thread() { return threadIdx.x; }
CF_DEV uint32_t cf_block_threads() { return blockDim.x; }
CF_DEV int cf_ctz32(uint32_t x) { return __builtin_ctz(x); }
CF_DEV int cf_ctz64(uint64_t x) { return __builtin_ctzll(x); }
// keeps the compiler from moving memory accesses across it (LDS ops of one wavefront retire in order,
// so this is all a same-wave LDS write -> read hand-off between lanes needs)
CF_DEV void cf_compiler_fence() { asm volatile("" ::: "memory"); }
// value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1,0,3,2]), no LDS crossbar
CF_DEV uint32_t cf_swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
CF_DEV int cf_popc32(uint32_t x) { return __popc(x); }
CF_DEV uint32_t cf_brev32(uint32_t x) { return __brev(x); }                 // v_bfrev_b32
CF_DEV uint64_t cf_brev64(uint64_t x) { return __brevll(x); }
CF_DEV uint64_t cf_ballot(bool p) { return __ballot(p); }
CF_DEV uint32_t cf_first_lane_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <typename T> CF_DEV T cf_shfl(T v, int src) { return __shfl(v, src, 64); }
template <typename T> CF_DEV T cf_shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
CF_DEV int cf_popc64(uint64_t x) { return __popcll(x); }
CF_DEV uint32_t cf_atomic_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
CF_DEV unsigned long long cf_atomic_add(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
CF_DEV uint32_t cf_atomic_cas(uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); }   // returns the old value
CF_DEV void cf_atomic_or(uint32_t *p, uint32_t v) { (void)atomicOr(p, v); }
CF_DEV void cf_atomic_or64(uint64_t *p, uint64_t v) { (void)atomicOr(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
CF_DEV void cf_atomic_max(uint32_t *p, uint32_t v) { (void)atomicMax(p, v); }       // result unused: global_atomic_or without return
struct u64x2 { uint64_t x, y; };
// one global_load_dwordx4 / dwordx2
CF_DEV u64x2 cf_load16(const uint8_t *p) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
    return u64x2{v.x, v.y};
}
CF_DEV uint64_t cf_load8(const uint8_t *p) { return *reinterpret_cast<const uint64_t *>(p); }
// s_waitcnt vmcnt(0) (expcnt, lgkmcnt left alone), stated explicitly.  For loops whose loads sit in divergent branches: the
// compiler's own waits live in the branches that use the data, so along a path that skips them the loads still count as
// pending at the loop's back edge, and the next iteration's first write to one of their registers gets a vmcnt(0) — which,
// the counter being shared, also waits for the STORES the iteration ended with, before the new loads are even issued.
CF_DEV void cf_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0f70); }
CF_DEV void cf_block_sync() { __syncthreads(); }
// 16-byte store that is not read again by this kernel: non-temporal (global_store_dwordx4 ... nt), keeps
// the scattered hit records from displacing index lines in L2
CF_DEV void cf_store16_stream(void *p, uint64_t a, uint64_t b) {
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    v2 v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
}  // namespace cfamd
#endif
