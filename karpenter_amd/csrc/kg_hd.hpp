// kg_hd.hpp -- build-mode glue for the solve core.
//
// The solve core (solve_core.hpp) is written as block-cooperative phases.  In the product build
// (hipcc, gfx950) a phase body runs once per thread of the workgroup and phases are separated by
// __syncthreads().  With -DKG_EMU (g++, TEST-ONLY library tests/emu) the same source runs a
// workgroup as a host loop over thread ids, so the host compile logic and the kernel's control
// flow can be debugged with gdb / sanitizers in a container that has no GPU.  The emulation is
// never part of libkarpgpu.so.
#pragma once
#include <cstdint>

#if defined(KG_EMU) || !defined(__HIP__)
// KG_EMU: test-only emulation.  Plain C++ translation units of the product (host compile, C ABI)
// also land here: they only use the structs and the small meta helpers.
#define KG_HD inline
#define KG_DEVICE_ONLY 0
#define KG_FOR_THREADS(tid, nthreads) for (int tid = 0; tid < (nthreads); ++tid)
#define KG_SYNC() ((void)0)
#define KG_GLOBAL
#define KG_OPAQUE64(x) ((void)0)
#define KG_UNI(x) (x)
#define KG_UNI64(x) (x)
#define KG_UNIV(v) ((void)0)
#define KG_UNIB(v) ((void)0)
struct alignas(16) uint4 { uint32_t x, y, z, w; };
namespace kg {
inline void atomic_min_u32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
inline void atomic_min_u64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
inline void atomic_add_u32(uint32_t* p, uint32_t v) { *p += v; }
inline uint32_t atomic_inc_u32(uint32_t* p) { return (*p)++; }
inline void atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
inline void atomic_add_u64(unsigned long long* p, unsigned long long v) { *p += v; }
inline unsigned long long now_ticks() { return 0; }
// word `tid / 64` of slots <- the predicate of the 64 threads tid belongs to (the emulation runs them in ascending order)
// returns the number of lower-numbered threads of the same 64 whose predicate holds
inline uint32_t wave_ballot_store(unsigned long long* slots, int tid, bool pred) {
  if ((tid & 63) == 0) slots[tid >> 6] = 0;
  const uint32_t rank = (uint32_t)__builtin_popcountll(slots[tid >> 6]);
  if (pred) slots[tid >> 6] |= 1ull << (tid & 63);
  return rank;
}
inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
inline int ctz64(uint64_t x) { return __builtin_ctzll(x); }
}  // namespace kg
#else
#include <hip/hip_runtime.h>
#define KG_HD __device__ __forceinline__
#define KG_DEVICE_ONLY 1
#define KG_FOR_THREADS(tid, nthreads) for (int tid = (int)threadIdx.x, kg_once_ = 0; kg_once_ == 0; kg_once_ = 1)
#define KG_SYNC() __syncthreads()
// Pointers into device memory that the kernels load from memory (not from the kernel arguments) are typed as global
// address space in the device pass: otherwise the compiler emits flat loads, whose completion also counts on the LDS
// counter, and every LDS read in the same phase then waits for all outstanding memory loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define KG_GLOBAL __attribute__((address_space(1)))
#else
#define KG_GLOBAL
#endif
// keeps the compiler from merging an LDS load and a global load of two branches into one flat load behind a pointer select
#define KG_OPAQUE64(x) asm volatile("" : "+v"(x))
// A value every thread of the workgroup computes identically (read from the same LDS / global word): tell the compiler,
// so that it lives in a scalar register and steers scalar branches instead of exec-mask bookkeeping.
#define KG_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define KG_UNI64(x) kg::uni64((unsigned long long)(x))
#define KG_UNIV(v) ((v) = KG_UNI(v))
#define KG_UNIB(v) ((v) = KG_UNI((uint32_t)(v)) != 0)
namespace kg {
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// workgroup-shared (LDS) minimum
__device__ __forceinline__ void atomic_min_u32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void atomic_min_u64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
__device__ __forceinline__ void atomic_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ uint32_t atomic_inc_u32(uint32_t* p) { return atomicAdd(p, 1u); }
__device__ __forceinline__ void atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }
__device__ __forceinline__ void atomic_add_u64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
__device__ __forceinline__ unsigned long long now_ticks() { return wall_clock64(); }  // 100 MHz
// word `tid / 64` of slots <- the wave's ballot (every thread of the workgroup must call it)
// returns the number of lower lanes whose predicate holds
__device__ __forceinline__ uint32_t wave_ballot_store(unsigned long long* slots, int tid, bool pred) {
  const unsigned long long b = __ballot(pred);
  if ((tid & 63) == 0) slots[tid >> 6] = b;
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int ctz64(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
}  // namespace kg
#endif
