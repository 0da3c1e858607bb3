// debug_probes.hip -- DEBUG ONLY: latency micro-probes of the device (kg_set_collect_stats(ctx, 2): the numbers behind
// DESIGN.md section 3 "What the hardware taught").  A kernel of its own, launched before a Solve only when the debug timers
// are on; it writes its cycle counts into the solve state's counter block, where the host prints them with the timers.
// Nothing here is on a production path.
#include <hip/hip_runtime.h>

#include "hal.hpp"

namespace kg {

__global__ void __launch_bounds__(256) k_latency_probes(DevProblem P, const SimState* sims) {
  const SimState S = sims[blockIdx.x];
    // micro-probes (debug timers only): cost of a bare barrier, of a dependent global load chain, of a flat LDS access
    __shared__ uint32_t probe[64];
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 1000; ++i) __syncthreads();
    unsigned long long t1 = wall_clock64();
    uint32_t idx = 0;
    if (threadIdx.x == 0) for (int i = 0; i < 1000; ++i) idx = S.queue[(idx & 3) * 4] & 1;
    __syncthreads();
    unsigned long long t2 = wall_clock64();
    for (int i = 0; i < 1000; ++i) { if (threadIdx.x == 0) probe[i & 63] = (uint32_t)i; __syncthreads(); idx += probe[(i + threadIdx.x) & 63]; __syncthreads(); }
    unsigned long long t3 = wall_clock64();
    for (int i = 0; i < 1000; ++i) { if (threadIdx.x == 0) S.scratch[i & 7] = i + idx; __syncthreads(); }
    unsigned long long t4 = wall_clock64();
    if (threadIdx.x == 0) { S.counters[26] = t1 - t0; S.counters[27] = t2 - t1; S.counters[28] = t3 - t2 + (idx & 1); S.counters[29] = t4 - t3; }
    __syncthreads();
    // latency of a dependent chain over 4096 lines (256 KiB: misses the 32 KiB L1): first walk = from HBM / MALL, second = L2;
    // then a store -> load of the same line (what a phase sees of the previous phase's writes)
    if ((size_t)(P.TW + 1) * P.Cmax * 8 >= 4096 * 64 && threadIdx.x == 0) {
      for (uint32_t i = 0; i < 4096; ++i) S.scratch[(size_t)i * 8] = ((i * 1237u + 1u) & 4095u);  // i -> (1237 i + 1) mod 4096: one cycle
      __threadfence();
      uint64_t j = 0;
      const unsigned long long a0 = wall_clock64();
      for (int i = 0; i < 2048; ++i) j = S.scratch[j * 8];
      const unsigned long long a1 = wall_clock64();
      for (int i = 0; i < 2048; ++i) j = S.scratch[j * 8];
      const unsigned long long a2 = wall_clock64();
      uint64_t v = j;
      for (int i = 0; i < 1000; ++i) { S.scratch[8] = v + 1; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); v = S.scratch[8]; }
      const unsigned long long a3 = wall_clock64();
      S.counters[60] = a1 - a0; S.counters[61] = a2 - a1; S.counters[62] = a3 - a2 + (v & 1); 
      for (uint32_t i = 0; i < 4096; ++i) S.scratch[(size_t)i * 8] = 0;
    }
    __syncthreads();
    // one round trip of 24 independent 8-byte loads per lane (the prefilter's shape: rows Cmax apart), 200 times:
    // by one wave, by every wave, and the same with lanes scattered over the row
    {
      const size_t stride = (size_t)P.Cmax;
      uint64_t acc = idx;
      for (int variant = 0; variant < 3; ++variant) {
        __syncthreads();
        const unsigned long long b0 = wall_clock64();
        const bool act = variant == 0 ? threadIdx.x < 64 : true;
        const uint32_t lane = variant == 2 ? (threadIdx.x * 37u) % 311u : threadIdx.x % 311u;
        if (act)
          for (int it = 0; it < 200; ++it) {
            uint64_t v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = S.scratch[(size_t)(k % (P.TW + 1)) * stride + ((lane + (acc & 1)) % 311u)];
#pragma unroll
            for (int k = 0; k < 24; ++k) acc ^= v[k];
          }
        __syncthreads();
        const unsigned long long b1 = wall_clock64();
        if (threadIdx.x == 0) S.counters[56 + variant] = b1 - b0 + (acc == 77);
      }
    }
    __syncthreads();
}

namespace hal {
void launch_probes(const DevProblem& P, const SimState* sims, void* stream) {
  hipLaunchKernelGGL(k_latency_probes, dim3(1), dim3(256), 0, (hipStream_t)stream, P, sims);
}
}  // namespace hal
}  // namespace kg
