// dev_part.hip -- the solve kernels of libkarpgpu (gfx950).  Compiled once per KG_PART, each part instantiating a few of
// them (dev_kernels.hpp lists the launch points), so the device code builds in parallel.
#include <hip/hip_runtime.h>

#include "dev_kernels.hpp"
#include "solve_spec.hpp"

#ifndef KG_PART
#error "compile with -DKG_PART=0..7"
#endif

namespace kg {

#if KG_PART == 0 || KG_PART == 1
// k_solve: one workgroup = one Scheduler.Solve (persistent over the whole pod queue)
template <int NT, bool DBG, bool TPF = true>
__global__ void __launch_bounds__(NT) k_solve(DevProblem Pin, const SimState* sims, const uint32_t* nqueue) {
  __shared__ Shared sh;
  const DevProblem& P = Pin;
  const SimState S = sims[blockIdx.x];
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  solve_block<NT, DBG, TPF>(P, S, sh, nqueue[blockIdx.x], (const KG_CONST4 DevProblem*)__builtin_amdgcn_kernarg_segment_ptr(), (const KG_CONST4 SimState*)(sims + blockIdx.x));
  if (DBG && (P.stats & 2) && threadIdx.x == 0) { S.counters[24] = clock64() - c0; S.counters[25] = wall_clock64() - w0; }
}
#if KG_PART == 0
void enqueue_k_solve_main(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dbg) hipLaunchKernelGGL((k_solve<256, true>), dim3(nsims), dim3(256), 0, st, P, sims, nqueue);
  else if (P.G > 0 || P.GH > 0) hipLaunchKernelGGL((k_solve<256, false, true>), dim3(nsims), dim3(256), 0, st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_solve<256, false, false>), dim3(nsims), dim3(256), 0, st, P, sims, nqueue);  // (no topology groups: without their prefilter code)
}
#else
void enqueue_k_solve_wide(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int block, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (block >= 1024) hipLaunchKernelGGL((k_solve<1024, false>), dim3(nsims), dim3(1024), 0, st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_solve<512, false>), dim3(nsims), dim3(512), 0, st, P, sims, nqueue);
}
#endif
#endif

#if KG_PART == 2 || KG_PART == 3
// k_simulate: the same solve loop, one workgroup per consolidation simulation (a kernel of its own so that profiles keep
// the one persistent Solve and the batches of simulations apart)
// Its LDS is dynamic: a simulation rarely opens NodeClaims, so it gets sim_gp_cap(NT) evaluation groups instead of
// KG_MAXGP (DevProblem::gp_cap tells solve_block), 52 KB instead of 74, and THREE 256-thread workgroups share a CU; the
// second launch bound is the register budget that goes with it (168 per thread).  Simulations do not contend with each
// other (tools/gpu_sim_sweep.py): each is one chain of dependent placements, so residency is throughput.
constexpr uint32_t sim_gp_cap(int nt) { return (uint32_t)(nt / 64 > KG_GPMIN ? nt / 64 : KG_GPMIN); }
constexpr size_t sim_lds_bytes(int nt) { return sizeof(Shared) - (size_t)(KG_MAXGP - sim_gp_cap(nt)) * sizeof(Row); }
static_assert(sim_lds_bytes(256) * 3 <= 160 * 1024, "three workgroups of k_simulate share a CU's 160 KB of LDS");
static_assert(sim_lds_bytes(512) <= 64 * 1024, "dynamic LDS without an opt-in attribute");
extern __shared__ __align__(16) unsigned char sim_lds[];
// OCC: workgroups per CU the register budget allows.  3 (168 registers, a few dozen spilled) for a batch that fills the
// chip more than twice over; 2 (256 registers, none spilled) for a small batch, whose time is its longest simulation.
// TPF: solve_block's topology prefilter compiled in (batches whose problem has topology groups)
template <int NT, bool DBG, int OCC, bool TPF>
__global__ void __launch_bounds__(NT, OCC) k_simulate(DevProblem Pin, const SimState* sims, const uint32_t* nqueue) {
  Shared& sh = *reinterpret_cast<Shared*>(sim_lds);
  const DevProblem& P = Pin;
  const SimState S = sims[blockIdx.x];
  solve_block<NT, DBG, TPF, false>(P, S, sh, nqueue[blockIdx.x], (const KG_CONST4 DevProblem*)__builtin_amdgcn_kernarg_segment_ptr(), (const KG_CONST4 SimState*)(sims + blockIdx.x));
}
#if KG_PART == 2
void enqueue_k_simulate_main(const DevProblem& Pin, const SimState* sims, int nsims, const uint32_t* nqueue, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DevProblem P = Pin;
  P.gp_cap = sim_gp_cap(256);
  const bool topo = P.G > 0 || P.GH > 0;
  if (nsims > 2 * 256) {  // (256 CUs: more simulations than two per CU)
    if (topo) hipLaunchKernelGGL((k_simulate<256, false, 3, true>), dim3(nsims), dim3(256), sim_lds_bytes(256), st, P, sims, nqueue);
    else hipLaunchKernelGGL((k_simulate<256, false, 3, false>), dim3(nsims), dim3(256), sim_lds_bytes(256), st, P, sims, nqueue);
  } else {
    if (topo) hipLaunchKernelGGL((k_simulate<256, false, 2, true>), dim3(nsims), dim3(256), sim_lds_bytes(256), st, P, sims, nqueue);
    else hipLaunchKernelGGL((k_simulate<256, false, 2, false>), dim3(nsims), dim3(256), sim_lds_bytes(256), st, P, sims, nqueue);
  }
}
#else
void enqueue_k_simulate_other(const DevProblem& Pin, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DevProblem P = Pin;
  P.gp_cap = sim_gp_cap(dbg ? 256 : 512);
  if (dbg) hipLaunchKernelGGL((k_simulate<256, true, 1, true>), dim3(nsims), dim3(256), sim_lds_bytes(256), st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_simulate<512, false, 1, true>), dim3(nsims), dim3(512), sim_lds_bytes(512), st, P, sims, nqueue);
}
#endif
#endif

#if KG_PART == 4
// k_solve_wave: one WAVE = one Scheduler.Solve (solve_wave.hpp: batches without topology groups / existing nodes / minValues)
template <bool DBG>
__global__ void __launch_bounds__(64) k_solve_wave(DevProblem Pin, const SimState* sims, const uint32_t* nqueue) {
  __shared__ SharedBase sh;
  __shared__ __attribute__((aligned(16))) uint64_t pool[KG_POOL_WORDS];  // fit-threshold rows + prefilter records: the rest of the CU's 160 KiB
  static_assert(sizeof(SharedBase) + sizeof(uint64_t) * KG_POOL_WORDS <= 160 * 1024, "LDS of one CU");
  const DevProblem& P = Pin;
  const SimState S = sims[blockIdx.x];
  // the same two descriptors in memory, for the rare paths (solve_wave.hpp: rare_view): Pin is the first kernel argument, so
  // the kernel-argument segment starts with it; the state descriptors are written by the host before the launch and never
  // by a kernel
  const KG_CONST DevProblem* Pk = (const KG_CONST DevProblem*)__builtin_amdgcn_kernarg_segment_ptr();
  const KG_CONST SimState* Sk = (const KG_CONST SimState*)(sims + blockIdx.x);
  solve_wave<DBG>(P, S, Pk, Sk, sh, pool, nqueue[blockIdx.x]);
}
void enqueue_k_solve_wave(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dbg) hipLaunchKernelGGL((k_solve_wave<true>), dim3(nsims), dim3(64), 0, st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_solve_wave<false>), dim3(nsims), dim3(64), 0, st, P, sims, nqueue);
}
#endif

#if KG_PART >= 5
// k_solve_spec: W waves of one workgroup = one Scheduler.Solve, W pods in flight (solve_spec.hpp: same batches as k_solve_wave,
// minus those with negative operators on custom label keys)
constexpr uint32_t KG_SPEC_POOL_WORDS = (160 * 1024 - (uint32_t)sizeof(SharedBase) - (uint32_t)sizeof(SpecCtl)) / 8 - 4;
template <int W, bool DBG>
__global__ void __launch_bounds__(64 * W) k_solve_spec(DevProblem Pin, const SimState* sims, const uint32_t* nqueue) {
  __shared__ SharedBase sh;
  __shared__ SpecCtl ctl;
  __shared__ __attribute__((aligned(16))) uint64_t pool[KG_SPEC_POOL_WORDS];
  static_assert(sizeof(SharedBase) + sizeof(SpecCtl) + sizeof(uint64_t) * KG_SPEC_POOL_WORDS <= 160 * 1024, "LDS of one CU");
  const DevProblem& P = Pin;
  const SimState S = sims[blockIdx.x];
  const KG_CONST DevProblem* Pk = (const KG_CONST DevProblem*)__builtin_amdgcn_kernarg_segment_ptr();
  const KG_CONST SimState* Sk = (const KG_CONST SimState*)(sims + blockIdx.x);
  solve_spec<W, DBG>(P, S, Pk, Sk, sh, ctl, pool, KG_SPEC_POOL_WORDS, nqueue[blockIdx.x]);
}
#if KG_PART == 5
void enqueue_k_solve_spec4(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, void* stream) {
  hipLaunchKernelGGL((k_solve_spec<4, false>), dim3(nsims), dim3(256), 0, (hipStream_t)stream, P, sims, nqueue);
}
#elif KG_PART == 6
void enqueue_k_solve_spec_n(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (waves >= 8) hipLaunchKernelGGL((k_solve_spec<8, false>), dim3(nsims), dim3(512), 0, st, P, sims, nqueue);
  else if (waves >= 6) hipLaunchKernelGGL((k_solve_spec<6, false>), dim3(nsims), dim3(384), 0, st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_solve_spec<2, false>), dim3(nsims), dim3(128), 0, st, P, sims, nqueue);
}
#else
void enqueue_k_solve_spec_dbg(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (waves >= 8) hipLaunchKernelGGL((k_solve_spec<8, true>), dim3(nsims), dim3(512), 0, st, P, sims, nqueue);
  else hipLaunchKernelGGL((k_solve_spec<4, true>), dim3(nsims), dim3(256), 0, st, P, sims, nqueue);
}
#endif
#endif

}  // namespace kg
