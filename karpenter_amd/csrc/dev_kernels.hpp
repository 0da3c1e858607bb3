// dev_kernels.hpp -- the solve kernels' launch points.  The kernels are instantiated in dev_part.hip, which is compiled once
// per KG_PART (a handful of instantiations each) so that the heavy device code builds in parallel; device.hip (HAL: streams,
// events, launch groups) only sees these functions.  `stream` is a hipStream_t.
#pragma once
#include "solve_core.hpp"

namespace kg {
void enqueue_k_solve_main(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream);     // 256 threads (+ debug timers)
void enqueue_k_solve_wide(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int block, void* stream);   // 512 / 1024 threads (measurements)
void enqueue_k_simulate_main(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, void* stream);           // 256 threads, 3 or 2 workgroups per CU
void enqueue_k_simulate_other(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream);  // 512 threads / per-wave cycle accounting
void enqueue_k_solve_wave(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, bool dbg, void* stream);
void enqueue_k_solve_spec4(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, void* stream);
void enqueue_k_solve_spec_n(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, void* stream);  // 2 / 6 / 8 waves (measurements)
void enqueue_k_solve_spec_dbg(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, void* stream);

inline void enqueue_k_solve(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int block, bool dbg, void* stream) {
  if (dbg || block < 512) enqueue_k_solve_main(P, sims, nsims, nqueue, dbg, stream);
  else enqueue_k_solve_wide(P, sims, nsims, nqueue, block, stream);
}
inline void enqueue_k_simulate(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int block, bool dbg, void* stream) {
  if (dbg || block >= 512) enqueue_k_simulate_other(P, sims, nsims, nqueue, dbg, stream);
  else enqueue_k_simulate_main(P, sims, nsims, nqueue, stream);
}
inline void enqueue_k_solve_spec(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, bool dbg, void* stream) {
  if (dbg) enqueue_k_solve_spec_dbg(P, sims, nsims, nqueue, waves, stream);
  else if (waves >= 4 && waves < 6) enqueue_k_solve_spec4(P, sims, nsims, nqueue, stream);
  else enqueue_k_solve_spec_n(P, sims, nsims, nqueue, waves, stream);
}
}  // namespace kg
