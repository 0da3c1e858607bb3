// hal.hpp -- the few device operations the host side needs.  Implemented by device.hip (HIP,
// product) and by emu.cpp (host loops, TEST-ONLY emulation build).
#pragma once
#include <cstddef>
#include <mutex>
#include <string>

#include "solve_core.hpp"

namespace kg::hal {
// returns empty string on success, else the error text.  device < 0 = the current device; *device_out = the device the
// context lives on.  Set-up per device (once): the address of this device's copy of the library's static scratch array
// (kg_solve_noalloc) and a few timing events, so that the no-allocation entry point never creates anything.
std::string init(int device, int* device_out);
// makes `device` current for the calling thread and `stream` (a context's own, from stream_create) the stream every HAL
// operation of this thread goes to -- every entry point, before it touches the GPU
void use_device(int device, void* stream);
int device_count();     // visible GPUs (0 when there is none)
void* stream_create();  // on the current device; never the null stream
void stream_destroy(void* stream);
// the current device's copy of the static scratch array and its size (device.hip: `scratch`)
void* scratch_ptr(size_t* bytes);
// ... and the mutex that serialises its users: every context on a GPU shares that GPU's array, so kg_solve_noalloc holds
// this from carving its tables until its results are decoded (contexts on other GPUs are not affected)
std::mutex& scratch_mutex();
// the calling thread is inside kg_solve_noalloc: its launch groups take the timing events reserved for that entry point at
// kg_init (never hipEventCreate, whatever other contexts have taken from the device's pool)
void set_noalloc(bool on);
void* dmalloc(size_t bytes);
void* host_pinned(size_t bytes);  // page-locked host memory: a D2H into it is one DMA, no staging copy
void host_pinned_free(void* p);
void dfree(void* p);
void h2d(void* dst, const void* src, size_t bytes);
void d2h(void* dst, const void* src, size_t bytes);
// from page-locked host memory, NOT waited for: ordered before whatever the stream runs next; the source must stay untouched
// until the stream has been synchronised (sync(), or any of the waiting calls above)
void h2d_async(void* dst, const void* src_pinned, size_t bytes);
void sync();
// `rows` rows of `width` bytes, `spitch` bytes apart on the device, packed back to back on the host: one strided copy
void d2h_rows(void* dst, const void* src, size_t width, size_t spitch, size_t rows);
void dzero(void* dst, size_t bytes);
void dfill(void* dst, int byte, size_t bytes);  // debugging aid (KG_POISON): device memory filled with a byte pattern
// k_type_filter + k_node_static + k_pod_mask
void launch_prepass(const DevProblem& P, const PrepassTables& X, float* ms);
// one workgroup per simulation; sims = DEVICE array of SimState
void launch_solve(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue_dev, int block, float* ms, bool batch = false);
// one wave per Solve (device build; the emulation runs the workgroup loop): see Compiled::wave_ok for when it applies
void launch_solve_wave(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue_dev, float* ms);
// `waves` waves of one workgroup per Solve, that many pods in flight (solve_spec.hpp); see Compiled::spec_ok
void launch_solve_spec(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue_dev, int waves, float* ms);
// one workgroup per NodeClaim of the Solve that just ran on this stream: price-ordered, truncated type lists (TruncArgs)
void launch_truncate(const DevProblem& P, const SimState* sims, const TruncArgs& X, int max_claims);
// consolidation batches: k_sim_init (state slots from the pristine image + per-simulation patches), k_sim_gather
void launch_sim_init(const DevProblem& P, const SimBatch& B, float* ms);
void launch_sim_gather(const DevProblem& P, const SimBatch& B, const SimState* sims, float* ms);
// the three of a batch back to back with one host synchronisation; ms = {init, simulate, gather}
// ... and the batch's outcomes copied to `res_host` (page-locked) behind k_sim_gather, before that one synchronisation
void launch_sim_round(const DevProblem& P, const SimBatch& B, const SimState* sims, int block, float ms[3], void* res_host = nullptr, const void* res_dev = nullptr,
                      size_t res_bytes = 0);
// debug_probes.hip: the latency micro-probes (debug timers only)
void launch_probes(const DevProblem& P, const SimState* sims, void* stream);
// roctx ranges around the host phases of an entry point (the launch groups above carry their own)
void range_push(const char* name);
void range_pop();
struct Range {
  explicit Range(const char* name) { range_push(name); }
  ~Range() { range_pop(); }
};
const char* backend();
}  // namespace kg::hal
