// emu.cpp -- TEST-ONLY host emulation of hal.hpp (built with -DKG_EMU into tests/emu; never linked
// into libkarpgpu.so).  Runs each workgroup as a loop over thread ids so the host compile logic and
// the kernels' control flow can be debugged without a GPU.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "hal.hpp"

namespace kg::hal {
std::string init(int, int* device_out) { *device_out = 0; return ""; }
void use_device(int, void*) {}
int device_count() { return 1; }
void* stream_create() { return nullptr; }
void stream_destroy(void*) {}
void* scratch_ptr(size_t* bytes) {
  static void* p = std::calloc(32u << 20, 1);  // (stands in for the device array of device.hip)
  *bytes = 32u << 20;
  return p;
}
std::mutex& scratch_mutex() { static std::mutex m; return m; }
void set_noalloc(bool) {}
void* dmalloc(size_t bytes) { return std::calloc(bytes ? bytes : 8, 1); }
void dfree(void* p) { std::free(p); }
void* host_pinned(size_t bytes) { return std::malloc(bytes ? bytes : 8); }
void host_pinned_free(void* p) { std::free(p); }
void h2d(void* dst, const void* src, size_t bytes) { if (bytes) std::memcpy(dst, src, bytes); }
void h2d_async(void* dst, const void* src_pinned, size_t bytes) { if (bytes) std::memcpy(dst, src_pinned, bytes); }
void sync() {}
void d2h(void* dst, const void* src, size_t bytes) { if (bytes) std::memcpy(dst, src, bytes); }
void d2h_rows(void* dst, const void* src, size_t width, size_t spitch, size_t rows) {
  for (size_t r = 0; r < rows; ++r) std::memcpy((char*)dst + r * width, (const char*)src + r * spitch, width);
}
void dzero(void* dst, size_t bytes) { if (bytes) std::memset(dst, 0, bytes); }
void dfill(void* dst, int byte, size_t bytes) { if (bytes) std::memset(dst, byte, bytes); }

void launch_prepass(const DevProblem& P, const PrepassTables& X, float* ms) {
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t e = 0; e < X.E; ++e)
    for (int w = 0; w < P.TW; ++w) X.e_amask_out[(size_t)e * P.TW + w] = type_filter_word(P, e, w);
  for (int s = 0; s < P.S; ++s) {
    for (int w = 0; w < P.NW; ++w) X.nlab_out[(size_t)s * P.NW + w] = 0;
    for (int n = 0; n < P.N; ++n)
      if (node_static_bit(P, X, s, n)) X.nlab_out[(size_t)s * P.NW + (n >> 6)] |= 1ull << (n & 63);
  }
  for (int vi = 0; vi < P.NV; ++vi)
    for (int w = 0; w < P.TW; ++w) X.podmask_out[(size_t)vi * P.TW + w] = pod_mask_word(P, (uint32_t)vi, w);
  if (X.pctx_out && P.TW <= KG_TWLDS)
    for (uint32_t i = 0; i < (uint32_t)P.NV * (uint32_t)(sizeof(PodCtx) / 8); ++i) ((uint64_t*)X.pctx_out)[i] = pod_ctx_word(P, i / (sizeof(PodCtx) / 8), i % (sizeof(PodCtx) / 8));
  *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void launch_solve(const DevProblem& Pin, const SimState* sims, int nsims, const uint32_t* nqueue, int block, float* ms, bool batch) {
  auto t0 = std::chrono::steady_clock::now();
  DevProblem P = Pin;
  if (batch) P.gp_cap = (uint32_t)std::max(block / 64, KG_GPMIN);  // (what the device's k_simulate launch gives a simulation)
  std::unique_ptr<Shared> shp(new Shared());  // (the workgroup's LDS; per call: contexts run on several threads at once)
  Shared& sh = *shp;
  for (int i = 0; i < nsims; ++i) {
    std::memset(&sh, 0, sizeof sh);
    if (block >= 1024) solve_block<1024, false>(P, sims[i], sh, nqueue[i]);
    else if (block >= 512) solve_block<512, false>(P, sims[i], sh, nqueue[i]);
    else solve_block<256, false>(P, sims[i], sh, nqueue[i]);
  }
  *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
void launch_solve_wave(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, float* ms) {
  launch_solve(P, sims, nsims, nqueue, 256, ms, false);
}
void launch_solve_spec(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int, float* ms) {
  launch_solve(P, sims, nsims, nqueue, 256, ms, false);
}
void launch_truncate(const DevProblem& P, const SimState* sims, const TruncArgs& X, int max_claims) {
  const SimState& S = sims[0];
  const uint32_t T = (uint32_t)P.T;
  if (T > KG_TRUNC_T) return;
  std::vector<double> price(T);
  for (uint32_t c = 0; c < S.header[0] && c < (uint32_t)max_claims; ++c) {
    uint64_t zmask, ctmask;
    claim_offering_masks(P, S, c, &zmask, &ctmask);
    uint32_t nset = 0;
    for (uint32_t t = 0; t < T; ++t) {
      const bool in = (S.c_mask[(size_t)(t >> 6) * P.Cmax + c] >> (t & 63)) & 1;
      price[t] = in ? type_cheapest(X, t, zmask, ctmask) : -1.0;
      nset += in;
    }
    for (uint32_t t = 0; t < T; ++t) {
      if (price[t] < 0) continue;
      uint32_t rank = 0;
      for (uint32_t u = 0; u < T; ++u) rank += (price[u] >= 0 && price_before(price[u], X.name_rank[u], price[t], X.name_rank[t])) ? 1u : 0u;
      if (rank < X.max_types) X.rec[(size_t)c * X.rec_words + 4 + rank] = t;
    }
    trunc_pack(P, S, X, c, 0, 1, nset);
  }
}
void launch_sim_init(const DevProblem& P, const SimBatch& B, float* ms) {
  for (int i = 0; i < B.nsims; ++i) {
    for (uint32_t t = 0; t < 64; ++t) sim_init_copy(B, (uint32_t)i, t, 64);
    for (uint32_t t = 0; t < 64; ++t) sim_init_patch(P, B, (uint32_t)i, t, 64);
  }
  *ms = 0;
}
void launch_sim_gather(const DevProblem& P, const SimBatch& B, const SimState* sims, float* ms) {
  for (int i = 0; i < B.nsims; ++i) sim_gather(P, B, sims[i], (uint32_t)i);
  *ms = 0;
}
void launch_sim_round(const DevProblem& P, const SimBatch& B, const SimState* sims, int block, float ms[3], void* res_host, const void* res_dev, size_t res_bytes) {
  launch_sim_init(P, B, &ms[0]);
  launch_solve(P, sims, B.nsims, B.nqueue, block, &ms[1], true);
  launch_sim_gather(P, B, sims, &ms[2]);
  if (res_host && res_bytes) d2h(res_host, res_dev, res_bytes);
}
void launch_probes(const DevProblem&, const SimState*, void*) {}
void range_push(const char*) {}
void range_pop() {}
const char* backend() { return "emu"; }
}  // namespace kg::hal
