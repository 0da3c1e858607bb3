// ctx.hpp -- library context shared by catalog.cpp / karpgpu.cpp.
#pragma once
#include <mutex>
#include <string>

#include <memory>

#include "hal.hpp"
#include "host_model.hpp"

namespace kg {
struct ParsedPods;  // compile.hpp
void parsed_pods_delete(ParsedPods* p);
}  // namespace kg

struct kg_ctx {
  kg::Vocab V;
  kg::StrMap lab;  // pod-label / namespace / uid strings (selector space, never on the device)
  std::string last_error;
  kg_stats stats{};
  bool collect_timers = false;
  bool wave_timers = false;   // debug: per-wave work / barrier-wait cycles per phase kind (kg_set_collect_stats bit 5)
  bool no_fast_path = false;  // debug: keep every placement on the workgroup-wide evaluation (kg_set_collect_stats bit 4)
  bool no_spec = false;       // debug: never the speculative multi-wave kernel (kg_set_collect_stats bit 6)
  bool spec_fake_fail = false;  // debug: treat every speculative solve as one whose waves gave up (exercises the fallback; kg_set_collect_stats bit 8)
  bool spec_serial = false;   // debug: the speculative kernel evaluates every pod at its turn (kg_set_collect_stats bit 7)
  int spec_waves = 4;         // waves (= pods in flight) of the speculative kernel: 2, 4, 6 or 8 (KG_SPEC_WAVES overrides, for measurements)
  int sim_block_threads = 256;  // workgroup size of k_solve for consolidation batches (one workgroup per simulation)
  int block_threads = 256;  // workgroup size of k_solve (256, 512 or 1024): 256 measured best (one wave per SIMD)
  bool collect_stats = true;  // maintain the reference-visit counters (exact; costs an extra pass per placement)
  std::mutex mu;
  int device = 0;  // the GPU this context lives on (made current by every entry point)
  void* stream = nullptr;  // the context's own HIP stream: all of its copies and kernels (two contexts overlap on one GPU)
  // Device buffers of finished solves, kept for the next one (a Solve needs three; hipMalloc / hipFree per call cost more
  // than the kernels of a small batch).  Used under `mu` only.
  std::vector<kg::DeviceBuffer> pool;
  void* pool_take(size_t bytes, size_t* got) {
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < pool.size(); ++i)
      if (pool[i].bytes >= bytes && pool[i].bytes <= 4 * bytes + (1u << 20) && (best == SIZE_MAX || pool[i].bytes < pool[best].bytes)) best = i;
    if (best != SIZE_MAX) { void* p = pool[best].ptr; *got = pool[best].bytes; pool.erase(pool.begin() + best); return p; }
    *got = bytes;
    return kg::hal::dmalloc(bytes);
  }
  void pool_give(void* p, size_t bytes) {
    if (!p) return;
    if (pool.size() >= 12) { kg::hal::dfree(pool.front().ptr); pool.erase(pool.begin()); }  // (a Solve holds three, a consolidation plan six)
    pool.push_back(kg::DeviceBuffer{p, bytes});
  }
  // ... and one page-locked host buffer (a consolidation plan's outcome records)
  void* pinned_spare = nullptr;
  size_t pinned_spare_bytes = 0;
  void* pinned_take(size_t bytes, size_t* got) {
    if (pinned_spare && pinned_spare_bytes >= bytes) { void* p = pinned_spare; *got = pinned_spare_bytes; pinned_spare = nullptr; pinned_spare_bytes = 0; return p; }
    *got = bytes;
    return kg::hal::host_pinned(bytes);
  }
  void pinned_give(void* p, size_t bytes) {
    if (!p) return;
    if (bytes > pinned_spare_bytes) { std::swap(p, pinned_spare); std::swap(bytes, pinned_spare_bytes); }
    kg::hal::host_pinned_free(p);
  }
  // Page-locked staging for a call's uploads (problem image, state image, descriptors): copied here and sent without waiting,
  // ordered on the context's stream before the kernels that read them.  One call at a time uses it (under `mu`); a call starts
  // at offset 0 after the previous one's last wait.  `may_grow` false (kg_solve_noalloc): only what is already there.
  void* stage = nullptr;
  size_t stage_cap = 0, stage_off = 0;
  void upload(void* dst, const void* src, size_t bytes, bool may_grow) {
    if (!bytes) return;
    const size_t need = (stage_off + bytes + 255) & ~size_t(255);
    if (need > stage_cap) {
      if (!may_grow) { kg::hal::h2d(dst, src, bytes); return; }
      kg::hal::sync();  // (copies out of the old buffer may still be on their way)
      kg::hal::host_pinned_free(stage);
      stage_cap = std::max(need * 2, (size_t)4 << 20);
      stage = kg::hal::host_pinned(stage_cap);
      stage_off = 0;
    }
    std::memcpy((uint8_t*)stage + stage_off, src, bytes);
    kg::hal::h2d_async(dst, (uint8_t*)stage + stage_off, bytes);
    stage_off = (stage_off + bytes + 255) & ~size_t(255);
  }
  // the pod pools of the last cluster snapshot freed on this context, kept for the next upload: their capacity only -- a controller
  // that uploads a snapshot per poll otherwise faults the same megabytes in again every time (glibc unmaps them at the free)
  std::unique_ptr<kg::ParsedPods, void (*)(kg::ParsedPods*)> spare_bound{nullptr, kg::parsed_pods_delete};
  // merged views of the catalog sets recent solves named (at most 2 kept); owned by the context
  std::vector<kg_catalog*> merged;
  uint64_t merged_clock = 0;
  ~kg_ctx() {
    for (auto& b : pool) kg::hal::dfree(b.ptr);
    kg::hal::host_pinned_free(stage);
    kg::hal::host_pinned_free(pinned_spare);
    for (auto* m : merged) { kg::hal::dfree(m->tab.dev.ptr); delete m; }
  }
  uint32_t kHostname, kZone, kCT, kZid, kNodePool, kInstanceType;
  uint32_t kPart;  // hidden label key of merged catalog views ("karpgpu.internal/catalog")
  uint32_t rCPU, rMemory, rPods;
};

namespace kg {
// context lock + the context's device made current for the calling thread
struct CtxLock {
  std::lock_guard<std::mutex> lk;
  explicit CtxLock(kg_ctx* c) : lk(c->mu) { hal::use_device(c->device, c->stream); }
};
// A fixed region of device memory that a solve carves its tables from instead of the context's pool (kg_solve_noalloc: the
// library's static scratch array).  Nothing is ever allocated or freed; exceeding it is an error.
struct FixedArena {
  uint8_t* base = nullptr;
  size_t cap = 0, used = 0;
  void* take(size_t bytes, const char* what) {
    const size_t off = (used + 255) & ~size_t(255);
    if (off + bytes > cap) throw std::length_error(std::string("kg_solve_noalloc: the batch needs more than the ") + std::to_string(cap >> 20) + " MiB static device scratch (" + what + ")");
    used = off + bytes;
    return base + off;
  }
};
HReq parse_req(Vocab& V, const kg_pools& p, const kg_requirement& q);
HReqs parse_reqs(Vocab& V, const kg_pools& p, kg_range r);
std::vector<std::pair<uint32_t, int64_t>> parse_rl(Vocab& V, const kg_pools& p, kg_range r);
void parse_catalog(kg_ctx* ctx, const kg_catalog_desc* d, kg_catalog* cat);
void build_catalog_tables(kg_ctx* ctx, kg_catalog* cat);
void update_offerings(kg_ctx* ctx, kg_catalog* cat, const kg_strtab& strings, const kg_offering_update* ups, uint32_t n);
}  // namespace kg
