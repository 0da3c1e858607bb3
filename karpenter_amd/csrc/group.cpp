// group.cpp -- several GPUs behind the C ABI in ONE process: a kg_group holds one context per device and a
// worker thread bound to each; catalogs and cluster snapshots are replicated to every device, the units of a consolidation
// round are dealt round-robin (unit k -> device k mod N), every device runs its simulations and the decision logic of its
// shard, and the records meet in this process's memory, where kg_consolidate_select replays the reference's selection.
// Written against the C ABI only (include/karpgpu.h): the same file serves the product and the TEST-ONLY emulation build.
//
// No collective is needed here: a record is produced on the HOST from a simulation's device result (prices are float64
// catalog data), so the "gather" of 8(e) is the workers' results landing in one address space.  RCCL serves the
// multi-process variant (karpenter_amd/distributed.py: one all_gather of the fixed-size records).
#include <condition_variable>
#include <cstring>
#include <algorithm>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "karpgpu.h"

struct kg_group {
  std::vector<kg_ctx*> ctx;
  std::vector<int32_t> device;
  std::mutex call_mu;  // one group call at a time (the per-device contexts stay usable on their own)
  std::string last_error;
  // workers: one per device, each runs the job it is handed and reports its status
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int32_t()> job;
    bool has_job = false, done = false, quit = false;
    int32_t status = 0;
  };
  std::vector<std::unique_ptr<Worker>> workers;
  std::vector<std::vector<kg_candidate_record>> shard_records;

  static void loop(Worker* w) {
    for (;;) {
      std::function<int32_t()> job;
      {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        job = std::move(w->job);
        w->has_job = false;
      }
      const int32_t st = job();
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->status = st;
        w->done = true;
      }
      w->cv.notify_all();
    }
  }
  // fn(i) on every device's worker at once; the first failing device's status and error text win
  int32_t run_all(const std::function<int32_t(uint32_t)>& fn) {
    const uint32_t n = (uint32_t)workers.size();
    for (uint32_t i = 0; i < n; ++i) {
      Worker* w = workers[i].get();
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->job = [&fn, i] { return fn(i); };
        w->has_job = true;
        w->done = false;
      }
      w->cv.notify_all();
    }
    int32_t status = KG_OK;
    for (uint32_t i = 0; i < n; ++i) {
      Worker* w = workers[i].get();
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->done; });
      if (w->status != KG_OK && status == KG_OK) {
        status = w->status;
        char buf[512];
        buf[0] = 0;
        kg_last_error(ctx[i], buf, sizeof buf);
        last_error = "device " + std::to_string(device[i]) + ": " + buf;
      }
    }
    return status;
  }
};

extern "C" {

int32_t kg_group_init(const int32_t* devices, uint32_t n_devices, kg_group** out) {
  if (!out || (!devices && n_devices > 64)) return KG_EINVAL;
  *out = nullptr;
  std::vector<int32_t> devs;
  if (devices) devs.assign(devices, devices + n_devices);
  else {
    int32_t visible = 0;
    kg_device_count(&visible);
    if (visible <= 0 || n_devices > (uint32_t)visible) return KG_EDEVICE;
    for (uint32_t d = 0; d < (n_devices ? n_devices : (uint32_t)visible); ++d) devs.push_back((int32_t)d);
  }
  auto g = std::unique_ptr<kg_group>(new kg_group());
  for (size_t i = 0; i < devs.size(); ++i) {
    kg_ctx* c = nullptr;
    const int32_t st = kg_init(devs[i], &c);
    if (st != KG_OK) {
      for (kg_ctx* x : g->ctx) kg_destroy(x);
      return st;
    }
    g->ctx.push_back(c);
    g->device.push_back(devs[i]);
  }
  if (g->ctx.empty()) return KG_EDEVICE;
  g->shard_records.resize(g->ctx.size());
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    g->workers.emplace_back(new kg_group::Worker());
    kg_group::Worker* w = g->workers.back().get();
    w->th = std::thread(kg_group::loop, w);
  }
  *out = g.release();
  return KG_OK;
}

int32_t kg_group_destroy(kg_group* g) {
  if (!g) return KG_OK;
  for (auto& w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->quit = true;
    }
    w->cv.notify_all();
    w->th.join();
  }
  for (kg_ctx* c : g->ctx) kg_destroy(c);
  delete g;
  return KG_OK;
}

int32_t kg_group_size(kg_group* g, uint32_t* n_devices) {
  if (!g || !n_devices) return KG_EINVAL;
  *n_devices = (uint32_t)g->ctx.size();
  return KG_OK;
}

int32_t kg_group_context(kg_group* g, uint32_t i, kg_ctx** out) {
  if (!g || !out || i >= g->ctx.size()) return KG_EINVAL;
  *out = g->ctx[i];
  return KG_OK;
}

int32_t kg_group_last_error(kg_group* g, char* buf, size_t n) {
  if (!g || !buf || n == 0) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  const size_t m = std::min(n - 1, g->last_error.size());
  std::memcpy(buf, g->last_error.data(), m);
  buf[m] = 0;
  return KG_OK;
}

int32_t kg_group_catalog_upload(kg_group* g, const kg_catalog_desc* desc, uint64_t generation, kg_catalog** out) {
  if (!g || !desc || !out) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  for (size_t i = 0; i < g->ctx.size(); ++i) out[i] = nullptr;
  const int32_t st = g->run_all([&](uint32_t i) { return kg_catalog_upload(g->ctx[i], desc, generation, &out[i]); });
  if (st != KG_OK) for (size_t i = 0; i < g->ctx.size(); ++i) { kg_catalog_free(out[i]); out[i] = nullptr; }
  return st;
}

int32_t kg_group_catalog_update_offerings(kg_group* g, kg_catalog* const* catalogs, const kg_strtab* strings, const kg_offering_update* updates,
                                          uint32_t n_updates, uint64_t generation) {
  if (!g || !catalogs) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  return g->run_all([&](uint32_t i) { return kg_catalog_update_offerings(g->ctx[i], catalogs[i], strings, updates, n_updates, generation); });
}

int32_t kg_group_catalog_free(kg_group* g, kg_catalog* const* catalogs) {
  if (!g || !catalogs) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  for (size_t i = 0; i < g->ctx.size(); ++i) kg_catalog_free(catalogs[i]);
  return KG_OK;
}

int32_t kg_group_cluster_upload(kg_group* g, const kg_cluster_desc* desc, kg_cluster** out) {
  if (!g || !desc || !out) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  for (size_t i = 0; i < g->ctx.size(); ++i) out[i] = nullptr;
  const int32_t st = g->run_all([&](uint32_t i) { return kg_cluster_upload(g->ctx[i], desc, &out[i]); });
  if (st != KG_OK) for (size_t i = 0; i < g->ctx.size(); ++i) { kg_cluster_free(out[i]); out[i] = nullptr; }
  return st;
}

int32_t kg_group_cluster_free(kg_group* g, kg_cluster* const* clusters) {
  if (!g || !clusters) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  for (size_t i = 0; i < g->ctx.size(); ++i) kg_cluster_free(clusters[i]);
  return KG_OK;
}

int32_t kg_group_consolidate(kg_group* g, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* const* clusters, const kg_pods_desc* pending,
                             const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, int32_t multi, kg_candidate_record* records,
                             uint32_t cap_records, uint32_t* n_records, kg_decision* out, double* device_ms) {
  if (!g || !catalogs || n_catalogs == 0 || !clusters || !nodepools || !cands || !out || (!records && cap_records)) return KG_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  const uint32_t nd = (uint32_t)g->ctx.size();
  const uint32_t n = cands->n_candidates;
  const uint32_t units = multi ? (n < 2 ? 0 : std::min<uint32_t>(n, 101) - 1) : n;
  std::vector<uint32_t> got(nd, 0);
  std::vector<double> ms(nd, 0);
  for (uint32_t i = 0; i < nd; ++i) if (g->shard_records[i].size() < units / nd + 1) g->shard_records[i].resize(units / nd + 1);
  int32_t st = g->run_all([&](uint32_t i) {
    kg_candidates_desc cd = *cands;
    cd.shard = (int32_t)i;
    cd.n_shards = (int32_t)nd;
    auto fn = multi ? kg_consolidate_multi : kg_consolidate_single;
    return fn(g->ctx[i], catalogs + (size_t)i * n_catalogs, n_catalogs, clusters[i], pending, nodepools, &cd, g->shard_records[i].data(),
              (uint32_t)g->shard_records[i].size(), &got[i], &ms[i]);
  });
  if (st != KG_OK) return st;
  // the gather: every shard's records, side by side (selection does not depend on their order)
  std::vector<kg_candidate_record> local;
  uint32_t total = 0;
  for (uint32_t i = 0; i < nd; ++i) total += got[i];
  kg_candidate_record* all = records;
  if (total > cap_records) {
    if (records) { if (n_records) *n_records = total; g->last_error = "record buffer too small"; return KG_ENOMEM; }
    local.resize(total);
    all = local.data();
  }
  uint32_t at = 0;
  double ms_max = 0;
  for (uint32_t i = 0; i < nd; ++i) {
    if (got[i]) std::memcpy(all + at, g->shard_records[i].data(), (size_t)got[i] * sizeof(kg_candidate_record));
    at += got[i];
    ms_max = std::max(ms_max, ms[i]);
  }
  if (n_records) *n_records = total;
  if (device_ms) *device_ms = ms_max;
  st = kg_consolidate_select(g->ctx[0], catalogs, n_catalogs, multi, all, total, n, out);
  if (st != KG_OK) {
    char buf[512];
    buf[0] = 0;
    kg_last_error(g->ctx[0], buf, sizeof buf);
    g->last_error = buf;
    return st;
  }
  out->device_ms = ms_max;
  return KG_OK;
}

}  // extern "C"
