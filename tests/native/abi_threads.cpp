// abi_threads.cpp -- TEST-ONLY: the concurrency contract of the C ABI (SURVEY.md 8(b) "Threading"), run natively so that it
// can be built with -fsanitize=thread and -fsanitize=address,undefined (tests/test_sanitizers.py builds this file together
// with the library's host sources and the emulation HAL; no GPU, no Python in the process).
//
// Mirrors pkg/providers/instancetype/suite_test.go:2445-2479 (List() from 10 000 goroutines at once, every caller sees the
// same catalog) in the shape this library has: two contexts as a controller would hold them (provisioning, disruption),
// N threads x kg_solve on the first, consolidation rounds on the second, readers of the resident catalog
// (kg_catalog_query / _query_requirements / kg_stats_get / kg_last_error) and a writer applying in-place offering updates
// to another catalog of the same context -- all at once; every result must equal the serial run's, bit for bit.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "karpgpu.h"

namespace {

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) { fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); std::abort(); } \
  } while (0)

struct PoolBuilder {
  std::string data;
  std::vector<uint32_t> offs{0};
  std::map<std::string, uint32_t> ids;
  std::vector<kg_requirement> reqs;
  std::vector<uint32_t> req_values;
  std::vector<kg_quantity> quantities;
  std::vector<kg_label> labels;
  std::vector<kg_range> ranges;
  uint32_t s(const std::string& x) {
    auto f = ids.find(x);
    if (f != ids.end()) return f->second;
    const uint32_t id = (uint32_t)offs.size() - 1;
    data += x;
    offs.push_back((uint32_t)data.size());
    ids[x] = id;
    return id;
  }
  using Req = std::tuple<std::string, int, std::vector<std::string>>;
  kg_range req_list(const std::vector<Req>& rs) {
    kg_range r{(uint32_t)reqs.size(), (uint32_t)rs.size()};
    for (auto& [key, op, vals] : rs) {
      kg_requirement q{};
      q.key = s(key);
      q.op = (uint8_t)op;
      q.min_values = -1;
      q.values = {(uint32_t)req_values.size(), (uint32_t)vals.size()};
      for (auto& v : vals) req_values.push_back(s(v));
      reqs.push_back(q);
    }
    return r;
  }
  kg_range values(const std::vector<std::string>& vs) {
    kg_range r{(uint32_t)req_values.size(), (uint32_t)vs.size()};
    for (auto& v : vs) req_values.push_back(s(v));
    return r;
  }
  kg_range qty(const std::vector<std::pair<std::string, int64_t>>& qs) {
    kg_range r{(uint32_t)quantities.size(), (uint32_t)qs.size()};
    for (auto& kv : qs) quantities.push_back({s(kv.first), kv.second});
    return r;
  }
  kg_range labs(const std::vector<std::pair<std::string, std::string>>& ls) {
    kg_range r{(uint32_t)labels.size(), (uint32_t)ls.size()};
    for (auto& kv : ls) labels.push_back({s(kv.first), s(kv.second)});
    return r;
  }
  kg_pools pools() const {
    kg_pools p{};
    p.strings = {data.data(), offs.data(), (uint32_t)offs.size() - 1};
    p.reqs = reqs.data(); p.n_reqs = (uint32_t)reqs.size();
    p.req_values = req_values.data(); p.n_req_values = (uint32_t)req_values.size();
    p.quantities = quantities.data(); p.n_quantities = (uint32_t)quantities.size();
    p.labels = labels.data(); p.n_labels = (uint32_t)labels.size();
    p.ranges = ranges.data(); p.n_ranges = (uint32_t)ranges.size();
    return p;
  }
};

const char* kZone = "topology.kubernetes.io/zone";
const char* kCT = "karpenter.sh/capacity-type";
const char* kIT = "node.kubernetes.io/instance-type";
const char* kArch = "kubernetes.io/arch";
const char* kFam = "karpenter.k8s.aws/instance-family";
const char* kPool = "karpenter.sh/nodepool";
const char* kHost = "kubernetes.io/hostname";
const std::vector<std::string> ZONES = {"z-1a", "z-1b", "z-1c"};
constexpr int64_t GI = 1024ll * 1024 * 1024 * 1000;

std::string type_name(int i) { return "f" + std::to_string(i / 4) + "." + std::to_string(1 << (i % 4)) + "x"; }

struct Catalog {
  PoolBuilder pb;
  std::vector<kg_instance_type> types;
  std::vector<kg_offering> offerings;
  kg_catalog_desc desc{};
  explicit Catalog(int n_types) {
    for (int i = 0; i < n_types; ++i) {
      kg_instance_type t{};
      const std::string name = type_name(i);
      t.name = pb.s(name);
      t.requirements = pb.req_list({{kIT, KG_OP_IN, {name}}, {kArch, KG_OP_IN, {i % 8 < 6 ? "amd64" : "arm64"}}, {kFam, KG_OP_IN, {"f" + std::to_string(i / 4)}},
                                    {kZone, KG_OP_IN, ZONES}, {kCT, KG_OP_IN, {"on-demand", "spot"}}});
      const int64_t cpu = 2000ll << (i % 4);
      t.capacity = pb.qty({{"cpu", cpu}, {"memory", cpu * 4 * GI / 1000}, {"pods", 58000 + 1000 * (i % 4)}});
      t.overhead = pb.qty({{"cpu", 80 + 10 * (i % 4)}, {"memory", GI / 2}});
      t.offerings = {(uint32_t)offerings.size(), 6};
      for (auto& z : ZONES)
        for (int ct = 0; ct < 2; ++ct) {
          kg_offering o{};
          o.zone = pb.s(z);
          o.capacity_type = pb.s(ct ? "spot" : "on-demand");
          o.zone_id = KG_NONE;
          o.price = 0.02 * (double)(cpu / 1000) * (1.0 + 0.01 * (i / 4)) * (ct ? 0.4 : 1.0);
          o.available = 1;
          offerings.push_back(o);
        }
      types.push_back(t);
    }
    desc.pools = pb.pools();
    desc.types = types.data(); desc.n_types = (uint32_t)types.size();
    desc.offerings = offerings.data(); desc.n_offerings = (uint32_t)offerings.size();
  }
};

struct Pods {
  PoolBuilder pb;
  std::vector<kg_pod> pods;
  kg_pods_desc desc{};
  // `node_of`: bound pods (node index per pod); empty = pending pods
  Pods(int n, uint64_t seed, const std::vector<int>& node_of = {}) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&](uint32_t m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)((x >> 11) % m); };
    const int64_t cpus[] = {100, 250, 500, 1000, 2000};
    for (int i = 0; i < n; ++i) {
      kg_pod p{};
      p.uid = pb.s("pod-" + std::to_string(seed) + "-" + std::to_string(i));
      p.ns = pb.s("default");
      p.creation_ts = 1700000000 + (int64_t)rnd(1000);
      p.labels = pb.labs({{"app", "app-" + std::to_string(rnd(7))}});
      const int64_t cpu = cpus[rnd(5)];
      p.requests = pb.qty({{"cpu", cpu}, {"memory", cpu * (1 + rnd(4)) * GI / 1000}});
      if (node_of.empty()) {
        const uint32_t k = rnd(10);
        if (k < 2) p.node_selector = pb.labs({{kZone, ZONES[rnd(3)]}});
        else if (k < 4) p.node_selector = pb.labs({{kArch, k == 2 ? "amd64" : "arm64"}});
        else if (k == 4) {
          pb.ranges.push_back(pb.req_list({{kFam, KG_OP_NOT_IN, {"f0", "f1"}}}));
          p.required_terms = {(uint32_t)pb.ranges.size() - 1, 1};
        }
        p.node = -1;
      } else {
        p.node = node_of[i];
      }
      p.reschedulable = 1;
      pods.push_back(p);
    }
    desc.pools = pb.pools();
    desc.pods = pods.data(); desc.n_pods = (uint32_t)pods.size();
  }
};

struct Cluster {
  std::vector<int> node_of;
  std::unique_ptr<Pods> bound;
  std::vector<kg_node> nodes;
  kg_cluster_desc desc{};
  Cluster(int n_nodes, int n_types) {
    for (int i = 0; i < n_nodes; ++i) for (int k = 0; k < 3 + i % 3; ++k) node_of.push_back(i);
    bound.reset(new Pods((int)node_of.size(), 77, node_of));
    PoolBuilder& pb = bound->pb;  // (the cluster descriptor shares the bound pods' pools)
    for (int i = 0; i < n_nodes; ++i) {
      const int t = (i * 7) % n_types;
      const int64_t cpu = 2000ll << (t % 4);
      kg_node nd{};
      const std::string name = "node-" + std::to_string(i);
      nd.name = pb.s(name);
      nd.labels = pb.labs({{kIT, type_name(t)}, {kArch, t % 8 < 6 ? "amd64" : "arm64"}, {kFam, "f" + std::to_string(t / 4)}, {kZone, ZONES[i % 3]},
                           {kCT, i % 2 ? "spot" : "on-demand"}, {kPool, "default"}, {kHost, name}});
      nd.available = pb.qty({{"cpu", cpu / 2}, {"memory", cpu * 2 * GI / 1000}, {"pods", 40000}});
      nd.capacity = pb.qty({{"cpu", cpu}, {"memory", cpu * 4 * GI / 1000}, {"pods", 58000}});
      nd.initialized = 1; nd.managed = 1; nd.deleting = 0;
      nodes.push_back(nd);
    }
    bound->desc.pools = pb.pools();  // (re-take the pointers: the node strings were added after the pods')
    desc.pods = bound->desc;
    desc.nodes = nodes.data(); desc.n_nodes = (uint32_t)nodes.size();
  }
};

struct NodePools {
  PoolBuilder pb;
  kg_nodepool np{};
  kg_nodepools_desc desc{};
  NodePools() {
    np.name = pb.s("default");
    np.requirements = pb.req_list({{kCT, KG_OP_IN, {"on-demand", "spot"}}});
    desc.well_known_labels = pb.values({kPool, kZone, kIT, kArch, kCT, kFam, "kubernetes.io/os", "topology.k8s.aws/zone-id"});
    desc.pools = pb.pools();
    desc.nodepools = &np; desc.n_nodepools = 1;
    desc.max_instance_types = 60;
  }
};

struct SolveOut {
  std::vector<uint8_t> pod_kind;
  std::vector<int32_t> pod_index;
  std::vector<uint32_t> pod_reason, type_ids, value_ids, str_offsets;
  std::vector<kg_claim_out> claims;
  std::vector<kg_req_out> reqs;
  std::vector<kg_quantity> quantities;
  std::vector<char> str_data;
  kg_results res{};
  explicit SolveOut(uint32_t n) : pod_kind(n), pod_index(n), pod_reason(n), type_ids(n * 64 + 1024), value_ids(n * 24 + 256), str_offsets(n * 48 + 1024),
                                  claims(n), reqs(n * 12 + 64), quantities(n * 12 + 64), str_data(n * 48 * 24 + 65536) {
    res.pod_kind = pod_kind.data(); res.pod_index = pod_index.data(); res.pod_reason = pod_reason.data(); res.cap_pods = n;
    res.claims = claims.data(); res.cap_claims = n;
    res.type_ids = type_ids.data(); res.cap_type_ids = (uint32_t)type_ids.size();
    res.reqs = reqs.data(); res.cap_reqs = (uint32_t)reqs.size();
    res.value_ids = value_ids.data(); res.cap_value_ids = (uint32_t)value_ids.size();
    res.quantities = quantities.data(); res.cap_quantities = (uint32_t)quantities.size();
    res.str_data = str_data.data(); res.cap_str_data = (uint32_t)str_data.size();
    res.str_offsets = str_offsets.data(); res.cap_strs = (uint32_t)str_offsets.size();
  }
  // what a caller reads back (timings aside)
  std::string digest() const {
    std::string d;
    auto add = [&](const void* p, size_t n) { d.append((const char*)p, n); };
    add(pod_kind.data(), pod_kind.size());
    add(pod_index.data(), pod_index.size() * 4);
    add(&res.n_claims, 4);
    for (uint32_t c = 0; c < res.n_claims; ++c) {
      const kg_claim_out& k = claims[c];
      add(&k.nodepool, 4); add(&k.n_pods, 4);
      add(type_ids.data() + k.types.begin, (size_t)k.types.count * 4);
      for (uint32_t r = 0; r < k.requirements.count; ++r) {
        const kg_req_out& q = reqs[k.requirements.begin + r];
        add(str_data.data() + str_offsets[q.key], str_offsets[q.key + 1] - str_offsets[q.key]);
        add(&q.complement, 1);
        for (uint32_t v = 0; v < q.values.count; ++v) {
          const uint32_t sid = value_ids[q.values.begin + v];
          add(str_data.data() + str_offsets[sid], str_offsets[sid + 1] - str_offsets[sid]);
        }
      }
      for (uint32_t r = 0; r < k.requests.count; ++r) add(&quantities[k.requests.begin + r].milli, 8);
    }
    return d;
  }
};

std::string solve_digest(kg_ctx* ctx, kg_catalog* cat, kg_cluster* cl, const Pods& pods, const NodePools& nps) {
  SolveOut out(pods.desc.n_pods);
  kg_catalog* cats[1] = {cat};
  const int32_t st = kg_solve(ctx, cats, 1, cl, &pods.desc, &nps.desc, &out.res);
  if (st != KG_OK) { char buf[512]; kg_last_error(ctx, buf, sizeof buf); fprintf(stderr, "kg_solve: %d %s\n", st, buf); std::abort(); }
  return out.digest();
}

std::string records_digest(kg_ctx* ctx, kg_catalog* cat, kg_cluster* cl, const Pods& pending, const NodePools& nps, const std::vector<int32_t>& order, bool multi,
                           int shard, int n_shards) {
  kg_candidates_desc cd{};
  cd.candidates = order.data(); cd.n_candidates = (uint32_t)order.size();
  cd.spot_to_spot_enabled = 1; cd.shard = shard; cd.n_shards = n_shards;
  std::vector<kg_candidate_record> recs(order.size() + 1);
  uint32_t n = 0;
  double ms = 0;
  kg_catalog* cats[1] = {cat};
  auto fn = multi ? kg_consolidate_multi : kg_consolidate_single;
  const int32_t st = fn(ctx, cats, 1, cl, &pending.desc, &nps.desc, &cd, recs.data(), (uint32_t)recs.size(), &n, &ms);
  if (st != KG_OK) { char buf[512]; kg_last_error(ctx, buf, sizeof buf); fprintf(stderr, "kg_consolidate: %d %s\n", st, buf); std::abort(); }
  return std::string((const char*)recs.data(), (size_t)n * sizeof(kg_candidate_record));
}

}  // namespace

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 6;
  const int n_types = 48;
  Catalog cat(n_types);
  NodePools nps;
  Cluster cluster(60, n_types);
  std::vector<std::unique_ptr<Pods>> batches;
  for (int k = 0; k < 4; ++k) batches.emplace_back(new Pods(120 + 40 * k, 11 + k));
  Pods none(0, 1);
  std::vector<int32_t> order;
  for (int i = 0; i < 60; ++i) order.push_back((i * 13) % 60);

  kg_ctx *prov = nullptr, *disr = nullptr;
  CHECK(kg_init(-1, &prov) == KG_OK && kg_init(-1, &disr) == KG_OK);
  kg_catalog *cat_p = nullptr, *cat_d = nullptr, *cat_w = nullptr;
  CHECK(kg_catalog_upload(prov, &cat.desc, 1, &cat_p) == KG_OK);
  CHECK(kg_catalog_upload(disr, &cat.desc, 1, &cat_d) == KG_OK);
  CHECK(kg_catalog_upload(prov, &cat.desc, 1, &cat_w) == KG_OK);  // the writer's catalog: same context, other handle
  kg_cluster *cl_p = nullptr, *cl_d = nullptr;
  CHECK(kg_cluster_upload(prov, &cluster.desc, &cl_p) == KG_OK);
  CHECK(kg_cluster_upload(disr, &cluster.desc, &cl_d) == KG_OK);

  // ---- the serial run
  std::vector<std::string> want_solve, want_solve_nodes;
  for (auto& b : batches) { want_solve.push_back(solve_digest(prov, cat_p, nullptr, *b, nps)); want_solve_nodes.push_back(solve_digest(prov, cat_p, cl_p, *b, nps)); }
  CHECK(want_solve[0] != want_solve[1]);
  const std::string want_single = records_digest(disr, cat_d, cl_d, none, nps, order, false, 0, 1);
  const std::string want_multi = records_digest(disr, cat_d, cl_d, none, nps, order, true, 0, 1);
  const std::string want_s0 = records_digest(disr, cat_d, cl_d, none, nps, order, false, 0, 2), want_s1 = records_digest(disr, cat_d, cl_d, none, nps, order, false, 1, 2);
  CHECK(!want_single.empty() && !want_multi.empty());
  char want_reqs[4096];
  size_t need = 0;
  CHECK(kg_catalog_query_requirements(cat_p, 5, want_reqs, sizeof want_reqs, &need) == KG_OK);

  // ---- everything at once
  std::atomic<int> failures{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> th;
  for (int k = 0; k < 4; ++k)
    th.emplace_back([&, k] {
      for (int r = 0; r < rounds; ++r) {
        if (solve_digest(prov, cat_p, nullptr, *batches[k], nps) != want_solve[k]) failures++;
        if (solve_digest(prov, cat_p, cl_p, *batches[k], nps) != want_solve_nodes[k]) failures++;
      }
    });
  for (int k = 0; k < 2; ++k)
    th.emplace_back([&, k] {
      for (int r = 0; r < rounds; ++r) {
        if (records_digest(disr, cat_d, cl_d, none, nps, order, k == 1, 0, 1) != (k == 1 ? want_multi : want_single)) failures++;
        if (records_digest(disr, cat_d, cl_d, none, nps, order, false, k, 2) != (k ? want_s1 : want_s0)) failures++;
      }
    });
  th.emplace_back([&] {  // writer: ICE marks and price moves on its own catalog, back and forth
    PoolBuilder pb;
    std::vector<kg_offering_update> ups;
    for (int t = 0; t < n_types; t += 3) ups.push_back({(uint32_t)t, pb.s("z-1a"), pb.s("spot"), 0, -1.0});
    kg_strtab st = pb.pools().strings;
    uint64_t gen = 2;
    while (!stop.load()) {
      for (auto& u : ups) u.available = 0;
      if (kg_catalog_update_offerings(prov, cat_w, &st, ups.data(), (uint32_t)ups.size(), gen++) != KG_OK) failures++;
      for (auto& u : ups) { u.available = 1; u.price = 0.01 + 0.001 * (double)(gen % 7); }
      if (kg_catalog_update_offerings(prov, cat_w, &st, ups.data(), (uint32_t)ups.size(), gen++) != KG_OK) failures++;
    }
  });
  for (int k = 0; k < 2; ++k)
    th.emplace_back([&] {  // readers of the resident catalog and of the context
      char buf[4096], name[64];
      const char* res[2] = {"cpu", "memory"};
      while (!stop.load()) {
        size_t nd = 0;
        if (kg_catalog_query_requirements(cat_p, 5, buf, sizeof buf, &nd) != KG_OK || strcmp(buf, want_reqs) != 0) failures++;
        int64_t cap[2], alloc[2];
        if (kg_catalog_query(cat_p, 7, res, 2, cap, alloc, name, sizeof name) != KG_OK || cap[0] != (2000ll << 3) || alloc[0] != cap[0] - 110) failures++;
        if (kg_catalog_query_requirements(cat_w, 3, buf, sizeof buf, &nd) != KG_OK) failures++;  // (the catalog being updated: any consistent state)
        uint32_t nt = 0;
        uint64_t g = 0;
        if (kg_catalog_size(cat_w, &nt, &g) != KG_OK || nt != (uint32_t)n_types) failures++;
        kg_stats s;
        if (kg_stats_get(prov, &s) != KG_OK || kg_stats_get(disr, &s) != KG_OK) failures++;
        kg_last_error(prov, buf, sizeof buf);
      }
    });
  for (size_t i = 0; i < 6; ++i) th[i].join();
  stop.store(true);
  for (size_t i = 6; i < th.size(); ++i) th[i].join();

  // ---- a device group on top: three contexts and their worker threads, the same decision as the serial run
  {
    kg_group* g = nullptr;
    const int32_t devs[3] = {0, 0, 0};
    CHECK(kg_group_init(devs, 3, &g) == KG_OK);
    kg_catalog* gc[3];
    kg_cluster* gl[3];
    CHECK(kg_group_catalog_upload(g, &cat.desc, 1, gc) == KG_OK && kg_group_cluster_upload(g, &cluster.desc, gl) == KG_OK);
    kg_candidates_desc cd{};
    cd.candidates = order.data(); cd.n_candidates = (uint32_t)order.size(); cd.spot_to_spot_enabled = 1; cd.n_shards = 1;
    for (int multi = 0; multi < 2; ++multi) {
      std::vector<uint32_t> t1(256), t2(256);
      kg_decision d1{}, d2{};
      d1.type_ids = t1.data(); d1.cap_type_ids = 256; d2.type_ids = t2.data(); d2.cap_type_ids = 256;
      uint32_t n = 0;
      double ms = 0;
      CHECK(kg_group_consolidate(g, gc, 1, gl, &none.desc, &nps.desc, &cd, multi, nullptr, 0, &n, &d1, &ms) == KG_OK);
      const std::string& all = multi ? want_multi : want_single;
      kg_catalog* cats[1] = {cat_d};
      CHECK(kg_consolidate_select(disr, cats, 1, multi, (const kg_candidate_record*)all.data(), (uint32_t)(all.size() / sizeof(kg_candidate_record)),
                                  (uint32_t)order.size(), &d2) == KG_OK);
      if (d1.decision != d2.decision || d1.n_candidates != d2.n_candidates || d1.first_candidate != d2.first_candidate || d1.n_types != d2.n_types ||
          memcmp(t1.data(), t2.data(), 4 * std::min<uint32_t>(d1.n_types, 256)) != 0) failures++;
    }
    kg_group_cluster_free(g, gl);
    kg_group_catalog_free(g, gc);
    kg_group_destroy(g);
  }

  kg_cluster_free(cl_p); kg_cluster_free(cl_d);
  kg_catalog_free(cat_p); kg_catalog_free(cat_d); kg_catalog_free(cat_w);
  kg_destroy(prov); kg_destroy(disr);
  if (failures.load()) { fprintf(stderr, "abi_threads: %d result(s) differ from the serial run\n", failures.load()); return 1; }
  printf("abi_threads ok: %d rounds x (4 solve + 2 consolidation threads), 1 writer, 2 readers, a 3-context group\n", rounds);
  return 0;
}
