// karpgpu.cpp -- C ABI of libkarpgpu + host compile of a scheduling problem + result decode.
// See include/karpgpu.h for the reference interfaces each entry point replaces.
#include <chrono>
#include <cmath>
#include <limits>
#include <unordered_set>

#include "compile.hpp"
#include "host_threads.hpp"

using namespace kg;
void kg_cluster_drop_plans(kg_cluster* cl);
using kgdesc::check;

namespace kg {
// KG_HOST_PROF=1: wall time of every section of the host's longer functions to stderr (host-side tuning aid)
struct HostProf {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  const char* name = "start";
  bool on = getenv("KG_HOST_PROF") != nullptr;
  void mark(const char* next) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[kg host] %-42s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(n - t).count());
    t = n; name = next;
  }
};
#define KG_PROF(x) prof_.mark(x)

// ------------------------------------------------------------------ selectors
bool PSel::matches(const std::vector<std::pair<uint32_t, uint32_t>>& labels) const {
  if (nil) return false;
  auto get = [&](uint32_t k, uint32_t* v) {
    for (auto& kv : labels) if (kv.first == k) { *v = kv.second; return true; }
    return false;
  };
  for (auto& kv : ml) {
    uint32_t v;
    if (!get(kv.first, &v) || v != kv.second) return false;
  }
  for (auto& e : ex) {
    uint32_t v = 0;
    bool has = get(e.key, &v);
    bool in = has && std::find(e.vals.begin(), e.vals.end(), v) != e.vals.end();
    if (e.op == KG_OP_IN && !in) return false;
    if (e.op == KG_OP_NOT_IN && in) return false;
    if (e.op == KG_OP_EXISTS && !has) return false;
    if (e.op == KG_OP_DOES_NOT_EXIST && has) return false;
  }
  return true;
}
std::string PSel::canon() const {
  if (nil) return "nil";
  auto m = ml;
  std::sort(m.begin(), m.end());
  std::string s = "L";
  for (auto& kv : m) s += std::to_string(kv.first) + "=" + std::to_string(kv.second) + ",";
  std::vector<std::string> es;
  for (auto& e : ex) {
    auto v = e.vals;
    std::sort(v.begin(), v.end());
    std::string t = std::to_string(e.key) + ":" + std::to_string(e.op) + ":";
    for (auto x : v) t += std::to_string(x) + ",";
    es.push_back(t);
  }
  std::sort(es.begin(), es.end());
  s += "E";
  for (auto& t : es) s += t + ";";
  return s;
}

// ------------------------------------------------------------------ pod parsing (deduplicated by spec signature)
namespace {
struct Sig {
  std::vector<uint32_t> w = std::vector<uint32_t>(256);  // the words appended so far (grown by doubling: no call per word)
  size_t n = 0;
  uint64_t h = 0;  // running hash of those words (the class table is probed with it; the words settle ties)
  void clear() { n = 0; h = 0x9E3779B97F4A7C15ull; }
  void u32(uint32_t v) {
    if (n == w.size()) w.resize(n * 2);
    w[n++] = v;
    h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 29;
  }
  bool same(const std::vector<uint32_t>& o) const { return o.size() == n && memcmp(o.data(), w.data(), n * 4) == 0; }
  std::vector<uint32_t> words() const { return std::vector<uint32_t>(w.begin(), w.begin() + n); }
  void range_labels(const kg_pools& p, kg_range r) { u32(r.count); for (auto& l : kgdesc::labels(p, r)) { u32(l.key); u32(l.value); } }
  void range_reqs(const kg_pools& p, kg_range r) {
    u32(r.count);
    for (auto& q : kgdesc::reqs(p, r)) {
      u32(q.key); u32(q.op); u32((uint32_t)q.min_values); u32(q.values.count);
      for (uint32_t v : kgdesc::values(p, q.values)) u32(v);
    }
  }
  void selector(const kg_pools& p, const kg_selector& s_) { u32(s_.is_nil); range_labels(p, s_.match_labels); range_reqs(p, s_.match_exprs); }
};

PSel parse_selector(kg_ctx* ctx, const kg_pools& p, const kg_selector& s) {
  PSel out;
  out.nil = s.is_nil != 0;
  for (auto& l : kgdesc::labels(p, s.match_labels)) out.ml.push_back({ctx->lab.id(kgdesc::str(p.strings, l.key)), ctx->lab.id(kgdesc::str(p.strings, l.value))});
  for (auto& q : kgdesc::reqs(p, s.match_exprs)) {
    PSel::Ex e{ctx->lab.id(kgdesc::str(p.strings, q.key)), q.op, {}};
    for (uint32_t v : kgdesc::values(p, q.values)) e.vals.push_back(ctx->lab.id(kgdesc::str(p.strings, v)));
    out.ex.push_back(std::move(e));
  }
  return out;
}
std::vector<HReq> parse_req_list(kg_ctx* ctx, const kg_pools& p, kg_range r) {
  std::vector<HReq> out;
  for (auto& q : kgdesc::reqs(p, r)) out.push_back(parse_req(ctx->V, p, q));
  return out;
}
std::vector<PTerm> parse_terms(kg_ctx* ctx, const kg_pods_desc* d, kg_range r) {
  std::vector<PTerm> out;
  for (auto& t : kgdesc::span(d->affinity_terms, d->n_affinity_terms, r, "affinity range")) {
    PTerm a{ctx->V.key(kgdesc::str(d->pools.strings, t.topology_key)), parse_selector(ctx, d->pools, t.selector), {}, t.weight};
    for (uint32_t v : kgdesc::values(d->pools, t.namespaces)) a.ns.push_back(ctx->lab.id(kgdesc::str(d->pools.strings, v)));
    out.push_back(std::move(a));
  }
  return out;
}
// The signature of a pod's scheduling-relevant spec (everything a PodClass holds), as words of descriptor string ids
static void pod_signature(const kg_pods_desc* d, const kg_pod& q, Sig& sg) {
  const kg_pools& p = d->pools;
  sg.clear();
  sg.u32(q.ns);
  sg.range_labels(p, q.labels);
  sg.range_labels(p, q.node_selector);
  sg.u32(q.required_terms.count);
  for (auto& r : kgdesc::ranges(p, q.required_terms)) sg.range_reqs(p, r);
  sg.u32(q.preferred_terms.count);
  for (auto& w : kgdesc::span(d->weighted_terms, d->n_weighted_terms, q.preferred_terms, "weighted range")) { sg.u32((uint32_t)w.weight); sg.range_reqs(p, w.reqs); }
  sg.u32(q.tolerations.count);
  for (auto& t : kgdesc::tolerations(p, q.tolerations)) { sg.u32(t.key); sg.u32(t.value); sg.u32(t.op); sg.u32(t.effect); }
  sg.u32(q.spreads.count);
  for (auto& s : kgdesc::span(d->spreads, d->n_spreads, q.spreads, "spread range")) {
    sg.u32(s.topology_key); sg.u32((uint32_t)s.max_skew); sg.u32((uint32_t)s.min_domains); sg.u32(s.when_unsatisfiable); sg.selector(p, s.selector);
  }
  for (kg_range r : {q.affinity_required, q.affinity_preferred, q.anti_affinity_required, q.anti_affinity_preferred}) {
    sg.u32(r.count);
    for (auto& t : kgdesc::span(d->affinity_terms, d->n_affinity_terms, r, "affinity range")) {
      sg.u32(t.topology_key); sg.selector(p, t.selector); sg.u32((uint32_t)t.weight); sg.u32(t.namespaces.count);
      for (uint32_t v : kgdesc::values(p, t.namespaces)) sg.u32(v);
    }
  }
  sg.u32(q.host_ports.count);
  for (auto& h : kgdesc::span(d->host_ports, d->n_host_ports, q.host_ports, "host port range")) { sg.u32(h.ip); sg.u32((uint32_t)h.port); sg.u32(h.protocol); }
}

// The class of pod `q` in vocabulary ids (the only part of pod parsing that interns strings besides resource names)
static PodClass pod_class(kg_ctx* ctx, const kg_pods_desc* d, const kg_pod& q) {
  const kg_pools& p = d->pools;
  PodClass c;
  c.ns = ctx->lab.id(kgdesc::str(p.strings, q.ns));
  for (auto& l : kgdesc::labels(p, q.labels)) c.labels.push_back({ctx->lab.id(kgdesc::str(p.strings, l.key)), ctx->lab.id(kgdesc::str(p.strings, l.value))});
  std::sort(c.labels.begin(), c.labels.end());
  for (auto& l : kgdesc::labels(p, q.node_selector)) {
    uint32_t k = ctx->V.key(kgdesc::str(p.strings, l.key));
    c.node_selector.push_back(hreq_new(ctx->V, k, KG_OP_IN, {ctx->V.val(k, kgdesc::str(p.strings, l.value))}, -1));
  }
  for (auto& r : kgdesc::ranges(p, q.required_terms)) c.required.push_back(parse_req_list(ctx, p, r));
  for (auto& w : kgdesc::span(d->weighted_terms, d->n_weighted_terms, q.preferred_terms, "weighted range"))
    c.preferred.push_back({w.weight, parse_req_list(ctx, p, w.reqs)});
  for (auto& t : kgdesc::tolerations(p, q.tolerations))
    c.tols.push_back({t.key == KG_NONE ? -1 : (int)ctx->lab.id(kgdesc::str(p.strings, t.key)),
                      t.value == KG_NONE ? ctx->lab.id("") : ctx->lab.id(kgdesc::str(p.strings, t.value)), t.op, t.effect});
  for (auto& s : kgdesc::span(d->spreads, d->n_spreads, q.spreads, "spread range"))
    c.spreads.push_back({ctx->V.key(kgdesc::str(p.strings, s.topology_key)), s.max_skew, s.min_domains, s.when_unsatisfiable, parse_selector(ctx, p, s.selector)});
  c.aff_req = parse_terms(ctx, d, q.affinity_required);
  c.aff_pref = parse_terms(ctx, d, q.affinity_preferred);
  c.anti_req = parse_terms(ctx, d, q.anti_affinity_required);
  c.anti_pref = parse_terms(ctx, d, q.anti_affinity_preferred);
  c.host_ports = q.host_ports.count != 0;
  for (auto& h : kgdesc::span(d->host_ports, d->n_host_ports, q.host_ports, "host port range"))
    c.ports.push_back({h.ip == KG_NONE ? -1 : (int)ctx->lab.id(kgdesc::str(p.strings, h.ip)), h.port, (int)h.protocol});
  return c;
}

// pod classes by signature: open addressing on the running hash, the signature words compared on a hit
struct ClassTable {
  std::vector<std::vector<uint32_t>> sig;
  std::vector<uint64_t> hash;
  uint32_t mask = 1023;
  std::vector<uint32_t> tab = std::vector<uint32_t>(1024, UINT32_MAX);
  // the class with this signature, or UINT32_MAX and *slot_out = where add() puts it
  template <class S>
  uint32_t find(uint64_t h, const S& same, uint32_t* slot_out) const {
    uint32_t slot = (uint32_t)(h >> 20) & mask;
    for (; tab[slot] != UINT32_MAX; slot = (slot + 1) & mask) {
      const uint32_t k = tab[slot];
      if (hash[k] == h && same(sig[k])) return k;
    }
    *slot_out = slot;
    return UINT32_MAX;
  }
  uint32_t add(uint64_t h, std::vector<uint32_t>&& words, uint32_t slot) {
    const uint32_t k = (uint32_t)sig.size();
    sig.push_back(std::move(words));
    hash.push_back(h);
    tab[slot] = k;
    if (sig.size() * 2 > tab.size()) {  // keep the table at most half full
      mask = mask * 2 + 1;
      tab.assign(mask + 1, UINT32_MAX);
      for (uint32_t j = 0; j < sig.size(); ++j) {
        uint32_t s2 = (uint32_t)(hash[j] >> 20) & mask;
        while (tab[s2] != UINT32_MAX) s2 = (s2 + 1) & mask;
        tab[s2] = j;
      }
    }
    return k;
  }
};
}  // namespace

// Three steps.  (1) serial, cheap: resource names -> ids (and the volumes' ids), in pod order.  (2) the pods in contiguous chunks,
// one thread each, nothing shared but the descriptor and step 1's table: signature, chunk-local class, requests, UID.
// (3) serial over the chunks' classes in order (= order of first appearance over all pods, what one serial pass numbers them
// by): the vocabulary is touched here only; then the chunks copy their pools into place.  One chunk below 4 096 pods.
void parsed_pods_delete(ParsedPods* p) { delete p; }

void parse_pods(kg_ctx* ctx, const kg_pods_desc* d, ParsedPods* out, bool add_pods_one) {
  if (!d) return;
  const kg_pools& p = d->pools;
  const uint32_t n = d->n_pods;
  const uint32_t NS = p.strings.count;
  HostProf prof_;
  KG_PROF("parse_pods: resource ids");
  std::vector<int32_t> rid_of_str(NS, -1), vol_rid_of_str;
  for (uint32_t i = 0; i < n; ++i) {
    const kg_pod& q = d->pods[i];
    for (auto& qq : kgdesc::quantities(p, q.requests)) {
      check(qq.name < NS, "quantity name");
      int32_t& rid = rid_of_str[qq.name];
      if (rid < 0) rid = (int32_t)ctx->V.resources.id(kgdesc::str(p.strings, qq.name));
    }
    if (q.volumes.count) {  // volumes per CSI driver -> requests of the driver's internal resource
      if (vol_rid_of_str.empty()) vol_rid_of_str.assign(NS, -1);
      for (auto& l : kgdesc::labels(p, q.volumes)) {
        check(l.key < NS, "volume driver name");
        const std::string drv(kgdesc::str(p.strings, l.key));
        int32_t& rid = vol_rid_of_str[l.key];
        if (rid < 0) rid = (int32_t)ctx->V.resources.id(kVolumeResourcePrefix() + drv);
        out->vols.push_back(ctx->lab.id(drv + "\x1f" + std::string(kgdesc::str(p.strings, l.value))));
      }
    }
  }
  struct Chunk {
    uint32_t lo = 0, hi = 0;
    ClassTable classes;
    std::vector<uint32_t> first_pod, global;  // per chunk-local class: the pod that introduced it; its class in `out`
    std::vector<std::pair<uint32_t, int64_t>> rq;
    std::string uid;
    size_t rq_base = 0, uid_base = 0;
  };
  KG_PROF("parse_pods: chunks");
  const unsigned T = host_chunks(n);
  std::vector<Chunk> chunks(T);
  for (unsigned t = 0; t < T; ++t) { chunks[t].lo = (uint32_t)((uint64_t)n * t / T); chunks[t].hi = (uint32_t)((uint64_t)n * (t + 1) / T); }
  const size_t pod_base = out->pods.size();
  const uint32_t rPods = ctx->rPods;
  out->pods.resize(pod_base + n);
  parallel_chunks(T, [&](unsigned t) {
    Chunk& ch = chunks[t];
    ch.rq.reserve((size_t)(ch.hi - ch.lo) * 4);
    ch.uid.reserve((size_t)(ch.hi - ch.lo) * 24);
    std::vector<std::pair<uint32_t, int64_t>> rl;
    Sig sg;
    for (uint32_t i = ch.lo; i < ch.hi; ++i) {
      const kg_pod& q = d->pods[i];
      pod_signature(d, q, sg);
      uint32_t slot = 0;
      uint32_t cls = ch.classes.find(sg.h, [&](const std::vector<uint32_t>& o) { return sg.same(o); }, &slot);
      if (cls == UINT32_MAX) {
        cls = ch.classes.add(sg.h, sg.words(), slot);
        ch.first_pod.push_back(i);
      }
      PodRef& pr = out->pods[pod_base + i];
      pr.cls = cls;  // (chunk-local until step 3)
      // the pod's requests, merged by resource and sorted by resource id
      rl.clear();
      for (auto& qq : kgdesc::quantities(p, q.requests)) {
        const uint32_t rid = (uint32_t)rid_of_str[qq.name];
        bool found = false;
        for (auto& kv : rl) if (kv.first == rid) { kv.second += qq.milli; found = true; }
        if (!found) rl.push_back({rid, qq.milli});
      }
      if (add_pods_one) {  // resources.RequestsForPods adds pods: 1
        bool found = false;
        for (auto& kv : rl) if (kv.first == rPods) { kv.second += 1000; found = true; }
        if (!found) rl.push_back({rPods, 1000});
      }
      for (auto& l : kgdesc::labels(p, q.volumes)) {
        const uint32_t rid = (uint32_t)vol_rid_of_str[l.key];
        bool found = false;
        for (auto& kv : rl) if (kv.first == rid) { kv.second += 1000; found = true; }
        if (!found) rl.push_back({rid, 1000});
      }
      std::sort(rl.begin(), rl.end());
      pr.rq_off = (uint32_t)ch.rq.size(); pr.rq_n = (uint32_t)rl.size();
      ch.rq.insert(ch.rq.end(), rl.begin(), rl.end());
      const std::string_view u = kgdesc::str(p.strings, q.uid);
      pr.uid_off = (uint32_t)ch.uid.size(); pr.uid_len = (uint32_t)u.size();
      ch.uid.append(u.data(), u.size());
      pr.creation = q.creation_ts;
      pr.node = q.node;
      pr.reschedulable = q.reschedulable != 0;
      pr.priority = q.priority;
      pr.deletion_cost = q.deletion_cost;
    }
  });
  KG_PROF("parse_pods: classes");
  ClassTable all;
  const uint32_t class_base = (uint32_t)out->classes.size();
  size_t rq_total = out->rq.size(), uid_total = out->uid_pool.size();
  for (Chunk& ch : chunks) {
    ch.global.resize(ch.first_pod.size());
    for (uint32_t k = 0; k < ch.first_pod.size(); ++k) {
      std::vector<uint32_t>& words = ch.classes.sig[k];
      uint32_t slot = 0;
      uint32_t g = all.find(ch.classes.hash[k], [&](const std::vector<uint32_t>& o) { return o == words; }, &slot);
      if (g == UINT32_MAX) {
        out->classes.push_back(pod_class(ctx, d, d->pods[ch.first_pod[k]]));
        g = all.add(ch.classes.hash[k], std::move(words), slot);
      }
      ch.global[k] = class_base + g;
    }
    ch.rq_base = rq_total; rq_total += ch.rq.size();
    ch.uid_base = uid_total; uid_total += ch.uid.size();
  }
  check(rq_total < UINT32_MAX && uid_total < UINT32_MAX, "pod request / UID pools exceed 4 GiB");
  KG_PROF("parse_pods: pools");
  out->rq.resize(rq_total);
  out->uid_pool.resize(uid_total);
  parallel_chunks(T, [&](unsigned t) {
    Chunk& ch = chunks[t];
    for (uint32_t i = ch.lo; i < ch.hi; ++i) {
      PodRef& pr = out->pods[pod_base + i];
      pr.cls = ch.global[pr.cls];
      pr.rq_off += (uint32_t)ch.rq_base;
      pr.uid_off += (uint32_t)ch.uid_base;
    }
    std::copy(ch.rq.begin(), ch.rq.end(), out->rq.begin() + ch.rq_base);
    memcpy(&out->uid_pool[0] + ch.uid_base, ch.uid.data(), ch.uid.size());
  });
  std::sort(out->vols.begin(), out->vols.end());
  out->vols_shared = std::adjacent_find(out->vols.begin(), out->vols.end()) != out->vols.end();
  KG_PROF("parse_pods: end");
}

// Volume counts are modelled as additive requests; that is exact as long as every volume belongs to one pod of the call.
static void check_volumes_unshared(const ParsedPods& pending, const kg_cluster* cluster) {
  bool shared = pending.vols_shared || (cluster && cluster->bound.vols_shared);
  if (!shared && cluster && !pending.vols.empty()) {
    const auto& a = pending.vols;
    const auto& b = cluster->bound.vols;
    for (size_t i = 0, j = 0; i < a.size() && j < b.size() && !shared;) { if (a[i] == b[j]) shared = true; else if (a[i] < b[j]) ++i; else ++j; }
  }
  if (shared) throw Unsupported("a volume is mounted by more than one pod of this call (volume-limit counts become set unions): not on the device path");
}

// ------------------------------------------------------------------ compile
namespace {

struct Template {
  uint32_t index;  // nodepool index in the descriptor
  uint32_t part = 0;  // which source catalog its instance types come from (merged views; 0 otherwise)
  std::string name;
  int weight;
  HReqs reqs;       // NodeClaimTemplate requirements
  HReqs spec_reqs;  // spec requirements + labels (topology domain universe)
  std::vector<PTaint> taints;
  bool has_limits;
  std::vector<std::pair<uint32_t, int64_t>> limits, daemon;
};

// state of one pod class after k Relax steps
struct RelaxState {
  std::vector<std::vector<HReq>> required;
  std::vector<std::pair<int, std::vector<HReq>>> preferred;
  std::vector<PTol> tols;
  std::vector<PSpread> spreads;
  std::vector<PTerm> aff_pref, anti_pref;
};

struct HGroup {
  int type;
  uint32_t key;
  int max_skew, min_domains;
  std::set<uint32_t> ns;
  PSel sel;
  std::vector<HReqs> filter;  // spread only
  bool inverse = false;
  bool selects(const PodClass& c) const { return ns.count(c.ns) && sel.matches(c.labels); }
};

std::string reqs_canon(const HReqs& r) {
  std::string s;
  for (auto& q : r.rs) {
    s += std::to_string(q.key) + (q.compl_ ? "!" : "=");
    for (auto v : q.vals) s += std::to_string(v) + ",";
    if (q.has_gt) s += ">" + std::to_string(q.gt);
    if (q.has_lt) s += "<" + std::to_string(q.lt);
    if (q.min_values >= 0) s += "m" + std::to_string(q.min_values);
    s += ";";
  }
  return s;
}
std::string group_canon(const HGroup& g) {
  std::string s = std::to_string(g.key) + "|" + std::to_string(g.type) + "|" + std::to_string(g.max_skew) + "|" + std::to_string(g.min_domains) + "|";
  for (auto n : g.ns) s += std::to_string(n) + ",";
  s += "|" + g.sel.canon() + "|";
  std::vector<std::string> fs;
  for (auto& f : g.filter) fs.push_back(reqs_canon(f));
  std::sort(fs.begin(), fs.end());
  for (auto& f : fs) s += f + "#";
  return s;
}

bool tolerates(const std::vector<PTaint>& taints, const std::vector<PTol>& tols) {
  for (auto& taint : taints) {
    bool ok = false;
    for (auto& t : tols) {
      if (t.effect != KG_EFFECT_NONE && t.effect != taint.effect) continue;
      if (t.key >= 0 && (uint32_t)t.key != taint.key) continue;
      if (t.op == KG_TOL_EXISTS || t.value == taint.value) { ok = true; break; }
    }
    if (!ok) return false;
  }
  return true;
}

struct Interner {
  std::unordered_map<std::string, uint32_t> ids;
  uint32_t get(const std::string& k, bool* fresh) {
    auto it = ids.find(k);
    if (it != ids.end()) { *fresh = false; return it->second; }
    uint32_t i = (uint32_t)ids.size();
    ids.emplace(k, i);
    *fresh = true;
    return i;
  }
};

}  // namespace

// Solve-local numbering of the values of one active label key: the catalog's own numbering first (so the catalog's resident
// value->types rows, offering masks and zone -> zone-id table apply unchanged), then the values only this batch mentions
// (pod / NodePool requirements, labels of the cluster's nodes) in order of appearance.  Requirement rows, topology domain
// counters and node label ids on the device use these ids: their size is a property of the batch and the catalog, never
// of what the context has seen before.
struct LocalVocab {
  const std::vector<uint32_t>* cat_vals = nullptr;                  // catalog-local id -> context value id
  const std::unordered_map<uint32_t, uint32_t>* cat_map = nullptr;
  std::vector<uint32_t> extra;
  std::unordered_map<uint32_t, uint32_t> xmap;
  uint32_t ncat() const { return cat_vals ? (uint32_t)cat_vals->size() : 0; }
  uint32_t size() const { return ncat() + (uint32_t)extra.size(); }
  uint32_t global(uint32_t l) const { return l < ncat() ? (*cat_vals)[l] : extra[l - ncat()]; }
  int find(uint32_t g) const {
    if (cat_map) { auto f = cat_map->find(g); if (f != cat_map->end()) return (int)f->second; }
    auto f = xmap.find(g);
    return f == xmap.end() ? -1 : (int)f->second;
  }
  uint32_t id(uint32_t g) {
    const int l = find(g);
    if (l >= 0) return (uint32_t)l;
    const uint32_t n = size();
    extra.push_back(g);
    xmap.emplace(g, n);
    return n;
  }
};

struct Compiled {
  kg_ctx* ctx;
  kg_catalog* cat;
  std::vector<LocalVocab> lv;        // per active slot
  std::vector<Template> templates;
  std::vector<uint32_t> node_order;  // scan position -> cluster node index
  std::vector<uint32_t> slot_key;    // active slot -> global key
  std::vector<uint16_t> aoff, aw;
  std::vector<int8_t> slot2spec;
  std::vector<uint8_t> spec_slot;
  std::vector<HReqs> reqsets;        // WITH hostname entries (identity)
  std::vector<uint32_t> act_res;     // active resource -> global resource id
  int R = 0, Ka = 0, Wa = 0, P = 0, N = 0, G = 0, GH = 0, Cmax = 0, TW = 0, T = 0, NP = 0, S = 0;
  DevProblem dp{};
  PrepassTables pre{};
  DeviceBuffer prob_dev, state_dev, sims_dev;
  size_t prob_cap = 0, state_cap = 0, sims_cap = 0;  // real sizes of the (possibly pooled, larger) allocations
  SimState sim{};  // device pointers
  std::vector<uint8_t> state_image;
  size_t state_bytes = 0, state_init_bytes = 0;  // whole state slot / its initialised prefix (the only part uploaded for a Solve)
  uint32_t qcap = 0;
  bool wave_ok = false;  // the single-wave solve applies (no topology groups, no existing nodes, no minValues, tables fit LDS)
  bool spec_ok = false;  // ... and so does the speculative multi-wave one: no requirement set holds a negative operator on a custom label key
  std::vector<uint32_t> queue_order, pod_variant0;  // pods in Queue order; initial variant of every pod
  // simulations: what every bound pod of the universe adds to the initial topology counts (it is taken out of the
  // simulations that reschedule it): gcount index g * 64 + d, or 0x80000000 | hcount index
  std::vector<std::vector<uint32_t>> contrib;
  std::vector<uint64_t> gdom_univ;   // [G] domains registered from the NodePools / instance types alone
  size_t o_nports, o_cports, o_crow = 0;  // o_crow: behind the state image (device-only, never uploaded)
  size_t o_ntouch = 0, o_tnode = 0, o_tadd = 0, o_tports = 0, o_thash = 0;
  uint32_t thash_mask = 0;
  size_t o_nreq, o_creq, o_chead, o_chstamp, o_cmask, o_cbits, o_cmeta, o_cnpods, o_ctmpl, o_order, o_remaining, o_gcount, o_gdom, o_gpos, o_hcount, o_hpos, o_queue,
      o_podkind, o_excl, o_podindex, o_podreason, o_scratch, o_ordg, o_header, o_counters, o_nqueue;
  // price-ordered truncation on the device (k_price_order_truncate): records of the first trunc_claims NodeClaims behind the state
  size_t o_trec = 0;
  uint32_t trunc_words = 0, trunc_claims = 0, trunc_max = 0;
  TruncArgs trunc{};
  bool fixed = false;  // the device buffers are carved from a FixedArena (kg_solve_noalloc): nothing to give back
  ~Compiled() {
    // (back to the context's pool; every path that builds or drops a Compiled holds ctx->mu)
    if (fixed) return;
    ctx->pool_give(prob_dev.ptr, prob_cap);
    ctx->pool_give(state_dev.ptr, state_cap);
    ctx->pool_give(sims_dev.ptr, sims_cap);
  }
};

static SimState sim_at(const Compiled& C, uint8_t* base);

static int find_slot(const Compiled& C, uint32_t key) {
  for (size_t a = 0; a < C.slot_key.size(); ++a) if (C.slot_key[a] == key) return (int)a;
  return -1;
}

// Preferences.Relax on the class state; returns false when nothing is left to relax
static bool relax_step(RelaxState& st, bool tolerate_pns, kg_ctx* ctx) {
  if (st.required.size() > 1) { st.required.erase(st.required.begin()); return true; }
  auto drop_heaviest = [](auto& terms, auto weight) {
    if (terms.empty()) return false;
    std::stable_sort(terms.begin(), terms.end(), [&](auto& a, auto& b) { return weight(a) > weight(b); });
    terms.erase(terms.begin());
    return true;
  };
  if (drop_heaviest(st.aff_pref, [](const PTerm& t) { return t.weight; })) return true;
  if (drop_heaviest(st.anti_pref, [](const PTerm& t) { return t.weight; })) return true;
  if (drop_heaviest(st.preferred, [](const std::pair<int, std::vector<HReq>>& t) { return t.first; })) return true;
  for (size_t i = 0; i < st.spreads.size(); ++i) {
    if (st.spreads[i].when == KG_SCHEDULE_ANYWAY) {
      st.spreads[i] = st.spreads.back();
      st.spreads.pop_back();
      return true;
    }
  }
  if (tolerate_pns) {
    for (auto& t : st.tols) if (t.key < 0 && t.op == KG_TOL_EXISTS && t.effect == KG_EFFECT_PREFER_NO_SCHEDULE) return false;
    st.tols.push_back({-1, ctx->lab.id(""), KG_TOL_EXISTS, KG_EFFECT_PREFER_NO_SCHEDULE});
    return true;
  }
  return false;
}

static HReqs pod_requirements(const Vocab& V, const PodClass& c, const RelaxState& st, bool strict) {
  HReqs r;
  for (auto& q : c.node_selector) r.add(V, q);
  if (!strict && !st.preferred.empty()) {
    size_t best = 0;
    for (size_t i = 1; i < st.preferred.size(); ++i) if (st.preferred[i].first > st.preferred[best].first) best = i;
    for (auto& q : st.preferred[best].second) r.add(V, q);
  }
  if (!st.required.empty()) for (auto& q : st.required[0]) r.add(V, q);
  return r;
}

// The catalog a call compiles against: the one catalog passed, or a merged view of the several passed (GetInstanceTypes is
// resolved per NodePool through its own EC2NodeClass, cloudprovider.go:172-190, 254-265: NodePools may name different
// catalogs).  The view is kept with the context, keyed on the identity of what it was built from.
static kg_catalog* catalog_view(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs) {
  for (uint32_t k = 0; k < n_catalogs; ++k) check(catalogs[k] && catalogs[k]->ctx == ctx && catalogs[k]->part_begin.empty(), "catalog handle");
  if (n_catalogs == 1) return catalogs[0];
  std::vector<uint64_t> key;
  for (uint32_t k = 0; k < n_catalogs; ++k) { key.push_back(catalogs[k]->uid); key.push_back(catalogs[k]->generation); key.push_back(catalogs[k]->tab_epoch); }
  for (auto* m : ctx->merged) if (m->part_key == key) { m->last_use = ++ctx->merged_clock; return m; }
  if (n_catalogs > 60) throw Unsupported("more than 60 catalogs in one call");
  std::unique_ptr<kg_catalog> m(new kg_catalog());
  m->ctx = ctx;
  m->generation = 0;
  m->part_key = key;
  Vocab& V = ctx->V;
  for (uint32_t k = 0; k < n_catalogs; ++k) {
    m->part_begin.push_back((uint32_t)m->types.size());
    const uint32_t pv = V.val(ctx->kPart, "c" + std::to_string(k));
    for (const CatType& t : catalogs[k]->types) {
      m->types.push_back(t);
      m->types.back().reqs.add(V, hreq_new(V, ctx->kPart, KG_OP_IN, {pv}, -1));
    }
  }
  m->part_begin.push_back((uint32_t)m->types.size());
  build_catalog_tables(ctx, m.get());
  if (ctx->merged.size() >= 2) {
    const size_t old = ctx->merged[0]->last_use <= ctx->merged[1]->last_use ? 0 : 1;
    hal::dfree(ctx->merged[old]->tab.dev.ptr);
    delete ctx->merged[old];
    ctx->merged.erase(ctx->merged.begin() + old);
  }
  m->last_use = ++ctx->merged_clock;
  ctx->merged.push_back(m.release());
  return ctx->merged.back();
}
// ... for a call that names NodePools: their catalog indices and the generation each expects are checked first (KG_ESTALE)
static kg_catalog* effective_catalog(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, const kg_nodepools_desc* nps) {
  for (uint32_t i = 0; i < nps->n_nodepools; ++i) {
    const kg_nodepool& q = nps->nodepools[i];
    check(q.catalog < n_catalogs && catalogs[q.catalog], "nodepool catalog index");
    const kg_catalog* c = catalogs[q.catalog];
    if (q.catalog_generation != 0 && q.catalog_generation != c->generation)
      throw Stale("nodepool " + std::to_string(i) + " expects catalog generation " + std::to_string(q.catalog_generation) + ", the resident catalog has " +
                  std::to_string(c->generation) + ": upload or update it first");
  }
  return catalog_view(ctx, catalogs, n_catalogs);
}

struct SolveInputs {
  kg_ctx* ctx;
  kg_catalog* cat;
  kg_cluster* cluster;
  const kg_nodepools_desc* nps;
  ParsedPods* pending;
  int sim_cmax = 0;  // > 0: compile for a batch of simulations over subsets of `pending` with at most this many pods each
  FixedArena* fixed = nullptr;  // device tables from this region instead of the context's pool; no allocation of any kind
  const std::vector<int32_t>* universe_bound = nullptr;  // simulations: bound-pod index of every pod of `pending`, -1 = a pending pod
  uint32_t n_shared = 0;                                  // simulations: pods [0, n_shared) take part in every simulation
};

// The Queue's order (byCPUAndMemoryDescending, then creation time, then UID) over `keys`: an LSD radix sort on the three
// numeric fields -- 16-bit digits, the digits no key differs in skipped (cpu and memory requests are a few round values,
// creation times of a snapshot lie within days: five or six passes in practice) -- then the runs of full ties ordered by UID.
// Stable, so equal UIDs keep their input order: the permutation std::stable_sort with the same comparison gives.
struct QKey { int64_t cpu, mem, creation; uint32_t idx; };
template <class UidOf>
static void queue_sort(std::vector<QKey>& keys, UidOf uid_of) {
  const size_t n = keys.size();
  if (n < 2) return;
  if (n < 256) {
    std::stable_sort(keys.begin(), keys.end(), [&](const QKey& a, const QKey& b) {
      if (a.cpu != b.cpu) return a.cpu > b.cpu;
      if (a.mem != b.mem) return a.mem > b.mem;
      if (a.creation != b.creation) return a.creation < b.creation;
      return uid_of(a.idx) < uid_of(b.idx);
    });
    return;
  }
  // ascending sort fields: descending ones complemented, signed ones biased
  auto f_cre = [](const QKey& k) { return (uint64_t)k.creation ^ (1ull << 63); };
  auto f_mem = [](const QKey& k) { return ~((uint64_t)k.mem ^ (1ull << 63)); };
  auto f_cpu = [](const QKey& k) { return ~((uint64_t)k.cpu ^ (1ull << 63)); };
  std::vector<QKey> tmp(n);
  std::vector<uint32_t> hist(65536);
  QKey* src = keys.data();
  QKey* dst = tmp.data();
  auto pass_field = [&](auto field) {
    uint64_t diff = 0;
    const uint64_t first = field(src[0]);
    for (size_t i = 1; i < n; ++i) diff |= field(src[i]) ^ first;
    for (int sh = 0; sh < 64; sh += 16) {
      if (!((diff >> sh) & 0xFFFF)) continue;
      std::fill(hist.begin(), hist.end(), 0u);
      for (size_t i = 0; i < n; ++i) hist[(field(src[i]) >> sh) & 0xFFFF]++;
      uint32_t run = 0;
      for (auto& h : hist) { const uint32_t c = h; h = run; run += c; }
      for (size_t i = 0; i < n; ++i) dst[hist[(field(src[i]) >> sh) & 0xFFFF]++] = src[i];
      std::swap(src, dst);
    }
  };
  pass_field(f_cre);
  pass_field(f_mem);
  pass_field(f_cpu);
  if (src != keys.data()) memcpy(keys.data(), src, n * sizeof(QKey));
  for (size_t i = 0; i < n;) {
    size_t j = i + 1;
    while (j < n && keys[j].cpu == keys[i].cpu && keys[j].mem == keys[i].mem && keys[j].creation == keys[i].creation) ++j;
    if (j - i > 1) std::stable_sort(keys.begin() + i, keys.begin() + j, [&](const QKey& a, const QKey& b) { return uid_of(a.idx) < uid_of(b.idx); });
    i = j;
  }
}
// Place of every bound pod of a snapshot in the Queue order among the bound pods (a property of the snapshot: simulations sort
// their few pending pods only and merge).  Reads `bound` and nothing else: kg_cluster_upload runs it beside the caller's thread.
static void bound_queue_rank(const ParsedPods& bound, uint32_t rCPU, uint32_t rMemory, std::vector<uint32_t>* rank) {
  const size_t nb = bound.pods.size();
  std::vector<QKey> bk(nb);
  for (size_t i = 0; i < nb; ++i) {
    int64_t c = 0, m = 0;
    const PodRef& pr = bound.pods[i];
    for (uint32_t k = pr.rq_off; k < pr.rq_off + pr.rq_n; ++k) { if (bound.rq[k].first == rCPU) c = bound.rq[k].second; if (bound.rq[k].first == rMemory) m = bound.rq[k].second; }
    bk[i] = QKey{c, m, pr.creation, (uint32_t)i};
  }
  queue_sort(bk, [&](uint32_t i) { return bound.uid(bound.pods[i]); });
  rank->assign(nb, 0);
  for (size_t r = 0; r < nb; ++r) (*rank)[bk[r].idx] = (uint32_t)r;
}
// Builds every device table for one Solve over `in.pending`.
static std::unique_ptr<Compiled> compile_problem(const SolveInputs& in, bool stats) {
  HostProf prof_;
  kg_ctx* ctx = in.ctx;
  Vocab& V = ctx->V;
  auto Cp = std::make_unique<Compiled>();
  Compiled& C = *Cp;
  C.ctx = ctx;
  C.cat = in.cat;
  const kg_nodepools_desc* nps = in.nps;
  const kg_pools& npp = nps->pools;

  KG_PROF("templates (NodePools ordered by weight d");
  // ---- templates (NodePools ordered by weight desc, stable)
  std::set<uint32_t> well_known;
  for (uint32_t v : kgdesc::values(npp, nps->well_known_labels)) well_known.insert(V.key(kgdesc::str(npp.strings, v)));
  bool tolerate_pns = false;
  for (uint32_t i = 0; i < nps->n_nodepools; ++i) {
    const kg_nodepool& q = nps->nodepools[i];
    Template t;
    t.index = i;
    t.name = std::string(kgdesc::str(npp.strings, q.name));
    t.weight = q.weight;
    t.spec_reqs = parse_reqs(V, npp, q.requirements);
    for (auto& l : kgdesc::labels(npp, q.labels)) {
      uint32_t k = V.key(kgdesc::str(npp.strings, l.key));
      t.spec_reqs.add(V, hreq_new(V, k, KG_OP_IN, {V.val(k, kgdesc::str(npp.strings, l.value))}, -1));
    }
    t.reqs = t.spec_reqs;
    t.reqs.add(V, hreq_new(V, ctx->kNodePool, KG_OP_IN, {V.val(ctx->kNodePool, t.name)}, -1));
    for (auto& tn : kgdesc::taints(npp, q.taints)) {
      t.taints.push_back({ctx->lab.id(kgdesc::str(npp.strings, tn.key)), tn.value == KG_NONE ? ctx->lab.id("") : ctx->lab.id(kgdesc::str(npp.strings, tn.value)), tn.effect});
      if (tn.effect == KG_EFFECT_PREFER_NO_SCHEDULE) tolerate_pns = true;
    }
    t.has_limits = q.has_limits != 0;
    t.limits = parse_rl(V, npp, q.limits);
    t.daemon = parse_rl(V, npp, q.daemon_overhead);
    if (!in.cat->part_begin.empty()) {  // merged view: the template's types are those of its own catalog
      t.part = q.catalog;
      t.reqs.add(V, hreq_new(V, ctx->kPart, KG_OP_IN, {V.val(ctx->kPart, "c" + std::to_string(q.catalog))}, -1));
    }
    C.templates.push_back(std::move(t));
  }
  std::stable_sort(C.templates.begin(), C.templates.end(), [](const Template& a, const Template& b) { return a.weight > b.weight; });
  const int NP = (int)C.templates.size();
  if (NP == 0 || NP > 64) throw Unsupported("need 1..64 nodepools");
  C.NP = NP;

  KG_PROF("relax chains per pod class");
  // ---- relax chains per pod class
  const ParsedPods& PP = *in.pending;
  const size_t NC = PP.classes.size();
  std::vector<std::vector<RelaxState>> chains(NC);
  for (size_t ci = 0; ci < NC; ++ci) {
    const PodClass& c = PP.classes[ci];
    RelaxState st{c.required, c.preferred, c.tols, c.spreads, c.aff_pref, c.anti_pref};
    chains[ci].push_back(st);
    while (relax_step(st, tolerate_pns, ctx)) {
      chains[ci].push_back(st);
      if (chains[ci].size() > 64) throw Unsupported("relaxation chain longer than 64");
    }
  }

  KG_PROF("topology groups (all created at NewTopol");
  // ---- topology groups (all created at NewTopology from the unrelaxed specs)
  std::vector<HGroup> groups;
  Interner ginterner, iinterner;
  std::vector<uint32_t> inv_ids;  // group index of each inverse group (by inverse interner id)
  auto make_filter = [&](const PodClass& c) {
    std::vector<HReqs> f;
    HReqs sel;
    for (auto& q : c.node_selector) sel.add(V, q);
    if (c.required.empty()) { f.push_back(sel); return f; }
    for (auto& term : c.required) {
      HReqs r = sel;
      for (auto& q : term) r.add(V, q);
      f.push_back(std::move(r));
    }
    return f;
  };
  auto ns_of = [&](const PodClass& c, const PTerm& t) {
    std::set<uint32_t> ns;
    if (t.ns.empty()) ns.insert(c.ns); else ns.insert(t.ns.begin(), t.ns.end());
    return ns;
  };
  auto regular_group = [&](HGroup g) {
    bool fresh;
    std::string k = group_canon(g);
    uint32_t id = ginterner.get(k, &fresh);
    static thread_local std::vector<uint32_t>* map = nullptr;
    (void)map;
    if (fresh) { groups.push_back(std::move(g)); return (uint32_t)groups.size() - 1; }
    (void)id;
    // find existing: linear map kept alongside
    return UINT32_MAX;
  };
  (void)regular_group;
  std::unordered_map<std::string, uint32_t> gmap, imap;
  auto get_group = [&](HGroup g, std::unordered_map<std::string, uint32_t>& m) {
    std::string k = group_canon(g);
    auto it = m.find(k);
    if (it != m.end()) return it->second;
    uint32_t id = (uint32_t)groups.size();
    groups.push_back(std::move(g));
    m.emplace(k, id);
    return id;
  };
  auto class_groups = [&](const PodClass& c, const RelaxState& st, std::vector<uint32_t>* owned) {
    for (auto& s : st.spreads) {
      HGroup g{TG_SPREAD, s.key, s.max_skew, s.min_domains, {c.ns}, s.sel, make_filter(c), false};
      // nodeFilter is derived from the pod's CURRENT required terms (MakeTopologyNodeFilter)
      owned->push_back(get_group(std::move(g), gmap));
    }
    auto add_terms = [&](const std::vector<PTerm>& ts, int type) {
      for (auto& t : ts) {
        HGroup g{type, t.key, INT32_MAX, -1, ns_of(c, t), t.sel, {}, false};
        owned->push_back(get_group(std::move(g), gmap));
      }
    };
    add_terms(c.aff_req, TG_AFFINITY);
    add_terms(st.aff_pref, TG_AFFINITY);
    add_terms(c.anti_req, TG_ANTI);
    add_terms(st.anti_pref, TG_ANTI);
  };
  // regular groups of every class, level 0 first so the static set equals NewTopology's
  std::vector<std::vector<std::vector<uint32_t>>> owned(NC);
  for (size_t ci = 0; ci < NC; ++ci) {
    owned[ci].resize(chains[ci].size());
    for (size_t lv = 0; lv < chains[ci].size(); ++lv) {
      PodClass cc = PP.classes[ci];
      cc.required = chains[ci][lv].required;  // node filter follows the relaxed required terms
      class_groups(cc, chains[ci][lv], &owned[ci][lv]);
    }
  }
  const size_t n_regular = groups.size();
  // inverse anti-affinity groups: pending pods' required terms, then bound pods' required terms
  std::vector<std::vector<uint32_t>> inv_owned(NC);
  for (size_t ci = 0; ci < NC; ++ci) {
    const PodClass& c = PP.classes[ci];
    for (auto& t : c.anti_req) {
      HGroup g{TG_ANTI, t.key, INT32_MAX, -1, ns_of(c, t), t.sel, {}, true};
      inv_owned[ci].push_back(get_group(std::move(g), imap));
    }
  }
  const kg_cluster* cl = in.cluster;
  std::vector<std::vector<uint32_t>> bound_inv_owned;
  if (cl) {
    bound_inv_owned.resize(cl->bound.classes.size());
    for (size_t ci = 0; ci < cl->bound.classes.size(); ++ci) {
      const PodClass& c = cl->bound.classes[ci];
      for (auto& t : c.anti_req) {
        HGroup g{TG_ANTI, t.key, INT32_MAX, -1, ns_of(c, t), t.sel, {}, true};
        bound_inv_owned[ci].push_back(get_group(std::move(g), imap));
      }
    }
  }
  (void)n_regular;
  const int G = (int)groups.size();
  if (G > 65535) throw Unsupported("too many topology groups");
  C.G = G;

  KG_PROF("requirement sets of every variant, templ");
  // ---- requirement sets of every variant, template and node filter
  std::vector<HReqs>& reqsets = C.reqsets;
  std::unordered_map<std::string, uint32_t> rsmap;
  auto rs_id = [&](const HReqs& r) {
    std::string k = reqs_canon(r);
    auto it = rsmap.find(k);
    if (it != rsmap.end()) return it->second;
    uint32_t id = (uint32_t)reqsets.size();
    reqsets.push_back(r);
    rsmap.emplace(k, id);
    return id;
  };
  std::vector<uint32_t> tmpl_rs(NP);
  for (int t = 0; t < NP; ++t) tmpl_rs[t] = rs_id(C.templates[t].reqs);
  struct VarTmp { uint32_t rs, rs_strict, tolset; };
  std::vector<std::vector<VarTmp>> vartmp(NC);
  std::vector<std::vector<PTol>> tolsets;
  auto tol_id = [&](const std::vector<PTol>& t) {
    for (size_t i = 0; i < tolsets.size(); ++i) {
      if (tolsets[i].size() != t.size()) continue;
      bool eq = true;
      for (size_t j = 0; j < t.size(); ++j)
        eq = eq && tolsets[i][j].key == t[j].key && tolsets[i][j].value == t[j].value && tolsets[i][j].op == t[j].op && tolsets[i][j].effect == t[j].effect;
      if (eq) return (uint32_t)i;
    }
    tolsets.push_back(t);
    return (uint32_t)tolsets.size() - 1;
  };
  for (size_t ci = 0; ci < NC; ++ci) {
    for (auto& st : chains[ci]) {
      HReqs all = pod_requirements(V, PP.classes[ci], st, false);
      HReqs strict = st.preferred.empty() ? all : pod_requirements(V, PP.classes[ci], st, true);
      vartmp[ci].push_back({rs_id(all), rs_id(strict), tol_id(st.tols)});
    }
  }
  std::vector<uint32_t> filters;
  std::vector<std::pair<uint32_t, uint32_t>> gfilter(G, {0, 0});
  for (int g = 0; g < G; ++g) {
    if (groups[g].type != TG_SPREAD || groups[g].inverse) continue;
    bool trivial = true;
    for (auto& f : groups[g].filter) if (!f.rs.empty()) trivial = false;
    // a filter with an empty term always matches (empty Requirements are compatible with anything)
    bool has_empty = false;
    for (auto& f : groups[g].filter) if (f.rs.empty()) has_empty = true;
    if (trivial || has_empty) continue;
    gfilter[g].first = (uint32_t)filters.size();
    for (auto& f : groups[g].filter) filters.push_back(rs_id(f));
    gfilter[g].second = (uint32_t)groups[g].filter.size();
  }
  const int S = (int)reqsets.size();
  C.S = S;
  if (tolsets.size() > 4096) throw Unsupported("too many toleration sets");
  const int TS = (int)std::max<size_t>(tolsets.size(), 1);

  KG_PROF("active keys: every key of a requirement ");
  // ---- active keys: every key of a requirement set (except hostname) + non-hostname topology keys
  std::vector<uint8_t> is_active(V.keys.size(), 0), is_topo(V.keys.size(), 0);
  for (auto& r : reqsets) for (auto& q : r.rs) if (q.key != ctx->kHostname) is_active[q.key] = 1;
  for (auto& g : groups) if (g.key != ctx->kHostname) { is_active[g.key] = 1; is_topo[g.key] = 1; }
  // keys on which a type may carry several values are always handled as merged sets
  std::vector<uint8_t> is_multi(V.keys.size(), 0);
  for (auto& t : in.cat->types) for (auto& r : t.reqs.rs) if (r.vals.size() > 1) is_multi[r.key] = 1;
  is_multi[ctx->kZone] = is_multi[ctx->kCT] = is_multi[ctx->kZid] = 1;
  for (uint32_t k = 0; k < V.keys.size(); ++k) {
    if (!is_active[k]) continue;
    C.slot_key.push_back(k);
  }
  const int Ka = (int)C.slot_key.size();
  if (Ka > KG_MAXKA) throw Unsupported("more than 32 active label keys");
  C.Ka = Ka;
  // ---- solve-local value numbering of every active key: the catalog's values, then what this batch adds (requirement
  // sets, labels of the cluster's nodes -- deleting nodes too: their bound pods count in topology domains)
  if (!in.cat->tab.dev.ptr || in.cat->tab.R != (uint32_t)V.resources.size()) {
    if (in.fixed) throw Unsupported("kg_solve_noalloc: the resident catalog tables must be rebuilt for a resource first seen in this batch (needs an allocation): run kg_solve once");
    build_catalog_tables(ctx, in.cat);
  }
  C.lv.resize(std::max(Ka, 1));
  {
    const CatalogTables& CT0 = in.cat->tab;
    std::vector<int> slot_of(V.keys.size(), -1);
    for (int a = 0; a < Ka; ++a) {
      const uint32_t k = C.slot_key[a];
      slot_of[k] = a;
      if (k < CT0.cvals.size()) { C.lv[a].cat_vals = &CT0.cvals[k]; C.lv[a].cat_map = &CT0.cmap[k]; }
    }
    for (auto& r : reqsets) for (auto& q : r.rs) if (slot_of[q.key] >= 0) for (uint32_t v : q.vals) C.lv[slot_of[q.key]].id(v);
    for (auto& tp : C.templates) for (auto& q : tp.spec_reqs.rs) if (slot_of[q.key] >= 0) for (uint32_t v : q.vals) C.lv[slot_of[q.key]].id(v);
    if (in.cluster)
      for (auto& nd : in.cluster->nodes) for (auto& kv : nd.labels) if (slot_of[kv.first] >= 0) C.lv[slot_of[kv.first]].id(kv.second);
  }
  auto lid = [&](int a, uint32_t g) -> uint32_t {  // solve-local id of a value the batch is known to mention
    const int l = C.lv[a].find(g);
    if (l < 0) throw std::logic_error("value outside the solve-local vocabulary");
    return (uint32_t)l;
  };
  C.aoff.resize(std::max(Ka, 1)); C.aw.resize(std::max(Ka, 1)); C.slot2spec.assign(std::max(Ka, 1), -1);
  int Wa = 0;
  uint32_t wk_slots = 0;
  for (int a = 0; a < Ka; ++a) {
    const uint32_t k = C.slot_key[a];
    const uint32_t nlocal = C.lv[a].size();
    if (nlocal >= 0xFFFF) throw Unsupported("more than 65534 values on one label key in one solve");
    const int w = (int)((nlocal + 63) / 64);
    C.aoff[a] = (uint16_t)Wa;
    C.aw[a] = (uint16_t)std::max(w, 1);
    Wa += C.aw[a];
    if (well_known.count(k)) wk_slots |= 1u << a;
    if (is_multi[k] || is_topo[k]) {
      if (nlocal > 64) throw Unsupported("topology / multi-valued key with more than 64 values in this solve (catalog + batch)");
      if (C.spec_slot.size() >= KG_MAXSPEC) throw Unsupported("more than 6 special keys");
      C.slot2spec[a] = (int8_t)C.spec_slot.size();
      C.spec_slot.push_back((uint8_t)a);
    }
  }
  C.Wa = std::max(Wa, 1);
  if (C.Wa > KG_ROWW) throw Unsupported("requirement rows wider than 64 words (label vocabulary too large)");
  const int nspec = (int)C.spec_slot.size();
  auto spec_of_key = [&](uint32_t key) { int a = find_slot(C, key); return a < 0 ? -1 : (int)C.slot2spec[a]; };

  KG_PROF("bounds tables");
  // ---- bounds tables
  std::vector<std::vector<int64_t>> gts(Ka), lts(Ka);
  for (auto& r : reqsets)
    for (auto& q : r.rs) {
      int a = find_slot(C, q.key);
      if (a < 0) continue;
      if (q.has_gt) gts[a].push_back(q.gt);
      if (q.has_lt) lts[a].push_back(q.lt);
    }
  std::vector<uint32_t> bnd_off(std::max(Ka, 1), 0), bndmask_off(std::max(Ka, 1), 0);
  std::vector<int64_t> gtval((size_t)std::max(Ka, 1) * 16, 0), ltval((size_t)std::max(Ka, 1) * 16, 0);
  std::vector<uint64_t> gtmask, ltmask;
  for (int a = 0; a < Ka; ++a) {
    auto uniq = [](std::vector<int64_t>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(gts[a]); uniq(lts[a]);
    if (gts[a].size() > 16 || lts[a].size() > 16) throw Unsupported("more than 16 distinct Gt/Lt bounds on one key");
    bnd_off[a] = a * 16;
    bndmask_off[a] = (uint32_t)gtmask.size();
    const uint32_t k = C.slot_key[a];
    const int w = C.aw[a];
    gtmask.resize(gtmask.size() + (size_t)16 * w, 0);
    ltmask.resize(ltmask.size() + (size_t)16 * w, 0);
    const uint32_t nl = C.lv[a].size();
    for (size_t i = 0; i < gts[a].size(); ++i) {
      gtval[a * 16 + i] = gts[a][i];
      for (uint32_t l = 0; l < nl; ++l) {
        const uint32_t v = C.lv[a].global(l);
        if (V.isint[k][v] && V.ival[k][v] > gts[a][i]) gtmask[bndmask_off[a] + i * w + (l >> 6)] |= 1ull << (l & 63);
      }
    }
    for (size_t i = 0; i < lts[a].size(); ++i) {
      ltval[a * 16 + i] = lts[a][i];
      for (uint32_t l = 0; l < nl; ++l) {
        const uint32_t v = C.lv[a].global(l);
        if (V.isint[k][v] && V.ival[k][v] < lts[a][i]) ltmask[bndmask_off[a] + i * w + (l >> 6)] |= 1ull << (l & 63);
      }
    }
  }
  auto gt_idx = [&](int a, int64_t x) { return (uint32_t)(std::lower_bound(gts[a].begin(), gts[a].end(), x) - gts[a].begin()) + 1; };
  auto lt_idx = [&](int a, int64_t x) { return (uint32_t)(std::lower_bound(lts[a].begin(), lts[a].end(), x) - lts[a].begin()) + 1; };

  KG_PROF("requirement-set rows");
  // ---- requirement-set rows
  std::vector<uint32_t> rs_off(S + 1, 0), e_meta, e_bits, rs_host_meta(std::max(S, 1), 0), rs_host_off(S + 1, 0), rs_host_nodes;
  std::vector<uint8_t> e_slot;
  std::vector<uint64_t> rs_bits;
  std::vector<uint8_t> rs_host_claim_ok(std::max(S, 1), 1);
  KG_PROF("existing nodes (needed for hostname list");
  // ---- existing nodes (needed for hostname lists): scan order
  std::vector<const CNode*> nodes;
  if (cl) {
    for (size_t i = 0; i < cl->nodes.size(); ++i) if (!cl->nodes[i].deleting) C.node_order.push_back((uint32_t)i);
    std::stable_sort(C.node_order.begin(), C.node_order.end(), [&](uint32_t a, uint32_t b) {
      const CNode &x = cl->nodes[a], &y = cl->nodes[b];
      if (x.initialized != y.initialized) return x.initialized;
      return x.name < y.name;
    });
    for (uint32_t i : C.node_order) nodes.push_back(&cl->nodes[i]);
  }
  const int N = (int)nodes.size(), NW = std::max((N + 63) / 64, 1);
  C.N = N;
  bool neg_custom = false;
  for (int s = 0; s < S; ++s) {
    rs_off[s] = (uint32_t)e_slot.size();
    rs_host_off[s] = (uint32_t)rs_host_nodes.size();
    int ne = 0;
    for (auto& q : reqsets[s].rs) {
      if (q.key == ctx->kHostname) {
        rs_host_meta[s] = rm_make(q.compl_, q.has_gt ? 1 : 0, q.has_lt ? 1 : 0);
        rs_host_claim_ok[s] = q.compl_ && !q.has_gt && !q.has_lt;
        for (int n = 0; n < N; ++n)
          if (std::binary_search(q.vals.begin(), q.vals.end(), nodes[n]->hostname_val)) rs_host_nodes.push_back((uint32_t)n);
        continue;
      }
      const int a = find_slot(C, q.key);
      // NotIn / DoesNotExist on a label key that is not well known: a NodeClaim that lacks the key rejects pods that select
      // on it until such a pod has joined -- the one case where NodeClaim.Add can BECOME feasible, which the speculative
      // kernel's argument excludes (solve_spec.hpp)
      if (!((wk_slots >> a) & 1) && ((q.compl_ && !q.vals.empty()) || (!q.compl_ && q.vals.empty() && !q.has_gt && !q.has_lt))) neg_custom = true;
      e_slot.push_back((uint8_t)a);
      e_meta.push_back(rm_make(q.compl_, q.has_gt ? gt_idx(a, q.gt) : 0, q.has_lt ? lt_idx(a, q.lt) : 0));
      e_bits.push_back((uint32_t)rs_bits.size());
      size_t base = rs_bits.size();
      rs_bits.resize(base + C.aw[a], 0);
      for (uint32_t v : q.vals) { const uint32_t l = lid(a, v); rs_bits[base + (l >> 6)] |= 1ull << (l & 63); }
      ne++;
    }
    if (ne > KG_MAXE) throw Unsupported("requirement set with more than 8 keys");
  }
  rs_off[S] = (uint32_t)e_slot.size();
  rs_host_off[S] = (uint32_t)rs_host_nodes.size();
  const uint32_t E = (uint32_t)e_slot.size();

  KG_PROF("resources and catalog tables");
  // ---- resources and catalog tables
  // intern every resource name first, then make sure the catalog tables cover them
  // active resources: requested by a pending pod, part of a template's daemon overhead, or named by a limit.
  // Everything else can never decide a placement (Fits only looks at requested resources).
  const int RG = (int)V.resources.size();
  std::vector<int> ract(RG, -1);
  {
    std::vector<uint8_t> used(RG, 0);
    for (auto& kv : in.pending->rqs()) if (kv.second != 0) used[kv.first] = 1;  // (every pair in the pool belongs to a pod of this solve)
    for (auto& tp : C.templates) {
      for (auto& kv : tp.daemon) if (kv.second != 0) used[kv.first] = 1;
      if (tp.has_limits) for (auto& kv : tp.limits) used[kv.first] = 1;
    }
    used[ctx->rCPU] = used[ctx->rMemory] = used[ctx->rPods] = 1;
    for (int r = 0; r < RG; ++r) if (used[r]) { ract[r] = (int)C.act_res.size(); C.act_res.push_back((uint32_t)r); }
  }
  const int R = (int)C.act_res.size();
  if (R > KG_MAXR) throw Unsupported("more than 8 distinct resources requested / limited in one solve");
  C.R = R;
  const CatalogTables& CT = in.cat->tab;  // (rebuilt above when the batch named a resource the tables do not cover)
  const int T = (int)CT.T, TW = (int)CT.TW;
  C.T = T; C.TW = TW;

  KG_PROF("lists, variants");
  // ---- lists, variants
  std::vector<ListEntry> lists;
  std::vector<Variant> variants;
  std::vector<uint32_t> variant_class;  // variant -> pod class (host ports are a property of the class)
  std::vector<uint32_t> class_variant0(NC);
  auto poddom = [&](const HReqs& strict, uint32_t key) {
    const HReq* q = strict.find(key);
    if (!q) return (uint64_t)~0ull;
    const int a = find_slot(C, key);
    uint64_t m = 0;
    for (uint32_t l = 0; a >= 0 && l < C.lv[a].size() && l < 64; ++l) if (q->has(V, C.lv[a].global(l))) m |= 1ull << l;
    return m;
  };
  std::vector<uint16_t> ghidx(G, 0xFFFF);
  int GH = 0;
  for (int g = 0; g < G; ++g) if (groups[g].key == ctx->kHostname) ghidx[g] = (uint16_t)GH++;
  C.GH = GH;
  for (size_t ci = 0; ci < NC; ++ci) {
    const PodClass& c = PP.classes[ci];
    // Record list is the same at every relax level
    std::vector<ListEntry> rec;
    for (int g = 0; g < G; ++g) {
      if (groups[g].inverse) continue;
      if (groups[g].selects(c)) rec.push_back({(uint16_t)g, 0, 0, 0, 0});
    }
    for (uint32_t g : inv_owned[ci]) rec.push_back({(uint16_t)g, LF_INVERSE, 0, 0, 0});
    const uint32_t rec_off = (uint32_t)lists.size();
    lists.insert(lists.end(), rec.begin(), rec.end());
    int32_t next = -1;
    for (int lv = (int)chains[ci].size() - 1; lv >= 0; --lv) {
      const VarTmp& vt = vartmp[ci][lv];
      Variant v{};
      v.rs = vt.rs; v.rs_strict = vt.rs_strict; v.tolset = vt.tolset;
      v.match_off = (uint32_t)lists.size();
      const HReqs& strict = reqsets[vt.rs_strict];
      std::set<uint32_t> seen;
      for (uint32_t g : owned[ci][lv]) {
        if (!seen.insert(g).second) continue;
        if (groups[g].key == ctx->kHostname && strict.find(ctx->kHostname)) throw Unsupported("hostname topology with a hostname node selector");
        uint64_t pd = groups[g].key == ctx->kHostname ? ~0ull : poddom(strict, groups[g].key);
        lists.push_back({(uint16_t)g, (uint8_t)(groups[g].selects(c) ? LF_SELF : 0), 0, (uint32_t)pd, (uint32_t)(pd >> 32)});
      }
      for (int g = 0; g < G; ++g) {
        if (!groups[g].inverse || !groups[g].selects(c)) continue;
        if (groups[g].key == ctx->kHostname && strict.find(ctx->kHostname)) throw Unsupported("hostname topology with a hostname node selector");
        uint64_t pd = groups[g].key == ctx->kHostname ? ~0ull : poddom(strict, groups[g].key);
        lists.push_back({(uint16_t)g, (uint8_t)(LF_INVERSE | LF_SELF), 0, (uint32_t)pd, (uint32_t)(pd >> 32)});
      }
      v.match_cnt = (uint32_t)lists.size() - v.match_off;
      v.record_off = rec_off; v.record_cnt = (uint32_t)rec.size();
      v.next = next;
      v.host_claim_ok = rs_host_claim_ok[vt.rs];
      v.pm_all = reqsets[vt.rs].rs.empty() ? 1u : 0u;
      next = (int32_t)variants.size();
      variants.push_back(v);
      variant_class.push_back((uint32_t)ci);
    }
    class_variant0[ci] = (uint32_t)next;
  }

  KG_PROF("pods");
  // ---- pods
  const int P = (int)PP.pods.size();
  C.P = P;
  if (P >= (1 << 20)) throw Unsupported("more than 2^20 pods in one solve");
  std::vector<int64_t> preq((size_t)std::max(R, 1) * std::max(P, 1), 0);
  std::vector<uint32_t> pod_variant0(std::max(P, 1), 0), queue0(std::max(P, 1), 0);
  for (int p = 0; p < P; ++p) {
    for (uint32_t k = PP.pods[p].rq_off; k < PP.pods[p].rq_off + PP.pods[p].rq_n; ++k) { const auto& kv = PP.rqs()[k]; if (ract[kv.first] >= 0) preq[(size_t)p * R + ract[kv.first]] = kv.second; }
    pod_variant0[p] = class_variant0[PP.pods[p].cls];
    queue0[p] = (uint32_t)p;
  }
  KG_PROF("pods: queue order");
  {
    const int rc = ract[ctx->rCPU], rm = ract[ctx->rMemory];
    auto cpu = [&](uint32_t i) { return preq[(size_t)i * R + rc]; };
    auto mem = [&](uint32_t i) { return preq[(size_t)i * R + rm]; };
    // (keys gathered into one contiguous array first: the order then touches 32 bytes per pod instead of three tables)
    auto qless = [&](const QKey& a, const QKey& b) {
      if (a.cpu != b.cpu) return a.cpu > b.cpu;
      if (a.mem != b.mem) return a.mem > b.mem;
      if (a.creation != b.creation) return a.creation < b.creation;
      return PP.uid(PP.pods[a.idx]) < PP.uid(PP.pods[b.idx]);
    };
    auto qkey = [&](uint32_t p) { return QKey{cpu(p), mem(p), PP.pods[p].creation, p}; };
    std::vector<QKey> keys;
    keys.reserve((size_t)P);
    if (in.universe_bound && cl) {
      // a universe of bound pods: their mutual order is a property of the cluster snapshot (ranked once per kg_cluster),
      // so only the pending pods are sorted here and merged into the bound pods taken in rank order
      const size_t nb = cl->bound.pods.size();
      in.cluster->wait_rank();  // (kg_cluster_upload started it on a thread of its own for large snapshots)
      if (cl->bound_rank.size() != nb) bound_queue_rank(cl->bound, ctx->rCPU, ctx->rMemory, &in.cluster->bound_rank);
      KG_PROF("pods: merge pending into ranked bound");
      std::vector<uint32_t> at_rank(nb, UINT32_MAX);
      std::vector<QKey> pend_keys, bound_keys;
      for (int p = 0; p < P; ++p) {
        const int32_t bi = (*in.universe_bound)[p];
        if (bi >= 0) at_rank[cl->bound_rank[bi]] = (uint32_t)p; else pend_keys.push_back(qkey((uint32_t)p));
      }
      if (pend_keys.empty()) {  // (nothing to merge: the Queue is the bound pods of the universe in rank order, no key needed)
        size_t k = 0;
        for (size_t r = 0; r < nb; ++r) if (at_rank[r] != UINT32_MAX) queue0[k++] = at_rank[r];
      } else {
        bound_keys.reserve((size_t)P);
        for (size_t r = 0; r < nb; ++r) if (at_rank[r] != UINT32_MAX) bound_keys.push_back(qkey(at_rank[r]));
        queue_sort(pend_keys, [&](uint32_t i) { return PP.uid(PP.pods[i]); });
        keys.resize(pend_keys.size() + bound_keys.size());
        std::merge(pend_keys.begin(), pend_keys.end(), bound_keys.begin(), bound_keys.end(), keys.begin(), qless);
        for (int p = 0; p < P; ++p) queue0[p] = keys[p].idx;
      }
    } else {
      for (int p = 0; p < P; ++p) keys.push_back(qkey((uint32_t)p));
      queue_sort(keys, [&](uint32_t i) { return PP.uid(PP.pods[i]); });
      for (int p = 0; p < P; ++p) queue0[p] = keys[p].idx;
    }
  }

  KG_PROF("templates -> device rows");
  // ---- templates -> device rows
  std::vector<int64_t> tmpl_req((size_t)std::max(R, 1) * NP, 0), tmpl_limits0((size_t)std::max(R, 1) * NP, std::numeric_limits<int64_t>::max());
  std::vector<uint8_t> tmpl_has_limits(NP, 0), tmpl_limit_res((size_t)std::max(R, 1) * NP, 0);
  std::vector<uint64_t> tol_tmpl(TS, 0);
  std::vector<uint32_t> tmpl_minv_off(NP + 1, 0), minv_min;
  std::vector<uint8_t> minv_slot;
  for (int t = 0; t < NP; ++t) {
    const Template& tp = C.templates[t];
    for (auto& kv : tp.daemon) if (ract[kv.first] >= 0) tmpl_req[(size_t)ract[kv.first] * NP + t] = kv.second;
    tmpl_has_limits[t] = tp.has_limits && !tp.limits.empty();
    if (tmpl_has_limits[t]) for (auto& kv : tp.limits) { tmpl_limits0[(size_t)ract[kv.first] * NP + t] = kv.second; tmpl_limit_res[(size_t)ract[kv.first] * NP + t] = 1; }
    for (size_t ts = 0; ts < tolsets.size(); ++ts) if (tolerates(tp.taints, tolsets[ts])) tol_tmpl[ts] |= 1ull << t;
    tmpl_minv_off[t] = (uint32_t)minv_slot.size();
    for (auto& q : tp.reqs.rs) {
      if (q.min_values < 0) continue;
      minv_slot.push_back((uint8_t)find_slot(C, q.key));
      minv_min.push_back((uint32_t)q.min_values);
    }
  }
  tmpl_minv_off[NP] = (uint32_t)minv_slot.size();
  unsigned long long tmpl_minv_mask = 0;
  for (int t = 0; t < NP; ++t) if (tmpl_minv_off[t + 1] > tmpl_minv_off[t]) tmpl_minv_mask |= 1ull << t;
  // remaining limits: subtract the capacity of every existing node of the nodepool
  for (int n = 0; n < N; ++n) {
    uint32_t npv = UINT32_MAX;
    for (auto& kv : nodes[n]->labels) if (kv.first == ctx->kNodePool) npv = kv.second;
    if (npv == UINT32_MAX) continue;
    for (int t = 0; t < NP; ++t) {
      if (V.vals[ctx->kNodePool].strs[npv] != C.templates[t].name || !tmpl_has_limits[t]) continue;
      for (auto& kv : nodes[n]->capacity)
        if (ract[kv.first] >= 0 && tmpl_limit_res[(size_t)ract[kv.first] * NP + t]) tmpl_limits0[(size_t)ract[kv.first] * NP + t] -= kv.second;
    }
  }

  KG_PROF("nodes -> device rows");
  // ---- nodes -> device rows
  std::vector<uint8_t> vol_res(std::max(R, 1), 0);
  for (int r = 0; r < R; ++r) vol_res[r] = is_volume_resource(V.resources.strs[C.act_res[r]]);
  std::vector<int64_t> navail((size_t)std::max(R, 1) * std::max(N, 1), 0), nreq0(navail.size(), 0);
  std::vector<uint16_t> nval((size_t)std::max(Ka, 1) * std::max(N, 1), 0xFFFF), ndom((size_t)std::max(nspec, 1) * std::max(N, 1), 0xFFFF);
  std::vector<uint64_t> ntol((size_t)TS * NW, 0);
  for (int n = 0; n < N; ++n) {
    const CNode& nd = *nodes[n];
    bool valid = true;
    for (int r = 0; r < R; ++r) if (vol_res[r]) navail[(size_t)r * N + n] = INT64_MAX / 4;  // (no CSINode limit for the driver: unlimited)
    for (auto& kv : nd.available) { if (ract[kv.first] >= 0) navail[(size_t)ract[kv.first] * N + n] = kv.second; if (kv.second < 0) valid = false; }
    for (auto& kv : nd.daemon) {
      if (ract[kv.first] >= 0) nreq0[(size_t)ract[kv.first] * N + n] = kv.second;
      int64_t av = 0;
      for (auto& a : nd.available) if (a.first == kv.first) av = a.second;
      if (kv.second > av) valid = false;
    }
    for (auto& kv : nd.labels) {
      int a = find_slot(C, kv.first);
      if (a < 0) continue;
      const uint32_t l = lid(a, kv.second);
      nval[(size_t)a * N + n] = (uint16_t)l;
      if (C.slot2spec[a] >= 0) ndom[(size_t)C.slot2spec[a] * N + n] = (uint16_t)l;
    }
    if (valid)
      for (size_t ts = 0; ts < tolsets.size(); ++ts)
        if (tolerates(nd.taints, tolsets[ts])) ntol[ts * NW + (n >> 6)] |= 1ull << (n & 63);
  }
  // unsupported reference quirks on existing nodes (requirements that accumulate on undefined keys)
  if (N > 0) {
    for (int a = 0; a < Ka; ++a) {
      bool some_node_lacks = false;
      for (int n = 0; n < N; ++n) if (nval[(size_t)a * N + n] == 0xFFFF) { some_node_lacks = true; break; }
      if (!some_node_lacks) continue;
      bool neg = false, pos = false;
      for (size_t ci = 0; ci < NC; ++ci)
        for (auto& vt : vartmp[ci]) {
          const HReq* q = reqsets[vt.rs].find(C.slot_key[a]);
          if (!q) continue;
          int oc = q->opclass();
          if (oc == 1 || oc == 3) neg = true; else pos = true;
        }
      if (neg && pos) throw Unsupported("NotIn/DoesNotExist and In/Exists pods on a label some existing node lacks");
      // A node without the key of a topology group is outside every domain of it: ExistingNode.Add cannot bind the
      // group's In{domain} to an undefined key (Requirements.Compatible without AllowUndefined), so the node refuses every
      // pod that consults the group -- what the scan does on the 0xFFFF domain id -- and countDomains skips its pods.
      // Only a pod that first defines the key on the node through a negative operator escapes that rule upstream.
      if (neg) for (auto& g : groups) if (g.key == C.slot_key[a]) throw Unsupported("NotIn/DoesNotExist pods on a topology key some existing node lacks");
    }
  }

  KG_PROF("topology: domain universe, initial count");
  // ---- topology: domain universe, initial counts
  std::vector<Group> dgroups(std::max(G, 1));
  std::vector<uint64_t> gdom0(std::max(G, 1), 0), gpos0(std::max(G, 1), 0);
  std::vector<int32_t> gcount0((size_t)std::max(G, 1) * 64, 0);
  const bool sim_mode = in.sim_cmax > 0;
  const int Cmax = sim_mode ? in.sim_cmax : std::max(P, 1);
  C.Cmax = Cmax;
  const int H = N + Cmax;
  std::vector<int32_t> hcount0(GH > 0 ? (size_t)GH * H : 2, 0), hpos0(std::max(GH, 1), 0);  // (no hostname group: nothing to count -- every simulation copies this)
  if (G > 0) {
    // universe per special key: union over templates x types of (template ∩ type) values, plus template In values
    std::vector<uint64_t> universe(std::max(nspec, 1), 0);
    for (int sp = 0; sp < nspec; ++sp) {
      const int sa = C.spec_slot[sp];
      const uint32_t key = C.slot_key[sa];
      for (auto& tp : C.templates) {
        const HReq* tq = tp.spec_reqs.find(key);
        for (auto& ty : in.cat->types) {
          const HReq* yq = ty.reqs.find(key);
          if (yq && tq) { HReq m = hreq_intersect(V, *tq, *yq); for (uint32_t v : m.vals) universe[sp] |= 1ull << lid(sa, v); }
          else if (yq) { for (uint32_t v : yq->vals) universe[sp] |= 1ull << lid(sa, v); }
        }
        if (tq) for (uint32_t v : tq->vals) universe[sp] |= 1ull << lid(sa, v);  // (.values inserted whatever the operator)
      }
    }
    for (int g = 0; g < G; ++g) {
      Group& d = dgroups[g];
      d.type = (uint8_t)groups[g].type;
      d.max_skew = groups[g].max_skew;
      d.min_domains = groups[g].min_domains;
      d.filter_off = gfilter[g].first; d.filter_cnt = gfilter[g].second;
      d.hidx = ghidx[g];
      if (groups[g].key == ctx->kHostname) d.spec = 0xFF;
      else { d.spec = (uint8_t)spec_of_key(groups[g].key); gdom0[g] = universe[d.spec]; }
    }
    // initial counts from bound pods (Topology.countDomains / updateInverseAffinities)
    if (cl) {
      std::vector<int> pos_of_node(cl->nodes.size(), -1);
      for (int n = 0; n < N; ++n) pos_of_node[C.node_order[n]] = n;
      std::vector<std::vector<uint32_t>> pods_of_class(cl->bound.classes.size());
      for (size_t i = 0; i < cl->bound.pods.size(); ++i) pods_of_class[cl->bound.pods[i].cls].push_back((uint32_t)i);
      auto node_filter_ok = [&](int g, const CNode& nd) {
        if (groups[g].type != TG_SPREAD || groups[g].inverse) return true;
        for (auto& f : groups[g].filter) {
          bool ok = true;
          for (auto& q : f.rs) {
            uint32_t val = UINT32_MAX;
            for (auto& kv : nd.labels) if (kv.first == q.key) val = kv.second;
            if (val == UINT32_MAX) { int oc = q.opclass(); if (!(oc == 1 || oc == 3)) ok = false; }
            else if (!q.has(V, val)) ok = false;
          }
          if (ok) return true;
        }
        return false;
      };
      // Topology.excludedPods = the pods a Solve schedules.  In a batch of simulations the shared pods are excluded
      // from every simulation; a candidate's pods are counted here and taken out again per simulation (C.contrib).
      std::vector<int32_t> univ_of_bound(cl->bound.pods.size(), -1);
      if (in.universe_bound) {
        C.contrib.resize(in.universe_bound->size());
        for (size_t u = 0; u < in.universe_bound->size(); ++u) if ((*in.universe_bound)[u] >= 0) univ_of_bound[(*in.universe_bound)[u]] = (int32_t)u;
      }
      C.gdom_univ = gdom0;
      if ((size_t)std::max(GH, 1) * H >= (1ull << 31)) throw Unsupported("hostname topology table over 2^31 entries");
      auto record = [&](int g, const CNode& nd, int cluster_idx, int32_t u) {
        if (groups[g].key == ctx->kHostname) {
          // hostname domains of nodes outside the scan (deleting) are not tracked
          int pos = pos_of_node[cluster_idx];
          if (pos < 0) return;
          const size_t idx = (size_t)ghidx[g] * H + pos;
          int32_t& c = hcount0[idx];
          if (c == 0) hpos0[ghidx[g]]++;
          c++;
          if (u >= 0) C.contrib[u].push_back(0x80000000u | (uint32_t)idx);
          return;
        }
        for (auto& kv : nd.labels) {
          if (kv.first != groups[g].key) continue;
          const uint32_t d = lid(find_slot(C, kv.first), kv.second);
          if (d > 63) throw Unsupported("topology domain id over 63");
          gcount0[(size_t)g * 64 + d]++;
          gdom0[g] |= 1ull << d;
          gpos0[g] |= 1ull << d;
          if (u >= 0) C.contrib[u].push_back((uint32_t)(g * 64 + d));
        }
      };
      for (size_t ci = 0; ci < cl->bound.classes.size(); ++ci) {
        const PodClass& bc = cl->bound.classes[ci];
        for (int g = 0; g < G; ++g) {
          if (groups[g].inverse) continue;
          if (!groups[g].selects(bc)) continue;
          for (uint32_t pi : pods_of_class[ci]) {
            const int32_t u = univ_of_bound[pi];
            if (u >= 0 && (uint32_t)u < in.n_shared) continue;
            const CNode& nd = cl->nodes[cl->bound.pods[pi].node];
            if (!node_filter_ok(g, nd)) continue;
            record(g, nd, cl->bound.pods[pi].node, u);
          }
        }
        for (uint32_t g : bound_inv_owned[ci])
          for (uint32_t pi : pods_of_class[ci]) {
            const int32_t u = univ_of_bound[pi];
            if (u >= 0 && (uint32_t)u < in.n_shared) continue;
            record((int)g, cl->nodes[cl->bound.pods[pi].node], cl->bound.pods[pi].node, u);
          }
      }
    }
  }
  // lexicographic ranks of the special keys' values
  std::vector<uint16_t> vrank((size_t)std::max(nspec, 1) * 64, 0);
  for (int sp = 0; sp < nspec; ++sp) {
    const LocalVocab& lvs = C.lv[C.spec_slot[sp]];
    const uint32_t key = C.slot_key[C.spec_slot[sp]];
    std::vector<uint32_t> idx(lvs.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return V.vals[key].strs[lvs.global(a)] < V.vals[key].strs[lvs.global(b)]; });
    for (size_t i = 0; i < idx.size(); ++i) if (idx[i] < 64) vrank[sp * 64 + idx[i]] = (uint16_t)i;
  }

  KG_PROF("host ports (scheduling.HostPortUsage): t");
  // ---- host ports (scheduling.HostPortUsage): the distinct (ip, port, protocol) entries of this solve's pods form a
  // vocabulary of <= 64 ids; a node / NodeClaim holds the ids reserved on it in one word, a pod class the ids it reserves
  // and the ids that conflict with one of them (same port and protocol, equal ips or either one 0.0.0.0)
  std::vector<std::array<int, 3>> port_vocab;
  auto port_id = [&](const std::array<int, 3>& h) {
    for (size_t i = 0; i < port_vocab.size(); ++i) if (port_vocab[i] == h) return (int)i;
    if (port_vocab.size() >= 64) throw Unsupported("more than 64 distinct host ports in one solve");
    port_vocab.push_back(h);
    return (int)port_vocab.size() - 1;
  };
  for (auto& c : PP.classes) for (auto& h : c.ports) port_id(h);
  if (cl) for (auto& c : cl->bound.classes) for (auto& h : c.ports) port_id(h);
  auto port_use = [&](const PodClass& c) { uint64_t m = 0; for (auto& h : c.ports) m |= 1ull << port_id(h); return m; };
  auto port_conf = [&](const PodClass& c) {
    uint64_t m = 0;
    for (auto& h : c.ports)
      for (size_t i = 0; i < port_vocab.size(); ++i) {
        const auto& o = port_vocab[i];
        if (o[1] == h[1] && o[2] == h[2] && (o[0] == h[0] || o[0] == -1 || h[0] == -1)) m |= 1ull << i;
      }
    return m;
  };
  std::vector<uint64_t> nports0(std::max(N, 1), 0);
  if (cl && !port_vocab.empty()) {
    std::vector<int> pos_of(cl->nodes.size(), -1);
    for (int n = 0; n < N; ++n) pos_of[C.node_order[n]] = n;
    for (auto& bp : cl->bound.pods) {
      if (bp.node < 0 || (size_t)bp.node >= cl->nodes.size() || pos_of[bp.node] < 0) continue;
      nports0[pos_of[bp.node]] |= port_use(cl->bound.classes[bp.cls]);
    }
  }

  KG_PROF("one precompiled record per pod variant (");
  // ---- one precompiled record per pod variant (what NodeClaim.Add / ExistingNode.Add need of the pod)
  const int NV = (int)variants.size();
  std::vector<VariantCtx> vctx(std::max(NV, 1));
  for (int vi = 0; vi < NV; ++vi) {
    VariantCtx& vc = vctx[vi];
    std::memset(&vc, 0, sizeof vc);
    vc.v = variants[vi];
    vc.hp_use = port_use(PP.classes[variant_class[vi]]);
    vc.hp_conf = port_conf(PP.classes[variant_class[vi]]);
    const uint32_t rs = vc.v.rs;
    vc.ne = rs_off[rs + 1] - rs_off[rs];
    uint32_t used = 0;
    for (uint32_t j = 0; j < vc.ne; ++j) {
      const uint32_t e = rs_off[rs] + j;
      PodEntry& pe = vc.ent[j];
      pe.e = e;
      pe.meta = e_meta[e];
      pe.slot = e_slot[e];
      pe.aoff = C.aoff[pe.slot];
      pe.w = (uint8_t)C.aw[pe.slot];
      pe.spec = C.slot2spec[pe.slot];
      bool anyb = false;
      for (int i = 0; i < pe.w; ++i) anyb |= rs_bits[e_bits[e] + i] != 0;
      pe.opc = (uint8_t)opclass(pe.meta & RM_COMPL, anyb);
      if (used + pe.w <= (uint32_t)KG_EBITS) {
        pe.bits = (uint16_t)used;
        for (int i = 0; i < pe.w; ++i) vc.ebits[used + i] = rs_bits[e_bits[e] + i];
        used += pe.w;
      } else {
        pe.bits = 0xFFFF;
      }
    }
    vc.pzone = vc.pct = ~0ull;
    {
      const int sz = spec_of_key(ctx->kZone), sc = spec_of_key(ctx->kCT);
      for (uint32_t j = 0; j < vc.ne; ++j) {
        const PodEntry& pe = vc.ent[j];
        if (pe.spec < 0 || pe.bits == 0xFFFF || (pe.spec != sz && pe.spec != sc)) continue;
        const uint64_t w = vc.ebits[pe.bits];
        const uint64_t pw = ((pe.meta & (RM_PRESENT | RM_COMPL)) == RM_PRESENT && w != 0) ? w : ~0ull;
        if (pe.spec == sz) vc.pzone = pw; else vc.pct = pw;
      }
    }
    if (vc.v.match_cnt > (uint32_t)KG_MAXLIST) throw Unsupported("pod consults more than 8 topology groups");
    vc.nlist = vc.v.match_cnt;
    for (uint32_t li = 0; li < vc.nlist; ++li) {
      vc.lst[li].le = lists[vc.v.match_off + li];
      vc.lst[li].g = dgroups[vc.lst[li].le.g];
    }
  }

  KG_PROF("per-slot catalog views");
  // ---- per-slot catalog views
  std::vector<uint64_t> kU((size_t)std::max(Ka, 1) * TW, 0), kN((size_t)std::max(Ka, 1) * TW, 0);
  std::vector<uint32_t> ktwv_off(std::max(Ka, 1), 0);
  std::vector<uint16_t> ktwv_n(std::max(Ka, 1), 0);
  for (int a = 0; a < Ka; ++a) {
    const uint32_t k = C.slot_key[a];
    // (a key first seen after the tables were built: no type mentions it)
    const bool known = k < CT.key_row_off.size();
    for (int w = 0; w < TW; ++w) { kU[(size_t)a * TW + w] = known ? CT.kU[k][w] : 0; kN[(size_t)a * TW + w] = known ? CT.kN[k][w] : CT.all_types[w]; }
    if (known && CT.key_row_off[k] != UINT32_MAX) { ktwv_off[a] = CT.key_row_off[k]; ktwv_n[a] = (uint16_t)CT.key_nvals[k]; }
  }

  KG_PROF("pack the problem image");
  // ---- pack the problem image
  // (ExistingNode.Add's fit test requests + pod <= available, kept as pod <= available - requests: one word per resource)
  std::vector<int64_t> nfree0(navail.size());
  for (size_t i = 0; i < navail.size(); ++i) nfree0[i] = navail[i] - nreq0[i];
  Arena A;
  // (one allocation for the image: the big tables are known by now)
  A.bytes.reserve(vctx.size() * sizeof(vctx[0]) + preq.size() * 8 + (pod_variant0.size() + queue0.size()) * 4 + rs_bits.size() * 8 + (kU.size() + kN.size()) * 8 +
                  (gtmask.size() + ltmask.size()) * 8 + nfree0.size() * 8 + (ntol.size() + nports0.size()) * 8 + (ndom.size() + nval.size()) * 2 + (size_t)64 * 1024);
  const size_t o_aoff = A.add(C.aoff), o_aw = A.add(C.aw), o_s2s = A.add(C.slot2spec), o_ss = A.add(C.spec_slot), o_kU = A.add(kU), o_kN = A.add(kN),
               o_ktoff = A.add(ktwv_off), o_ktn = A.add(ktwv_n), o_vrank = A.add(vrank), o_bnd = A.add(bnd_off), o_gtval = A.add(gtval),
               o_ltval = A.add(ltval), o_gtmask = A.add(gtmask), o_ltmask = A.add(ltmask), o_bmo = A.add(bndmask_off), o_rsoff = A.add(rs_off),
               o_eslot = A.add(e_slot), o_emeta = A.add(e_meta), o_ebits = A.add(e_bits), o_rsbits = A.add(rs_bits),
               o_vctx = A.add(vctx), o_pv0 = A.add(pod_variant0), o_preq = A.add(preq),
               o_queue0 = A.add(queue0), o_lists = A.add(lists), o_tmplrs = A.add(tmpl_rs), o_tmplreq = A.add(tmpl_req),
               o_thl = A.add(tmpl_has_limits), o_tl0 = A.add(tmpl_limits0), o_tlr = A.add(tmpl_limit_res), o_toltmpl = A.add(tol_tmpl),
               o_minvoff = A.add(tmpl_minv_off), o_minvslot = A.add(minv_slot), o_minvmin = A.add(minv_min), o_nfree0 = A.add(nfree0),
               o_ntol = A.add(ntol), o_ndom = A.add(ndom), o_groups = A.add(dgroups), o_nports0 = A.add(nports0),
               o_filters = A.add(filters), o_nval = A.add(nval), o_rhm = A.add(rs_host_meta), o_rho = A.add(rs_host_off), o_rhn = A.add(rs_host_nodes);
  // kernel outputs (and the small table uploaded below) live behind the host image: never materialised on the host
  const size_t o_amask = A.reserve_tail((size_t)std::max(E, 1u) * TW * 8), o_podmask = A.reserve_tail((size_t)std::max(NV, 1) * TW * 8),
               o_pctx = A.reserve_tail((size_t)std::max(NV, 1) * sizeof(PodCtx)), o_atab = A.reserve_tail(5 * KG_MAXR * 4),
               o_nlab = A.reserve_tail((size_t)std::max(S, 1) * NW * 8);
  KG_PROF("pack: upload");
  C.prob_dev.bytes = A.total();
  C.fixed = in.fixed != nullptr;
  if (in.fixed) { C.prob_dev.ptr = in.fixed->take(C.prob_dev.bytes, "problem tables"); C.prob_cap = C.prob_dev.bytes; }
  else C.prob_dev.ptr = ctx->pool_take(C.prob_dev.bytes, &C.prob_cap);
  ctx->stage_off = 0;  // (the previous call on this context ended with a wait: nothing of it is on its way any more)
  ctx->upload(C.prob_dev.ptr, A.bytes.data(), A.bytes.size(), !in.fixed);
  hal::dzero((uint8_t*)C.prob_dev.ptr + A.bytes.size(), C.prob_dev.bytes - A.bytes.size());
  KG_PROF("pack: descriptor");
  auto dptr = [&](size_t off) { return (uint8_t*)C.prob_dev.ptr + off; };
  auto cptr = [&](size_t off) { return (uint8_t*)CT.dev.ptr + off; };
  DevProblem& D = C.dp;
  std::vector<uint32_t> atab(5 * KG_MAXR, 0);
  int TWP = 16;
  while (TWP < TW) TWP *= 2;
  if (TWP > 256) throw Unsupported("more than 16384 instance types");
  D.tmpl_minv_mask = tmpl_minv_mask;
  D.has_ports = port_vocab.empty() ? 0 : 1;
  {
    uint32_t thr_total = 0;
    for (int r = 0; r < R; ++r) thr_total += ((const uint32_t*)(CT.image.bytes.data() + CT.o_thr_n))[C.act_res[r]];
    bool ok = !sim_mode && G == 0 && N == 0 && tmpl_minv_mask == 0 && TW <= KG_TWLDS && Ka <= KG_MAXKA && C.Wa <= 64 && thr_total <= (uint32_t)KG_THR_LDS &&
              CT.n_zone_vals * CT.n_ct_vals <= (uint32_t)KG_OFFROWS;
    for (int sp = 0; sp < nspec; ++sp) ok = ok && ktwv_n[C.spec_slot[sp]] <= KG_SPV;
    C.wave_ok = ok;
    // (... and a batch small enough that its NodeClaims' prefilter records stay in LDS: the speculative evaluation tests every
    // claim, where the single-wave kernel stops at the first feasible one)
    C.spec_ok = ok && !neg_custom && P <= 16384;
  }
  D.TWP = TWP; D.NV = NV; D.stats = (stats ? 1 : 0) | (ctx->collect_timers ? 2 : 0) | (ctx->no_fast_path ? 4 : 0) | (ctx->wave_timers ? 8 : 0) |
            (ctx->collect_timers && sim_mode ? 16 : 0) | (ctx->spec_serial ? 64 : 0);  // (DevProblem::stats bit 2 = no wave-local evaluation, bit 4 = simulations leave their time stamps)
  for (int r = 0; r < R; ++r) {
    const uint32_t g = C.act_res[r];
    const uint32_t* toff = (const uint32_t*)(CT.image.bytes.data() + CT.o_thr_off);
    const uint32_t* tn = (const uint32_t*)(CT.image.bytes.data() + CT.o_thr_n);
    const uint32_t* coff = (const uint32_t*)(CT.image.bytes.data() + CT.o_cthr_off);
    const uint32_t* cn = (const uint32_t*)(CT.image.bytes.data() + CT.o_cthr_n);
    atab[0 * KG_MAXR + r] = toff[g]; atab[1 * KG_MAXR + r] = tn[g]; atab[2 * KG_MAXR + r] = coff[g]; atab[3 * KG_MAXR + r] = cn[g]; atab[4 * KG_MAXR + r] = g;
  }
  D.atab = (const uint32_t*)dptr(o_atab);
  ctx->upload(dptr(o_atab), atab.data(), atab.size() * 4, !in.fixed);
  D.R = R; D.Ka = Ka; D.Wa = C.Wa; D.T = T; D.TW = TW; D.NP = NP; D.N = N; D.NW = NW; D.P = P; D.S = S; D.G = G; D.GH = GH; D.H = H;
  D.Cmax = Cmax; D.nspec = nspec; D.NL = (int)lists.size();
  D.slot_zone = spec_of_key(ctx->kZone); D.slot_ct = spec_of_key(ctx->kCT); D.slot_zid = spec_of_key(ctx->kZid);
  D.well_known_slots = wk_slots;
  D.aoff = (const uint16_t*)dptr(o_aoff); D.aw = (const uint16_t*)dptr(o_aw); D.slot2spec = (const int8_t*)dptr(o_s2s);
  D.spec_slot = (const uint8_t*)dptr(o_ss); D.kU = (const uint64_t*)dptr(o_kU); D.kN = (const uint64_t*)dptr(o_kN);
  D.ktwv_off = (const uint32_t*)dptr(o_ktoff); D.ktwv_n = (const uint16_t*)dptr(o_ktn); D.twv = (const uint64_t*)cptr(CT.o_twv);
  D.vrank = (const uint16_t*)dptr(o_vrank); D.bnd_off = (const uint32_t*)dptr(o_bnd); D.gtval = (const int64_t*)dptr(o_gtval);
  D.ltval = (const int64_t*)dptr(o_ltval); D.gtmask = (const uint64_t*)dptr(o_gtmask); D.ltmask = (const uint64_t*)dptr(o_ltmask);
  D.bndmask_off = (const uint32_t*)dptr(o_bmo); D.rs_off = (const uint32_t*)dptr(o_rsoff); D.e_slot = (const uint8_t*)dptr(o_eslot);
  D.e_meta = (const uint32_t*)dptr(o_emeta); D.e_bits = (const uint32_t*)dptr(o_ebits); D.rs_bits = (const uint64_t*)dptr(o_rsbits);
  D.e_amask = (const uint64_t*)dptr(o_amask); D.vctx = (const VariantCtx*)dptr(o_vctx); D.podmask = (const uint64_t*)dptr(o_podmask); D.pctx = (const PodCtx*)dptr(o_pctx); D.pod_variant0 = (const uint32_t*)dptr(o_pv0);
  D.preq = (const int64_t*)dptr(o_preq); D.queue0 = (const uint32_t*)dptr(o_queue0); D.lists = (const ListEntry*)dptr(o_lists);
  D.tmpl_rs = (const uint32_t*)dptr(o_tmplrs); D.tmpl_req = (const int64_t*)dptr(o_tmplreq); D.tmpl_has_limits = (const uint8_t*)dptr(o_thl);
  D.tmpl_limits0 = (const int64_t*)dptr(o_tl0); D.tmpl_limit_res = (const uint8_t*)dptr(o_tlr); D.tol_tmpl = (const uint64_t*)dptr(o_toltmpl);
  D.tmpl_minv_off = (const uint32_t*)dptr(o_minvoff); D.minv_slot = (const uint8_t*)dptr(o_minvslot); D.minv_min = (const uint32_t*)dptr(o_minvmin);
  D.thr_val = (const int64_t*)cptr(CT.o_thr_val); D.thr_mask = (const uint64_t*)cptr(CT.o_thr_mask);
  D.cthr_val = (const int64_t*)cptr(CT.o_cthr_val); D.cthr_mask = (const uint64_t*)cptr(CT.o_cthr_mask); D.valid_types = (const uint64_t*)cptr(CT.o_valid);
  D.cap = (const int64_t*)cptr(CT.o_cap); D.alloc = (const int64_t*)cptr(CT.o_alloc); D.offmask = (const uint64_t*)cptr(CT.o_offmask); D.zone_zid = (const uint8_t*)cptr(CT.o_zone_zid);
  D.n_zone_vals = (int)CT.n_zone_vals; D.n_ct_vals = (int)CT.n_ct_vals;
  D.gp_cap = KG_MAXGP;  // (launch_sim_round lowers it for the batch kernel)
  D.sparse_nodes = sim_mode ? 1u : 0u; D.nfree0 = (const int64_t*)dptr(o_nfree0); D.nports0 = (const uint64_t*)dptr(o_nports0);
  D.nlab = (const uint64_t*)dptr(o_nlab); D.ntol = (const uint64_t*)dptr(o_ntol);
  D.ndom = (const uint16_t*)dptr(o_ndom); D.groups = (const Group*)dptr(o_groups); D.filters = (const uint32_t*)dptr(o_filters);
  PrepassTables& X = C.pre;
  X.E = E; X.nval = (const uint16_t*)dptr(o_nval); X.rs_host_meta = (const uint32_t*)dptr(o_rhm); X.rs_host_off = (const uint32_t*)dptr(o_rho);
  X.rs_host_nodes = (const uint32_t*)dptr(o_rhn); X.e_amask_out = (uint64_t*)dptr(o_amask); X.nlab_out = (uint64_t*)dptr(o_nlab); X.podmask_out = (uint64_t*)dptr(o_podmask); X.pctx_out = (PodCtx*)dptr(o_pctx);

  KG_PROF("simulation state image");
  // ---- simulation state image
  Arena SA;
  std::vector<int64_t> remaining0(tmpl_limits0);
  std::vector<uint32_t> header(16, 0), nqueue{(uint32_t)P};
  std::vector<uint64_t> counters(in.sim_cmax > 0 && !ctx->wave_timers ? 64 : 64 + 16 * 32, 0);  // [64..): per-wave cycle accounting (16 waves x 16 slots; debug kernels of a Solve only)
  // initialised state first (it is what gets uploaded), per-claim / per-pod arrays behind it (zeroed on the device)
  // existing nodes: one Solve keeps its own [R][N] requests / [N] port words; a simulation of a batch keeps a short list of
  // the nodes it touched (behind the image) and a bit per node here
  std::vector<uint64_t> ntouch0(NW, 0), excl0(NW, 0);
  if (!sim_mode) { C.o_nreq = SA.add(nfree0); C.o_nports = SA.add(nports0); C.o_ntouch = 0; }
  else {
    C.o_nreq = C.o_nports = 0; C.o_ntouch = SA.add(ntouch0);
    // node -> entry of the touched list: a simulation touches at most Cmax nodes (one per placed pod); load factor <= 1/2
    uint32_t H = 16;
    while (H < 2u * (uint32_t)Cmax) H *= 2;
    std::vector<uint64_t> thash0(H, 0);
    C.o_thash = SA.add(thash0);
    C.thash_mask = H - 1;
  }
  C.o_excl = SA.add(excl0);
  C.o_remaining = SA.add(remaining0);
  C.o_gcount = SA.add(gcount0); C.o_gdom = SA.add(gdom0); C.o_gpos = SA.add(gpos0); C.o_hcount = SA.add(hcount0); C.o_hpos = SA.add(hpos0);
  // Queue ring entries {pod, variant, lastLen epoch, lastLen value}
  const int PR = sim_mode ? Cmax : std::max(P, 1);  // pod result rows
  C.qcap = (uint32_t)(sim_mode ? Cmax : P) + 1;
  std::vector<uint32_t> queue((size_t)C.qcap * 4, 0);
  if (!sim_mode) for (int i = 0; i < P; ++i) { queue[(size_t)i * 4] = queue0[i]; queue[(size_t)i * 4 + 1] = pod_variant0[queue0[i]]; }
  C.queue_order = queue0;
  C.pod_variant0 = pod_variant0;
  C.o_queue = SA.add(queue);
  C.o_header = SA.add(header);
  C.o_counters = SA.add(counters);
  C.o_nqueue = SA.add(nqueue);
  C.state_init_bytes = (SA.bytes.size() + 255) & ~size_t(255);
  C.o_cports = SA.reserve_tail((size_t)Cmax * 8);
  C.o_creq = SA.reserve_tail((size_t)std::max(R, 1) * Cmax * 8);
  C.o_chead = SA.reserve_tail((size_t)std::max(R, 1) * Cmax * 8);
  C.o_chstamp = SA.reserve_tail((size_t)Cmax * 4);
  C.o_cmask = SA.reserve_tail((size_t)TW * Cmax * 8);
  C.o_cbits = SA.reserve_tail((size_t)C.Wa * Cmax * 8);
  C.o_cmeta = SA.reserve_tail((size_t)std::max(Ka, 1) * Cmax * 4);
  C.o_cnpods = SA.reserve_tail((size_t)Cmax * 4);
  C.o_ctmpl = SA.reserve_tail((size_t)Cmax * 2);
  C.o_order = SA.reserve_tail((size_t)Cmax * 4);
  C.o_podkind = SA.reserve_tail((size_t)PR);
  C.o_podindex = SA.reserve_tail((size_t)PR * 4);
  C.o_podreason = SA.reserve_tail((size_t)PR * 4);
  C.o_tnode = SA.reserve_tail(sim_mode ? (size_t)Cmax * 4 : 8);
  C.o_tadd = SA.reserve_tail(sim_mode ? (size_t)std::max(R, 1) * Cmax * 8 : 8);
  C.o_tports = SA.reserve_tail(sim_mode ? (size_t)Cmax * 8 : 8);
  C.o_scratch = SA.reserve_tail((size_t)(TW + 1) * Cmax * 8);
  C.o_ordg = SA.reserve_tail((size_t)2 * Cmax * 8);
  C.state_bytes = SA.total();
  SA.bytes.resize(C.state_init_bytes, 0);  // (a batch copies this initialised prefix into every simulation's slot)
  C.state_image = std::move(SA.bytes);
  C.o_crow = C.state_bytes;
  if (sim_mode) return Cp;  // the caller lays out one slot per simulation
  size_t tail = C.wave_ok ? (size_t)Cmax * 896 : 8;  // claim rows of the single-wave solve (KG_ROW_WORDS * 8)
  // Results.TruncateInstanceTypes on the device: NodePools without minValues (SatisfiesMinValues may widen the list: host),
  // a catalog whose prices fit the kernel's LDS
  const int max_types = in.nps->max_instance_types;
  if (max_types > 0 && max_types <= 1024 && tmpl_minv_mask == 0 && T <= (int)KG_TRUNC_T) {
    C.trunc_max = (uint32_t)max_types;
    C.trunc_words = trunc_rec_words(C.trunc_max, Ka, C.Wa, R);
    C.trunc_claims = (uint32_t)std::min(Cmax, 8192);
    C.o_trec = (C.state_bytes + tail + 255) & ~size_t(255);
    tail = C.o_trec - C.state_bytes + (size_t)C.trunc_claims * C.trunc_words * 4;
  }
  C.state_dev.bytes = C.state_bytes + tail;
  if (in.fixed) { C.state_dev.ptr = in.fixed->take(C.state_dev.bytes, "solve state"); C.state_cap = C.state_dev.bytes; }
  else C.state_dev.ptr = ctx->pool_take(C.state_dev.bytes, &C.state_cap);
  ctx->upload(C.state_dev.ptr, C.state_image.data(), C.state_init_bytes, !in.fixed);
  hal::dzero((uint8_t*)C.state_dev.ptr + C.state_init_bytes, C.state_bytes - C.state_init_bytes);
  C.sim = sim_at(C, (uint8_t*)C.state_dev.ptr);
  if (C.trunc_claims) {
    TruncArgs& X = C.trunc;
    X.off_begin = (const uint32_t*)cptr(CT.o_off_begin); X.off_zone = (const uint16_t*)cptr(CT.o_off_zone); X.off_ct = (const uint16_t*)cptr(CT.o_off_ct);
    X.off_price = (const double*)cptr(CT.o_off_price); X.off_avail = (const uint8_t*)cptr(CT.o_off_avail); X.name_rank = (const uint16_t*)cptr(CT.o_name_rank);
    X.max_types = C.trunc_max; X.rec = (uint32_t*)((uint8_t*)C.state_dev.ptr + C.o_trec); X.rec_words = C.trunc_words;
  }
  SimState& St = C.sim;
  C.sims_dev.bytes = sizeof(SimState);
  if (in.fixed) { C.sims_dev.ptr = in.fixed->take(sizeof(SimState), "state descriptor"); C.sims_cap = sizeof(SimState); }
  else C.sims_dev.ptr = ctx->pool_take(sizeof(SimState), &C.sims_cap);
  ctx->upload(C.sims_dev.ptr, &St, sizeof(SimState), !in.fixed);
  KG_PROF("end");
  return Cp;
}

// SimState whose arrays live in the state slot starting at `base`
static SimState sim_at(const Compiled& C, uint8_t* base) {
  auto sptr = [&](size_t off) { return base + off; };
  SimState St{};
  St.nports = (uint64_t*)sptr(C.o_nports); St.c_ports = (uint64_t*)sptr(C.o_cports); St.c_row = (uint64_t*)sptr(C.o_crow);
  St.nfree = (int64_t*)sptr(C.o_nreq); St.c_req = (int64_t*)sptr(C.o_creq); St.c_head = (int64_t*)sptr(C.o_chead); St.c_hstamp = (uint32_t*)sptr(C.o_chstamp); St.c_mask = (uint64_t*)sptr(C.o_cmask); St.c_bits = (uint64_t*)sptr(C.o_cbits);
  St.c_meta = (uint32_t*)sptr(C.o_cmeta); St.c_npods = (uint32_t*)sptr(C.o_cnpods); St.c_tmpl = (uint16_t*)sptr(C.o_ctmpl); St.c_stamp = (uint32_t*)sptr(C.o_order);
  St.remaining = (int64_t*)sptr(C.o_remaining); St.gcount = (int32_t*)sptr(C.o_gcount); St.gdom = (uint64_t*)sptr(C.o_gdom); St.gpos = (uint64_t*)sptr(C.o_gpos);
  St.hcount = (int32_t*)sptr(C.o_hcount); St.hpos = (int32_t*)sptr(C.o_hpos); St.queue = (uint32_t*)sptr(C.o_queue); St.pod_kind = (uint8_t*)sptr(C.o_podkind);
  St.pod_index = (int32_t*)sptr(C.o_podindex); St.pod_reason = (uint32_t*)sptr(C.o_podreason); St.scratch = (uint64_t*)sptr(C.o_scratch);
  St.header = (uint32_t*)sptr(C.o_header); St.counters = (uint64_t*)sptr(C.o_counters);
  St.excl = (const uint64_t*)sptr(C.o_excl); St.ordg = (unsigned long long*)sptr(C.o_ordg);
  St.ntouch = (uint64_t*)sptr(C.o_ntouch); St.tnode = (uint32_t*)sptr(C.o_tnode); St.tadd = (int64_t*)sptr(C.o_tadd); St.tports = (uint64_t*)sptr(C.o_tports);
  St.thash = (uint64_t*)sptr(C.o_thash); St.thash_mask = C.thash_mask;
  St.n_shared = (uint32_t)C.P; St.pod_base = 0; St.qcap = C.qcap;
  return St;
}

// ------------------------------------------------------------------ truncation (InstanceTypes.OrderByPrice / Truncate / SatisfiesMinValues)
struct ClaimView {
  std::vector<uint32_t> types;   // catalog indices
  uint64_t zmask, ctmask;        // allowed zone / capacity-type value ids (zone-id folded into zmask)
  std::vector<std::pair<uint32_t, int>> minvals;  // (global key, min)
};

static double cheapest_price(const CatType& t, uint64_t zmask, uint64_t ctmask) {
  double best = std::numeric_limits<double>::max();
  bool any = false;
  for (auto& o : t.offerings) {
    if (!o.available || !((zmask >> o.lz) & 1) || !((ctmask >> o.lct) & 1)) continue;  // (masks over the catalog's own value numbering)
    if (!any || o.price < best) { best = o.price; any = true; }
  }
  return best;
}
// InstanceTypes.SatisfiesMinValues; *needed = how many of the (ordered) types it takes
static bool satisfies_min_values(const kg_catalog* cat, const std::vector<uint32_t>& types, const std::vector<std::pair<uint32_t, int>>& minvals,
                                 int* needed = nullptr) {
  if (needed) *needed = 0;
  if (minvals.empty() || types.empty()) return true;
  std::vector<std::set<uint32_t>> seen(minvals.size());
  int n = 0;
  for (uint32_t t : types) {
    bool ok = true;
    n++;
    for (size_t i = 0; i < minvals.size(); ++i) {
      const HReq* r = cat->types[t].reqs.find(minvals[i].first);
      if (r) seen[i].insert(r->vals.begin(), r->vals.end());
      if ((int)seen[i].size() < minvals[i].second) ok = false;
    }
    if (ok) { if (needed) *needed = n; return true; }
  }
  if (needed) *needed = n;
  return false;
}
// returns false when minValues breaks under truncation (the caller keeps the ordered, untruncated list)
static bool truncate_types(const kg_catalog* cat, ClaimView& cv, int max_types) {
  std::vector<std::pair<double, uint32_t>> keyed;
  for (uint32_t t : cv.types) keyed.push_back({cheapest_price(cat->types[t], cv.zmask, cv.ctmask), t});
  std::stable_sort(keyed.begin(), keyed.end(), [&](auto& a, auto& b) {
    if (a.first == b.first) return cat->types[a.second].name < cat->types[b.second].name;
    return a.first < b.first;
  });
  std::vector<uint32_t> ordered, trunc;
  for (auto& kv : keyed) ordered.push_back(kv.second);
  trunc.assign(ordered.begin(), ordered.begin() + std::min<size_t>(ordered.size(), (size_t)std::max(max_types, 0)));
  if (!cv.minvals.empty() && !satisfies_min_values(cat, trunc, cv.minvals)) { cv.types = ordered; return false; }
  cv.types = trunc;
  return true;
}

}  // namespace kg

// ================================================================== C ABI
static int32_t fail(kg_ctx* ctx, int32_t code, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}

static int32_t solve_impl(kg_ctx* ctx, kg_catalog* const* catalogs, kg_cluster* cluster, const kg_pods_desc* pods,
                          const kg_nodepools_desc* nodepools, kg_results* out, bool add_pods_one, FixedArena* fixed = nullptr);

#define KG_GUARD_BEGIN try {
#define KG_GUARD_END(ctx)                                                                   \
  } catch (const kg::Unsupported& e) { return fail(ctx, KG_EUNSUPPORTED, e.what()); }        \
  catch (const kg::Stale& e) { return fail(ctx, KG_ESTALE, e.what()); }                      \
  catch (const kgdesc::DescError& e) { return fail(ctx, KG_EINVAL, e.what()); }              \
  catch (const std::bad_alloc&) { return fail(ctx, KG_ENOMEM, "out of host memory"); }       \
  catch (const std::length_error& e) { return fail(ctx, KG_ENOMEM, e.what()); }              \
  catch (const std::exception& e) { return fail(ctx, KG_EDEVICE, e.what()); }

extern "C" {

int32_t kg_init(int32_t device, kg_ctx** out) {
  if (!out) return KG_EINVAL;
  *out = nullptr;
  int dev = 0;
  std::string err = hal::init(device, &dev);
  if (!err.empty()) { fprintf(stderr, "kg_init: %s\n", err.c_str()); return KG_EDEVICE; }
  auto* ctx = new kg_ctx();
  ctx->device = dev;
  if (const char* sw = getenv("KG_SPEC_WAVES")) { const int w = atoi(sw); if (w >= 2 && w <= 8) ctx->spec_waves = w; }  // (measurements only)
  try {
    hal::use_device(dev, nullptr);
    ctx->stream = hal::stream_create();
  } catch (const std::exception& e) {
    fprintf(stderr, "kg_init: %s\n", e.what());
    delete ctx;
    return KG_EDEVICE;
  }
  Vocab& V = ctx->V;
  ctx->kHostname = V.key("kubernetes.io/hostname");
  ctx->kZone = V.key("topology.kubernetes.io/zone");
  ctx->kCT = V.key("karpenter.sh/capacity-type");
  ctx->kZid = V.key("topology.k8s.aws/zone-id");
  ctx->kPart = V.key("karpgpu.internal/catalog");
  ctx->kNodePool = V.key("karpenter.sh/nodepool");
  ctx->kInstanceType = V.key("node.kubernetes.io/instance-type");
  ctx->rCPU = V.resources.id("cpu");
  ctx->rMemory = V.resources.id("memory");
  ctx->rPods = V.resources.id("pods");
  *out = ctx;
  return KG_OK;
}

int32_t kg_device_count(int32_t* n) {
  if (!n) return KG_EINVAL;
  *n = hal::device_count();
  return KG_OK;
}

int32_t kg_destroy(kg_ctx* ctx) {
  if (!ctx) return KG_OK;
  void* stream = ctx->stream;
  {
    CtxLock lk(ctx);  // (the pooled device buffers are freed on the context's GPU; a call still running finishes first)
  }
  delete ctx;
  hal::stream_destroy(stream);
  return KG_OK;
}

int32_t kg_last_error(kg_ctx* ctx, char* buf, size_t n) {
  if (!ctx || !buf || n == 0) return KG_EINVAL;
  CtxLock lk(ctx);
  size_t m = std::min(n - 1, ctx->last_error.size());
  std::memcpy(buf, ctx->last_error.data(), m);
  buf[m] = 0;
  return KG_OK;
}

int32_t kg_stats_get(kg_ctx* ctx, kg_stats* out) {
  if (!ctx || !out) return KG_EINVAL;
  CtxLock lk(ctx);
  *out = ctx->stats;
  return KG_OK;
}

const char* kg_backend(void) { return hal::backend(); }

int32_t kg_set_collect_stats(kg_ctx* ctx, int32_t enabled) {
  if (!ctx) return KG_EINVAL;
  CtxLock lk(ctx);
  ctx->collect_stats = (enabled & 1) != 0;
  ctx->collect_timers = (enabled & 2) != 0;
  ctx->no_fast_path = (enabled & 16) != 0;
  ctx->no_spec = (enabled & 64) != 0;
  ctx->spec_serial = (enabled & 128) != 0;
  ctx->spec_fake_fail = (enabled & 256) != 0;
  ctx->wave_timers = (enabled & 32) != 0;
  ctx->block_threads = (enabled & 4) ? 1024 : ((enabled & 8) ? 512 : 256);  // debugging aid: per-phase ticks of k_solve printed to stderr
  return KG_OK;
}

// sizeof of every element struct, in the order karpenter_amd/abi.py lists them (layout check for bindings)
int32_t kg_abi_sizes(uint32_t* out, uint32_t cap) {
  const uint32_t s[] = {sizeof(kg_range), sizeof(kg_requirement), sizeof(kg_quantity), sizeof(kg_label), sizeof(kg_taint), sizeof(kg_toleration),
                        sizeof(kg_selector), sizeof(kg_offering), sizeof(kg_instance_type), sizeof(kg_spread), sizeof(kg_pod_affinity_term),
                        sizeof(kg_weighted_term), sizeof(kg_host_port), sizeof(kg_pod), sizeof(kg_node), sizeof(kg_nodepool), sizeof(kg_claim_out),
                        sizeof(kg_req_out), sizeof(kg_candidate_record), sizeof(kg_ec2_device), sizeof(kg_ec2_type), sizeof(kg_subnet),
                        sizeof(kg_spot_price), sizeof(kg_od_price), sizeof(kg_unavailable), sizeof(kg_offering_update)};
  const uint32_t n = sizeof(s) / sizeof(s[0]);
  for (uint32_t i = 0; i < n && i < cap; ++i) out[i] = s[i];
  return (int32_t)n;
}

int32_t kg_catalog_upload(kg_ctx* ctx, const kg_catalog_desc* desc, uint64_t generation, kg_catalog** out) {
  if (!ctx || !desc || !out) return KG_EINVAL;
  CtxLock lk(ctx);
  std::unique_ptr<kg_catalog> cat(new kg_catalog());
  KG_GUARD_BEGIN
  cat->ctx = ctx;
  cat->generation = generation;
  parse_catalog(ctx, desc, cat.get());
  build_catalog_tables(ctx, cat.get());
  *out = cat.release();
  return KG_OK;
  KG_GUARD_END(ctx)
}

int32_t kg_catalog_size(kg_catalog* cat, uint32_t* n_types, uint64_t* generation) {
  if (!cat) return KG_EINVAL;
  CtxLock lk(cat->ctx);  // (kg_catalog_update_offerings moves the generation)
  if (n_types) *n_types = (uint32_t)cat->types.size();
  if (generation) *generation = cat->generation;
  return KG_OK;
}

int32_t kg_catalog_query(kg_catalog* cat, uint32_t type_index, const char* const* resource_names, uint32_t n, int64_t* capacity_milli,
                         int64_t* allocatable_milli, char* name_buf, size_t name_cap) {
  if (!cat || type_index >= cat->types.size()) return KG_EINVAL;
  CtxLock lk(cat->ctx);
  const CatType& t = cat->types[type_index];
  for (uint32_t i = 0; i < n; ++i) {
    int rid = cat->ctx->V.resources.find(resource_names[i]);
    int64_t c = 0, o = 0;
    bool has = false;
    for (auto& kv : t.capacity) if ((int)kv.first == rid) { c = kv.second; has = true; }
    for (auto& kv : t.overhead) if ((int)kv.first == rid) o = kv.second;
    if (capacity_milli) capacity_milli[i] = c;
    if (allocatable_milli) allocatable_milli[i] = has ? c - o : 0;
  }
  if (name_buf && name_cap) {
    size_t m = std::min(name_cap - 1, t.name.size());
    std::memcpy(name_buf, t.name.data(), m);
    name_buf[m] = 0;
  }
  return KG_OK;
}

int32_t kg_catalog_query_requirements(kg_catalog* cat, uint32_t type_index, char* buf, size_t cap, size_t* needed) {
  if (!cat || type_index >= cat->types.size() || (!buf && cap)) return KG_EINVAL;
  CtxLock lk(cat->ctx);
  const Vocab& V = cat->ctx->V;
  std::vector<std::string> lines;
  for (auto& r : cat->types[type_index].reqs.rs) {
    std::vector<std::string> vs;
    for (uint32_t v : r.vals) vs.push_back(V.vals[r.key].strs[v]);
    std::sort(vs.begin(), vs.end());
    std::string l = V.keys.strs[r.key] + "\t" + (vs.empty() ? "DoesNotExist" : "In") + "\t";
    for (size_t i = 0; i < vs.size(); ++i) l += (i ? "," : "") + vs[i];
    lines.push_back(l + "\n");
  }
  std::sort(lines.begin(), lines.end());
  std::string all;
  for (auto& l : lines) all += l;
  if (needed) *needed = all.size() + 1;
  if (cap) {
    const size_t m = std::min(cap - 1, all.size());
    std::memcpy(buf, all.data(), m);
    buf[m] = 0;
  }
  return cap >= all.size() + 1 ? KG_OK : KG_ENOMEM;
}

int32_t kg_catalog_update_offerings(kg_ctx* ctx, kg_catalog* cat, const kg_strtab* strings, const kg_offering_update* updates, uint32_t n_updates,
                                    uint64_t generation) {
  if (!ctx || !cat || cat->ctx != ctx || !strings || (!updates && n_updates)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  update_offerings(ctx, cat, *strings, updates, n_updates);
  cat->generation = generation;
  cat->tab_epoch += 1;
  return KG_OK;
  KG_GUARD_END(ctx)
}

int32_t kg_catalog_free(kg_catalog* cat) {
  if (!cat) return KG_OK;
  {
    CtxLock lk(cat->ctx);
    hal::dfree(cat->tab.dev.ptr);
  }
  delete cat;
  return KG_OK;
}

int32_t kg_cluster_upload(kg_ctx* ctx, const kg_cluster_desc* desc, kg_cluster** out) {
  if (!ctx || !desc || !out) return KG_EINVAL;
  CtxLock lk(ctx);
  std::unique_ptr<kg_cluster> cl(new kg_cluster());
  KG_GUARD_BEGIN
  cl->ctx = ctx;
  kg::HostProf prof_;
  KG_PROF("upload: nodes");
  Vocab& V = ctx->V;
  const kg_pools& p = desc->pods.pools;
  // A descriptor interns its strings: the label keys, label values and resource names of 5 000 nodes are a few hundred
  // distinct string ids.  Each is looked up in the vocabulary once; after that a label costs two array reads.
  const uint32_t NS = p.strings.count;
  std::vector<uint32_t> key_of(NS, UINT32_MAX), res_of(NS, UINT32_MAX);
  std::vector<std::pair<uint32_t, uint32_t>> val_of(NS, {UINT32_MAX, 0});  // (key it was resolved under, value id)
  auto rl = [&](kg_range r) {
    std::vector<std::pair<uint32_t, int64_t>> out;
    out.reserve(r.count);
    for (auto& q : kgdesc::quantities(p, r)) {
      kgdesc::check(q.name < NS, "string id out of range");
      uint32_t& rid = res_of[q.name];
      if (rid == UINT32_MAX) rid = V.resources.id(kgdesc::str(p.strings, q.name));
      bool merged = false;
      for (auto& kv : out) if (kv.first == rid) { kv.second += q.milli; merged = true; }
      if (!merged) out.push_back({rid, q.milli});
    }
    std::sort(out.begin(), out.end());  // (ascending resource id, as parse_rl's map gives)
    return out;
  };
  cl->nodes.reserve(desc->n_nodes);
  for (uint32_t i = 0; i < desc->n_nodes; ++i) {
    const kg_node& n = desc->nodes[i];
    CNode cn;
    cn.name = std::string(kgdesc::str(p.strings, n.name));
    bool has_hostname = false;
    cn.labels.reserve(n.labels.count);
    for (auto& l : kgdesc::labels(p, n.labels)) {
      kgdesc::check(l.key < NS && l.value < NS, "string id out of range");
      uint32_t& k = key_of[l.key];
      if (k == UINT32_MAX) k = V.key(kgdesc::str(p.strings, l.key));
      auto& kv = val_of[l.value];
      if (kv.first != k) kv = {k, V.val(k, kgdesc::str(p.strings, l.value))};
      const uint32_t v = kv.second;
      cn.labels.push_back({k, v});
      if (k == ctx->kHostname) { has_hostname = true; cn.hostname_val = v; }
    }
    if (!has_hostname) cn.hostname_val = V.val(ctx->kHostname, cn.name);
    for (auto& t : kgdesc::taints(p, n.taints))
      cn.taints.push_back({ctx->lab.id(kgdesc::str(p.strings, t.key)), t.value == KG_NONE ? ctx->lab.id("") : ctx->lab.id(kgdesc::str(p.strings, t.value)), t.effect});
    cn.available = rl(n.available);
    cn.capacity = rl(n.capacity);
    cn.daemon = rl(n.daemon_requests);
    cn.initialized = n.initialized; cn.managed = n.managed; cn.deleting = n.deleting;
    cl->nodes.push_back(std::move(cn));
  }
  KG_PROF("upload: bound pods");
  if (ctx->spare_bound) {  // (the pools of the last snapshot freed here: emptied, their pages still mapped)
    ParsedPods& sp = *ctx->spare_bound;
    cl->bound.pods = std::move(sp.pods); cl->bound.rq = std::move(sp.rq); cl->bound.uid_pool = std::move(sp.uid_pool);
    cl->bound.pods.clear(); cl->bound.rq.clear(); cl->bound.uid_pool.clear();
    ctx->spare_bound.reset();
  }
  parse_pods(ctx, &desc->pods, &cl->bound, true);
  KG_PROF("upload: volume limits");
  for (auto& bp : cl->bound.pods) check(bp.node >= 0 && bp.node < (int)cl->nodes.size(), "bound pod node index");
  // volume limits: what the node can still attach per CSI driver = CSINode count - volumes of the pods bound to it.  A node
  // already OVER a driver's limit refuses every pod, with or without volumes: VolumeUsage.ExceedsLimits walks the union of the
  // node's and the pod's volumes over all drivers (a negative availability marks the node infeasible in compile_problem)
  for (uint32_t i = 0; i < desc->n_nodes; ++i) {
    if (desc->nodes[i].volume_limits.count == 0) continue;
    for (auto& q : kgdesc::quantities(p, desc->nodes[i].volume_limits)) {
      const uint32_t rid = V.resources.id(kVolumeResourcePrefix() + std::string(kgdesc::str(p.strings, q.name)));
      int64_t left = q.milli;
      for (auto& bp : cl->bound.pods)
        if (bp.node == (int)i)
          for (uint32_t k = 0; k < bp.rq_n; ++k) if (cl->bound.rq[bp.rq_off + k].first == rid) left -= cl->bound.rq[bp.rq_off + k].second;
      cl->nodes[i].available.push_back({rid, left});  // (negative = over its limit: the node takes no pod at all, see below)
      cl->nodes[i].volume_limited.push_back(rid);
    }
  }
  KG_PROF("upload: end");
  // the bound pods' Queue ranks (first needed by the first consolidation plan over this snapshot, milliseconds at 10^4+ pods)
  // are worked out on a thread of their own from here on: it reads cl->bound only, which nothing changes after this point
  if (cl->bound.pods.size() >= 2 * (size_t)kg::host_chunk_min() && kg::host_threads() > 1) {
    kg_cluster* c = cl.get();
    const uint32_t rc = ctx->rCPU, rm = ctx->rMemory;
    try {
      c->rank_job = std::async(std::launch::async, [c, rc, rm] { kg::bound_queue_rank(c->bound, rc, rm, &c->bound_rank); });
    } catch (const std::system_error&) {}  // no thread to be had: ranked at first use instead
  }
  *out = cl.release();
  return KG_OK;
  KG_GUARD_END(ctx)
}

int32_t kg_cluster_free(kg_cluster* cl) {
  if (cl) {
    CtxLock lk(cl->ctx);  // (the plans' device buffers are freed on the context's GPU)
    kg_cluster_drop_plans(cl);
    try { cl->wait_rank(); } catch (...) {}
    kg_ctx* ctx = cl->ctx;
    if (!ctx->spare_bound || ctx->spare_bound->pods.capacity() < cl->bound.pods.capacity()) {
      ParsedPods* sp = new (std::nothrow) ParsedPods();
      if (sp) {
        sp->pods = std::move(cl->bound.pods); sp->rq = std::move(cl->bound.rq); sp->uid_pool = std::move(cl->bound.uid_pool);
        ctx->spare_bound.reset(sp);
      }
    }
  }
  delete cl;
  return KG_OK;
}

int32_t kg_candidates_order(kg_ctx* ctx, kg_catalog* catalog, kg_cluster* cluster, const kg_candidate_inputs* in, int32_t* out_nodes,
                            double* out_costs, uint32_t cap, uint32_t* n_out) {
  if (!ctx || !cluster || !n_out || (!out_nodes && cap)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  const size_t N = cluster->nodes.size();
  if (in && (in->lifetime_remaining || in->blocked)) check(in->n_nodes == N, "candidate inputs: node count");
  // disruption cost: bound pods in descriptor order (the order fixes the floating-point sum)
  std::vector<double> cost(N, 0.0);
  for (auto& p : cluster->bound.pods) {
    if (p.node < 0 || (size_t)p.node >= N) continue;
    double c = 1.0 + (double)p.deletion_cost / 134217728.0 + (double)p.priority / 33554432.0;
    c = std::max(-10.0, std::min(10.0, c));
    cost[p.node] += c;
  }
  std::unordered_set<std::string> names;
  if (catalog) for (auto& t : catalog->types) names.insert(t.name);
  std::vector<int32_t> order;
  for (size_t i = 0; i < N; ++i) {
    const CNode& nd = cluster->nodes[i];
    if (in && in->lifetime_remaining) cost[i] *= in->lifetime_remaining[i];
    if (!nd.initialized || !nd.managed || nd.deleting) continue;
    if (in && in->blocked && in->blocked[i]) continue;
    bool has_pool = false, has_type = false, type_ok = !catalog;
    for (auto& kv : nd.labels) {
      if (kv.first == ctx->kNodePool) has_pool = true;
      if (kv.first == ctx->kInstanceType) { has_type = true; if (catalog) type_ok = names.count(ctx->V.vals[ctx->kInstanceType].strs[kv.second]) != 0; }
    }
    if (!has_pool || !has_type || !type_ok) continue;
    order.push_back((int32_t)i);
  }
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[a] < cost[b]; });
  *n_out = (uint32_t)order.size();
  if (order.size() > cap) return fail(ctx, KG_ENOMEM, "candidate buffer too small");
  for (size_t k = 0; k < order.size(); ++k) { out_nodes[k] = order[k]; if (out_costs) out_costs[k] = cost[order[k]]; }
  return KG_OK;
  KG_GUARD_END(ctx)
}

int32_t kg_solve(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pods,
                 const kg_nodepools_desc* nodepools, kg_results* out) {
  if (!ctx || !catalogs || n_catalogs == 0 || !pods || !nodepools || !out) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  kg_catalog* view = effective_catalog(ctx, catalogs, n_catalogs, nodepools);
  return solve_impl(ctx, &view, cluster, pods, nodepools, out, true);
  KG_GUARD_END(ctx)
}

// kg_solve without any allocation of device memory: the per-solve tables go into the library's static scratch array
// (device.hip: `scratch`, set up per device at kg_init) instead of pooled hipMalloc buffers.  Same kernels, same results.
int32_t kg_solve_noalloc(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pods,
                         const kg_nodepools_desc* nodepools, kg_results* out) {
  if (!ctx || !catalogs || n_catalogs == 0 || !pods || !nodepools || !out) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  if (n_catalogs != 1) throw Unsupported("kg_solve_noalloc: one catalog per solve (a merged view of several has to be built: an allocation)");
  kg_catalog* view = effective_catalog(ctx, catalogs, n_catalogs, nodepools);
  // every context on this GPU shares the GPU's one array: one user at a time, from carving the tables until the results
  // have been decoded (two contexts of one device, each under its own context lock, would otherwise overwrite each
  // other's problem image and state mid-kernel)
  std::lock_guard<std::mutex> scratch_guard(hal::scratch_mutex());
  struct NoAllocScope { NoAllocScope() { hal::set_noalloc(true); } ~NoAllocScope() { hal::set_noalloc(false); } } noalloc_scope;
  FixedArena fa;
  fa.base = (uint8_t*)hal::scratch_ptr(&fa.cap);
  return solve_impl(ctx, &view, cluster, pods, nodepools, out, true, &fa);
  KG_GUARD_END(ctx)
}

}  // extern "C"

// Scheduler.Solve on the device + result decode; the caller holds the context lock and the exception guard.
static int32_t solve_impl(kg_ctx* ctx, kg_catalog* const* catalogs, kg_cluster* cluster, const kg_pods_desc* pods,
                          const kg_nodepools_desc* nodepools, kg_results* out, bool add_pods_one, FixedArena* fixed) {
  {
  auto t0 = std::chrono::steady_clock::now();
  std::optional<hal::Range> phase;  // (roctx: rocprofv3 --marker-trace shows the host phases next to the launch groups)
  phase.emplace("kg_solve: host compile");
  ParsedPods pending;
  {
    HostProf prof_;
    KG_PROF("parse_pods");
    parse_pods(ctx, pods, &pending, add_pods_one);
    KG_PROF("parse_pods end");
  }
  check(out->cap_pods >= pending.pods.size(), "pod result capacity");
  check_volumes_unshared(pending, cluster);
  SolveInputs in{ctx, catalogs[0], cluster, nodepools, &pending};
  in.fixed = fixed;
  const bool stats = ctx->collect_stats;
  auto C = compile_problem(in, stats);
  phase.reset();
  auto t1 = std::chrono::steady_clock::now();
  float ms_pre = 0, ms_solve = 0;
  hal::launch_prepass(C->dp, C->pre, &ms_pre);
  const uint32_t* nq_dev = (const uint32_t*)((uint8_t*)C->state_dev.ptr + C->o_nqueue);
  const bool use_wave = C->wave_ok && !stats && !ctx->collect_timers && !ctx->no_fast_path;
  bool use_spec = use_wave && C->spec_ok && !ctx->no_spec;
  if (use_spec) { hal::launch_solve_spec(C->dp, (const SimState*)C->sims_dev.ptr, 1, nq_dev, ctx->spec_waves, &ms_solve); ctx->stats.wave_solves += 1; ctx->stats.spec_solves += 1; }
  else if (use_wave) { hal::launch_solve_wave(C->dp, (const SimState*)C->sims_dev.ptr, 1, nq_dev, &ms_solve); ctx->stats.wave_solves += 1; }
  else hal::launch_solve(C->dp, (const SimState*)C->sims_dev.ptr, 1, nq_dev, ctx->block_threads, &ms_solve);
  ctx->stats.kernel_launches += 3;
  // InstanceTypes.OrderByPrice / Truncate per new NodeClaim, behind the solve kernel on the same stream (no host round trip
  // in between: the kernel reads the claim count from the state's header)
  if (C->trunc_claims) { hal::launch_truncate(C->dp, (const SimState*)C->sims_dev.ptr, C->trunc, (int)C->trunc_claims); ctx->stats.kernel_launches += 1; }
  // ---- results
  phase.emplace("kg_solve: decode");
  const int P = C->P, Cmax = C->Cmax, TW = C->TW, R = C->R, Ka = C->Ka;
  std::vector<uint8_t> pod_kind(std::max(P, 1));
  std::vector<int32_t> pod_index(std::max(P, 1));
  std::vector<uint32_t> pod_reason(std::max(P, 1)), header(16);
  std::vector<uint64_t> counters(64 + 16 * 32);
  auto sp = [&](size_t off) { return (uint8_t*)C->state_dev.ptr + off; };
  auto fetch = [&]() {
    // header + counters lie next to each other in the state image, and so do the three per-pod result arrays: two copies
    check(C->o_counters >= C->o_header + 64 && C->o_counters - C->o_header <= 256, "state layout: header / counters");
    std::vector<uint8_t> hc((C->o_counters - C->o_header) + counters.size() * 8);
    hal::d2h(hc.data(), sp(C->o_header), hc.size());
    memcpy(header.data(), hc.data(), 64);
    memcpy(counters.data(), hc.data() + (C->o_counters - C->o_header), counters.size() * 8);
    check(C->o_podindex > C->o_podkind && C->o_podreason > C->o_podindex, "state layout: pod results");
    std::vector<uint8_t> pr((C->o_podreason - C->o_podkind) + (size_t)std::max(P, 1) * 4);
    hal::d2h(pr.data(), sp(C->o_podkind), pr.size());
    memcpy(pod_kind.data(), pr.data(), P);
    memcpy(pod_index.data(), pr.data() + (C->o_podindex - C->o_podkind), (size_t)P * 4);
    memcpy(pod_reason.data(), pr.data() + (C->o_podreason - C->o_podkind), (size_t)P * 4);
  };
  fetch();
  if (use_spec && (header[15] || ctx->spec_fake_fail)) {
    // The speculative kernel's workgroup drained without finishing (a wave gave up waiting for its turn: solve_spec.hpp;
    // never observed -- debug bit 8 of kg_set_collect_stats pretends it happened).  Nothing of its state is kept: the same
    // batch again from a fresh state image on the single-wave kernel, which has no wave to wait for.
    ctx->stats.spec_fallbacks += 1;
    ctx->upload(C->state_dev.ptr, C->state_image.data(), C->state_init_bytes, !fixed);
    hal::dzero((uint8_t*)C->state_dev.ptr + C->state_init_bytes, C->state_bytes - C->state_init_bytes);
    float ms2 = 0;
    hal::launch_solve_wave(C->dp, (const SimState*)C->sims_dev.ptr, 1, nq_dev, &ms2);
    ms_solve += ms2;
    ctx->stats.kernel_launches += 1;
    if (C->trunc_claims) { hal::launch_truncate(C->dp, (const SimState*)C->sims_dev.ptr, C->trunc, (int)C->trunc_claims); ctx->stats.kernel_launches += 1; }
    use_spec = false;
    fetch();
  }
  if (ctx->collect_timers) {
    static const char* names[] = {"pop+ctx", "nodes", "prefilter", "load rows", "merge(+topology)", "type filter", "pick", "commit", "fresh setup", "relax"};
    for (int i = 0; i < 10; ++i) fprintf(stderr, "[kg timers] %-15s %10.3f ms\n", names[i], counters[8 + i] / 1e5);
    static const char* sub[] = {"pf: entry", "pf: loads+test", "pf: ballot+ctx fetch", "pf: barrier", "load: select", "load: loads", "merge: thread 0", "filter: thread 0", "pick: body", "commit: body"};
    for (int i = 0; i < 10; ++i) fprintf(stderr, "[kg timers]   sub %-20s %10.3f ms\n", sub[i], counters[32 + i] / 1e5);
    static const char* fsub[] = {"fast: select", "fast: row loads", "fast: entries", "fast: resources", "fast: type filter", "fast: write / refresh", "fast: barrier wait"};
    for (int i = 0; i < 7; ++i) fprintf(stderr, "[kg timers]   sub %-20s %10.3f ms\n", fsub[i], counters[42 + i] / 1e5);
    fprintf(stderr, "[kg timers] shader clock %.0f MHz over the launch\n", counters[25] ? counters[24] / (counters[25] / 100.0) : 0.0);
    fprintf(stderr, "[kg timers] probes (ns each): barrier %.0f, dependent global load %.0f, LDS store+2 barriers+LDS load %.0f, global store+barrier %.0f\n",
            counters[26] * 10.0 / 1000, counters[27] * 10.0 / 1000, counters[28] * 10.0 / 1000, counters[29] * 10.0 / 1000);
    fprintf(stderr, "[kg timers] probes (ns each): chain over 256 KiB first walk %.0f, second walk %.0f, store->load same line %.0f\n",
            counters[60] * 10.0 / 2048, counters[61] * 10.0 / 2048, counters[62] * 10.0 / 1000);
    fprintf(stderr, "[kg timers] probes (ns each): round trip of 24 independent loads per lane: one wave %.0f, all waves %.0f, all waves scattered lanes %.0f\n",
            counters[56] * 10.0 / 200, counters[57] * 10.0 / 200, counters[58] * 10.0 / 200);
    fprintf(stderr, "[kg timers] survivors %llu rounds %llu false_pos %llu passes %llu\n", (unsigned long long)counters[20],
            (unsigned long long)counters[21], (unsigned long long)counters[22], (unsigned long long)counters[23]);
  }
  if (ctx->wave_timers && use_wave && !use_spec) {
    static const char* sec[] = {"pop + pod context", "prefilter", "row load + group end", "NodeClaim.Add", "commit", "refresh after a false survivor", "new NodeClaim", "loop back"};
    for (int k = 0; k < 8; ++k) fprintf(stderr, "[kg wave] %-32s %9.0f kcycles\n", sec[k], counters[64 + k] / 1e3);
    fprintf(stderr, "[kg wave] prefilter steps %llu, survivors %llu, exact evaluations %llu\n", (unsigned long long)counters[72], (unsigned long long)counters[73],
            (unsigned long long)counters[74]);
    static const char* nsec[] = {"limits + template row", "NodeClaim.Add (fresh)", "row stores", "headroom / subtractMax", "scan order"};
    for (int k = 0; k < 5; ++k) fprintf(stderr, "[kg wave]   new NodeClaim: %-24s %9.0f kcycles\n", nsec[k], counters[80 + k] / 1e3);
  } else if (ctx->wave_timers && !use_spec) {
    static const char* ph[] = {"prefilter", "wave-local Add", "commit", "fresh commit", "fresh setup", "workgroup Add", "nodes", "relax / ctx"};
    for (int k = 0; k < 8; ++k) {
      fprintf(stderr, "[kg waves] %-15s work / wait kcycles:", ph[k]);
      for (int w = 0; w < ctx->block_threads / 64; ++w) fprintf(stderr, "  %8.0f / %-8.0f", counters[64 + w * 32 + 2 * k] / 1e3, counters[64 + w * 32 + 2 * k + 1] / 1e3);
      fprintf(stderr, "\n");
    }
    for (int k = 16; k < 32; ++k) {
      fprintf(stderr, "[kg waves] slot %2d kcycles:", k);
      for (int w = 0; w < ctx->block_threads / 64; ++w) fprintf(stderr, "  %8.0f", counters[64 + w * 32 + k] / 1e3);
      fprintf(stderr, "\n");
    }
  }
  if (use_spec) {
    ctx->stats.spec_fast_commits += counters[48]; ctx->stats.spec_challengers += counters[49]; ctx->stats.spec_rescans += counters[50];
    if (getenv("KG_SPEC_DEBUG") || ctx->wave_timers)
      fprintf(stderr, "[kg spec] commits from speculation %llu, challengers %llu, rescans %llu, new NodeClaims %llu, failures %llu, exact evaluations %llu, "
              "new NodeClaims / failures evaluated again at the turn %llu; LDS: %llu prefilter records, pool %llu of %llu words before them\n",
              (unsigned long long)counters[48], (unsigned long long)counters[49], (unsigned long long)counters[50], (unsigned long long)counters[51],
              (unsigned long long)counters[52], (unsigned long long)counters[53], (unsigned long long)counters[54], (unsigned long long)counters[55],
              (unsigned long long)counters[56], (unsigned long long)counters[57]);
    if (ctx->wave_timers) {
      static const char* nm[] = {"wait: queue entry", "speculative evaluation", "wait: turn", "token: commit from speculation", "token: after a challenger",
                                 "token: after a rescan", "token: new NodeClaim", "token: failure", "after the token", "tickets",
                                 "fast: control words", "fast: window check", "fast: commit to LDS", "fast: release", "rescan: decide", "rescan: commit"};
      const int nw = ctx->spec_waves >= 8 ? 8 : 4;
      for (int k = 0; k < 16; ++k) {
        fprintf(stderr, "[kg spec] %-30s", nm[k]);
        unsigned long long tot = 0;
        for (int w = 0; w < nw; ++w) { fprintf(stderr, " %9.0f", counters[64 + 16 * w + k] / (k == 9 ? 1.0 : 1e3)); tot += counters[64 + 16 * w + k]; }
        fprintf(stderr, "   sum %10.0f %s\n", tot / (k == 9 ? 1.0 : 1e3), k == 9 ? "" : "kcycles");
      }
      static const char* rn[] = {"turn: first candidate outside", "turn: window claims", "turn: after (not first choice)", "spec: survivors snapshot", "gap: release -> next token",
                                 "(count) not first choice", "(count) late for the turn", "(count) scan behind candidates"};
      for (int k = 0; k < 8; ++k) {
        fprintf(stderr, "[kg spec] %-30s", rn[k]);
        unsigned long long tot = 0;
        for (int w = 0; w < nw; ++w) { fprintf(stderr, " %9.0f", counters[64 + 16 * 8 + 8 * w + k] / 1e3); tot += counters[64 + 16 * 8 + 8 * w + k]; }
        fprintf(stderr, "   sum %10.0f\n", tot / 1e3);
      }
    }
  }
  const uint32_t nclaims = header[0];
  std::vector<uint32_t> order(nclaims), c_npods(nclaims), c_meta((size_t)Ka * nclaims);
  std::vector<uint16_t> c_tmpl(nclaims);
  std::vector<uint64_t> c_mask((size_t)TW * nclaims), c_bits((size_t)C->Wa * nclaims);
  std::vector<int64_t> c_req((size_t)R * nclaims);
  // the NodeClaims' records as k_price_order_truncate packed them (one copy), or the state's columns (one copy per table)
  const bool dev_trunc = C->trunc_claims != 0 && nclaims <= C->trunc_claims;
  std::vector<uint32_t> trec, trunc_n(nclaims, 0);
  if (nclaims && dev_trunc) {
    const uint32_t RW = C->trunc_words, MT = C->trunc_max;
    trec.resize((size_t)nclaims * RW);
    hal::d2h(trec.data(), sp(C->o_trec), trec.size() * 4);
    std::vector<uint32_t> c_stamp(nclaims);
    for (uint32_t c = 0; c < nclaims; ++c) {
      const uint32_t* rec = trec.data() + (size_t)c * RW;
      c_npods[c] = rec[0]; c_tmpl[c] = (uint16_t)rec[1]; c_stamp[c] = rec[2]; trunc_n[c] = rec[3];
      const uint32_t* q = rec + 4 + MT;
      for (int a = 0; a < Ka; ++a) c_meta[(size_t)a * nclaims + c] = q[a];
      q += Ka;
      for (int w = 0; w < C->Wa; ++w) c_bits[(size_t)w * nclaims + c] = (uint64_t)q[2 * w] | ((uint64_t)q[2 * w + 1] << 32);
      q += 2 * C->Wa;
      for (int r = 0; r < R; ++r) c_req[(size_t)r * nclaims + c] = (int64_t)((uint64_t)q[2 * r] | ((uint64_t)q[2 * r + 1] << 32));
    }
    for (uint32_t c = 0; c < nclaims; ++c) order[c] = c;
    auto okey = [&](uint32_t c) -> unsigned long long {
      if (c == header[2]) return header[5] ? ~0ull : claim_key(header[3], header[4], c);
      return claim_key(c_npods[c], c_stamp[c], c);
    };
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return okey(a) < okey(b); });
  } else if (nclaims) {
    std::vector<uint32_t> c_stamp(nclaims);
    hal::d2h(c_stamp.data(), sp(C->o_order), (size_t)nclaims * 4);
    hal::d2h(c_npods.data(), sp(C->o_cnpods), (size_t)nclaims * 4);
    // Results.NewNodeClaims comes back in the slice's final (sorted) order
    for (uint32_t c = 0; c < nclaims; ++c) order[c] = c;
    // ... as of the last sort.Slice: the claim touched after it keeps its old place (or stays appended at the end)
    auto okey = [&](uint32_t c) -> unsigned long long {
      if (c == header[2]) return header[5] ? ~0ull : claim_key(header[3], header[4], c);
      return claim_key(c_npods[c], c_stamp[c], c);
    };
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return okey(a) < okey(b); });
    hal::d2h(c_tmpl.data(), sp(C->o_ctmpl), (size_t)nclaims * 2);
    // the first nclaims entries of every [row][Cmax] column: one strided copy per table (not one blocking copy per row)
    hal::d2h_rows(c_mask.data(), sp(C->o_cmask), (size_t)nclaims * 8, (size_t)Cmax * 8, (size_t)TW);
    hal::d2h_rows(c_bits.data(), sp(C->o_cbits), (size_t)nclaims * 8, (size_t)Cmax * 8, (size_t)C->Wa);
    hal::d2h_rows(c_meta.data(), sp(C->o_cmeta), (size_t)nclaims * 4, (size_t)Cmax * 4, (size_t)Ka);
    hal::d2h_rows(c_req.data(), sp(C->o_creq), (size_t)nclaims * 8, (size_t)Cmax * 8, (size_t)R);
  }
  Vocab& V = ctx->V;
  kgdesc::ResultWriter W(out);
  std::vector<int32_t> out_index(nclaims, -1);
  std::vector<uint8_t> trunc_failed(nclaims, 0);
  const int zslot = find_slot(*C, ctx->kZone), ctslot = find_slot(*C, ctx->kCT), zidslot = find_slot(*C, ctx->kZid);
  const uint8_t* zone_zid = catalogs[0]->tab.image.bytes.data() + catalogs[0]->tab.o_zone_zid;
  for (uint32_t pos = 0; pos < nclaims; ++pos) {
    const uint32_t c = order[pos];
    const Template& tp = C->templates[c_tmpl[c]];
    ClaimView cv;
    if (dev_trunc) {
      const uint32_t* ids = trec.data() + (size_t)c * C->trunc_words + 4;
      cv.types.assign(ids, ids + std::min(trunc_n[c], C->trunc_max));
    } else {
      for (int w = 0; w < TW; ++w)
        for (uint64_t m = c_mask[(size_t)w * nclaims + c]; m; m &= m - 1) cv.types.push_back((uint32_t)(w * 64 + __builtin_ctzll(m)));
    }
    // decode requirements
    std::vector<HReq> creqs;
    for (int a = 0; a < Ka; ++a) {
      const uint32_t meta = c_meta[(size_t)a * nclaims + c];
      if (!(meta & RM_PRESENT)) continue;
      HReq r;
      r.key = C->slot_key[a];
      r.compl_ = meta & RM_COMPL;
      for (int i = 0; i < C->aw[a]; ++i)
        for (uint64_t m = c_bits[(size_t)(C->aoff[a] + i) * nclaims + c]; m; m &= m - 1) r.vals.push_back(C->lv[a].global((uint32_t)(i * 64 + __builtin_ctzll(m))));
      std::sort(r.vals.begin(), r.vals.end());  // (HReq keeps context ids sorted)
      // bounds: indices refer to the per-slot sorted tables the compile built; recover the values from the device tables' host copy
      creqs.push_back(std::move(r));
      // (gt/lt filled below)
      const uint32_t gi = rm_gt(meta), li = rm_lt(meta);
      if (gi || li) {
        // rebuild the sorted distinct bound lists exactly as compile_problem did
        std::vector<int64_t> gts, lts;
        for (auto& rs : C->reqsets) for (auto& q : rs.rs) if (q.key == C->slot_key[a]) { if (q.has_gt) gts.push_back(q.gt); if (q.has_lt) lts.push_back(q.lt); }
        auto uniq = [](std::vector<int64_t>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
        uniq(gts); uniq(lts);
        if (gi) { creqs.back().has_gt = true; creqs.back().gt = gts[gi - 1]; }
        if (li) { creqs.back().has_lt = true; creqs.back().lt = lts[li - 1]; }
      }
    }
    auto has_mask = [&](int slot) -> uint64_t {  // allowed values of the claim on the key, over the catalog's value numbering
      if (slot < 0) return ~0ull;
      for (auto& r : creqs) {
        if (r.key != C->slot_key[slot]) continue;
        const auto& cv = catalogs[0]->tab.cvals[r.key];
        uint64_t m = 0;
        for (uint32_t l = 0; l < cv.size() && l < 64; ++l) if (r.has(V, cv[l])) m |= 1ull << l;
        return m;
      }
      return ~0ull;
    };
    if (!dev_trunc) {  // (the device kernel derived the same masks from the claim's requirement words)
      cv.zmask = has_mask(zslot);
      cv.ctmask = has_mask(ctslot);
      const uint64_t zidm = has_mask(zidslot);
      if (zidslot >= 0)
        for (int z = 0; z < 64; ++z) if (zone_zid[z] != 0xFF && !((zidm >> zone_zid[z]) & 1)) cv.zmask &= ~(1ull << z);
      for (auto& q : tp.reqs.rs) if (q.min_values >= 0) cv.minvals.push_back({q.key, q.min_values});
    }
    if (!dev_trunc && nodepools->max_instance_types > 0) trunc_failed[c] = !truncate_types(catalogs[0], cv, nodepools->max_instance_types);
    kg_claim_out co{};
    co.nodepool = tp.index;
    co.n_pods = c_npods[c];
    co.truncation_failed = trunc_failed[c];
    co.types.begin = out->needed_type_ids;
    co.types.count = (uint32_t)cv.types.size();
    const uint32_t type_base = catalogs[0]->part_begin.empty() ? 0 : catalogs[0]->part_begin[tp.part];  // (merged view -> ids of the NodePool's own catalog)
    for (uint32_t t : cv.types) W.add_type(t - type_base);
    // requirements, keys and values in lexicographic order
    std::sort(creqs.begin(), creqs.end(), [&](const HReq& a, const HReq& b) { return V.keys.strs[a.key] < V.keys.strs[b.key]; });
    std::vector<kg_req_out> tmp;
    for (auto& r : creqs) {
      if (r.key == ctx->kPart) continue;  // (the merged view's own bookkeeping, not a label)
      kg_req_out q{};
      q.key = W.add_string(V.keys.strs[r.key]);
      q.complement = r.compl_;
      q.has_gt = r.has_gt; q.has_lt = r.has_lt; q.gt = r.gt; q.lt = r.lt;
      q.min_values = -1;  // (pods carry none: a claim's minValues are its NodePool's)
      for (auto& t : tp.reqs.rs) if (t.key == r.key) q.min_values = t.min_values;
      std::vector<std::string> vs;
      for (uint32_t v : r.vals) vs.push_back(V.vals[r.key].strs[v]);
      std::sort(vs.begin(), vs.end());
      q.values.begin = out->needed_value_ids;
      q.values.count = (uint32_t)vs.size();
      for (auto& s : vs) W.add_value(W.add_string(s));
      tmp.push_back(q);
    }
    co.requirements.begin = out->needed_reqs;
    co.requirements.count = (uint32_t)tmp.size();
    for (auto& q : tmp) W.add_req(q);
    std::vector<std::pair<std::string, int64_t>> rq;
    for (int r = 0; r < R; ++r)
      if (c_req[(size_t)r * nclaims + c] != 0 && !is_volume_resource(V.resources.strs[C->act_res[r]])) rq.push_back({V.resources.strs[C->act_res[r]], c_req[(size_t)r * nclaims + c]});
    std::sort(rq.begin(), rq.end());
    co.requests.begin = out->needed_quantities;
    co.requests.count = (uint32_t)rq.size();
    for (auto& kv : rq) W.add_quantity({W.add_string(kv.first), kv.second});
    out_index[c] = (int32_t)W.add_claim(co);
  }
  for (int p = 0; p < P; ++p) {
    out->pod_kind[p] = pod_kind[p];
    out->pod_reason[p] = pod_kind[p] == KG_POD_UNSCHEDULABLE ? pod_reason[p] : 0;
    if (pod_kind[p] == KG_POD_EXISTING_NODE) out->pod_index[p] = (int32_t)C->node_order[pod_index[p]];
    else if (pod_kind[p] == KG_POD_NEW_CLAIM) {
      const int32_t c = pod_index[p];
      if (trunc_failed[c]) { out->pod_kind[p] = KG_POD_UNSCHEDULABLE; out->pod_index[p] = -1; out->pod_reason[p] = KG_REASON_MIN_VALUES; }
      else out->pod_index[p] = out_index[c];
    } else out->pod_index[p] = -1;
  }
  out->visited_existing = counters[0];
  out->visited_claims = counters[1];
  out->visited_claim_types = counters[2];
  out->new_claim_scans = counters[3];
  out->relaxations = header[1];
  out->algorithmic_bytes = (double)KG_ROW_BYTES_EXISTING_NODE * counters[0] + (double)KG_ROW_BYTES_NODE_CLAIM * (counters[1] + counters[3]) +
                           (double)KG_ROW_BYTES_INSTANCE_TYPE * counters[2];
  out->device_ms = ms_pre + ms_solve;
  out->prepass_ms = ms_pre;
  out->solve_ms = ms_solve;
  auto t2 = std::chrono::steady_clock::now();
  out->host_compile_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  out->host_decode_ms = std::chrono::duration<double, std::milli>(t2 - t1).count() - ms_pre - ms_solve;
  ctx->stats.solves += 1;
  ctx->stats.pods += P;
  ctx->stats.device_ms_total += ms_pre + ms_solve;
  ctx->stats.host_compile_ms_total += std::chrono::duration<double, std::milli>(t1 - t0).count();
  return W.overflow ? fail(ctx, KG_ENOMEM, "result buffers too small (see needed_*)") : KG_OK;
  }
}

extern "C" {

}  // extern "C"

// ------------------------------------------------------------------ consolidation (disruption simulation batches)
namespace {

struct CandInfo {
  bool ok = false;       // price known (instance type resolvable, a compatible offering exists)
  double price = 0;      // cheapest offering compatible with the node's labels (getCandidatePrices)
  int type = -1;
  bool spot = false;
};

// the replacement NodeClaim of one simulation after computeConsolidation's filters
struct Replacement {
  int decision = KG_DECISION_NOOP;
  uint32_t nodepool = 0;
  std::vector<uint32_t> types;  // price ordered
  uint64_t zmask = ~0ull, ctmask = ~0ull;
  double candidate_price = 0;
  bool spot_only = false;
};

uint64_t allowed1(uint32_t meta, uint64_t word) {
  if (!(meta & RM_PRESENT)) return ~0ull;
  return (meta & RM_COMPL) ? ~word : word;
}

void order_by_price(const kg_catalog* cat, std::vector<uint32_t>* types, uint64_t zmask, uint64_t ctmask) {
  std::vector<std::pair<double, uint32_t>> keyed;
  for (uint32_t t : *types) keyed.push_back({cheapest_price(cat->types[t], zmask, ctmask), t});
  std::stable_sort(keyed.begin(), keyed.end(), [&](auto& a, auto& b) {
    if (a.first == b.first) return cat->types[a.second].name < cat->types[b.second].name;
    return a.first < b.first;
  });
  types->clear();
  for (auto& kv : keyed) types->push_back(kv.second);
}

// Offerings.Available().WorstLaunchPrice(reqs)
double worst_launch_price(const CatType& t, uint64_t zmask, uint64_t ctmask, int spot, int od) {
  for (int want : {spot, od}) {
    if (want < 0 || !((ctmask >> want) & 1)) continue;
    bool any = false;
    double worst = 0;
    for (auto& o : t.offerings) {
      if (!o.available || (int)o.lct != want || !((zmask >> o.lz) & 1)) continue;
      if (!any || o.price > worst) { worst = o.price; any = true; }
    }
    if (any) return worst;
  }
  return std::numeric_limits<double>::max();
}

void remove_by_price(const kg_catalog* cat, Replacement* r, double max_price, int spot, int od) {
  std::vector<uint32_t> kept;
  for (uint32_t t : r->types) if (worst_launch_price(cat->types[t], r->zmask, r->ctmask, spot, od) < max_price) kept.push_back(t);
  r->types.swap(kept);
}

}  // namespace

// what a validation re-simulation reports (kg_consolidate_validate)
struct ValidateOut {
  bool all_scheduled = false;
  uint32_t n_claims = 0;          // new NodeClaims left after Results.TruncateInstanceTypes
  std::vector<uint32_t> types;    // options of the one NodeClaim, truncated
};

// ------------------------------------------------------------------ consolidation plans
// Everything of a consolidation round that does not depend on the shard or on the call: the pod universe, the compiled
// problem (resident on the device), the queue / excluded-node / limit / topology patches of EVERY unit (resident too), the
// candidates' prices.  Built once per (cluster snapshot, catalog generation, NodePools, pending pods, candidate order, mode)
// and kept with the kg_cluster; a round that finds its plan only launches the three kernels over its shard's units and runs
// the price logic on what comes back.  (Round 1 rebuilt all of it in every call on every rank: 6 of 8 ms on C4.)
namespace {
struct Hasher {
  // two independent 64-bit mixes of the same bytes and their count: a plan is reused only when all three agree (and its
  // candidate list compares equal) -- no single hash decides alone
  uint64_t h = 1469598103934665603ull, h2 = 0x9E3779B97F4A7C15ull, total = 0;
  void bytes(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    total += n;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      std::memcpy(&w, b + i, 8);
      h = (h ^ w) * 1099511628211ull; h ^= h >> 29;
      h2 = (h2 + w) * 0xD6E8FEB86659FD93ull; h2 ^= h2 >> 32;
    }
    for (; i < n; ++i) { h = (h ^ b[i]) * 1099511628211ull; h2 = (h2 + b[i]) * 0xD6E8FEB86659FD93ull; h2 ^= h2 >> 32; }
  }
  void u64(uint64_t v) { bytes(&v, 8); }
  template <class T> void arr(const T* p, size_t n) { u64(n); if (n && p) bytes(p, n * sizeof(T)); }
  void pools(const kg_pools& p) {
    u64(p.strings.count);
    if (p.strings.count) { arr(p.strings.offsets, p.strings.count + 1); arr(p.strings.data, p.strings.offsets[p.strings.count]); }
    arr(p.reqs, p.n_reqs); arr(p.req_values, p.n_req_values); arr(p.quantities, p.n_quantities); arr(p.labels, p.n_labels);
    arr(p.taints, p.n_taints); arr(p.tolerations, p.n_tolerations); arr(p.ranges, p.n_ranges);
  }
};
}  // namespace

struct ConsUnit { uint32_t id, pod_lo, pod_hi; bool skip; };
struct kg_cons_plan {
  uint64_t key = 0, key2 = 0, key_bytes = 0;  // Hasher::h / h2 / total of what the plan was built from
  bool multi = false;
  uint32_t n = 0, n_used = 0, n_shared = 0;
  std::vector<int32_t> candidates;
  std::vector<uint32_t> cand_begin;
  std::vector<ConsUnit> units;  // every unit of the mode (single: candidate u; multi: prefix [0..u]), index = u - lo
  std::unique_ptr<Compiled> C;
  bool any_limits = false;
  DeviceBuffer tables;          // per-unit patches of every unit + the universe's flags
  size_t o_qoff = 0, o_qent = 0, o_xoff = 0, o_xnode = 0, o_rem = 0, o_np = 0, o_un = 0, o_toff = 0, o_tent = 0, o_guniv = 0;
  std::vector<CandInfo> cinfo;  // getCandidatePrices inputs per candidate
  int spot = -1, od = -1;
  float ms_prepass = 0;
  // the shard last run: its live units, their state slots and launch tables (kept: the next round of the same shard reuses them)
  int shard = -1, n_shards = 0;
  std::vector<uint32_t> live;   // unit indices
  DeviceBuffer arena, run;
  size_t max_batch = 0, o_sims = 0, o_unit = 0, o_nq = 0, o_res = 0;
  uint64_t last_use = 0;
  SimResult* results_host = nullptr;  // pinned: the simulations' outcomes come back in one DMA
  // device buffers and the pinned one come from the context's pools and go back there (a controller that uploads a snapshot per
  // poll builds a plan per poll: hipMalloc / hipFree / hipHostMalloc of these cost more than the round's kernels)
  kg_ctx* ctx = nullptr;
  size_t tables_cap = 0, arena_cap = 0, run_cap = 0, results_cap = 0;
  void drop_shard() {
    ctx->pool_give(arena.ptr, arena_cap); ctx->pool_give(run.ptr, run_cap); ctx->pinned_give(results_host, results_cap);
    arena = DeviceBuffer(); run = DeviceBuffer(); results_host = nullptr;
  }
  ~kg_cons_plan() {
    if (!ctx) return;
    ctx->pool_give(tables.ptr, tables_cap);
    drop_shard();
  }
};
void kg_cluster_drop_plans(kg_cluster* cl) { for (auto* p : cl->plans) delete p; cl->plans.clear(); }

static std::unique_ptr<kg_cons_plan> build_cons_plan(kg_ctx* ctx, kg_catalog* cat, kg_cluster* cluster, const kg_pods_desc* pending,
                                                     const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, bool multi, bool validate) {
  HostProf prof_;
  Vocab& V = ctx->V;
  auto Pp = std::make_unique<kg_cons_plan>();
  kg_cons_plan& PL = *Pp;
  PL.ctx = ctx;
  PL.multi = multi || validate;  // (a validated command is a prefix: every candidate of it is taken out of the cluster)
  const uint32_t n = cands->n_candidates;
  PL.n = n;
  KG_PROF("plan: pod universe");
  // ---- pod universe: pending pods, pods of deleting nodes, then the candidates' reschedulable pods in candidate order
  ParsedPods pend;
  parse_pods(ctx, pending, &pend, true);
  check_volumes_unshared(pend, cluster);
  ParsedPods U;
  U.classes = pend.classes;
  const uint32_t cls_off = (uint32_t)U.classes.size();
  U.classes.insert(U.classes.end(), cluster->bound.classes.begin(), cluster->bound.classes.end());
  std::vector<uint8_t> nonpending;
  std::vector<int32_t> universe_bound;
  // the bound pods keep their request pairs and UIDs where the cluster snapshot holds them: both pools are appended whole
  // behind the pending pods' and a bound pod's offsets move by the pending pools' sizes (no copy per pod)
  const uint32_t rq_shift = (uint32_t)pend.rq.size(), uid_shift = (uint32_t)pend.uid_pool.size();
  if (pend.pods.empty()) U.pools_of = &cluster->bound;  // (nothing in front of them: borrowed, not copied)
  else {
    U.rq.reserve(pend.rq.size() + cluster->bound.rq.size());
    U.rq = pend.rq;
    U.rq.insert(U.rq.end(), cluster->bound.rq.begin(), cluster->bound.rq.end());
    U.uid_pool.reserve(pend.uid_pool.size() + cluster->bound.uid_pool.size());
    U.uid_pool = pend.uid_pool;
    U.uid_pool += cluster->bound.uid_pool;
  }
  const size_t n_bound = cluster->bound.pods.size();
  U.pods.reserve(pend.pods.size() + n_bound);
  nonpending.reserve(pend.pods.size() + n_bound);
  universe_bound.reserve(pend.pods.size() + n_bound);
  for (auto& p : pend.pods) { U.pods.push_back(p); nonpending.push_back(0); universe_bound.push_back(-1); }
  // reschedulable bound pods by node (counting sort: descriptor order within a node)
  std::vector<uint32_t> node_begin(cluster->nodes.size() + 1, 0), node_pods;
  for (size_t i = 0; i < n_bound; ++i) if (cluster->bound.pods[i].reschedulable) node_begin[cluster->bound.pods[i].node + 1]++;
  for (size_t nd = 0; nd < cluster->nodes.size(); ++nd) node_begin[nd + 1] += node_begin[nd];
  node_pods.resize(node_begin[cluster->nodes.size()]);
  {
    std::vector<uint32_t> fill(node_begin.begin(), node_begin.end() - 1);
    for (size_t i = 0; i < n_bound; ++i) if (cluster->bound.pods[i].reschedulable) node_pods[fill[cluster->bound.pods[i].node]++] = (uint32_t)i;
  }
  auto add_node = [&](size_t nd) {
    for (uint32_t k = node_begin[nd]; k < node_begin[nd + 1]; ++k) {
      const uint32_t bi = node_pods[k];
      PodRef pr = cluster->bound.pods[bi];
      pr.cls += cls_off;
      pr.rq_off += rq_shift;
      pr.uid_off += uid_shift;
      U.pods.push_back(pr);
      nonpending.push_back(1);
      universe_bound.push_back((int32_t)bi);
    }
  };
  std::vector<uint8_t> is_cand(cluster->nodes.size(), 0);
  // multi: firstNConsolidationOption gets max = Clamp(len, 0, 100) and only lowers it to len - 1 when len <= max, so with more than
  // 100 candidates the binary search probes mid in [1, 100] and the longest prefix is candidates[0:101]
  const uint32_t n_used = (multi && !validate) ? std::min<uint32_t>(n, 101) : n;
  PL.n_used = n_used;
  PL.candidates.assign(cands->candidates, cands->candidates + n_used);
  for (uint32_t i = 0; i < n_used; ++i) is_cand[cands->candidates[i]] = 1;
  for (size_t nd = 0; nd < cluster->nodes.size(); ++nd)
    if (cluster->nodes[nd].deleting && !is_cand[nd]) add_node(nd);
  const uint32_t n_shared = (uint32_t)U.pods.size();
  PL.n_shared = n_shared;
  std::vector<uint32_t>& cand_begin = PL.cand_begin;
  cand_begin.assign(n_used + 1, n_shared);
  for (uint32_t i = 0; i < n_used; ++i) {
    cand_begin[i] = (uint32_t)U.pods.size();
    add_node((size_t)cands->candidates[i]);
  }
  cand_begin[n_used] = (uint32_t)U.pods.size();

  KG_PROF("plan: units");
  const uint32_t lo = multi ? 1 : 0, hi = multi ? n_used : n;
  if (validate) { if (n_used > 0) PL.units.push_back(ConsUnit{n_used - 1, n_shared, cand_begin[n_used], false}); }  // one simulation: every candidate of the command
  else for (uint32_t u = lo; u < hi; ++u) {
    ConsUnit un{u, multi ? n_shared : cand_begin[u], cand_begin[u + 1], false};
    if (!multi && un.pod_hi == un.pod_lo) un.skip = true;  // no reschedulable pods: emptiness handles the node
    PL.units.push_back(un);
  }
  int cmax = 1;
  bool any_live = false;
  for (auto& un : PL.units) if (!un.skip) { any_live = true; cmax = std::max(cmax, (int)(n_shared + un.pod_hi - un.pod_lo)); }
  if (!any_live) return Pp;

  KG_PROF("plan: compile_problem");
  // ---- compile once; every simulation shares the problem tables
  SolveInputs in{ctx, cat, cluster, nodepools, &U};
  in.sim_cmax = cmax;
  in.universe_bound = &universe_bound;
  in.n_shared = n_shared;
  // (with the visit counters on, every simulation scans exactly and reports the rows the reference visits: bench.py's
  // algorithmic bytes for the consolidation roofline; production callers switch them off)
  PL.C = compile_problem(in, ctx->collect_stats && !validate);
  Compiled* C = PL.C.get();
  const int N = C->N, NW = std::max((N + 63) / 64, 1), R = C->R, NP = C->NP;
  std::vector<int> pos_of_node(cluster->nodes.size(), -1);
  for (int i = 0; i < N; ++i) pos_of_node[C->node_order[i]] = i;
  std::vector<uint32_t> rank(std::max<size_t>(U.pods.size(), 1), 0);
  for (size_t i = 0; i < C->queue_order.size() && i < U.pods.size(); ++i) rank[C->queue_order[i]] = (uint32_t)i;
  std::vector<uint64_t> node_uninit(NW, 0);
  for (int i = 0; i < N; ++i) if (!cluster->nodes[C->node_order[i]].initialized) node_uninit[i >> 6] |= 1ull << (i & 63);
  bool any_limits = false;
  for (auto& tp : C->templates) any_limits |= tp.has_limits && !tp.limits.empty();
  PL.any_limits = any_limits;
  std::vector<int> ract(V.resources.size(), -1);
  for (int r = 0; r < R; ++r) ract[C->act_res[r]] = r;
  hal::launch_prepass(C->dp, C->pre, &PL.ms_prepass);

  KG_PROF("plan: per-unit patches");
  // ---- the patches of every unit: its queue (the universe's Queue order restricted to its pods), the candidate nodes it
  // leaves out, the limits they give back, the topology counts of the pods it reschedules
  const size_t nu = PL.units.size();
  std::vector<uint32_t> q_off(nu + 1, 0), x_off(nu + 1, 0), x_node, t_off(nu + 1, 0), t_ent, q_ent;
  std::vector<int64_t> rem_adj(any_limits ? nu * (size_t)R * NP : 0, 0);
  std::vector<uint32_t> shared_sorted;
  for (uint32_t p = 0; p < n_shared; ++p) shared_sorted.push_back(p);
  std::sort(shared_sorted.begin(), shared_sorted.end(), [&](uint32_t a, uint32_t b) { return rank[a] < rank[b]; });
  // (units in contiguous chunks on the host's threads, each into its own vectors: offsets are chunk-relative until the chunks'
  // sizes are known, then every chunk copies its part into place)
  size_t patch_work = 0;
  for (auto& un : PL.units) if (!un.skip) patch_work += n_shared + un.pod_hi - un.pod_lo;
  const unsigned TC = (unsigned)std::min<size_t>(host_chunks(patch_work), std::max<size_t>(nu, 1));
  struct PatchChunk { size_t k0 = 0, k1 = 0, q_base = 0, x_base = 0, t_base = 0; std::vector<uint32_t> q_ent, x_node, t_ent; };
  std::vector<PatchChunk> pch(TC);
  for (unsigned t = 0; t < TC; ++t) { pch[t].k0 = nu * t / TC; pch[t].k1 = nu * (t + 1) / TC; }
  parallel_chunks(TC, [&](unsigned t) {
    PatchChunk& ch = pch[t];
    std::vector<uint32_t> own, pods;
    for (size_t k = ch.k0; k < ch.k1; ++k) {
      const ConsUnit& un = PL.units[k];
      q_off[k] = (uint32_t)(ch.q_ent.size() / 4);
      x_off[k] = (uint32_t)ch.x_node.size();
      t_off[k] = (uint32_t)ch.t_ent.size();
      if (un.skip) continue;
      own.clear();
      for (uint32_t p = un.pod_lo; p < un.pod_hi; ++p) own.push_back(p);
      std::sort(own.begin(), own.end(), [&](uint32_t a, uint32_t b) { return rank[a] < rank[b]; });
      pods.resize(shared_sorted.size() + own.size());
      std::merge(shared_sorted.begin(), shared_sorted.end(), own.begin(), own.end(), pods.begin(), [&](uint32_t a, uint32_t b) { return rank[a] < rank[b]; });
      for (uint32_t p : pods) { ch.q_ent.push_back(p); ch.q_ent.push_back(C->pod_variant0[p]); ch.q_ent.push_back(0); ch.q_ent.push_back(0); }
      if (!C->contrib.empty())
        for (uint32_t p = un.pod_lo; p < un.pod_hi; ++p) ch.t_ent.insert(ch.t_ent.end(), C->contrib[p].begin(), C->contrib[p].end());
      const uint32_t c0 = PL.multi ? 0 : un.id, c1 = un.id + 1;
      for (uint32_t ci = c0; ci < c1; ++ci) {
        const int nd = cands->candidates[ci];
        if (pos_of_node[nd] >= 0) ch.x_node.push_back((uint32_t)pos_of_node[nd]);
        if (any_limits && pos_of_node[nd] >= 0) {
          // the scheduler of a simulation does not charge the candidates against their NodePool's limits
          uint32_t npv = UINT32_MAX;
          for (auto& kv : cluster->nodes[nd].labels) if (kv.first == ctx->kNodePool) npv = kv.second;
          if (npv == UINT32_MAX) continue;
          for (int tpi = 0; tpi < NP; ++tpi) {
            const auto& tp = C->templates[tpi];
            if (!(tp.has_limits && !tp.limits.empty()) || V.vals[ctx->kNodePool].strs[npv] != tp.name) continue;
            for (auto& kv : cluster->nodes[nd].capacity) {
              bool limited = false;
              for (auto& lv : tp.limits) limited |= lv.first == kv.first;
              if (limited && ract[kv.first] >= 0) rem_adj[k * (size_t)R * NP + (size_t)ract[kv.first] * NP + tpi] += kv.second;
            }
          }
        }
      }
    }
  });
  {
    size_t qn = 0, xn = 0, tn = 0;
    for (auto& ch : pch) { ch.q_base = qn; ch.x_base = xn; ch.t_base = tn; qn += ch.q_ent.size(); xn += ch.x_node.size(); tn += ch.t_ent.size(); }
    check(qn / 4 < UINT32_MAX && xn < UINT32_MAX && tn < UINT32_MAX, "consolidation patches exceed 2^32 entries");
    q_ent.resize(qn); x_node.resize(xn); t_ent.resize(tn);
  }
  parallel_chunks(TC, [&](unsigned t) {
    PatchChunk& ch = pch[t];
    for (size_t k = ch.k0; k < ch.k1; ++k) { q_off[k] += (uint32_t)(ch.q_base / 4); x_off[k] += (uint32_t)ch.x_base; t_off[k] += (uint32_t)ch.t_base; }
    std::copy(ch.q_ent.begin(), ch.q_ent.end(), q_ent.begin() + ch.q_base);
    std::copy(ch.x_node.begin(), ch.x_node.end(), x_node.begin() + ch.x_base);
    std::copy(ch.t_ent.begin(), ch.t_ent.end(), t_ent.begin() + ch.t_base);
  });
  q_off[nu] = (uint32_t)(q_ent.size() / 4);
  x_off[nu] = (uint32_t)x_node.size();
  t_off[nu] = (uint32_t)t_ent.size();
  {
    Arena PA;
    PL.o_qoff = PA.add(q_off); PL.o_qent = PA.add(q_ent, 4); PL.o_xoff = PA.add(x_off); PL.o_xnode = PA.add(x_node); PL.o_rem = PA.add(rem_adj);
    PL.o_np = PA.add(nonpending); PL.o_un = PA.add(node_uninit); PL.o_toff = PA.add(t_off); PL.o_tent = PA.add(t_ent); PL.o_guniv = PA.add(C->gdom_univ);
    PL.tables.bytes = PA.bytes.size();
    PL.tables.ptr = ctx->pool_take(PL.tables.bytes, &PL.tables_cap);
    ctx->upload(PL.tables.ptr, PA.bytes.data(), PL.tables.bytes, true);  // (copied to the staging buffer: PA may go)
  }

  KG_PROF("plan: candidate prices");
  // capacity types in the catalog's value numbering (ctmask bits, CatOffering::lct); gspot: the context id node labels carry
  const int gspot = V.vals[ctx->kCT].find("spot"), god = V.vals[ctx->kCT].find("on-demand");
  PL.spot = gspot < 0 ? -1 : cat->tab.local(ctx->kCT, (uint32_t)gspot);
  PL.od = god < 0 ? -1 : cat->tab.local(ctx->kCT, (uint32_t)god);
  // (a candidate's instance type is resolved among the types of ITS NodePool: instanceTypeMap[nodePool][name] upstream)
  std::unordered_map<std::string, int> type_by_name;
  for (size_t t = 0; t < cat->types.size(); ++t) type_by_name.emplace(std::to_string(cat->part_begin.empty() ? 0 : cat->part_of((uint32_t)t)) + "/" + cat->types[t].name, (int)t);
  std::unordered_map<uint64_t, int> type_of_value;
  std::unordered_map<std::string, uint32_t> part_of_pool;
  for (uint32_t i = 0; i < nodepools->n_nodepools; ++i)
    part_of_pool.emplace(std::string(kgdesc::str(nodepools->pools.strings, nodepools->nodepools[i].name)), cat->part_begin.empty() ? 0 : nodepools->nodepools[i].catalog);
  PL.cinfo.assign(n_used, CandInfo());
  for (uint32_t i = 0; i < n_used; ++i) {
    const CNode& nd = cluster->nodes[cands->candidates[i]];
    int zone = -1, ctv = -1, zid = -1, itv = -1, npv = -1;
    for (auto& kv : nd.labels) {
      if (kv.first == ctx->kZone) zone = (int)kv.second;
      if (kv.first == ctx->kCT) ctv = (int)kv.second;
      if (kv.first == ctx->kZid) zid = (int)kv.second;
      if (kv.first == ctx->kInstanceType) itv = (int)kv.second;
      if (kv.first == ctx->kNodePool) npv = (int)kv.second;
    }
    CandInfo& ci = PL.cinfo[i];
    ci.spot = ctv >= 0 && ctv == gspot;
    if (itv < 0) continue;
    uint32_t part = 0;
    if (!cat->part_begin.empty()) {
      auto pf = npv < 0 ? part_of_pool.end() : part_of_pool.find(V.vals[ctx->kNodePool].strs[npv]);
      if (pf == part_of_pool.end()) continue;  // (not a node of one of these NodePools: no instance type to price)
      part = pf->second;
    }
    // (thousands of candidates, a few dozen instance types among them: the name is looked up once per (catalog, label value))
    const uint64_t tk = ((uint64_t)part << 32) | (uint32_t)itv;
    auto known = type_of_value.find(tk);
    if (known == type_of_value.end()) {
      auto f = type_by_name.find(std::to_string(part) + "/" + V.vals[ctx->kInstanceType].strs[itv]);
      known = type_of_value.emplace(tk, f == type_by_name.end() ? -1 : f->second).first;
    }
    if (known->second < 0) continue;
    ci.type = known->second;
    for (auto& o : cat->types[ci.type].offerings) {
      if (zone >= 0 && (int)o.zone != zone) continue;
      if (ctv >= 0 && (int)o.ct != ctv) continue;
      if (zid >= 0 && o.zid >= 0 && o.zid != zid) continue;
      if (!ci.ok || o.price < ci.price) { ci.price = o.price; ci.ok = true; }
    }
  }
  KG_PROF("plan: end");
  return Pp;
}

static int32_t consolidate_impl(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pending,
                                const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, bool multi, kg_candidate_record* records,
                                uint32_t cap_records, uint32_t* n_records, double* device_ms, ValidateOut* vout = nullptr) {
  if (!ctx || !catalogs || n_catalogs == 0 || !cluster || !nodepools || !cands || !n_records || (!records && cap_records)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  kg_catalog* cat = effective_catalog(ctx, catalogs, n_catalogs, nodepools);
  if (cat->types.size() > (size_t)KG_RECORD_TYPE_WORDS * 64) throw Unsupported("consolidation over more than 1024 instance types (all catalogs of the call together)");
  const uint32_t n = cands->n_candidates;
  const int shards = std::max(cands->n_shards, 1);
  check(cands->shard >= 0 && cands->shard < shards, "shard index");
  for (uint32_t i = 0; i < n; ++i) check(cands->candidates[i] >= 0 && cands->candidates[i] < (int)cluster->nodes.size(), "candidate node index");
  *n_records = 0;
  if (device_ms) *device_ms = 0;
  HostProf prof_;
  ctx->stage_off = 0;  // (every earlier call ended in a wait: its staged uploads have landed)
  std::optional<hal::Range> phase;
  phase.emplace("kg_consolidate: plan");
  KG_PROF("cons: plan lookup");
  // ---- the plan: found by the content of what it was built from, or built now
  kg_cons_plan* PLp = nullptr;
  std::unique_ptr<kg_cons_plan> one_off;
  if (vout) {
    one_off = build_cons_plan(ctx, cat, cluster, pending, nodepools, cands, multi, true);
    PLp = one_off.get();
  } else {
    Hasher H;
    for (uint32_t k = 0; k < n_catalogs; ++k) { H.u64(catalogs[k]->uid); H.u64(catalogs[k]->generation); H.u64(catalogs[k]->tab_epoch); }
    H.u64((uint64_t)(uintptr_t)cat->tab.dev.ptr); H.u64(ctx->V.resources.size());
    H.u64(multi); H.u64(ctx->collect_stats); H.u64((uint64_t)nodepools->max_instance_types);
    H.arr(cands->candidates, n);
    H.pools(nodepools->pools); H.arr(nodepools->nodepools, nodepools->n_nodepools); H.u64(nodepools->well_known_labels.begin); H.u64(nodepools->well_known_labels.count);
    if (pending) {
      H.pools(pending->pools); H.arr(pending->pods, pending->n_pods); H.arr(pending->weighted_terms, pending->n_weighted_terms); H.arr(pending->spreads, pending->n_spreads);
      H.arr(pending->affinity_terms, pending->n_affinity_terms); H.arr(pending->host_ports, pending->n_host_ports);
    }
    for (auto* p : cluster->plans)
      if (p->key == H.h && p->key2 == H.h2 && p->key_bytes == H.total && p->multi == multi && p->n == n && p->candidates.size() <= n &&
          (p->candidates.empty() || std::memcmp(p->candidates.data(), cands->candidates, p->candidates.size() * sizeof(int32_t)) == 0))
        PLp = p;  // (the plan keeps the candidates it uses: all of them, or the first 101 of a multi-node round)
    if (PLp) ctx->stats.cons_plan_hits += 1;
    else {
      ctx->stats.cons_plans_built += 1;
      if (cluster->plans.size() >= 4) {  // (single + multi of the current and of the previous candidate order)
        size_t old = 0;
        for (size_t i = 1; i < cluster->plans.size(); ++i) if (cluster->plans[i]->last_use < cluster->plans[old]->last_use) old = i;
        delete cluster->plans[old];
        cluster->plans.erase(cluster->plans.begin() + old);
      }
      auto np = build_cons_plan(ctx, cat, cluster, pending, nodepools, cands, multi, false);
      np->key = H.h; np->key2 = H.h2; np->key_bytes = H.total;
      PLp = np.release();
      cluster->plans.push_back(PLp);
    }
    PLp->last_use = ++cluster->plan_clock;
  }
  kg_cons_plan& PL = *PLp;
  Compiled* C = PL.C.get();
  const uint32_t n_used = PL.n_used;
  const std::vector<ConsUnit>& units = PL.units;
  const std::vector<CandInfo>& cinfo = PL.cinfo;
  const int spot = PL.spot, od = PL.od;
  if (vout) multi = true;

  KG_PROF("cons: shard");
  // ---- this shard's units (unit u belongs to shard (u - lo) mod n_shards = its index mod n_shards)
  std::vector<uint32_t> mine;
  for (size_t k = 0; k < units.size(); ++k) if (vout || (int)(k % (size_t)shards) == cands->shard) mine.push_back((uint32_t)k);
  *n_records = (uint32_t)mine.size();
  if (mine.size() > cap_records) return fail(ctx, KG_ENOMEM, "record buffer too small");
  for (size_t k = 0; k < mine.size(); ++k) {
    std::memset(&records[k], 0, sizeof(kg_candidate_record));
    records[k].candidate = (int32_t)units[mine[k]].id;
  }
  if (!C) return KG_OK;  // (no unit of the mode has pods to reschedule)
  if (PL.shard != cands->shard || PL.n_shards != shards || !PL.run.ptr) {
    // state slots and launch tables of this shard's live units: built when the shard is first run, reused by every later round
    PL.shard = cands->shard; PL.n_shards = shards;
    PL.live.clear();
    for (uint32_t k : mine) if (!units[k].skip) PL.live.push_back(k);
    // longest simulations first (the hardware hands workgroups out in launch order: the long ones must not start last)
    std::stable_sort(PL.live.begin(), PL.live.end(), [&](uint32_t a, uint32_t b) { return units[a].pod_hi - units[a].pod_lo > units[b].pod_hi - units[b].pod_lo; });
    PL.drop_shard();
    if (!PL.live.empty()) {
      const size_t stride = C->state_bytes, budget = (size_t)8 << 30, nl = PL.live.size();
      PL.max_batch = std::max<size_t>(1, std::min<size_t>(nl, budget / stride > 1 ? budget / stride - 1 : 1));
      PL.arena.bytes = (PL.max_batch + 1) * stride;
      KG_PROF("cons: shard: state slots");
      PL.arena.ptr = ctx->pool_take(PL.arena.bytes, &PL.arena_cap);
      // KG_POISON=1 (tests): whatever a simulation reads of its slot beyond the initialised prefix must have been written by it
      if (getenv("KG_POISON")) hal::dfill(PL.arena.ptr, 0xAB, PL.arena.bytes);
      ctx->upload(PL.arena.ptr, C->state_image.data(), C->state_init_bytes, true);
      KG_PROF("cons: shard: state descriptors");
      std::vector<SimState> sims(nl);
      for (size_t i = 0; i < nl; ++i) {
        sims[i] = sim_at(*C, (uint8_t*)PL.arena.ptr + (i % PL.max_batch + 1) * stride);
        sims[i].n_shared = PL.n_shared;
        sims[i].pod_base = units[PL.live[i]].pod_lo;
      }
      Arena RA;
      PL.o_sims = RA.add(sims); PL.o_unit = RA.add(PL.live); PL.o_nq = RA.reserve(nl * 4); PL.o_res = RA.reserve(nl * sizeof(SimResult));
      PL.run.bytes = RA.bytes.size();
      KG_PROF("cons: shard: upload");
      PL.run.ptr = ctx->pool_take(PL.run.bytes, &PL.run_cap);
      // (staged and not waited for: ordered on the context's stream before the round's kernels, which end in a wait)
      ctx->upload(PL.run.ptr, RA.bytes.data(), PL.run.bytes, true);
      PL.results_host = (SimResult*)ctx->pinned_take(nl * sizeof(SimResult), &PL.results_cap);
    }
  }
  const std::vector<uint32_t>& live = PL.live;
  if (live.empty()) return KG_OK;

  phase.reset();
  KG_PROF("cons: kernels");
  if (prof_.on) fprintf(stderr, "[kg host] simulation state: %zu bytes per slot, %zu initialised (ntouch %zu excl %zu remaining %zu gcount %zu queue %zu header %zu counters %zu)\n",
                        (size_t)C->state_bytes, (size_t)C->state_init_bytes, C->o_ntouch, C->o_excl, C->o_remaining, C->o_gcount, C->o_queue, C->o_header, C->o_counters);
  double ms_total = PL.ms_prepass, ms_init = 0, ms_sim = 0, ms_gather = 0;
  const float ms_pre = PL.ms_prepass;
  PL.ms_prepass = 0;  // (the prepass ran when the plan was built: charged to the first round only)
  const size_t stride = C->state_bytes;
  SimResult* const results = PL.results_host;
  auto tp = [&](size_t off) { return (uint8_t*)PL.tables.ptr + off; };
  auto rp_ = [&](size_t off) { return (uint8_t*)PL.run.ptr + off; };
  for (size_t b0 = 0; b0 < live.size(); b0 += PL.max_batch) {
    const size_t nb = std::min(PL.max_batch, live.size() - b0);
    SimBatch B{};
    B.arena = (uint8_t*)PL.arena.ptr; B.stride = stride; B.init_bytes = C->state_init_bytes; B.o_queue = C->o_queue; B.o_excl = C->o_excl; B.o_remaining = C->o_remaining;
    B.unit_of_sim = (const uint32_t*)rp_(PL.o_unit) + b0;
    B.q_off = (const uint32_t*)tp(PL.o_qoff); B.q_ent = (const uint4*)tp(PL.o_qent); B.x_off = (const uint32_t*)tp(PL.o_xoff); B.x_node = (const uint32_t*)tp(PL.o_xnode);
    B.rem_adj = PL.any_limits ? (const int64_t*)tp(PL.o_rem) : nullptr; B.nqueue = (uint32_t*)rp_(PL.o_nq) + b0; B.results = (SimResult*)rp_(PL.o_res) + b0;
    B.t_off = C->G > 0 ? (const uint32_t*)tp(PL.o_toff) : nullptr; B.t_ent = (const uint32_t*)tp(PL.o_tent); B.gdom_univ = (const uint64_t*)tp(PL.o_guniv);
    B.o_gcount = C->o_gcount; B.o_gdom = C->o_gdom; B.o_gpos = C->o_gpos; B.o_hcount = C->o_hcount; B.o_hpos = C->o_hpos;
    B.pod_nonpending = (const uint8_t*)tp(PL.o_np); B.node_uninit = (const uint64_t*)tp(PL.o_un); B.nsims = (int)nb;
    const SimState* sims_dev = (const SimState*)rp_(PL.o_sims) + b0;
    float ms3[3] = {0, 0, 0};
    hal::launch_sim_round(C->dp, B, sims_dev, ctx->sim_block_threads, ms3, results + b0, (const SimResult*)rp_(PL.o_res) + b0, nb * sizeof(SimResult));
    const float ms_i = ms3[0], ms_s = ms3[1], ms_g = ms3[2];
    ms_total += ms_i + ms_s + ms_g;
    ms_init += ms_i; ms_sim += ms_s; ms_gather += ms_g;
    ctx->stats.kernel_launches += 3;
    ctx->stats.solves += nb;
  }
  if (device_ms) *device_ms = ms_total;
  if (C->dp.stats & 8) {
    // debug (kg_set_collect_stats bit 5): per-wave cycle accounting of the batch's simulations, summed
    const size_t nb = std::min(PL.max_batch, live.size()), W = 16 * 32;
    std::vector<uint64_t> wc(nb * W), hdr(nb);
    hal::d2h_rows(wc.data(), (uint8_t*)PL.arena.ptr + C->o_counters + 64 * 8, W * 8, stride, nb);
    std::vector<double> sum(W, 0.0);
    for (size_t i = 0; i < nb; ++i) for (size_t k = 0; k < W; ++k) sum[k] += (double)wc[i * W + k];
    static const char* ph[] = {"prefilter", "wave-local Add", "commit", "fresh commit", "fresh setup", "workgroup Add", "nodes", "relax / ctx"};
    const int nw = ctx->sim_block_threads / 64;
    fprintf(stderr, "[kg sim waves] %zu simulations; cycles per simulation (work / wait at the barrier), per wave\n", nb);
    for (int k = 0; k < 8; ++k) {
      fprintf(stderr, "[kg sim waves] %-15s", ph[k]);
      for (int w = 0; w < nw; ++w) fprintf(stderr, "  %8.0f / %-8.0f", sum[w * 32 + 2 * k] / nb, sum[w * 32 + 2 * k + 1] / nb);
      fprintf(stderr, "\n");
    }
    for (int k = 16; k < 24; ++k) {
      fprintf(stderr, "[kg sim waves] slot %2d        ", k);
      for (int w = 0; w < nw; ++w) fprintf(stderr, "  %8.0f", sum[w * 32 + k] / nb);
      fprintf(stderr, "\n");
    }
  }
  if (C->dp.stats & 16) {
    // debug (kg_set_collect_stats bit 1): where the simulations of the last batch spent their time -- 100 MHz stamps each
    // workgroup left at kernel entry, after staging its tables and at its end
    const size_t nb = std::min(PL.max_batch, live.size());
    std::vector<uint64_t> st(nb * 4);
    hal::d2h_rows(st.data(), (uint8_t*)PL.arena.ptr + C->o_counters + 8 * 8, 32, stride, nb);
    uint64_t t0 = ~0ull, t1 = 0;
    double sx = 0, sy = 0, sxx = 0, sxy = 0, sstage = 0;
    for (size_t i = 0; i < nb; ++i) {
      const uint64_t* v = &st[i * 4];
      if (v[0]) t0 = std::min(t0, v[0]);  // (a slot the batch did not run keeps zeros)
      t1 = std::max(t1, v[2]);
      const double x = (double)v[3], y = (v[2] - v[1]) / 100.0;
      sx += x; sy += y; sxx += x * x; sxy += x * y; sstage += (v[1] - v[0]) / 100.0;
    }
    const double n = (double)nb, slope = (n * sxy - sx * sy) / std::max(1e-9, n * sxx - sx * sx), icpt = (sy - slope * sx) / n;
    fprintf(stderr, "[kg sims] %zu simulations, first entry -> last end %.1f us; staging %.2f us mean; loop = %.2f us + %.3f us x pods (mean pods %.1f, mean loop %.1f us)\n",
            nb, (t1 - t0) / 100.0, sstage / n, icpt, slope, sx / n, sy / n);
    std::vector<uint64_t> starts;
    for (size_t i = 0; i < nb; ++i) starts.push_back(st[i * 4] ? st[i * 4] - t0 : 0);
    std::sort(starts.begin(), starts.end());
    fprintf(stderr, "[kg sims] entry times (us after the first): p10 %.1f p50 %.1f p90 %.1f last %.1f\n", starts[nb / 10] / 100.0, starts[nb / 2] / 100.0,
            starts[nb * 9 / 10] / 100.0, starts[nb - 1] / 100.0);
  }
  ctx->stats.device_ms_total += ms_total;
  {
    uint64_t v[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < live.size(); ++i) for (int k = 0; k < 4; ++k) v[k] += results[i].visited[k];
    ctx->stats.cons_simulations = live.size();
    ctx->stats.cons_algorithmic_bytes = (double)KG_ROW_BYTES_EXISTING_NODE * v[0] + (double)KG_ROW_BYTES_NODE_CLAIM * (v[1] + v[3]) + (double)KG_ROW_BYTES_INSTANCE_TYPE * v[2];
    ctx->stats.cons_prepass_ms = ms_pre; ctx->stats.cons_sim_init_ms = ms_init; ctx->stats.cons_simulate_ms = ms_sim; ctx->stats.cons_gather_ms = ms_gather;
    ctx->stats.cons_state_bytes_per_simulation = stride;
  }

  if (vout) {
    // validation: AllNonPendingPodsScheduled, and the NodeClaims Results.TruncateInstanceTypes leaves
    const SimResult& sr = results[0];
    vout->all_scheduled = sr.errors == 0;
    vout->n_claims = sr.nclaims;
    if (sr.errors == 0 && sr.nclaims == 1) {
      uint64_t zmask = allowed1(sr.zone_meta, sr.zone_word);
      const uint64_t ctmask = allowed1(sr.ct_meta, sr.ct_word), zidm = allowed1(sr.zid_meta, sr.zid_word);
      const uint8_t* zz = cat->tab.image.bytes.data() + cat->tab.o_zone_zid;
      if (sr.zid_meta & RM_PRESENT)
        for (int z = 0; z < 64; ++z) if (zz[z] != 0xFF && !((zidm >> zz[z]) & 1)) zmask &= ~(1ull << z);
      std::vector<uint32_t> types;
      for (int w = 0; w < C->TW && w < KG_TWLDS; ++w)
        for (uint64_t m = sr.mask[w]; m; m &= m - 1) types.push_back((uint32_t)(w * 64 + __builtin_ctzll(m)));
      order_by_price(cat, &types, zmask, ctmask);
      std::vector<std::pair<uint32_t, int>> mv;
      for (auto& q : C->templates[sr.tmpl].reqs.rs) if (q.min_values >= 0) mv.push_back({q.key, q.min_values});
      if (nodepools->max_instance_types > 0) {
        std::vector<uint32_t> trunc(types.begin(), types.begin() + std::min<size_t>(types.size(), (size_t)nodepools->max_instance_types));
        if (!mv.empty() && !satisfies_min_values(cat, trunc, mv)) {
          // the claim is dropped and its pods fail
          if (sr.pad) vout->all_scheduled = false;
          vout->n_claims = 0;
          trunc.clear();
        }
        types.swap(trunc);
      }
      vout->types = types;
    }
    return KG_OK;
  }
  phase.emplace("kg_consolidate: decisions");
  KG_PROF("cons: decisions");
  // ---- computeConsolidation's decision logic per simulation (prices are float64 catalog data: host side)
  // (records[] is indexed by position in `mine`; live[li] is a unit index)
  std::vector<uint32_t> rec_of_unit(units.size(), 0);
  for (size_t k = 0; k < mine.size(); ++k) rec_of_unit[mine[k]] = (uint32_t)k;
  const uint8_t* zone_zid = cat->tab.image.bytes.data() + cat->tab.o_zone_zid;
  struct PriceOrder { uint64_t zmask, ctmask; std::vector<uint32_t> order; };
  std::vector<PriceOrder> orders;
  auto price_order = [&](uint64_t zmask, uint64_t ctmask) -> const std::vector<uint32_t>& {
    for (auto& o : orders) if (o.zmask == zmask && o.ctmask == ctmask) return o.order;
    orders.push_back({zmask, ctmask, {}});
    std::vector<uint32_t>& all = orders.back().order;
    all.resize(cat->types.size());
    for (uint32_t t = 0; t < all.size(); ++t) all[t] = t;
    order_by_price(cat, &all, zmask, ctmask);
    return all;
  };
  // the zones and capacity types a simulation's one NodeClaim still allows
  auto claim_masks = [&](const SimResult& sr, uint64_t* zmask, uint64_t* ctmask) {
    *zmask = allowed1(sr.zone_meta, sr.zone_word);
    *ctmask = allowed1(sr.ct_meta, sr.ct_word);
    const uint64_t zidm = allowed1(sr.zid_meta, sr.zid_word);
    if (sr.zid_meta & RM_PRESENT)
      for (int z = 0; z < 64; ++z) if (zone_zid[z] != 0xFF && !((zidm >> zone_zid[z]) & 1)) *zmask &= ~(1ull << z);
  };
  // the price orders of the round's distinct mask pairs first (a handful), so that the simulations below -- independent of
  // each other, one record each -- only read them and can be spread over the host's threads
  size_t n_priced = 0;
  for (size_t li = 0; li < live.size(); ++li) {
    const SimResult& sr = results[li];
    if (sr.errors != 0 || sr.nclaims != 1) continue;
    uint64_t zm, cm;
    claim_masks(sr, &zm, &cm);
    price_order(zm, cm);
    n_priced += 1;
  }
  const unsigned TD = (unsigned)std::min<size_t>(host_chunks(n_priced * 8), std::max<size_t>(live.size(), 1));
  parallel_chunks(TD, [&](unsigned td) {
  for (size_t li = live.size() * td / TD; li < live.size() * (td + 1) / TD; ++li) {
    const ConsUnit& un = units[live[li]];
    const SimResult& sr = results[li];
    kg_candidate_record& rec = records[rec_of_unit[live[li]]];
    Replacement rp;
    const uint32_t c0 = multi ? 0 : un.id, c1 = un.id + 1;
    if (sr.errors == 0 && sr.nclaims == 0) rp.decision = KG_DECISION_DELETE;
    else if (sr.errors == 0 && sr.nclaims == 1) {
      bool priced = true, all_spot = true;
      double price = 0;
      for (uint32_t ci = c0; ci < c1; ++ci) { priced &= cinfo[ci].ok; price += cinfo[ci].price; all_spot &= cinfo[ci].spot; }
      // Results.TruncateInstanceTypes comes before any price: the claim's options in price order, cut to the maximum,
      // minValues (the merged requirements of the claim carry its NodePool's) checked on what is left
      rp.nodepool = C->templates[sr.tmpl].index;
      claim_masks(sr, &rp.zmask, &rp.ctmask);
      // OrderByPrice of the claim's options = the catalog-wide order for these (zone, capacity-type) masks, filtered by the
      // claim's type mask (a stable sort of a subset keeps the relative order of the stable sort of the whole); a round
      // sees a handful of distinct mask pairs, so the sort runs that many times instead of once per simulation.  Only
      // the first max_instance_types of the order are ever looked at.
      const std::vector<uint32_t>* order_p = nullptr;
      for (auto& o : orders) if (o.zmask == rp.zmask && o.ctmask == rp.ctmask) order_p = &o.order;
      check(order_p != nullptr, "price order of a claim's masks");
      const std::vector<uint32_t>& order = *order_p;
      const size_t limit = nodepools->max_instance_types > 0 ? (size_t)nodepools->max_instance_types : order.size();
      for (uint32_t t : order) {
        if (t >= (uint32_t)(std::min(C->TW, (int)KG_TWLDS) * 64) || !((sr.mask[t >> 6] >> (t & 63)) & 1)) continue;
        rp.types.push_back(t);
        if (rp.types.size() == limit) break;
      }
      std::vector<std::pair<uint32_t, int>> mv;
      for (auto& q : C->templates[sr.tmpl].reqs.rs) if (q.min_values >= 0) mv.push_back({q.key, q.min_values});
      if (nodepools->max_instance_types > 0 && !mv.empty() && !satisfies_min_values(cat, rp.types, mv)) {
        // the claim is dropped and its pods fail: a rescheduled pod among them vetoes the command, otherwise no claim is left
        rp.types.clear();
        if (!sr.pad) rp.decision = KG_DECISION_DELETE;
        priced = false;
      }
      auto price_filter = [&](double max_price) {  // NodeClaim.RemoveInstanceTypeOptionsByPriceAndMinValues
        remove_by_price(cat, &rp, max_price, spot, od);
        return mv.empty() || satisfies_min_values(cat, rp.types, mv);
      };
      if (priced) {
        rp.candidate_price = price;
        {
        const bool has_spot = spot >= 0 && ((rp.ctmask >> spot) & 1), has_od = od >= 0 && ((rp.ctmask >> od) & 1);
        if (all_spot && has_spot) {
          // computeSpotToSpotConsolidation
          if (cands->spot_to_spot_enabled) {
            rp.ctmask = 1ull << spot;
            std::vector<uint32_t> compat;
            for (uint32_t t : rp.types) if (cheapest_price(cat->types[t], rp.zmask, rp.ctmask) != std::numeric_limits<double>::max()) compat.push_back(t);
            rp.types.swap(compat);
            if (!price_filter(price)) rp.types.clear();
            rp.spot_only = true;
            if (!rp.types.empty()) {
              if (c1 - c0 > 1) rp.decision = KG_DECISION_REPLACE;
              else if (rp.types.size() >= 15) {
                int keep = 15, need = 0;  // MinInstanceTypesForSpotToSpotConsolidation, or what minValues needs
                if (!mv.empty()) { satisfies_min_values(cat, rp.types, mv, &need); keep = std::max(keep, need); }
                if ((int)rp.types.size() > keep) rp.types.resize(keep);
                rp.decision = KG_DECISION_REPLACE;
              }
            }
          }
        } else {
          if (!price_filter(price)) rp.types.clear();
          if (!rp.types.empty()) {
            if (has_spot && has_od) { rp.ctmask = 1ull << spot; rp.spot_only = true; }
            rp.decision = KG_DECISION_REPLACE;
          }
        }
        if (multi && rp.decision == KG_DECISION_REPLACE) {
          // filterOutSameType
          double max_price = std::numeric_limits<double>::max();
          for (uint32_t t : rp.types)
            for (uint32_t ci = c0; ci < c1; ++ci)
              if (cinfo[ci].ok && cinfo[ci].price < max_price &&  // (upstream keys on the instance type NAME: across catalogs too)
                  (cinfo[ci].type == (int)t || (!cat->part_begin.empty() && cat->types[cinfo[ci].type].name == cat->types[t].name))) max_price = cinfo[ci].price;
          remove_by_price(cat, &rp, max_price, spot, od);  // (its minValues error is swallowed upstream)
          if (rp.types.empty()) rp.decision = KG_DECISION_NOOP;
        }
        }
      }
    }
    rec.decision = rp.decision;
    if (rp.decision == KG_DECISION_REPLACE) {
      rec.nodepool = rp.nodepool;
      rec.n_types = (uint32_t)rp.types.size();
      rec.candidate_price = rp.candidate_price;
      rec.spot_only = rp.spot_only;
      rec.zone_mask = rp.zmask;
      rec.ct_mask = rp.ctmask;
      for (uint32_t t : rp.types) {
        if (t >= KG_RECORD_TYPE_WORDS * 64) throw Unsupported("more than 1024 instance types in a consolidation record");
        rec.type_mask[t >> 6] |= 1ull << (t & 63);
      }
    }
  }
  });
  KG_PROF("cons: end");
  return KG_OK;
  KG_GUARD_END(ctx)
}

extern "C" {

int32_t kg_consolidate_single(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pending,
                              const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, kg_candidate_record* records,
                              uint32_t cap_records, uint32_t* n_records, double* device_ms) {
  return consolidate_impl(ctx, catalogs, n_catalogs, cluster, pending, nodepools, cands, false, records, cap_records, n_records, device_ms);
}

int32_t kg_consolidate_multi(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pending,
                             const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, kg_candidate_record* records,
                             uint32_t cap_records, uint32_t* n_records, double* device_ms) {
  return consolidate_impl(ctx, catalogs, n_catalogs, cluster, pending, nodepools, cands, true, records, cap_records, n_records, device_ms);
}

// Validation of a consolidation command against a fresh cluster snapshot: one more
// SimulateScheduling with the command's candidates; valid when every rescheduled pod still schedules and the outcome is
// the same kind -- no new NodeClaim for a DELETE; exactly one for a REPLACE, whose options still contain every instance
// type of the command.
int32_t kg_consolidate_validate(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pending,
                                const kg_nodepools_desc* nodepools, const int32_t* candidates, uint32_t n_candidates, int32_t decision,
                                const uint32_t* type_ids, uint32_t n_types, uint8_t* valid, double* device_ms) {
  if (!valid || (!candidates && n_candidates) || (!type_ids && n_types)) return KG_EINVAL;
  *valid = 0;
  if (n_candidates == 0 || n_candidates > 101 || decision == KG_DECISION_NOOP) return KG_EINVAL;
  kg_candidates_desc cd{};
  cd.candidates = candidates; cd.n_candidates = n_candidates; cd.shard = 0; cd.n_shards = 1;
  kg_candidate_record rec;
  uint32_t nrec = 0;
  ValidateOut vo;
  const int32_t st = consolidate_impl(ctx, catalogs, n_catalogs, cluster, pending, nodepools, &cd, true, &rec, 1, &nrec, device_ms, &vo);
  if (st != KG_OK) return st;
  if (!vo.all_scheduled) return KG_OK;
  if (vo.n_claims == 0) { *valid = decision == KG_DECISION_DELETE; return KG_OK; }
  if (vo.n_claims > 1 || decision != KG_DECISION_REPLACE) return KG_OK;
  std::set<uint32_t> have(vo.types.begin(), vo.types.end());
  bool subset = n_types > 0;
  for (uint32_t i = 0; i < n_types; ++i) subset = subset && have.count(type_ids[i]) != 0;
  *valid = subset;
  return KG_OK;
}

// SingleNodeConsolidation.ComputeCommand (first candidate in order with a command) / the binary search of
// MultiNodeConsolidation.firstNConsolidationOption, replayed over records gathered from every shard.
int32_t kg_consolidate_select(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, int32_t multi, const kg_candidate_record* records,
                              uint32_t n_records, uint32_t n_candidates, kg_decision* out) {
  if (!ctx || !catalogs || n_catalogs == 0 || !out || (!records && n_records)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  kg_catalog* catalog = catalog_view(ctx, catalogs, n_catalogs);  // (the view the records' type ids and masks refer to)
  const kg_candidate_record* win = nullptr;
  int win_n = 0;
  if (!multi) {
    for (uint32_t i = 0; i < n_records; ++i)
      if (records[i].decision != KG_DECISION_NOOP && (!win || records[i].candidate < win->candidate)) win = &records[i];
    win_n = 1;
  } else if (n_candidates >= 2) {
    std::unordered_map<int, const kg_candidate_record*> by_mid;
    for (uint32_t i = 0; i < n_records; ++i) by_mid[records[i].candidate] = &records[i];
    int mn = 1, mx = (int)std::min<uint32_t>(n_candidates, 101) - 1;  // (max = 100 unless len <= 100: then len - 1)
    while (mn <= mx) {
      const int mid = (mn + mx) / 2;
      auto f = by_mid.find(mid);
      if (f == by_mid.end()) throw kgdesc::DescError("multi-node selection: the record of a probed prefix is missing");
      if (f->second->decision != KG_DECISION_NOOP) { win = f->second; win_n = mid + 1; mn = mid + 1; } else mx = mid - 1;
    }
  }
  out->decision = win ? win->decision : KG_DECISION_NOOP;
  out->n_candidates = win ? win_n : 0;
  out->first_candidate = win ? (multi ? 0 : win->candidate) : -1;
  out->nodepool = win ? win->nodepool : 0;
  out->n_types = 0;
  out->replacement_spot_only = win ? win->spot_only : 0;
  out->candidate_price = win && win->decision == KG_DECISION_REPLACE ? win->candidate_price : 0;
  out->simulations = n_records;
  out->device_ms = 0;
  if (win && win->decision == KG_DECISION_REPLACE) {
    std::vector<uint32_t> types;
    for (int w = 0; w < KG_RECORD_TYPE_WORDS; ++w)
      for (uint64_t m = win->type_mask[w]; m; m &= m - 1) types.push_back((uint32_t)(w * 64 + __builtin_ctzll(m)));
    for (uint32_t t : types) check(t < catalog->types.size(), "record type id outside the catalog");
    order_by_price(catalog, &types, win->zone_mask, win->ct_mask);
    // (several catalogs: ids of the replacement NodePool's own catalog -- one claim, one NodePool, one catalog)
    const uint32_t base = catalog->part_begin.empty() || types.empty() ? 0 : catalog->part_begin[catalog->part_of(types[0])];
    for (uint32_t t : types) { if (out->n_types < out->cap_type_ids && out->type_ids) out->type_ids[out->n_types] = t - base; out->n_types++; }
  }
  return KG_OK;
  KG_GUARD_END(ctx)
}

}  // extern "C"

// ------------------------------------------------------------------ launch side (CloudProvider.Create)
namespace {

// zone / capacity-type admissible sets of a NodeClaim's requirements + its minValues, for price ordering
void launch_view(kg_ctx* ctx, kg_catalog* cat, const HReqs& reqs, ClaimView* cv) {
  Vocab& V = ctx->V;
  auto mask_of = [&](uint32_t key) -> uint64_t {  // over the catalog's value numbering
    const HReq* r = reqs.find(key);
    if (!r) return ~0ull;
    uint64_t m = 0;
    if (key < cat->tab.cvals.size())
      for (uint32_t l = 0; l < cat->tab.cvals[key].size() && l < 64; ++l) if (r->has(V, cat->tab.cvals[key][l])) m |= 1ull << l;
    return m;
  };
  cv->zmask = mask_of(ctx->kZone);
  cv->ctmask = mask_of(ctx->kCT);
  if (reqs.find(ctx->kZid)) {
    const uint64_t zidm = mask_of(ctx->kZid);
    const uint8_t* zone_zid = cat->tab.image.bytes.data() + cat->tab.o_zone_zid;
    for (int z = 0; z < 64; ++z) if (zone_zid[z] != 0xFF && !((zidm >> zone_zid[z]) & 1)) cv->zmask &= ~(1ull << z);
  }
  for (auto& q : reqs.rs) if (q.min_values >= 0) cv->minvals.push_back({q.key, q.min_values});
}

}  // namespace

extern "C" {

int32_t kg_truncate(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements, const uint32_t* type_ids, uint32_t n_types,
                    int32_t max_types, uint32_t* out_type_ids, uint32_t* n_out, uint8_t* min_values_failed) {
  if (!ctx || !catalog || !pools || (!type_ids && n_types) || !out_type_ids || !n_out) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  HReqs reqs = parse_reqs(ctx->V, *pools, requirements);
  ClaimView cv;
  launch_view(ctx, catalog, reqs, &cv);
  for (uint32_t i = 0; i < n_types; ++i) { check(type_ids[i] < catalog->types.size(), "type id"); cv.types.push_back(type_ids[i]); }
  const bool ok = truncate_types(catalog, cv, max_types);
  if (min_values_failed) *min_values_failed = !ok;
  *n_out = (uint32_t)cv.types.size();  // (on a minValues failure: the ordered, untruncated list; never more than n_types)
  for (size_t i = 0; i < cv.types.size(); ++i) out_type_ids[i] = cv.types[i];
  return KG_OK;
  KG_GUARD_END(ctx)
}

int32_t kg_launch_filter(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements, kg_range requests, int32_t max_types,
                         uint32_t* out_type_ids, uint32_t cap, uint32_t* n_out, uint8_t* min_values_failed) {
  if (!ctx || !catalog || !pools || !n_out || (!out_type_ids && cap)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  check(pools->strings.count > 0, "launch filter needs a non-empty string table");
  Vocab& V = ctx->V;
  // resolveInstanceTypes (cloudprovider.go:267-282) == NodeClaim.Add of one pod that carries the claim's requests on a
  // fresh claim whose template carries the claim's requirements: run exactly that on the device.
  kg_nodepool np{};
  np.name = 0;
  np.requirements = requirements;
  kg_nodepools_desc nps{};
  nps.pools = *pools;
  // (minValues play no part in resolveInstanceTypes; Truncate checks them afterwards)
  std::vector<kg_requirement> plain(pools->reqs, pools->reqs + pools->n_reqs);
  for (auto& q : plain) q.min_values = -1;
  nps.pools.reqs = plain.data();
  nps.nodepools = &np;
  nps.n_nodepools = 1;
  nps.max_instance_types = 0;
  kg_pod pod{};
  pod.uid = 0; pod.ns = 0;
  pod.requests = requests;
  pod.node = -1;
  kg_pods_desc pd{};
  pd.pools = *pools;
  pd.pods = &pod;
  pd.n_pods = 1;
  const uint32_t T = (uint32_t)catalog->types.size();
  std::vector<uint32_t> type_ids(std::max(T, 1u)), value_ids(4096), str_offsets(1024);
  std::vector<kg_claim_out> claims(1);
  std::vector<kg_req_out> rq(64);
  std::vector<kg_quantity> qs(64);
  std::vector<char> str_data(65536);
  uint8_t pod_kind = 0; int32_t pod_index = -1; uint32_t pod_reason = 0;
  kg_results res{};
  res.pod_kind = &pod_kind; res.pod_index = &pod_index; res.pod_reason = &pod_reason; res.cap_pods = 1;
  res.claims = claims.data(); res.cap_claims = 1;
  res.type_ids = type_ids.data(); res.cap_type_ids = (uint32_t)type_ids.size();
  res.reqs = rq.data(); res.cap_reqs = (uint32_t)rq.size();
  res.value_ids = value_ids.data(); res.cap_value_ids = (uint32_t)value_ids.size();
  res.quantities = qs.data(); res.cap_quantities = (uint32_t)qs.size();
  res.str_data = str_data.data(); res.cap_str_data = (uint32_t)str_data.size();
  res.str_offsets = str_offsets.data(); res.cap_strs = (uint32_t)str_offsets.size();
  kg_catalog* cats[1] = {catalog};
  const bool stats = ctx->collect_stats;
  ctx->collect_stats = false;
  const int32_t st = solve_impl(ctx, cats, nullptr, &pd, &nps, &res, false);
  ctx->collect_stats = stats;
  if (st != KG_OK && st != KG_ENOMEM) return st;
  ClaimView cv;
  HReqs reqs = parse_reqs(V, *pools, requirements);
  launch_view(ctx, catalog, reqs, &cv);
  if (pod_kind == KG_POD_NEW_CLAIM && res.n_claims == 1)
    for (uint32_t i = 0; i < claims[0].types.count; ++i) cv.types.push_back(type_ids[claims[0].types.begin + i]);
  if (cv.minvals.empty() && !cv.types.empty()) {
    // filterInstanceTypes (instance.go:389-399): only without minValues
    const int kSize = V.keys.find("karpenter.k8s.aws/instance-size");
    const char* exotic[] = {"aws.amazon.com/neuron", "amd.com/gpu", "nvidia.com/gpu", "habana.ai/gaudi"};
    std::vector<uint32_t> generic;
    for (uint32_t t : cv.types) {
      const CatType& ty = catalog->types[t];
      bool skip = false;
      if (kSize >= 0)
        if (const HReq* r = ty.reqs.find((uint32_t)kSize))
          for (uint32_t v : r->vals) skip |= V.vals[kSize].strs[v].find("metal") != std::string::npos;
      for (const char* rn : exotic) {
        const int rid = V.resources.find(rn);
        if (rid >= 0) for (auto& kv : ty.capacity) if ((int)kv.first == rid && kv.second != 0) skip = true;
      }
      if (!skip) generic.push_back(t);
    }
    if (!generic.empty()) cv.types.swap(generic);
    // isMixedCapacityLaunch + filterUnwantedSpot
    const int gspot = V.vals[ctx->kCT].find("spot"), god = V.vals[ctx->kCT].find("on-demand");
    const int spot = gspot < 0 ? -1 : catalog->tab.local(ctx->kCT, (uint32_t)gspot), od = god < 0 ? -1 : catalog->tab.local(ctx->kCT, (uint32_t)god);
    if (spot >= 0 && od >= 0 && ((cv.ctmask >> spot) & 1) && ((cv.ctmask >> od) & 1)) {
      bool has_spot = false, has_od = false;
      for (uint32_t t : cv.types)
        for (auto& o : catalog->types[t].offerings) {
          if (!o.available || !((cv.zmask >> o.lz) & 1) || !((cv.ctmask >> o.lct) & 1)) continue;
          if ((int)o.lct == spot) has_spot = true; else has_od = true;
        }
      if (has_spot && has_od) {
        double cheapest_od = std::numeric_limits<double>::max();
        for (uint32_t t : cv.types)
          for (auto& o : catalog->types[t].offerings) if (o.available && (int)o.lct == od && o.price < cheapest_od) cheapest_od = o.price;
        std::vector<uint32_t> kept;
        for (uint32_t t : cv.types) {
          bool any = false;
          double best = 0;
          for (auto& o : catalog->types[t].offerings) if (o.available && (!any || o.price < best)) { best = o.price; any = true; }
          if (any && best <= cheapest_od) kept.push_back(t);
        }
        cv.types.swap(kept);
      }
    }
  }
  const bool ok = truncate_types(catalog, cv, max_types);
  if (min_values_failed) *min_values_failed = !ok;
  *n_out = (uint32_t)cv.types.size();
  for (size_t i = 0; i < cv.types.size() && i < cap; ++i) out_type_ids[i] = cv.types[i];
  return cv.types.size() > cap ? fail(ctx, KG_ENOMEM, "type id buffer too small") : KG_OK;
  KG_GUARD_END(ctx)
}

// getCapacityType (instance.go:372-385) + getOverrides (instance.go:323-359).  An offering's requirements are its zone,
// capacity type and (when the zone has one) zone id, all well-known labels: Compatible(..., AllowUndefinedWellKnownLabels)
// reduces to "the NodeClaim's requirement on each of the three keys, if it has one, admits the offering's value" -- the
// zone / capacity-type masks launch_view derives (zone id folded into the zone mask).
int32_t kg_launch_overrides(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements, const uint32_t* type_ids, uint32_t n_types,
                            const uint32_t* subnet_zones, uint32_t n_subnet_zones, int32_t* capacity_type, kg_override* out, uint32_t cap, uint32_t* n_out) {
  if (!ctx || !catalog || !pools || (!type_ids && n_types) || (!subnet_zones && n_subnet_zones) || !capacity_type || !n_out || (!out && cap)) return KG_EINVAL;
  CtxLock lk(ctx);
  KG_GUARD_BEGIN
  Vocab& V = ctx->V;
  HReqs reqs = parse_reqs(V, *pools, requirements);
  ClaimView cv;
  launch_view(ctx, catalog, reqs, &cv);
  for (uint32_t i = 0; i < n_types; ++i) check(type_ids[i] < catalog->types.size(), "type id");
  const int gspot = V.vals[ctx->kCT].find("spot"), god = V.vals[ctx->kCT].find("on-demand");
  const int spot = gspot < 0 ? -1 : catalog->tab.local(ctx->kCT, (uint32_t)gspot), od = god < 0 ? -1 : catalog->tab.local(ctx->kCT, (uint32_t)god);
  // requirements.Get(capacity-type).Has("spot"): a NodeClaim without the key admits everything (Exists)
  const HReq* ctr = reqs.find(ctx->kCT);
  const bool spot_ok = !ctr || (gspot >= 0 ? ctr->has(V, (uint32_t)gspot) : ctr->has(V, V.val(ctx->kCT, "spot")));
  int ct = od;
  *capacity_type = KG_CT_ON_DEMAND;
  if (spot_ok && spot >= 0)
    for (uint32_t i = 0; i < n_types && ct == od; ++i)
      for (auto& o : catalog->types[type_ids[i]].offerings)
        if (o.available && (int)o.lct == spot && ((cv.zmask >> o.lz) & 1)) { ct = spot; *capacity_type = KG_CT_SPOT; break; }
  // zone -> first subnet entry naming it (catalog-local zone numbering; a zone the catalog has never listed matches nothing)
  int subnet_of_zone[64];
  for (int z = 0; z < 64; ++z) subnet_of_zone[z] = -1;
  for (uint32_t s = 0; s < n_subnet_zones; ++s) {
    const int g = V.vals[ctx->kZone].find(std::string(kgdesc::str(pools->strings, subnet_zones[s])));
    const int l = g < 0 ? -1 : catalog->tab.local(ctx->kZone, (uint32_t)g);
    if (l >= 0 && l < 64 && subnet_of_zone[l] < 0) subnet_of_zone[l] = (int)s;
  }
  uint32_t n = 0;
  if (ct >= 0)
    for (uint32_t i = 0; i < n_types; ++i)
      for (auto& o : catalog->types[type_ids[i]].offerings) {
        if (!o.available || (int)o.lct != ct || !((cv.zmask >> o.lz) & 1) || subnet_of_zone[o.lz] < 0) continue;
        if (n < cap) out[n] = kg_override{type_ids[i], (uint32_t)subnet_of_zone[o.lz]};
        n++;
      }
  *n_out = n;
  return n > cap ? fail(ctx, KG_ENOMEM, "override buffer too small") : KG_OK;
  KG_GUARD_END(ctx)
}

}  // extern "C"
