// oracle.cpp -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's bin-packing path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (libkarpgpu) never links or calls it.
//
// What it restates, and where the algorithm comes from:
//   * The scheduler core (Requirements algebra, Scheduler.Solve/add, Queue, NodeClaim.Add,
//     ExistingNode.Add, Topology/TopologyGroup, Preferences.Relax, InstanceTypes.OrderByPrice /
//     Truncate / SatisfiesMinValues, disruption SimulateScheduling / computeConsolidation /
//     firstNConsolidationOption) lives in the third-party module
//     sigs.k8s.io/karpenter v0.37.1-0.20240715214311-a122e25df3de (reference go.mod:34), whose
//     source is NOT in the reference checkout.  This file restates its published algorithm as
//     recorded in SURVEY.md Appendix A (A.1-A.10); each function names the upstream function it
//     follows.  PARITY UNPINNED for exact pod->node assignment (no reference test asserts it);
//     the set-level outcomes the reference's own tests pin (instancetype/suite_test.go:390-434,
//     693-884; cloudprovider/suite_test.go:365-384) are checked in tests/test_oracle_pins.py.
//   * The three launch predicates are the ones spelled out in-repo at
//     pkg/cloudprovider/cloudprovider.go:276-281; launch filters follow
//     pkg/providers/instance/instance.go:372-477.
//
// Declared canonical choices where the reference is nondeterministic (Go map order / unstable
// sort), applied identically by the device path (DESIGN.md "Canonical order"):
//   - sort.Slice of in-flight NodeClaims by len(Pods): STABLE, ties keep prior slice order;
//   - NodePools by weight desc: stable, ties keep input order;
//   - topology domain ties, Requirement.Any(): lexicographically smallest value;
//   - offerings: insertion order of the descriptor (never observable: only prices are used).
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "karpgpu.h"
#include "karpgpu_desc.hpp"

namespace ko {
using kgdesc::check;
using std::optional;
using std::string;
using std::vector;

// ------------------------------------------------------------------ interning
struct Intern {
  vector<string> strs;
  std::unordered_map<string, int> ids;
  int id(std::string_view s) {
    auto it = ids.find(string(s));
    if (it != ids.end()) return it->second;
    int i = (int)strs.size();
    strs.emplace_back(s);
    ids.emplace(strs.back(), i);
    return i;
  }
  const string& s(int i) const { return strs[i]; }
};
static thread_local Intern* G = nullptr;  // active intern table of the running call

static bool lexLess(int a, int b) { return G->s(a) < G->s(b); }

// ------------------------------------------------------------------ Requirement (Appendix A.1)
enum Op { In = KG_OP_IN, NotIn = KG_OP_NOT_IN, Exists = KG_OP_EXISTS, DoesNotExist = KG_OP_DOES_NOT_EXIST, Gt = KG_OP_GT, Lt = KG_OP_LT };

static bool withinIntPtrs(int value, const optional<int64_t>& gt, const optional<int64_t>& lt) {
  if (!gt && !lt) return true;
  int64_t v;
  if (!kgdesc::atoi_go(G->s(value), &v)) return false;  // non integer values are invalid when bounds are set
  if (gt && *gt >= v) return false;
  if (lt && *lt <= v) return false;
  return true;
}

struct Requirement {
  int key = -1;
  bool complement = false;
  std::set<int> values;
  optional<int64_t> gt, lt;
  optional<int> minValues;

  // scheduling.NewRequirementWithFlexibility
  static Requirement New(int key, int op, const vector<int>& vals, optional<int> minValues = {}) {
    Requirement r;
    r.key = key;
    r.minValues = minValues;
    if (op == In) {
      r.values.insert(vals.begin(), vals.end());
      return r;
    }
    r.complement = true;
    if (op == DoesNotExist) r.complement = false;
    if (op == NotIn) r.values.insert(vals.begin(), vals.end());
    if (op == Gt || op == Lt) {
      int64_t v = 0;
      check(!vals.empty() && kgdesc::atoi_go(G->s(vals[0]), &v), "Gt/Lt needs one integer value");
      if (op == Gt) r.gt = v; else r.lt = v;
    }
    return r;
  }
  bool Has(int value) const {
    if (complement) return !values.count(value) && withinIntPtrs(value, gt, lt);
    return values.count(value) && withinIntPtrs(value, gt, lt);
  }
  int64_t Len() const { return complement ? INT64_MAX - (int64_t)values.size() : (int64_t)values.size(); }
  int Operator() const {
    if (complement) return Len() < INT64_MAX ? NotIn : Exists;
    return Len() > 0 ? In : DoesNotExist;
  }
  // Any(): arbitrary in Go; canonical = lexicographically smallest
  int Any() const {
    int best = -1;
    for (int v : values) if (best < 0 || lexLess(v, best)) best = v;
    return best;
  }
  vector<int> SortedValues() const {
    vector<int> v(values.begin(), values.end());
    std::sort(v.begin(), v.end(), lexLess);
    return v;
  }
  // Requirement.Intersection
  Requirement Intersection(const Requirement& o) const {
    Requirement r;
    r.key = key;
    r.complement = complement && o.complement;
    if (gt && o.gt) r.gt = std::max(*gt, *o.gt); else r.gt = gt ? gt : o.gt;
    if (lt && o.lt) r.lt = std::min(*lt, *o.lt); else r.lt = lt ? lt : o.lt;
    if (minValues && o.minValues) r.minValues = std::max(*minValues, *o.minValues); else r.minValues = minValues ? minValues : o.minValues;
    if (r.gt && r.lt && *r.gt >= *r.lt) {
      Requirement dne = New(key, DoesNotExist, {}, r.minValues);
      return dne;
    }
    if (complement && o.complement) {
      r.values = values;
      r.values.insert(o.values.begin(), o.values.end());
    } else if (complement && !o.complement) {
      for (int v : o.values) if (!values.count(v)) r.values.insert(v);
    } else if (!complement && o.complement) {
      for (int v : values) if (!o.values.count(v)) r.values.insert(v);
    } else {
      for (int v : values) if (o.values.count(v)) r.values.insert(v);
    }
    for (auto it = r.values.begin(); it != r.values.end();) {
      if (!withinIntPtrs(*it, r.gt, r.lt)) it = r.values.erase(it); else ++it;
    }
    if (!r.complement) { r.gt.reset(); r.lt.reset(); }
    return r;
  }
};

// ------------------------------------------------------------------ Requirements (Appendix A.2)
struct Requirements {
  std::map<int, Requirement> m;
  bool Has(int key) const { return m.count(key) != 0; }
  Requirement Get(int key) const {
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    return Requirement::New(key, Exists, {});  // missing => Exists
  }
  void Add(const Requirement& in) {
    auto it = m.find(in.key);
    if (it != m.end()) it->second = in.Intersection(it->second); else m.emplace(in.key, in);
  }
  void Add(const Requirements& o) { for (auto& kv : o.m) Add(kv.second); }
  bool HasMinValues() const { for (auto& kv : m) if (kv.second.minValues) return true; return false; }
  // Requirements.Intersects
  bool Intersects(const Requirements& incoming) const {
    for (auto& kv : m) {
      auto it = incoming.m.find(kv.first);
      if (it == incoming.m.end()) continue;
      const Requirement& existing = kv.second;
      const Requirement& inc = it->second;
      if (existing.Intersection(inc).Len() == 0) {
        int io = inc.Operator(), eo = existing.Operator();
        if ((io == NotIn || io == DoesNotExist) && (eo == NotIn || eo == DoesNotExist)) continue;
        return false;
      }
    }
    return true;
  }
  // Requirements.Compatible; allowUndefined = nullptr means "no options"
  bool Compatible(const Requirements& incoming, const std::set<int>* allowUndefined) const {
    bool ok = true;
    for (auto& kv : incoming.m) {
      if (allowUndefined && allowUndefined->count(kv.first)) continue;
      int op = kv.second.Operator();
      if (Has(kv.first) || op == NotIn || op == DoesNotExist) continue;
      ok = false;  // label does not have known values
    }
    return Intersects(incoming) && ok;
  }
};

// ------------------------------------------------------------------ resources (Appendix A.10)
using RL = std::map<int, int64_t>;
static int64_t rlGet(const RL& r, int k) { auto it = r.find(k); return it == r.end() ? 0 : it->second; }
static RL Merge(const RL& a, const RL& b) { RL r = a; for (auto& kv : b) r[kv.first] += kv.second; return r; }
static RL Subtract(const RL& lhs, const RL& rhs) { RL r; for (auto& kv : lhs) r[kv.first] = kv.second - rlGet(rhs, kv.first); return r; }
static bool Fits(const RL& candidate, const RL& total) {
  for (auto& kv : total) if (kv.second < 0) return false;
  for (auto& kv : candidate) if (kv.second > rlGet(total, kv.first)) return false;
  return true;
}

// ------------------------------------------------------------------ catalog types
struct Offering {
  Requirements reqs;
  double price;
  bool available;
  int zone, ct;
};
struct InstanceType {
  int index;  // catalog-local
  int name;
  Requirements reqs;
  vector<Offering> offerings;
  RL capacity, overhead, allocatable;
};
struct Catalog {
  vector<std::unique_ptr<InstanceType>> types;
};

struct Parser {
  Intern& in;
  int sid(const kg_pools& p, uint32_t id) { return in.id(kgdesc::str(p.strings, id)); }
  Requirement req(const kg_pools& p, const kg_requirement& q) {
    vector<int> vals;
    for (uint32_t v : kgdesc::values(p, q.values)) vals.push_back(sid(p, v));
    optional<int> mv;
    if (q.min_values >= 0) mv = q.min_values;
    return Requirement::New(sid(p, q.key), q.op, vals, mv);
  }
  Requirements reqs(const kg_pools& p, kg_range r) {
    Requirements out;
    for (auto& q : kgdesc::reqs(p, r)) out.Add(req(p, q));
    return out;
  }
  RL rl(const kg_pools& p, kg_range r) {
    RL out;
    for (auto& q : kgdesc::quantities(p, r)) out[sid(p, q.name)] += q.milli;
    return out;
  }
};

static std::unique_ptr<Catalog> parseCatalog(Intern& in, const kg_catalog_desc* d, int keyCT, int keyZone, int keyZoneID) {
  Parser P{in};
  auto cat = std::make_unique<Catalog>();
  for (uint32_t i = 0; i < d->n_types; ++i) {
    const kg_instance_type& t = d->types[i];
    auto it = std::make_unique<InstanceType>();
    it->index = (int)i;
    it->name = P.sid(d->pools, t.name);
    it->reqs = P.reqs(d->pools, t.requirements);
    it->capacity = P.rl(d->pools, t.capacity);
    it->overhead = P.rl(d->pools, t.overhead);
    it->allocatable = Subtract(it->capacity, it->overhead);
    for (auto& o : kgdesc::span(d->offerings, d->n_offerings, t.offerings, "offering range")) {
      Offering of;
      of.zone = P.sid(d->pools, o.zone);
      of.ct = P.sid(d->pools, o.capacity_type);
      of.price = o.price;
      of.available = o.available != 0;
      of.reqs.Add(Requirement::New(keyCT, In, {of.ct}));
      of.reqs.Add(Requirement::New(keyZone, In, {of.zone}));
      if (o.zone_id != KG_NONE) of.reqs.Add(Requirement::New(keyZoneID, In, {P.sid(d->pools, o.zone_id)}));
      it->offerings.push_back(std::move(of));
    }
    cat->types.push_back(std::move(it));
  }
  return cat;
}

// ------------------------------------------------------------------ pods
struct ReqDesc { int key, op; vector<int> values; };
struct Selector {
  bool nil = false;
  vector<std::pair<int, int>> matchLabels;
  vector<ReqDesc> exprs;
  // metav1.LabelSelectorAsSelector(...).Matches(labels)
  bool Matches(const std::map<int, int>& labels) const {
    if (nil) return false;
    for (auto& kv : matchLabels) {
      auto it = labels.find(kv.first);
      if (it == labels.end() || it->second != kv.second) return false;
    }
    for (auto& e : exprs) {
      auto it = labels.find(e.key);
      bool has = it != labels.end();
      bool inSet = has && std::find(e.values.begin(), e.values.end(), it->second) != e.values.end();
      if (e.op == In && !inSet) return false;
      if (e.op == NotIn && inSet) return false;
      if (e.op == Exists && !has) return false;
      if (e.op == DoesNotExist && has) return false;
    }
    return true;
  }
  string canon() const {
    if (nil) return "nil";
    auto ml = matchLabels;
    std::sort(ml.begin(), ml.end());
    string s = "L";
    for (auto& kv : ml) s += std::to_string(kv.first) + "=" + std::to_string(kv.second) + ",";
    vector<string> es;
    for (auto& e : exprs) {
      auto v = e.values;
      std::sort(v.begin(), v.end());
      string t = std::to_string(e.key) + ":" + std::to_string(e.op) + ":";
      for (int x : v) t += std::to_string(x) + ",";
      es.push_back(t);
    }
    std::sort(es.begin(), es.end());
    s += "E";
    for (auto& t : es) s += t + ";";
    return s;
  }
};
struct Spread { int key; int maxSkew; optional<int> minDomains; int whenUnsat; Selector sel; };
struct AffTerm { int key; Selector sel; vector<int> namespaces; int weight; };
struct WTerm { int weight; vector<ReqDesc> reqs; };
struct Toleration { int key; int value; int op; int effect; };  // key -1 = empty
struct Taint { int key, value, effect; };
struct HostPort { int ip; int port; int proto; };

struct Pod {
  int index = -1;  // index in its descriptor
  bool pending = true;
  int uid, ns;
  int64_t creation;
  std::map<int, int> labels;
  RL requests;  // RequestsForPods incl. pods:1
  vector<std::pair<int, int>> nodeSelector;
  vector<vector<ReqDesc>> requiredTerms;
  vector<WTerm> preferredTerms;
  vector<Toleration> tolerations;
  vector<Spread> spreads;
  vector<AffTerm> affReq, affPref, antiReq, antiPref;
  vector<HostPort> hostPorts;
  std::map<int, std::set<int>> volumes;  // CSI driver -> ids of the PVC-backed volumes the pod mounts
  int priority = 0, deletionCost = 0, node = -1;
  bool reschedulable = true;
};

static vector<ReqDesc> parseReqDescs(Parser& P, const kg_pools& p, kg_range r) {
  vector<ReqDesc> out;
  for (auto& q : kgdesc::reqs(p, r)) {
    ReqDesc d{P.sid(p, q.key), q.op, {}};
    for (uint32_t v : kgdesc::values(p, q.values)) d.values.push_back(P.sid(p, v));
    out.push_back(std::move(d));
  }
  return out;
}
static Selector parseSelector(Parser& P, const kg_pools& p, const kg_selector& s) {
  Selector out;
  out.nil = s.is_nil != 0;
  for (auto& l : kgdesc::labels(p, s.match_labels)) out.matchLabels.push_back({P.sid(p, l.key), P.sid(p, l.value)});
  out.exprs = parseReqDescs(P, p, s.match_exprs);
  return out;
}
static vector<AffTerm> parseTerms(Parser& P, const kg_pods_desc* d, kg_range r) {
  vector<AffTerm> out;
  for (auto& t : kgdesc::span(d->affinity_terms, d->n_affinity_terms, r, "affinity range")) {
    AffTerm a{P.sid(d->pools, t.topology_key), parseSelector(P, d->pools, t.selector), {}, t.weight};
    for (uint32_t v : kgdesc::values(d->pools, t.namespaces)) a.namespaces.push_back(P.sid(d->pools, v));
    out.push_back(std::move(a));
  }
  return out;
}

static vector<std::unique_ptr<Pod>> parsePods(Intern& in, const kg_pods_desc* d, int resPods, bool pending) {
  Parser P{in};
  vector<std::unique_ptr<Pod>> out;
  if (!d) return out;
  const kg_pools& p = d->pools;
  for (uint32_t i = 0; i < d->n_pods; ++i) {
    const kg_pod& q = d->pods[i];
    auto pod = std::make_unique<Pod>();
    pod->index = (int)i;
    pod->pending = pending;
    pod->uid = P.sid(p, q.uid);
    pod->ns = P.sid(p, q.ns);
    pod->creation = q.creation_ts;
    for (auto& l : kgdesc::labels(p, q.labels)) pod->labels[P.sid(p, l.key)] = P.sid(p, l.value);
    pod->requests = P.rl(p, q.requests);
    pod->requests[resPods] += 1000;  // RequestsForPods adds pods: 1
    for (auto& l : kgdesc::labels(p, q.node_selector)) pod->nodeSelector.push_back({P.sid(p, l.key), P.sid(p, l.value)});
    for (auto& r : kgdesc::ranges(p, q.required_terms)) pod->requiredTerms.push_back(parseReqDescs(P, p, r));
    for (auto& w : kgdesc::span(d->weighted_terms, d->n_weighted_terms, q.preferred_terms, "weighted range"))
      pod->preferredTerms.push_back({w.weight, parseReqDescs(P, p, w.reqs)});
    for (auto& t : kgdesc::tolerations(p, q.tolerations))
      pod->tolerations.push_back({t.key == KG_NONE ? -1 : P.sid(p, t.key), t.value == KG_NONE ? P.in.id("") : P.sid(p, t.value), t.op, t.effect});
    for (auto& s : kgdesc::span(d->spreads, d->n_spreads, q.spreads, "spread range")) {
      Spread sp{P.sid(p, s.topology_key), s.max_skew, {}, s.when_unsatisfiable, parseSelector(P, p, s.selector)};
      if (s.min_domains >= 0) sp.minDomains = s.min_domains;
      pod->spreads.push_back(std::move(sp));
    }
    pod->affReq = parseTerms(P, d, q.affinity_required);
    pod->affPref = parseTerms(P, d, q.affinity_preferred);
    pod->antiReq = parseTerms(P, d, q.anti_affinity_required);
    pod->antiPref = parseTerms(P, d, q.anti_affinity_preferred);
    for (auto& h : kgdesc::span(d->host_ports, d->n_host_ports, q.host_ports, "host port range"))
      pod->hostPorts.push_back({h.ip == KG_NONE ? -1 : P.sid(p, h.ip), h.port, h.protocol});
    for (auto& l : kgdesc::labels(p, q.volumes)) pod->volumes[P.sid(p, l.key)].insert(P.sid(p, l.value));
    pod->priority = q.priority;
    pod->deletionCost = q.deletion_cost;
    pod->node = q.node;
    pod->reschedulable = q.reschedulable != 0;
    out.push_back(std::move(pod));
  }
  return out;
}

static Requirements fromReqDescs(const vector<ReqDesc>& ds) {
  Requirements r;
  for (auto& d : ds) r.Add(Requirement::New(d.key, d.op, d.values));
  return r;
}
static Requirements NewLabelRequirements(const vector<std::pair<int, int>>& labels) {
  Requirements r;
  for (auto& kv : labels) r.Add(Requirement::New(kv.first, In, {kv.second}));
  return r;
}
// scheduling.newPodRequirements: nodeSelector + heaviest preferred term (all) + first required term
static Requirements NewPodRequirements(const Pod& pod, bool strict) {
  Requirements r = NewLabelRequirements(pod.nodeSelector);
  if (!strict && !pod.preferredTerms.empty()) {
    // sort.Slice by weight desc (canonical: stable) then take [0]
    size_t best = 0;
    for (size_t i = 1; i < pod.preferredTerms.size(); ++i)
      if (pod.preferredTerms[i].weight > pod.preferredTerms[best].weight) best = i;
    r.Add(fromReqDescs(pod.preferredTerms[best].reqs));
  }
  if (!pod.requiredTerms.empty()) r.Add(fromReqDescs(pod.requiredTerms[0]));
  return r;
}

// Taints.Tolerates(pod): every taint needs a matching toleration (corev1 Toleration.ToleratesTaint)
static bool Tolerates(const vector<Taint>& taints, const Pod& pod) {
  for (auto& taint : taints) {
    bool ok = false;
    for (auto& t : pod.tolerations) {
      if (t.effect != KG_EFFECT_NONE && t.effect != taint.effect) continue;
      if (t.key >= 0 && t.key != taint.key) continue;
      if (t.op == KG_TOL_EXISTS) { ok = true; break; }
      if (t.value == taint.value) { ok = true; break; }
    }
    if (!ok) return false;
  }
  return true;
}

// scheduling.HostPortUsage
static bool hostPortMatches(const HostPort& a, const HostPort& b) {
  if (a.proto != b.proto || a.port != b.port) return false;
  if (a.ip != b.ip && a.ip != -1 && b.ip != -1) return false;
  return true;
}
struct HostPortUsage {
  vector<std::pair<int, HostPort>> reserved;  // (pod uid, port)
  bool Conflicts(const Pod& pod) const {
    for (auto& np : pod.hostPorts)
      for (auto& e : reserved)
        if (hostPortMatches(np, e.second) && e.first != pod.uid) return true;
    return false;
  }
  void Add(const Pod& pod) { for (auto& hp : pod.hostPorts) reserved.push_back({pod.uid, hp}); }
};

// ------------------------------------------------------------------ Topology (Appendix A.7)
enum TopoType { TSpread = 0, TAffinity = 1, TAntiAffinity = 2 };

struct TopologyGroup {
  int key;
  int type;
  int maxSkew;
  optional<int> minDomains;
  std::set<int> namespaces;
  Selector selector;
  bool hasFilter = false;            // spread only
  vector<Requirements> nodeFilter;   // OR terms
  std::set<int> owners;
  std::map<int, int> domains;
  std::set<int> emptyDomains;
  bool isHostname = false;

  bool selects(const Pod& p) const { return namespaces.count(p.ns) && selector.Matches(p.labels); }
  bool filterMatches(const Requirements& reqs, const std::set<int>* allowUndefined) const {
    if (!hasFilter || nodeFilter.empty()) return true;
    for (auto& f : nodeFilter) if (reqs.Compatible(f, allowUndefined)) return true;
    return false;
  }
  bool Counts(const Pod& p, const Requirements& reqs, const std::set<int>* allowUndefined) const {
    return selects(p) && filterMatches(reqs, allowUndefined);
  }
  void Record(int domain) { domains[domain]++; emptyDomains.erase(domain); }
  void Register(int domain) { if (!domains.count(domain)) { domains[domain] = 0; emptyDomains.insert(domain); } }
  void Unregister(int domain) { domains.erase(domain); emptyDomains.erase(domain); }

  int domainMinCount(const Requirement& podDomains) const {
    if (isHostname) return 0;  // hostname topologies always have min 0: a node can be created
    int mn = INT_MAX, supported = 0;
    for (auto& kv : domains) {
      if (podDomains.Has(kv.first)) {
        supported++;
        if (kv.second < mn) mn = kv.second;
      }
    }
    if (minDomains && supported < *minDomains) mn = 0;
    return mn;
  }
  Requirement nextDomainTopologySpread(const Pod& pod, const Requirement& podDomains, const Requirement& nodeDomains) const {
    int mn = domainMinCount(podDomains);
    bool self = selects(pod);
    int minDomain = -1, minCount = INT_MAX;
    for (auto& kv : domains) {
      if (!nodeDomains.Has(kv.first)) continue;
      int64_t count = kv.second + (self ? 1 : 0);
      if (count - mn <= maxSkew) {
        // strict '<' in Go with map order; canonical tie-break: lexicographically smallest domain
        if (count < minCount || (count == minCount && minDomain >= 0 && lexLess(kv.first, minDomain))) {
          minDomain = kv.first;
          minCount = (int)count;
        }
      }
    }
    if (minDomain < 0) return Requirement::New(key, DoesNotExist, {});
    return Requirement::New(key, In, {minDomain});
  }
  Requirement nextDomainAffinity(const Pod& pod, const Requirement& podDomains, const Requirement& nodeDomains) const {
    Requirement options = Requirement::New(key, DoesNotExist, {});
    for (auto& kv : domains) if (podDomains.Has(kv.first) && kv.second > 0) options.values.insert(kv.first);
    if (options.Len() == 0 && selects(pod)) {
      // bootstrap: first domain in (pod ∩ node), then first pod-viable domain; "first" = canonical smallest
      Requirement inter = podDomains.Intersection(nodeDomains);
      int a = -1, b = -1;
      for (auto& kv : domains) {
        if (inter.Has(kv.first) && (a < 0 || lexLess(kv.first, a))) a = kv.first;
        if (podDomains.Has(kv.first) && (b < 0 || lexLess(kv.first, b))) b = kv.first;
      }
      if (a >= 0) options.values.insert(a);
      if (b >= 0) options.values.insert(b);
    }
    return options;
  }
  Requirement nextDomainAntiAffinity(const Requirement& podDomains) const {
    Requirement options = Requirement::New(key, DoesNotExist, {});
    for (int d : emptyDomains) if (podDomains.Has(d)) options.values.insert(d);
    return options;
  }
  Requirement Get(const Pod& pod, const Requirement& podDomains, const Requirement& nodeDomains) const {
    if (type == TSpread) return nextDomainTopologySpread(pod, podDomains, nodeDomains);
    if (type == TAffinity) return nextDomainAffinity(pod, podDomains, nodeDomains);
    return nextDomainAntiAffinity(podDomains);
  }
};

static string reqsCanon(const Requirements& r) {
  string s;
  for (auto& kv : r.m) {
    s += std::to_string(kv.first) + (kv.second.complement ? "!" : "=");
    for (int v : kv.second.values) s += std::to_string(v) + ",";
    if (kv.second.gt) s += ">" + std::to_string(*kv.second.gt);
    if (kv.second.lt) s += "<" + std::to_string(*kv.second.lt);
    s += ";";
  }
  return s;
}

struct NodeView {  // what countDomains needs of a cluster node
  int name;
  std::map<int, int> labels;
};

struct Topology {
  int keyHostname;
  std::map<string, std::unique_ptr<TopologyGroup>> topologies, inverseTopologies;
  std::map<int, std::set<int>> domains;  // universe per key
  std::set<int> excludedPods;
  const vector<std::unique_ptr<Pod>>* clusterPods = nullptr;
  const vector<NodeView>* clusterNodes = nullptr;

  // scheduling.MakeTopologyNodeFilter
  static vector<Requirements> MakeTopologyNodeFilter(const Pod& p) {
    Requirements sel = NewLabelRequirements(p.nodeSelector);
    if (p.requiredTerms.empty()) return {sel};
    vector<Requirements> f;
    for (auto& term : p.requiredTerms) {
      Requirements r;
      r.Add(sel);
      r.Add(fromReqDescs(term));
      f.push_back(std::move(r));
    }
    return f;
  }
  std::unique_ptr<TopologyGroup> NewTopologyGroup(int type, int key, const Pod& pod, std::set<int> ns, const Selector& sel, int maxSkew,
                                                  optional<int> minDomains) {
    auto tg = std::make_unique<TopologyGroup>();
    tg->type = type;
    tg->key = key;
    tg->maxSkew = maxSkew;
    tg->minDomains = minDomains;
    tg->namespaces = std::move(ns);
    tg->selector = sel;
    tg->isHostname = key == keyHostname;
    auto it = domains.find(key);
    if (it != domains.end()) for (int d : it->second) { tg->domains[d] = 0; tg->emptyDomains.insert(d); }
    if (type == TSpread) { tg->hasFilter = true; tg->nodeFilter = MakeTopologyNodeFilter(pod); }
    return tg;
  }
  static string hashKey(const TopologyGroup& tg) {
    string s = std::to_string(tg.key) + "|" + std::to_string(tg.type) + "|" + std::to_string(tg.maxSkew) + "|" +
               (tg.minDomains ? std::to_string(*tg.minDomains) : "nil") + "|";
    for (int n : tg.namespaces) s += std::to_string(n) + ",";
    s += "|" + tg.selector.canon() + "|";
    vector<string> fs;
    for (auto& f : tg.nodeFilter) fs.push_back(reqsCanon(f));
    std::sort(fs.begin(), fs.end());
    for (auto& f : fs) s += f + "#";
    return s;
  }
  static std::set<int> buildNamespaceList(const Pod& p, const vector<int>& namespaces) {
    if (namespaces.empty()) return {p.ns};
    return std::set<int>(namespaces.begin(), namespaces.end());
  }
  // Topology.countDomains: count bound pods per domain
  void countDomains(TopologyGroup& tg) {
    if (!clusterPods) return;
    for (auto& bp : *clusterPods) {
      if (!tg.namespaces.count(bp->ns) || !tg.selector.Matches(bp->labels)) continue;
      if (excludedPods.count(bp->uid)) continue;
      if (bp->node < 0) continue;
      const NodeView& node = (*clusterNodes)[bp->node];
      int domain;
      auto it = node.labels.find(tg.key);
      if (it != node.labels.end()) domain = it->second;
      else if (tg.key == keyHostname) domain = node.name;
      else continue;
      if (tg.hasFilter && !tg.nodeFilter.empty()) {
        vector<std::pair<int, int>> ls(node.labels.begin(), node.labels.end());
        if (!tg.filterMatches(NewLabelRequirements(ls), nullptr)) continue;
      }
      tg.Record(domain);
    }
  }
  void updateInverseAntiAffinity(const Pod& pod, const std::map<int, int>* nodeLabels) {
    for (auto& term : pod.antiReq) {
      auto tg = NewTopologyGroup(TAntiAffinity, term.key, pod, buildNamespaceList(pod, term.namespaces), term.sel, INT_MAX, {});
      string h = hashKey(*tg);
      auto it = inverseTopologies.find(h);
      TopologyGroup* g;
      if (it == inverseTopologies.end()) { g = tg.get(); inverseTopologies.emplace(h, std::move(tg)); } else g = it->second.get();
      if (nodeLabels) {
        auto d = nodeLabels->find(g->key);
        if (d != nodeLabels->end()) g->Record(d->second);
      }
      g->owners.insert(pod.uid);
    }
  }
  // Topology.updateInverseAffinities: bound pods with required anti-affinity
  void updateInverseAffinities() {
    if (!clusterPods) return;
    for (auto& bp : *clusterPods) {
      if (bp->antiReq.empty() || excludedPods.count(bp->uid) || bp->node < 0) continue;
      updateInverseAntiAffinity(*bp, &(*clusterNodes)[bp->node].labels);
    }
  }
  // Topology.Update
  void Update(const Pod& p) {
    for (auto& kv : topologies) kv.second->owners.erase(p.uid);
    if (!p.antiReq.empty() || !p.antiPref.empty()) updateInverseAntiAffinity(p, nullptr);
    vector<std::unique_ptr<TopologyGroup>> groups;
    for (auto& cs : p.spreads) groups.push_back(NewTopologyGroup(TSpread, cs.key, p, {p.ns}, cs.sel, cs.maxSkew, cs.minDomains));
    for (auto& t : p.affReq) groups.push_back(NewTopologyGroup(TAffinity, t.key, p, buildNamespaceList(p, t.namespaces), t.sel, INT_MAX, {}));
    for (auto& t : p.affPref) groups.push_back(NewTopologyGroup(TAffinity, t.key, p, buildNamespaceList(p, t.namespaces), t.sel, INT_MAX, {}));
    for (auto& t : p.antiReq) groups.push_back(NewTopologyGroup(TAntiAffinity, t.key, p, buildNamespaceList(p, t.namespaces), t.sel, INT_MAX, {}));
    for (auto& t : p.antiPref) groups.push_back(NewTopologyGroup(TAntiAffinity, t.key, p, buildNamespaceList(p, t.namespaces), t.sel, INT_MAX, {}));
    for (auto& tg : groups) {
      string h = hashKey(*tg);
      auto it = topologies.find(h);
      TopologyGroup* g;
      if (it == topologies.end()) {
        countDomains(*tg);
        g = tg.get();
        topologies.emplace(h, std::move(tg));
      } else g = it->second.get();
      g->owners.insert(p.uid);
    }
  }
  void Register(int key, int domain) {
    for (auto& kv : topologies) if (kv.second->key == key) kv.second->Register(domain);
    for (auto& kv : inverseTopologies) if (kv.second->key == key) kv.second->Register(domain);
  }
  void Unregister(int key, int domain) {
    for (auto& kv : topologies) if (kv.second->key == key) kv.second->Unregister(domain);
    for (auto& kv : inverseTopologies) if (kv.second->key == key) kv.second->Unregister(domain);
  }
  // Topology.AddRequirements; returns false on a topology error
  bool AddRequirements(const Requirements& podReqs, const Requirements& nodeReqs, const Pod& p, const std::set<int>* allowUndefined,
                       Requirements* out) {
    Requirements requirements = nodeReqs;
    auto apply = [&](TopologyGroup& tg) {
      Requirement podDomains = podReqs.Has(tg.key) ? podReqs.Get(tg.key) : Requirement::New(tg.key, Exists, {});
      Requirement nodeDomains = nodeReqs.Has(tg.key) ? nodeReqs.Get(tg.key) : Requirement::New(tg.key, Exists, {});
      Requirement d = tg.Get(p, podDomains, nodeDomains);
      if (d.Len() == 0) return false;
      requirements.Add(d);
      return true;
    };
    for (auto& kv : topologies) if (kv.second->owners.count(p.uid)) if (!apply(*kv.second)) return false;
    for (auto& kv : inverseTopologies) if (kv.second->Counts(p, nodeReqs, allowUndefined)) if (!apply(*kv.second)) return false;
    *out = std::move(requirements);
    return true;
  }
  // Topology.Record
  void Record(const Pod& p, const Requirements& reqs, const std::set<int>* allowUndefined) {
    for (auto& kv : topologies) {
      TopologyGroup& tc = *kv.second;
      if (!tc.Counts(p, reqs, allowUndefined)) continue;
      Requirement domains = reqs.Get(tc.key);
      if (tc.type == TAntiAffinity) {
        for (int v : domains.values) tc.Record(v);
      } else if (domains.Len() == 1) {
        tc.Record(*domains.values.begin());
      }
    }
    for (auto& kv : inverseTopologies) {
      TopologyGroup& tc = *kv.second;
      if (tc.owners.count(p.uid)) for (int v : reqs.Get(tc.key).values) tc.Record(v);
    }
  }
};

// ------------------------------------------------------------------ scheduler objects
struct NodePool {
  int index, name, weight;
  Requirements requirements;  // template: spec reqs + labels + nodepool name
  vector<Taint> taints;
  RL limits;
  bool hasLimits;
  RL daemonOverhead;
  Catalog* catalog;
  vector<InstanceType*> instanceTypes;
};

struct Counters {
  uint64_t visited_existing = 0, visited_claims = 0, visited_claim_types = 0, new_claim_scans = 0, relaxations = 0;
};

struct SchedCtx {
  int keyHostname, keyZone, keyCT, keyZoneID, keyNodePool, keyInstanceType;
  int resCPU, resMemory, resPods;
  int valSpot, valOnDemand;
  std::set<int> wellKnown;
  Counters cnt;
};

static bool offeringCompatible(const Requirements& reqs, const Offering& o, const SchedCtx& C) {
  return reqs.Compatible(o.reqs, &C.wellKnown);
}
// scheduling.filterInstanceTypesByRequirements (no short circuit across the three predicates upstream;
// results identical)
static vector<InstanceType*> filterInstanceTypesByRequirements(const vector<InstanceType*>& its, const Requirements& reqs, const RL& requests,
                                                               const SchedCtx& C);
static bool SatisfiesMinValues(const vector<InstanceType*>& its, const Requirements& reqs, int* minNeeded);

struct NodeClaim {
  NodePool* pool;
  Requirements requirements;
  vector<InstanceType*> options;
  RL requests;
  RL daemonResources;
  vector<Pod*> pods;
  int hostname;
  HostPortUsage hostPorts;
};

struct ExistingNode {
  int index;  // cluster node index
  int name;
  bool initialized;
  RL available;
  vector<Taint> taints;
  RL requests;
  Requirements requirements;
  vector<Pod*> pods;
  HostPortUsage hostPorts;
  // scheduling.VolumeUsage: volumes in use per CSI driver, and the CSINode's attachable count (drivers without one: unlimited)
  std::map<int, std::set<int>> volumes;
  std::map<int, int64_t> volumeLimits;
  // (PARITY UNPINNED: the file is not in the reference checkout.  Restated as upstream's v.volumes.Union(vols) walk: every
  // driver of the node OR the pod is checked, so a node already over a driver's limit refuses pods without volumes too.)
  bool ExceedsLimits(const Pod& pod) const {
    std::map<int, std::set<int>> u = volumes;
    for (auto& kv : pod.volumes) u[kv.first].insert(kv.second.begin(), kv.second.end());
    for (auto& kv : u) {
      auto lim = volumeLimits.find(kv.first);
      if (lim != volumeLimits.end() && (int64_t)kv.second.size() > lim->second) return true;
    }
    return false;
  }
  void AddVolumes(const Pod& pod) { for (auto& kv : pod.volumes) volumes[kv.first].insert(kv.second.begin(), kv.second.end()); }
};

static vector<InstanceType*> filterInstanceTypesByRequirements(const vector<InstanceType*>& its, const Requirements& reqs, const RL& requests,
                                                               const SchedCtx& C) {
  vector<InstanceType*> remaining;
  for (InstanceType* it : its) {
    bool compat = it->reqs.Intersects(reqs);
    bool fits = Fits(requests, it->allocatable);
    bool hasOffering = false;
    for (auto& o : it->offerings) {
      if (!o.available) continue;
      if (offeringCompatible(reqs, o, C)) { hasOffering = true; break; }
    }
    if (compat && fits && hasOffering) remaining.push_back(it);
  }
  if (reqs.HasMinValues()) {
    int n;
    if (!SatisfiesMinValues(remaining, reqs, &n)) remaining.clear();
  }
  return remaining;
}

// InstanceTypes.SatisfiesMinValues
static bool SatisfiesMinValues(const vector<InstanceType*>& its, const Requirements& reqs, int* minNeeded) {
  if (!reqs.HasMinValues()) { *minNeeded = 0; return true; }
  std::map<int, std::set<int>> valuesForKey;
  bool incompatible = false;
  for (size_t i = 0; i < its.size(); ++i) {
    for (auto& kv : reqs.m) {
      if (!kv.second.minValues) continue;
      auto& s = valuesForKey[kv.first];
      Requirement tr = its[i]->reqs.Get(kv.first);
      s.insert(tr.values.begin(), tr.values.end());
    }
    incompatible = false;
    for (auto& kv : valuesForKey) {
      if ((int)kv.second.size() < *reqs.Get(kv.first).minValues) { incompatible = true; break; }
    }
    if (!incompatible) { *minNeeded = (int)i + 1; return true; }
  }
  *minNeeded = (int)its.size();
  // empty list with minValues: valuesForKey empty => incompatibleKey "" => upstream returns (len, nil)
  if (its.empty()) return true;
  return !incompatible;
}

// InstanceTypes.OrderByPrice
static double cheapestCompatiblePrice(const InstanceType* it, const Requirements& reqs, const SchedCtx& C) {
  double p = std::numeric_limits<double>::max();
  bool any = false;
  for (auto& o : it->offerings) {
    if (!o.available || !offeringCompatible(reqs, o, C)) continue;
    if (!any || o.price < p) { p = o.price; any = true; }
  }
  return any ? p : std::numeric_limits<double>::max();
}
static vector<InstanceType*> OrderByPrice(vector<InstanceType*> its, const Requirements& reqs, const SchedCtx& C) {
  vector<std::pair<double, InstanceType*>> keyed;
  for (auto* it : its) keyed.push_back({cheapestCompatiblePrice(it, reqs, C), it});
  std::stable_sort(keyed.begin(), keyed.end(), [&](auto& a, auto& b) {
    if (a.first == b.first) return G->s(a.second->name) < G->s(b.second->name);
    return a.first < b.first;
  });
  vector<InstanceType*> out;
  for (auto& kv : keyed) out.push_back(kv.second);
  return out;
}
// InstanceTypes.Truncate; returns false when minValues breaks (caller keeps `its`)
static bool Truncate(const vector<InstanceType*>& its, const Requirements& reqs, int maxItems, const SchedCtx& C, vector<InstanceType*>* out) {
  vector<InstanceType*> ordered = OrderByPrice(its, reqs, C);
  vector<InstanceType*> trunc(ordered.begin(), ordered.begin() + std::min<size_t>(ordered.size(), (size_t)std::max(maxItems, 0)));
  if (reqs.HasMinValues()) {
    int n;
    if (!SatisfiesMinValues(trunc, reqs, &n)) { *out = ordered; return false; }
  }
  *out = std::move(trunc);
  return true;
}

// ------------------------------------------------------------------ Scheduler (Appendix A.3-A.6)
struct Scheduler {
  SchedCtx& C;
  Topology topo;
  vector<NodePool*> templates;  // weight order
  std::map<int, RL> remainingResources;  // nodepool index -> remaining
  vector<std::unique_ptr<ExistingNode>> existingNodes;
  vector<std::unique_ptr<NodeClaim>> newNodeClaims;
  bool toleratePreferNoSchedule = false;
  int64_t nodeID = 0;
  std::map<const Pod*, uint32_t> reasons;
  explicit Scheduler(SchedCtx& c) : C(c) {}

  // NodeClaim.Add
  bool claimAdd(NodeClaim& n, Pod& pod, bool inflight, uint32_t* reason) {
    if (!Tolerates(n.pool->taints, pod)) { *reason |= KG_REASON_TAINTS; return false; }
    if (n.hostPorts.Conflicts(pod)) { *reason |= KG_REASON_REQUIREMENTS; return false; }
    Requirements claimReqs = n.requirements;
    Requirements podReqs = NewPodRequirements(pod, false);
    if (!claimReqs.Compatible(podReqs, &C.wellKnown)) { *reason |= KG_REASON_REQUIREMENTS; return false; }
    claimReqs.Add(podReqs);
    Requirements strict = pod.preferredTerms.empty() ? podReqs : NewPodRequirements(pod, true);
    Requirements topoReqs;
    if (!topo.AddRequirements(strict, claimReqs, pod, &C.wellKnown, &topoReqs)) { *reason |= KG_REASON_TOPOLOGY; return false; }
    if (!claimReqs.Compatible(topoReqs, &C.wellKnown)) { *reason |= KG_REASON_TOPOLOGY; return false; }
    claimReqs.Add(topoReqs);
    RL requests = Merge(n.requests, pod.requests);
    C.cnt.visited_claim_types += n.options.size();
    vector<InstanceType*> remaining = filterInstanceTypesByRequirements(n.options, claimReqs, requests, C);
    if (remaining.empty()) { *reason |= KG_REASON_NO_TYPE; return false; }
    n.pods.push_back(&pod);
    n.options = std::move(remaining);
    n.requests = std::move(requests);
    n.requirements = std::move(claimReqs);
    topo.Record(pod, n.requirements, &C.wellKnown);
    n.hostPorts.Add(pod);
    (void)inflight;
    return true;
  }
  // ExistingNode.Add
  bool nodeAdd(ExistingNode& n, Pod& pod, uint32_t* reason) {
    if (!Tolerates(n.taints, pod)) { *reason |= KG_REASON_TAINTS; return false; }
    if (n.ExceedsLimits(pod)) { *reason |= KG_REASON_RESOURCES; return false; }  // "would exceed node volume limits"
    if (n.hostPorts.Conflicts(pod)) { *reason |= KG_REASON_REQUIREMENTS; return false; }
    RL requests = Merge(n.requests, pod.requests);
    if (!Fits(requests, n.available)) { *reason |= KG_REASON_RESOURCES; return false; }
    Requirements nodeReqs = n.requirements;
    Requirements podReqs = NewPodRequirements(pod, false);
    if (!nodeReqs.Compatible(podReqs, nullptr)) { *reason |= KG_REASON_REQUIREMENTS; return false; }
    nodeReqs.Add(podReqs);
    Requirements strict = pod.preferredTerms.empty() ? podReqs : NewPodRequirements(pod, true);
    Requirements topoReqs;
    if (!topo.AddRequirements(strict, nodeReqs, pod, nullptr, &topoReqs)) { *reason |= KG_REASON_TOPOLOGY; return false; }
    if (!nodeReqs.Compatible(topoReqs, nullptr)) { *reason |= KG_REASON_TOPOLOGY; return false; }
    nodeReqs.Add(topoReqs);
    n.pods.push_back(&pod);
    n.requests = std::move(requests);
    n.requirements = std::move(nodeReqs);
    topo.Record(pod, n.requirements, nullptr);
    n.hostPorts.Add(pod);
    n.AddVolumes(pod);
    return true;
  }
  // Scheduler.add
  bool add(Pod& pod, uint32_t* reason) {
    for (auto& n : existingNodes) {
      C.cnt.visited_existing++;
      if (nodeAdd(*n, pod, reason)) return true;
    }
    // sort.Slice by len(Pods) ascending -- canonical: stable
    std::stable_sort(newNodeClaims.begin(), newNodeClaims.end(), [](auto& a, auto& b) { return a->pods.size() < b->pods.size(); });
    for (auto& c : newNodeClaims) {
      C.cnt.visited_claims++;
      if (claimAdd(*c, pod, true, reason)) return true;
    }
    for (NodePool* tmpl : templates) {
      vector<InstanceType*> its = tmpl->instanceTypes;
      auto rem = remainingResources.find(tmpl->index);
      if (rem != remainingResources.end()) {
        // filterByRemainingResources
        vector<InstanceType*> f;
        for (auto* it : its) {
          bool viable = true;
          for (auto& kv : rem->second) if (rlGet(it->capacity, kv.first) > kv.second) viable = false;
          if (viable) f.push_back(it);
        }
        its = std::move(f);
        if (its.empty()) { *reason |= KG_REASON_LIMITS; continue; }
      }
      // NewNodeClaim
      auto nc = std::make_unique<NodeClaim>();
      nc->pool = tmpl;
      char buf[64];
      snprintf(buf, sizeof buf, "hostname-placeholder-%04lld", (long long)++nodeID);
      nc->hostname = G->id(buf);
      topo.Register(C.keyHostname, nc->hostname);
      nc->requirements = tmpl->requirements;
      nc->requirements.Add(Requirement::New(C.keyHostname, In, {nc->hostname}));
      nc->options = its;
      nc->requests = tmpl->daemonOverhead;
      nc->daemonResources = tmpl->daemonOverhead;
      C.cnt.new_claim_scans++;
      if (!claimAdd(*nc, pod, false, reason)) {
        topo.Unregister(C.keyHostname, nc->hostname);  // NodeClaim.Destroy
        continue;
      }
      // subtractMax
      if (rem != remainingResources.end() && !nc->options.empty()) {
        RL mx;
        for (auto* it : nc->options) for (auto& kv : it->capacity) mx[kv.first] = std::max(rlGet(mx, kv.first), kv.second);
        RL res;
        for (auto& kv : rem->second) res[kv.first] = kv.second - rlGet(mx, kv.first);
        rem->second = std::move(res);
      }
      newNodeClaims.push_back(std::move(nc));
      return true;
    }
    return false;
  }

  // Preferences.Relax: first applicable relaxation, one per call
  bool Relax(Pod& pod) {
    // removeRequiredNodeAffinityTerm
    if (pod.requiredTerms.size() > 1) { pod.requiredTerms.erase(pod.requiredTerms.begin()); return true; }
    auto dropHeaviest = [](auto& terms) {
      if (terms.empty()) return false;
      std::stable_sort(terms.begin(), terms.end(), [](auto& a, auto& b) { return a.weight > b.weight; });
      terms.erase(terms.begin());
      return true;
    };
    if (dropHeaviest(pod.affPref)) return true;   // removePreferredPodAffinityTerm
    if (dropHeaviest(pod.antiPref)) return true;  // removePreferredPodAntiAffinityTerm
    if (dropHeaviest(pod.preferredTerms)) return true;  // removePreferredNodeAffinityTerm
    // removeTopologySpreadScheduleAnyway: swap-remove the first ScheduleAnyway constraint
    for (size_t i = 0; i < pod.spreads.size(); ++i) {
      if (pod.spreads[i].whenUnsat == KG_SCHEDULE_ANYWAY) {
        pod.spreads[i] = pod.spreads.back();
        pod.spreads.pop_back();
        return true;
      }
    }
    if (toleratePreferNoSchedule) {
      // toleratePreferNoScheduleTaints
      for (auto& t : pod.tolerations)
        if (t.key < 0 && t.op == KG_TOL_EXISTS && t.effect == KG_EFFECT_PREFER_NO_SCHEDULE) return false;
      pod.tolerations.push_back({-1, G->id(""), KG_TOL_EXISTS, KG_EFFECT_PREFER_NO_SCHEDULE});
      return true;
    }
    return false;
  }

  // Scheduler.Solve with scheduling.Queue
  void Solve(vector<Pod*> pods, std::map<const Pod*, uint32_t>* podErrors) {
    // byCPUAndMemoryDescending; total order so stability is moot
    std::stable_sort(pods.begin(), pods.end(), [&](Pod* a, Pod* b) {
      int64_t ac = rlGet(a->requests, C.resCPU), bc = rlGet(b->requests, C.resCPU);
      if (ac != bc) return ac > bc;
      int64_t am = rlGet(a->requests, C.resMemory), bm = rlGet(b->requests, C.resMemory);
      if (am != bm) return am > bm;
      if (a->creation != b->creation) return a->creation < b->creation;
      return G->s(a->uid) < G->s(b->uid);
    });
    std::vector<Pod*> q(pods.begin(), pods.end());
    size_t head = 0;
    std::map<int, size_t> lastLen;
    while (true) {
      size_t len = q.size() - head;
      if (len == 0) break;
      Pod* pod = q[head];
      auto ll = lastLen.find(pod->uid);
      if (ll != lastLen.end() && ll->second == len) break;
      head++;
      uint32_t reason = 0;
      if (add(*pod, &reason)) { podErrors->erase(pod); continue; }
      (*podErrors)[pod] = reason ? reason : KG_REASON_NO_TYPE;
      bool relaxed = Relax(*pod);
      q.push_back(pod);
      if (relaxed) {
        lastLen.clear();
        C.cnt.relaxations++;
        topo.Update(*pod);
      } else {
        lastLen[pod->uid] = q.size() - head;
      }
    }
    for (auto& c : newNodeClaims) c->requirements.m.erase(C.keyHostname);  // FinalizeScheduling
  }
};

// ------------------------------------------------------------------ problem assembly (NewScheduler)
struct Problem {
  Intern in;
  SchedCtx C;
  vector<std::unique_ptr<Catalog>> catalogs;
  vector<std::unique_ptr<NodePool>> pools;
  vector<std::unique_ptr<Pod>> pending, bound;
  struct CNode {
    int name;
    std::map<int, int> labels;
    vector<Taint> taints;
    RL available, capacity, daemonRequests;
    bool initialized, managed, deleting;
    vector<Pod*> pods;
    std::map<int, int64_t> volumeLimits;
  };
  vector<CNode> nodes;
  vector<NodeView> nodeViews;
  int maxInstanceTypes = 60;
};

static vector<Taint> parseTaints(Parser& P, const kg_pools& p, kg_range r) {
  vector<Taint> out;
  for (auto& t : kgdesc::taints(p, r)) out.push_back({P.sid(p, t.key), t.value == KG_NONE ? P.in.id("") : P.sid(p, t.value), t.effect});
  return out;
}

static std::unique_ptr<Problem> buildProblem(const kg_catalog_desc* const* catalogs, uint32_t n_catalogs, const kg_cluster_desc* cluster,
                                             const kg_pods_desc* pods, const kg_nodepools_desc* nps) {
  auto pr = std::make_unique<Problem>();
  G = &pr->in;
  Intern& in = pr->in;
  SchedCtx& C = pr->C;
  C.keyHostname = in.id("kubernetes.io/hostname");
  C.keyZone = in.id("topology.kubernetes.io/zone");
  C.keyCT = in.id("karpenter.sh/capacity-type");
  C.keyZoneID = in.id("topology.k8s.aws/zone-id");
  C.keyNodePool = in.id("karpenter.sh/nodepool");
  C.keyInstanceType = in.id("node.kubernetes.io/instance-type");
  C.resCPU = in.id("cpu");
  C.resMemory = in.id("memory");
  C.resPods = in.id("pods");
  C.valSpot = in.id("spot");
  C.valOnDemand = in.id("on-demand");
  Parser P{in};
  for (uint32_t i = 0; i < n_catalogs; ++i) pr->catalogs.push_back(parseCatalog(in, catalogs[i], C.keyCT, C.keyZone, C.keyZoneID));
  check(nps != nullptr, "nodepools required");
  for (uint32_t v : kgdesc::values(nps->pools, nps->well_known_labels)) C.wellKnown.insert(P.sid(nps->pools, v));
  pr->maxInstanceTypes = nps->max_instance_types;
  for (uint32_t i = 0; i < nps->n_nodepools; ++i) {
    const kg_nodepool& q = nps->nodepools[i];
    auto np = std::make_unique<NodePool>();
    np->index = (int)i;
    np->name = P.sid(nps->pools, q.name);
    np->weight = q.weight;
    // NewNodeClaimTemplate: spec requirements (with minValues) + labels + nodepool name
    np->requirements = P.reqs(nps->pools, q.requirements);
    for (auto& l : kgdesc::labels(nps->pools, q.labels)) np->requirements.Add(Requirement::New(P.sid(nps->pools, l.key), In, {P.sid(nps->pools, l.value)}));
    np->requirements.Add(Requirement::New(C.keyNodePool, In, {np->name}));
    np->taints = parseTaints(P, nps->pools, q.taints);
    np->hasLimits = q.has_limits != 0;
    np->limits = P.rl(nps->pools, q.limits);
    np->daemonOverhead = P.rl(nps->pools, q.daemon_overhead);
    check(q.catalog < n_catalogs, "nodepool catalog index");
    np->catalog = pr->catalogs[q.catalog].get();
    for (auto& t : np->catalog->types) np->instanceTypes.push_back(t.get());
    pr->pools.push_back(std::move(np));
  }
  pr->pending = parsePods(in, pods, C.resPods, true);
  if (cluster) {
    pr->bound = parsePods(in, &cluster->pods, C.resPods, false);
    const kg_pools& cp = cluster->pods.pools;
    for (uint32_t i = 0; i < cluster->n_nodes; ++i) {
      const kg_node& n = cluster->nodes[i];
      Problem::CNode cn;
      cn.name = P.sid(cp, n.name);
      for (auto& l : kgdesc::labels(cp, n.labels)) cn.labels[P.sid(cp, l.key)] = P.sid(cp, l.value);
      cn.taints = parseTaints(P, cp, n.taints);
      cn.available = P.rl(cp, n.available);
      cn.capacity = P.rl(cp, n.capacity);
      cn.daemonRequests = P.rl(cp, n.daemon_requests);
      cn.initialized = n.initialized;
      cn.managed = n.managed;
      cn.deleting = n.deleting;
      for (auto& q : kgdesc::quantities(cp, n.volume_limits)) cn.volumeLimits[P.sid(cp, q.name)] = q.milli / 1000;
      pr->nodes.push_back(std::move(cn));
    }
    for (auto& bp : pr->bound) {
      check(bp->node >= 0 && bp->node < (int)pr->nodes.size(), "bound pod node index");
      pr->nodes[bp->node].pods.push_back(bp.get());
    }
    for (auto& n : pr->nodes) pr->nodeViews.push_back({n.name, n.labels});
  }
  return pr;
}

// provisioner.NewScheduler + scheduling.NewScheduler for a set of pods and state nodes
static std::unique_ptr<Scheduler> newScheduler(Problem& pr, const vector<Pod*>& pods, const std::set<int>& excludedNodes) {
  SchedCtx& C = pr.C;
  auto s = std::make_unique<Scheduler>(C);
  // nodepools ordered by weight desc (canonical: stable)
  vector<NodePool*> order;
  for (auto& np : pr.pools) order.push_back(np.get());
  std::stable_sort(order.begin(), order.end(), [](NodePool* a, NodePool* b) { return a->weight > b->weight; });
  s->templates = order;
  // topology domain universe
  std::map<int, std::set<int>> domains;
  for (NodePool* np : order) {
    for (InstanceType* it : np->instanceTypes) {
      Requirements r = np->requirements;  // spec reqs + labels (+ nodepool key, harmless superset as upstream adds it to the template)
      r.Add(it->reqs);
      for (auto& kv : r.m) domains[kv.first].insert(kv.second.values.begin(), kv.second.values.end());
    }
    for (auto& kv : np->requirements.m)
      if (kv.second.Operator() == In) domains[kv.first].insert(kv.second.values.begin(), kv.second.values.end());
    for (auto& t : np->taints) if (t.effect == KG_EFFECT_PREFER_NO_SCHEDULE) s->toleratePreferNoSchedule = true;
    if (np->hasLimits) s->remainingResources[np->index] = np->limits;
  }
  s->topo.keyHostname = C.keyHostname;
  s->topo.domains = std::move(domains);
  s->topo.clusterPods = &pr.bound;
  s->topo.clusterNodes = &pr.nodeViews;
  for (Pod* p : pods) s->topo.excludedPods.insert(p->uid);
  s->topo.updateInverseAffinities();
  for (Pod* p : pods) s->topo.Update(*p);
  // calculateExistingNodeClaims
  for (size_t i = 0; i < pr.nodes.size(); ++i) {
    if (excludedNodes.count((int)i) || pr.nodes[i].deleting) continue;
    auto& cn = pr.nodes[i];
    auto en = std::make_unique<ExistingNode>();
    en->index = (int)i;
    en->name = cn.name;
    en->initialized = cn.initialized;
    en->available = cn.available;
    en->taints = cn.taints;
    en->requests = cn.daemonRequests;
    vector<std::pair<int, int>> ls(cn.labels.begin(), cn.labels.end());
    en->requirements = NewLabelRequirements(ls);
    int hostname = cn.labels.count(C.keyHostname) ? cn.labels[C.keyHostname] : cn.name;
    en->requirements.Add(Requirement::New(C.keyHostname, In, {hostname}));
    s->topo.Register(C.keyHostname, hostname);
    for (Pod* bp : cn.pods) { en->hostPorts.Add(*bp); en->AddVolumes(*bp); }
    en->volumeLimits = cn.volumeLimits;
    auto npl = cn.labels.find(C.keyNodePool);
    if (npl != cn.labels.end()) {
      for (auto& np : pr.pools) {
        if (np->name == npl->second) {
          auto rem = s->remainingResources.find(np->index);
          if (rem != s->remainingResources.end()) rem->second = Subtract(rem->second, cn.capacity);
        }
      }
    }
    s->existingNodes.push_back(std::move(en));
  }
  std::stable_sort(s->existingNodes.begin(), s->existingNodes.end(), [](auto& a, auto& b) {
    if (a->initialized != b->initialized) return a->initialized;
    return G->s(a->name) < G->s(b->name);
  });
  return s;
}

// Results.TruncateInstanceTypes; truncFailed[i] marks claims dropped by the reference
static void truncateAll(Scheduler& s, int maxTypes, vector<uint8_t>* truncFailed, std::map<const Pod*, uint32_t>* podErrors) {
  truncFailed->assign(s.newNodeClaims.size(), 0);
  if (maxTypes <= 0) return;
  for (size_t i = 0; i < s.newNodeClaims.size(); ++i) {
    auto& c = *s.newNodeClaims[i];
    vector<InstanceType*> out;
    bool ok = Truncate(c.options, c.requirements, maxTypes, s.C, &out);
    c.options = std::move(out);
    if (!ok) {
      (*truncFailed)[i] = 1;
      for (Pod* p : c.pods) (*podErrors)[p] = KG_REASON_MIN_VALUES;
    }
  }
}

static void writeReqs(kgdesc::ResultWriter& W, const Requirements& reqs, kg_range* out) {
  out->begin = W.r->needed_reqs;
  out->count = 0;
  // keys sorted by string for a canonical output order
  vector<int> keys;
  for (auto& kv : reqs.m) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end(), lexLess);
  vector<kg_req_out> tmp;
  for (int k : keys) {
    const Requirement& r = reqs.m.at(k);
    kg_req_out q{};
    q.key = W.add_string(G->s(k));
    q.complement = r.complement;
    q.has_gt = r.gt.has_value();
    q.has_lt = r.lt.has_value();
    q.gt = r.gt.value_or(0);
    q.lt = r.lt.value_or(0);
    q.min_values = r.minValues.value_or(-1);
    q.values.begin = W.r->needed_value_ids;
    q.values.count = 0;
    for (int v : r.SortedValues()) { W.add_value(W.add_string(G->s(v))); q.values.count++; }
    tmp.push_back(q);
  }
  out->begin = W.r->needed_reqs;
  for (auto& q : tmp) { W.add_req(q); out->count++; }
}

static void fillCounters(const SchedCtx& C, kg_results* out) {
  out->visited_existing = C.cnt.visited_existing;
  out->visited_claims = C.cnt.visited_claims;
  out->visited_claim_types = C.cnt.visited_claim_types;
  out->new_claim_scans = C.cnt.new_claim_scans;
  out->relaxations = C.cnt.relaxations;
  out->algorithmic_bytes = 212.0 * C.cnt.visited_existing + 400.0 * (C.cnt.visited_claims + C.cnt.new_claim_scans) + 336.0 * C.cnt.visited_claim_types;
}


// ------------------------------------------------------------------ disruption / consolidation (Appendix A.9)
// Offerings.WorstLaunchPrice over the AVAILABLE offerings: spot is preferred at launch, so the worst price is the
// most expensive compatible spot offering when the requirements allow spot, else the most expensive on-demand one.
static double worstLaunchPrice(const InstanceType* it, const Requirements& reqs, const SchedCtx& C) {
  const Requirement ct = reqs.Get(C.keyCT);
  for (int want : {C.valSpot, C.valOnDemand}) {
    if (!ct.Has(want)) continue;
    bool any = false;
    double worst = 0;
    for (auto& o : it->offerings) {
      if (!o.available || o.ct != want || !offeringCompatible(reqs, o, C)) continue;
      if (!any || o.price > worst) { worst = o.price; any = true; }
    }
    if (any) return worst;
  }
  return std::numeric_limits<double>::max();
}

// NodeClaim.RemoveInstanceTypeOptionsByPriceAndMinValues; false = minValues no longer satisfiable
static bool removeByPriceAndMinValues(vector<InstanceType*>* options, const Requirements& reqs, double maxPrice, const SchedCtx& C) {
  vector<InstanceType*> kept;
  for (auto* it : *options) if (worstLaunchPrice(it, reqs, C) < maxPrice) kept.push_back(it);
  *options = std::move(kept);
  int n;
  return SatisfiesMinValues(*options, reqs, &n);
}

struct Outcome {
  int decision = KG_DECISION_NOOP;
  int nodepool = 0;
  vector<InstanceType*> options;  // price ordered
  Requirements reqs;
  double candidatePrice = 0;
  bool spotOnly = false;
};

static const int kMinTypesSpotToSpot = 15;  // MinInstanceTypesForSpotToSpotConsolidation

// candidate price: cheapest offering of the node's instance type compatible with the node's labels
static bool candidatePrice(Problem& pr, int node, double* price) {
  SchedCtx& C = pr.C;
  auto& cn = pr.nodes[node];
  auto itl = cn.labels.find(C.keyInstanceType);
  if (itl == cn.labels.end()) return false;
  // NewCandidate: instanceTypeMap[nodePool.Name][instance-type label] -- the type as ITS NodePool's catalog lists it
  const InstanceType* ty = nullptr;
  const Catalog* cat = pr.catalogs.size() == 1 ? pr.catalogs[0].get() : nullptr;
  auto npl = cn.labels.find(C.keyNodePool);
  if (npl != cn.labels.end())
    for (auto& np : pr.pools) if (np->name == npl->second) cat = np->catalog;
  if (!cat) return false;
  for (auto& t : cat->types) if (t->name == itl->second) { ty = t.get(); break; }
  if (!ty) return false;
  vector<std::pair<int, int>> ls(cn.labels.begin(), cn.labels.end());
  Requirements nodeReqs = NewLabelRequirements(ls);
  bool any = false;
  double best = 0;
  for (auto& o : ty->offerings) {
    if (!offeringCompatible(nodeReqs, o, C)) continue;
    if (!any || o.price < best) { best = o.price; any = true; }
  }
  if (!any) return false;
  *price = best;
  return true;
}

// disruption.SimulateScheduling + consolidation.computeConsolidation for one set of candidate nodes
struct SimRaw {  // what validation needs of SimulateScheduling
  bool allScheduled = false;
  int nClaims = 0;
  vector<int> types;  // options of the one NodeClaim after TruncateInstanceTypes
};
static Outcome computeConsolidation(Problem& pr, const vector<int>& cands, bool spotToSpotEnabled, uint64_t* sims, SimRaw* raw = nullptr) {
  SchedCtx& C = pr.C;
  Outcome out;
  (*sims)++;
  std::set<int> excluded(cands.begin(), cands.end());
  vector<Pod*> pods;
  std::set<const Pod*> nonPending;
  for (auto& p : pr.pending) pods.push_back(p.get());
  for (size_t i = 0; i < pr.nodes.size(); ++i)
    if (pr.nodes[i].deleting && !excluded.count((int)i))
      for (Pod* bp : pr.nodes[i].pods) if (bp->reschedulable) { pods.push_back(bp); nonPending.insert(bp); }
  for (int n : cands)
    for (Pod* bp : pr.nodes[n].pods) if (bp->reschedulable) { pods.push_back(bp); nonPending.insert(bp); }
  // each simulation relaxes its own copies of the pods
  vector<std::unique_ptr<Pod>> copies;
  vector<Pod*> work;
  std::map<const Pod*, const Pod*> origin;
  for (Pod* p : pods) { copies.push_back(std::make_unique<Pod>(*p)); work.push_back(copies.back().get()); origin[copies.back().get()] = p; }
  auto s = newScheduler(pr, work, excluded);
  std::map<const Pod*, uint32_t> errs;
  s->Solve(work, &errs);
  vector<uint8_t> truncFailed;
  truncateAll(*s, pr.maxInstanceTypes, &truncFailed, &errs);
  // pods that would land on a node that is not initialised yet count as errors
  for (auto& n : s->existingNodes) if (!n->initialized) for (Pod* p : n->pods) errs[p] = KG_REASON_UNINITIALIZED_NODE;
  for (auto& kv : errs) if (nonPending.count(origin[kv.first])) return out;  // !AllNonPendingPodsScheduled
  vector<NodeClaim*> claims;
  for (size_t i = 0; i < s->newNodeClaims.size(); ++i) if (!truncFailed[i]) claims.push_back(s->newNodeClaims[i].get());
  if (raw) {
    raw->allScheduled = true;
    raw->nClaims = (int)claims.size();
    if (claims.size() == 1) for (auto* it : claims[0]->options) raw->types.push_back(it->index);
  }
  if (claims.empty()) { out.decision = KG_DECISION_DELETE; return out; }
  if (claims.size() != 1) return out;
  double price = 0;
  bool allSpot = true;
  for (int n : cands) {
    double p;
    if (!candidatePrice(pr, n, &p)) return out;
    price += p;
    auto ct = pr.nodes[n].labels.find(C.keyCT);
    if (ct == pr.nodes[n].labels.end() || ct->second != C.valSpot) allSpot = false;
  }
  NodeClaim& nc = *claims[0];
  out.candidatePrice = price;
  out.nodepool = nc.pool->index;
  out.reqs = nc.requirements;
  out.options = OrderByPrice(nc.options, out.reqs, C);
  const Requirement ct = out.reqs.Get(C.keyCT);
  if (allSpot && ct.Has(C.valSpot)) {
    // computeSpotToSpotConsolidation
    if (!spotToSpotEnabled) return out;
    out.reqs.Add(Requirement::New(C.keyCT, In, {C.valSpot}));
    vector<InstanceType*> compat;
    for (auto* it : out.options) {
      bool ok = false;
      for (auto& o : it->offerings) if (o.available && offeringCompatible(out.reqs, o, C)) { ok = true; break; }
      if (ok) compat.push_back(it);
    }
    out.options = std::move(compat);
    if (!removeByPriceAndMinValues(&out.options, out.reqs, price, C) || out.options.empty()) { out.options.clear(); return out; }
    out.spotOnly = true;
    if (cands.size() > 1) { out.decision = KG_DECISION_REPLACE; return out; }
    if ((int)out.options.size() < kMinTypesSpotToSpot) { out.options.clear(); return out; }
    int keep = kMinTypesSpotToSpot;
    if (out.reqs.HasMinValues()) {
      int need = 0;
      SatisfiesMinValues(out.options, out.reqs, &need);
      keep = std::max(keep, need);
    }
    if ((int)out.options.size() > keep) out.options.resize(keep);
    out.decision = KG_DECISION_REPLACE;
    return out;
  }
  if (!removeByPriceAndMinValues(&out.options, out.reqs, price, C) || out.options.empty()) { out.options.clear(); return out; }
  // OD -> [OD, spot]: the price filter assumed the spot variant launches; pin the replacement to spot
  if (ct.Has(C.valSpot) && ct.Has(C.valOnDemand)) { out.reqs.Add(Requirement::New(C.keyCT, In, {C.valSpot})); out.spotOnly = true; }
  out.decision = KG_DECISION_REPLACE;
  return out;
}

// multi-node: filterOutSameType -- a replacement must be cheaper than the cheapest candidate of a type it may launch
static void filterOutSameType(Problem& pr, Outcome* o, const vector<int>& cands) {
  SchedCtx& C = pr.C;
  std::map<int, double> priceByType;
  for (int n : cands) {
    double p;
    auto itl = pr.nodes[n].labels.find(C.keyInstanceType);
    if (itl == pr.nodes[n].labels.end() || !candidatePrice(pr, n, &p)) continue;
    auto f = priceByType.find(itl->second);
    if (f == priceByType.end() || p < f->second) priceByType[itl->second] = p;
  }
  double maxPrice = std::numeric_limits<double>::max();
  for (auto* it : o->options) {
    auto f = priceByType.find(it->name);
    if (f != priceByType.end() && f->second < maxPrice) maxPrice = f->second;
  }
  removeByPriceAndMinValues(&o->options, o->reqs, maxPrice, C);  // minValues error swallowed upstream
}

static Outcome evaluateUnit(Problem& pr, const kg_candidates_desc* cd, bool multi, uint32_t unit, uint64_t* sims) {
  vector<int> cands;
  if (multi) for (uint32_t i = 0; i <= unit; ++i) cands.push_back(cd->candidates[i]);
  else cands.push_back(cd->candidates[unit]);
  if (!multi && pr.nodes[cands[0]].pods.empty()) return Outcome();  // (empty nodes belong to emptiness; skipped)
  if (!multi) {
    bool any = false;
    for (Pod* p : pr.nodes[cands[0]].pods) any |= p->reschedulable;
    if (!any) return Outcome();
  }
  Outcome o = computeConsolidation(pr, cands, cd->spot_to_spot_enabled != 0, sims);
  if (multi && o.decision == KG_DECISION_REPLACE) {
    filterOutSameType(pr, &o, cands);
    if (o.options.empty()) o.decision = KG_DECISION_NOOP;
  }
  return o;
}

static void fillRecord(const Outcome& o, int unit, kg_candidate_record* r) {
  std::memset(r, 0, sizeof *r);
  r->candidate = unit;
  r->decision = o.decision;
  if (o.decision != KG_DECISION_REPLACE) return;
  r->nodepool = (uint32_t)o.nodepool;
  r->n_types = (uint32_t)o.options.size();
  r->candidate_price = o.candidatePrice;
  r->spot_only = o.spotOnly;
  for (auto* it : o.options) if (it->index < KG_RECORD_TYPE_WORDS * 64) r->type_mask[it->index >> 6] |= 1ull << (it->index & 63);
}
static void fillDecision(const Outcome& o, int nCands, int first, uint64_t sims, kg_decision* d) {
  d->decision = o.decision;
  d->n_candidates = o.decision == KG_DECISION_NOOP ? 0 : nCands;
  d->first_candidate = o.decision == KG_DECISION_NOOP ? -1 : first;
  d->nodepool = (uint32_t)o.nodepool;
  d->n_types = 0;
  d->replacement_spot_only = o.spotOnly;
  d->candidate_price = o.decision == KG_DECISION_REPLACE ? o.candidatePrice : 0;
  d->simulations = sims;
  if (o.decision == KG_DECISION_REPLACE)
    for (auto* it : o.options) { if (d->n_types < d->cap_type_ids && d->type_ids) d->type_ids[d->n_types] = (uint32_t)it->index; d->n_types++; }
}

}  // namespace ko

using namespace ko;

extern "C" {

// Scheduler.Solve + Results.TruncateInstanceTypes over descriptors; same outputs as kg_solve.
int32_t ko_solve(const kg_catalog_desc* const* catalogs, uint32_t n_catalogs, const kg_cluster_desc* cluster, const kg_pods_desc* pods,
                 const kg_nodepools_desc* nodepools, kg_results* out) {
  try {
    auto pr = buildProblem(catalogs, n_catalogs, cluster, pods, nodepools);
    auto t0 = std::chrono::steady_clock::now();
    vector<Pod*> pp;
    for (auto& p : pr->pending) pp.push_back(p.get());
    auto s = newScheduler(*pr, pp, {});
    std::map<const Pod*, uint32_t> errs;
    s->Solve(pp, &errs);
    vector<uint8_t> truncFailed;
    truncateAll(*s, pr->maxInstanceTypes, &truncFailed, &errs);
    auto t1 = std::chrono::steady_clock::now();
    out->device_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    check(out->cap_pods >= pr->pending.size(), "pod result capacity");
    for (size_t i = 0; i < pr->pending.size(); ++i) { out->pod_kind[i] = KG_POD_UNSCHEDULABLE; out->pod_index[i] = -1; out->pod_reason[i] = 0; }
    for (auto& kv : errs) out->pod_reason[kv.first->index] = kv.second;
    for (auto& n : s->existingNodes)
      for (Pod* p : n->pods) { out->pod_kind[p->index] = KG_POD_EXISTING_NODE; out->pod_index[p->index] = n->index; }
    kgdesc::ResultWriter W(out);
    for (size_t ci = 0; ci < s->newNodeClaims.size(); ++ci) {
      auto& c = *s->newNodeClaims[ci];
      kg_claim_out co{};
      co.nodepool = (uint32_t)c.pool->index;
      co.n_pods = (uint32_t)c.pods.size();
      co.truncation_failed = truncFailed[ci];
      co.types.begin = out->needed_type_ids;
      for (auto* it : c.options) W.add_type((uint32_t)it->index);
      co.types.count = (uint32_t)c.options.size();
      writeReqs(W, c.requirements, &co.requirements);
      co.requests.begin = out->needed_quantities;
      vector<int> rk;
      for (auto& kv : c.requests) rk.push_back(kv.first);
      std::sort(rk.begin(), rk.end(), lexLess);
      for (int k : rk) W.add_quantity({W.add_string(G->s(k)), c.requests.at(k)});
      co.requests.count = (uint32_t)rk.size();
      uint32_t idx = W.add_claim(co);
      for (Pod* p : c.pods) {
        if (truncFailed[ci]) { out->pod_kind[p->index] = KG_POD_UNSCHEDULABLE; out->pod_index[p->index] = -1; continue; }
        out->pod_kind[p->index] = KG_POD_NEW_CLAIM;
        out->pod_index[p->index] = (int32_t)idx;
      }
    }
    fillCounters(pr->C, out);
    return W.overflow ? KG_ENOMEM : KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_solve: %s\n", e.what());
    return KG_EINVAL;
  }
}

// Candidate eligibility, disruption cost and order (designs/consolidation.md:25-42; disruption.md:17-19), straight from
// the descriptors: cost = sum over the node's pods of clamp(1 + deletionCost/2^27 + priority/2^25, -10, 10) *
// lifetimeRemaining; eligible = initialised, managed (nodepool label), not deleting, instance type known to the catalog,
// not blocked; order = ascending cost, ties keep node order.
int32_t ko_candidates_order(const kg_catalog_desc* catalog, const kg_cluster_desc* cluster, const double* lifetime, const uint8_t* blocked,
                            int32_t* out_nodes, double* out_costs, uint32_t cap, uint32_t* n_out) {
  try {
    const uint32_t N = cluster->n_nodes;
    vector<double> cost(N, 0.0);
    for (uint32_t i = 0; i < cluster->pods.n_pods; ++i) {
      const kg_pod& p = cluster->pods.pods[i];
      if (p.node < 0 || (uint32_t)p.node >= N) continue;
      double c = 1.0 + p.deletion_cost / 134217728.0 + p.priority / 33554432.0;
      if (c < -10.0) c = -10.0;
      if (c > 10.0) c = 10.0;
      cost[p.node] += c;
    }
    std::set<string> typeNames;
    if (catalog) for (uint32_t t = 0; t < catalog->n_types; ++t) typeNames.insert(string(kgdesc::str(catalog->pools.strings, catalog->types[t].name)));
    vector<int32_t> order;
    for (uint32_t i = 0; i < N; ++i) {
      const kg_node& n = cluster->nodes[i];
      if (lifetime) cost[i] *= lifetime[i];
      if (!n.initialized || !n.managed || n.deleting || (blocked && blocked[i])) continue;
      bool pool = false, type = false;
      for (auto& l : kgdesc::span(cluster->pods.pools.labels, cluster->pods.pools.n_labels, n.labels, "node labels")) {
        const auto k = kgdesc::str(cluster->pods.pools.strings, l.key);
        if (k == "karpenter.sh/nodepool") pool = true;
        if (k == "node.kubernetes.io/instance-type") type = !catalog || typeNames.count(string(kgdesc::str(cluster->pods.pools.strings, l.value)));
      }
      if (pool && type) order.push_back((int32_t)i);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[a] < cost[b]; });
    *n_out = (uint32_t)order.size();
    if (order.size() > cap) return KG_ENOMEM;
    for (size_t k = 0; k < order.size(); ++k) { out_nodes[k] = order[k]; if (out_costs) out_costs[k] = cost[order[k]]; }
    return KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_candidates_order: %s\n", e.what());
    return KG_EINVAL;
  }
}

// computeConsolidation for every unit of this shard (same records as kg_consolidate_single / _multi)
int32_t ko_consolidate_records(const kg_catalog_desc* const* catalogs, uint32_t n_catalogs, const kg_cluster_desc* cluster,
                               const kg_pods_desc* pending, const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, int32_t multi,
                               kg_candidate_record* records, uint32_t cap_records, uint32_t* n_records, double* ms) {
  try {
    auto pr = buildProblem(catalogs, n_catalogs, cluster, pending, nodepools);
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = cands->n_candidates;
    const uint32_t lo = multi ? 1 : 0, hi = multi ? std::min<uint32_t>(n, 101) : n;  // multi: mid in [1, len <= 100 ? len - 1 : 100]
    const int shards = std::max(cands->n_shards, 1);
    uint32_t k = 0;
    uint64_t sims = 0;
    for (uint32_t u = lo; u < hi; ++u) {
      if ((int)((u - lo) % shards) != cands->shard) continue;
      Outcome o = evaluateUnit(*pr, cands, multi != 0, u, &sims);
      if (k < cap_records) fillRecord(o, (int)u, &records[k]);
      k++;
    }
    *n_records = k;
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return k > cap_records ? KG_ENOMEM : KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_consolidate_records: %s\n", e.what());
    return KG_EINVAL;
  }
}

// Validation (disruption validation: re-simulate the command against the current cluster, require the same outcome --
// SURVEY.md Appendix A.9): every rescheduled pod schedules; a delete creates no NodeClaim; a replace creates exactly one
// whose options still contain every instance type of the command.
int32_t ko_consolidate_validate(const kg_catalog_desc* const* catalogs, uint32_t n_catalogs, const kg_cluster_desc* cluster,
                                const kg_pods_desc* pending, const kg_nodepools_desc* nodepools, const int32_t* candidates, uint32_t n_candidates,
                                int32_t decision, const uint32_t* type_ids, uint32_t n_types, uint8_t* valid) {
  try {
    *valid = 0;
    auto pr = buildProblem(catalogs, n_catalogs, cluster, pending, nodepools);
    vector<int> cands(candidates, candidates + n_candidates);
    uint64_t sims = 0;
    SimRaw raw;
    computeConsolidation(*pr, cands, false, &sims, &raw);
    if (!raw.allScheduled) return KG_OK;
    if (raw.nClaims == 0) { *valid = decision == KG_DECISION_DELETE; return KG_OK; }
    if (raw.nClaims > 1 || decision != KG_DECISION_REPLACE) return KG_OK;
    std::set<int> have(raw.types.begin(), raw.types.end());
    bool subset = n_types > 0;
    for (uint32_t i = 0; i < n_types; ++i) subset = subset && have.count((int)type_ids[i]);
    *valid = subset;
    return KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_consolidate_validate: %s\n", e.what());
    return KG_EINVAL;
  }
}

// The reference's sequential algorithms: SingleNodeConsolidation.ComputeCommand (first candidate in order with a valid
// command) and MultiNodeConsolidation.firstNConsolidationOption (binary search over prefixes of at most 100 candidates).
int32_t ko_consolidate_decide(const kg_catalog_desc* const* catalogs, uint32_t n_catalogs, const kg_cluster_desc* cluster,
                              const kg_pods_desc* pending, const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, int32_t multi,
                              kg_decision* out) {
  try {
    auto pr = buildProblem(catalogs, n_catalogs, cluster, pending, nodepools);
    auto t0 = std::chrono::steady_clock::now();
    uint64_t sims = 0;
    Outcome none;
    const uint32_t n = cands->n_candidates;
    if (!multi) {
      bool done = false;
      for (uint32_t u = 0; u < n && !done; ++u) {
        Outcome o = evaluateUnit(*pr, cands, false, u, &sims);
        if (o.decision == KG_DECISION_NOOP) continue;
        fillDecision(o, 1, (int)u, sims, out);
        done = true;
      }
      if (!done) fillDecision(none, 0, -1, sims, out);
    } else {
      Outcome last;
      int lastN = 0;
      if (n >= 2) {
        // max = Clamp(len, 0, 100); if len <= max { max = len - 1 }: with more than 100 candidates mid ranges over [1, 100]
        int mn = 1, mx = n <= 100 ? (int)n - 1 : 100;
        while (mn <= mx) {
          const int mid = (mn + mx) / 2;
          Outcome o = evaluateUnit(*pr, cands, true, (uint32_t)mid, &sims);
          if (o.decision != KG_DECISION_NOOP) { last = o; lastN = mid + 1; mn = mid + 1; } else mx = mid - 1;
        }
      }
      fillDecision(last, lastN, 0, sims, out);
    }
    out->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_consolidate_decide: %s\n", e.what());
    return KG_EINVAL;
  }
}


// InstanceTypes.Truncate(reqs, max) over the given catalog types (pkg/providers/instance/instance.go:101)
int32_t ko_truncate(const kg_catalog_desc* catalog, const kg_pools* pools, kg_range requirements, const uint32_t* type_ids, uint32_t n_types,
                    int32_t max_types, uint32_t* out_type_ids, uint32_t* n_out, uint8_t* min_values_failed) {
  try {
    Problem pr;
    G = &pr.in;
    SchedCtx& C = pr.C;
    C.keyZone = pr.in.id("topology.kubernetes.io/zone"); C.keyCT = pr.in.id("karpenter.sh/capacity-type"); C.keyZoneID = pr.in.id("topology.k8s.aws/zone-id");
    Parser P{pr.in};
    auto cat = parseCatalog(pr.in, catalog, C.keyCT, C.keyZone, C.keyZoneID);
    Requirements reqs = P.reqs(*pools, requirements);
    vector<InstanceType*> its;
    for (uint32_t i = 0; i < n_types; ++i) { check(type_ids[i] < cat->types.size(), "type id"); its.push_back(cat->types[type_ids[i]].get()); }
    for (const char* k : {"karpenter.sh/nodepool", "topology.kubernetes.io/zone", "karpenter.sh/capacity-type", "topology.k8s.aws/zone-id"}) C.wellKnown.insert(pr.in.id(k));
    vector<InstanceType*> out;
    const bool ok = Truncate(its, reqs, max_types, C, &out);
    if (min_values_failed) *min_values_failed = !ok;
    *n_out = (uint32_t)out.size();
    for (size_t i = 0; i < out.size(); ++i) out_type_ids[i] = (uint32_t)out[i]->index;
    return KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_truncate: %s\n", e.what());
    return KG_EINVAL;
  }
}

// CloudProvider.Create's launch-side selection: resolveInstanceTypes (pkg/cloudprovider/cloudprovider.go:267-282), then
// without minValues filterExoticInstanceTypes + (isMixedCapacityLaunch ? filterUnwantedSpot) (instance.go:389-477), then
// Truncate(reqs, max) (instance.go:101).
int32_t ko_launch_filter(const kg_catalog_desc* catalog, const kg_pools* pools, kg_range requirements, kg_range requests, int32_t max_types,
                         const char* const* well_known, uint32_t n_well_known, uint32_t* out_type_ids, uint32_t cap, uint32_t* n_out,
                         uint8_t* min_values_failed) {
  try {
    Problem pr;
    G = &pr.in;
    SchedCtx& C = pr.C;
    C.keyZone = pr.in.id("topology.kubernetes.io/zone"); C.keyCT = pr.in.id("karpenter.sh/capacity-type"); C.keyZoneID = pr.in.id("topology.k8s.aws/zone-id");
    C.valSpot = pr.in.id("spot"); C.valOnDemand = pr.in.id("on-demand");
    for (uint32_t i = 0; i < n_well_known; ++i) C.wellKnown.insert(pr.in.id(well_known[i]));
    Parser P{pr.in};
    auto cat = parseCatalog(pr.in, catalog, C.keyCT, C.keyZone, C.keyZoneID);
    Requirements reqs = P.reqs(*pools, requirements);
    RL req = P.rl(*pools, requests);
    vector<InstanceType*> its;
    for (auto& t : cat->types) {
      InstanceType* it = t.get();
      if (!reqs.Compatible(it->reqs, &C.wellKnown)) continue;
      bool off = false;
      for (auto& o : it->offerings) if (o.available && offeringCompatible(reqs, o, C)) { off = true; break; }
      if (!off || !Fits(req, it->allocatable)) continue;
      its.push_back(it);
    }
    if (!reqs.HasMinValues()) {
      const int keySize = pr.in.id("karpenter.k8s.aws/instance-size");
      vector<InstanceType*> generic;
      for (auto* it : its) {
        bool skip = false;
        for (int v : it->reqs.Get(keySize).values) skip |= G->s(v).find("metal") != string::npos;
        for (const char* rn : {"aws.amazon.com/neuron", "amd.com/gpu", "nvidia.com/gpu", "habana.ai/gaudi"}) skip |= rlGet(it->capacity, pr.in.id(rn)) != 0;
        if (!skip) generic.push_back(it);
      }
      if (!generic.empty()) its = generic;
      const Requirement ct = reqs.Get(C.keyCT);
      if (ct.Has(C.valSpot) && ct.Has(C.valOnDemand)) {
        bool hasSpot = false, hasOD = false;
        for (auto* it : its)
          for (auto& o : it->offerings) {
            if (!o.available || !offeringCompatible(reqs, o, C)) continue;
            if (o.ct == C.valSpot) hasSpot = true; else hasOD = true;
          }
        if (hasSpot && hasOD) {
          double cheapestOD = std::numeric_limits<double>::max();
          for (auto* it : its) for (auto& o : it->offerings) if (o.available && o.ct == C.valOnDemand && o.price < cheapestOD) cheapestOD = o.price;
          vector<InstanceType*> kept;
          for (auto* it : its) {
            bool any = false;
            double best = 0;
            for (auto& o : it->offerings) if (o.available && (!any || o.price < best)) { best = o.price; any = true; }
            if (any && best <= cheapestOD) kept.push_back(it);
          }
          its = kept;
        }
      }
    }
    vector<InstanceType*> out;
    const bool ok = Truncate(its, reqs, max_types, C, &out);
    if (min_values_failed) *min_values_failed = !ok;
    *n_out = (uint32_t)out.size();
    for (size_t i = 0; i < out.size() && i < cap; ++i) out_type_ids[i] = (uint32_t)out[i]->index;
    return KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_launch_filter: %s\n", e.what());
    return KG_EINVAL;
  }
}

// getCapacityType + getOverrides (pkg/providers/instance/instance.go:372-385, 323-359): the capacity type CreateFleet is
// asked for and the (instance type, zonal subnet) override list, types in the given order, offerings in catalog order.
int32_t ko_launch_overrides(const kg_catalog_desc* catalog, const kg_pools* pools, kg_range requirements, const uint32_t* type_ids, uint32_t n_types,
                            const uint32_t* subnet_zones, uint32_t n_subnet_zones, const char* const* well_known, uint32_t n_well_known,
                            int32_t* capacity_type, kg_override* out, uint32_t cap, uint32_t* n_out) {
  try {
    Problem pr;
    G = &pr.in;
    SchedCtx& C = pr.C;
    C.keyZone = pr.in.id("topology.kubernetes.io/zone"); C.keyCT = pr.in.id("karpenter.sh/capacity-type"); C.keyZoneID = pr.in.id("topology.k8s.aws/zone-id");
    C.valSpot = pr.in.id("spot"); C.valOnDemand = pr.in.id("on-demand");
    for (uint32_t i = 0; i < n_well_known; ++i) C.wellKnown.insert(pr.in.id(well_known[i]));
    Parser P{pr.in};
    auto cat = parseCatalog(pr.in, catalog, C.keyCT, C.keyZone, C.keyZoneID);
    Requirements reqs = P.reqs(*pools, requirements);
    vector<InstanceType*> its;
    for (uint32_t i = 0; i < n_types; ++i) { check(type_ids[i] < cat->types.size(), "type id"); its.push_back(cat->types[type_ids[i]].get()); }
    // getCapacityType: spot if the requirements admit it and an available offering is compatible with them pinned to spot
    int ct = C.valOnDemand;
    if (reqs.Get(C.keyCT).Has(C.valSpot)) {
      Requirements pinned = reqs;
      pinned.m.erase(C.keyCT);
      pinned.m.emplace(C.keyCT, Requirement::New(C.keyCT, In, {C.valSpot}));
      for (auto* it : its) {
        for (auto& o : it->offerings) if (o.available && offeringCompatible(pinned, o, C)) { ct = C.valSpot; break; }
        if (ct == C.valSpot) break;
      }
    }
    *capacity_type = ct == C.valSpot ? KG_CT_SPOT : KG_CT_ON_DEMAND;
    // getLaunchTemplateConfigs: requirements[capacity-type] = In [capacityType]; getOverrides over the available offerings
    Requirements pinned = reqs;
    pinned.m.erase(C.keyCT);
    pinned.m.emplace(C.keyCT, Requirement::New(C.keyCT, In, {ct}));
    std::map<int, uint32_t> zonal;  // zone -> subnet entry
    for (uint32_t s = 0; s < n_subnet_zones; ++s) zonal.emplace(pr.in.id(string(kgdesc::str(pools->strings, subnet_zones[s]))), s);
    uint32_t n = 0;
    for (auto* it : its)
      for (auto& o : it->offerings) {
        if (!o.available || !offeringCompatible(pinned, o, C)) continue;
        auto f = zonal.find(o.reqs.Get(C.keyZone).Any());
        if (f == zonal.end()) continue;
        if (n < cap) out[n] = kg_override{(uint32_t)it->index, f->second};
        n++;
      }
    *n_out = n;
    return n > cap ? KG_ENOMEM : KG_OK;
  } catch (const std::exception& e) {
    fprintf(stderr, "ko_launch_overrides: %s\n", e.what());
    return KG_EINVAL;
  }
}

}  // extern "C"
