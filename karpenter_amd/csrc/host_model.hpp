// host_model.hpp -- host-side vocabulary, requirement algebra and catalog tables of libkarpgpu.
//
// "Requirement compiler (host C++)": pod / NodePool / node-label requirements become per-key
// value bitsets over a context-wide vocabulary; the instance-type catalog becomes SoA tables of
// int64 milli-quantities plus type bitmasks.  Semantics follow scheduling.Requirement(s) of
// sigs.k8s.io/karpenter and the catalog built by
// pkg/providers/instancetype/types.go:51-71 / instancetype.go:264-315.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "karpgpu.h"
#include "karpgpu_desc.hpp"

namespace kg {

struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// the internal resource a CSI driver's volume count rides on (compile.hpp: ParsedPods::vols)
inline const char* kVolumeResourcePrefix() { return "karpgpu.internal/volumes/"; }
inline bool is_volume_resource(const std::string& name) { return name.rfind(kVolumeResourcePrefix(), 0) == 0; }
struct Stale : std::runtime_error {  // -> KG_ESTALE
  using std::runtime_error::runtime_error;
};

struct StrMap {
  std::deque<std::string> strs;  // (a deque: the strings never move, so the index below can key on views of them and a lookup
                                 //  builds no temporary std::string)
  std::unordered_map<std::string_view, uint32_t> ids;
  StrMap() = default;
  StrMap(const StrMap& o) : strs(o.strs) { for (uint32_t i = 0; i < strs.size(); ++i) ids.emplace(strs[i], i); }
  StrMap& operator=(const StrMap& o) {
    if (this != &o) { strs = o.strs; ids.clear(); for (uint32_t i = 0; i < strs.size(); ++i) ids.emplace(strs[i], i); }
    return *this;
  }
  uint32_t id(std::string_view s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    uint32_t i = (uint32_t)strs.size();
    strs.emplace_back(s);
    ids.emplace(strs.back(), i);
    return i;
  }
  int find(std::string_view s) const {
    auto it = ids.find(s);
    return it == ids.end() ? -1 : (int)it->second;
  }
  size_t size() const { return strs.size(); }
};

// Context-wide vocabulary: a host-side string interner (ids never change once assigned).  VALUE ids never reach the device:
// a catalog numbers the values it mentions itself (CatalogTables::cvals, "catalog-local" ids) and every Solve numbers the
// values its own batch uses (Compiled::lv, "solve-local" ids = the catalog's, then the batch's extras), so no device table,
// row width or kernel choice depends on what a long-lived context has seen before.  Key and resource ids do index host
// tables; they are as few as the label keys and resource names of the cluster.
struct Vocab {
  StrMap keys;
  std::vector<StrMap> vals;                 // per key
  std::vector<std::vector<int64_t>> ival;   // per key per value: parsed integer
  std::vector<std::vector<uint8_t>> isint;  // strconv.Atoi succeeded
  StrMap resources;
  uint32_t key(std::string_view k) {
    uint32_t i = keys.id(k);
    if (i >= vals.size()) { vals.resize(i + 1); ival.resize(i + 1); isint.resize(i + 1); }
    return i;
  }
  uint32_t val(uint32_t k, std::string_view v) {
    uint32_t i = vals[k].id(v);
    if (i >= ival[k].size()) {
      int64_t x = 0;
      bool ok = kgdesc::atoi_go(v, &x);
      ival[k].push_back(x);
      isint[k].push_back(ok);
    }
    return i;
  }
};

// One requirement on one key (scheduling.Requirement).
struct HReq {
  uint32_t key = 0;
  bool compl_ = false;
  std::vector<uint32_t> vals;  // sorted unique value ids
  bool has_gt = false, has_lt = false;
  int64_t gt = 0, lt = 0;
  int min_values = -1;
  bool operator==(const HReq& o) const {
    return key == o.key && compl_ == o.compl_ && vals == o.vals && has_gt == o.has_gt && has_lt == o.has_lt && (!has_gt || gt == o.gt) &&
           (!has_lt || lt == o.lt) && min_values == o.min_values;
  }
  bool within(const Vocab& V, uint32_t v) const {
    if (!has_gt && !has_lt) return true;
    if (!V.isint[key][v]) return false;
    int64_t x = V.ival[key][v];
    if (has_gt && gt >= x) return false;
    if (has_lt && lt <= x) return false;
    return true;
  }
  bool has(const Vocab& V, uint32_t v) const {
    bool in = std::binary_search(vals.begin(), vals.end(), v);
    return (compl_ ? !in : in) && within(V, v);
  }
  int opclass() const { return compl_ ? (vals.empty() ? 2 : 1) : (vals.empty() ? 3 : 0); }  // In, NotIn, Exists, DoesNotExist
};

inline HReq hreq_new(Vocab& V, uint32_t key, int op, const std::vector<uint32_t>& vals, int min_values) {
  HReq r;
  r.key = key;
  r.min_values = min_values;
  if (op == KG_OP_IN) {
    r.vals = vals;
  } else {
    r.compl_ = op != KG_OP_DOES_NOT_EXIST;
    if (op == KG_OP_NOT_IN) r.vals = vals;
    if (op == KG_OP_GT || op == KG_OP_LT) {
      kgdesc::check(!vals.empty() && V.isint[key][vals[0]], "Gt/Lt needs one integer value");
      if (op == KG_OP_GT) { r.has_gt = true; r.gt = V.ival[key][vals[0]]; } else { r.has_lt = true; r.lt = V.ival[key][vals[0]]; }
    }
  }
  std::sort(r.vals.begin(), r.vals.end());
  r.vals.erase(std::unique(r.vals.begin(), r.vals.end()), r.vals.end());
  return r;
}

inline HReq hreq_intersect(const Vocab& V, const HReq& a, const HReq& b) {
  HReq r;
  r.key = a.key;
  r.compl_ = a.compl_ && b.compl_;
  r.has_gt = a.has_gt || b.has_gt;
  r.gt = a.has_gt && b.has_gt ? std::max(a.gt, b.gt) : (a.has_gt ? a.gt : b.gt);
  r.has_lt = a.has_lt || b.has_lt;
  r.lt = a.has_lt && b.has_lt ? std::min(a.lt, b.lt) : (a.has_lt ? a.lt : b.lt);
  r.min_values = std::max(a.min_values, b.min_values);
  if (r.has_gt && r.has_lt && r.gt >= r.lt) {
    r.compl_ = false; r.has_gt = r.has_lt = false; r.vals.clear();
    return r;
  }
  auto in = [](const std::vector<uint32_t>& s, uint32_t v) { return std::binary_search(s.begin(), s.end(), v); };
  if (a.compl_ && b.compl_) {
    std::set_union(a.vals.begin(), a.vals.end(), b.vals.begin(), b.vals.end(), std::back_inserter(r.vals));
  } else if (a.compl_) {
    for (uint32_t v : b.vals) if (!in(a.vals, v)) r.vals.push_back(v);
  } else if (b.compl_) {
    for (uint32_t v : a.vals) if (!in(b.vals, v)) r.vals.push_back(v);
  } else {
    for (uint32_t v : a.vals) if (in(b.vals, v)) r.vals.push_back(v);
  }
  std::vector<uint32_t> kept;
  for (uint32_t v : r.vals) if (r.within(V, v)) kept.push_back(v);
  r.vals.swap(kept);
  if (!r.compl_) r.has_gt = r.has_lt = false;
  return r;
}

struct HReqs {
  std::vector<HReq> rs;  // sorted by key
  const HReq* find(uint32_t key) const {
    for (auto& r : rs) if (r.key == key) return &r;
    return nullptr;
  }
  void add(const Vocab& V, const HReq& in) {
    for (auto& r : rs) if (r.key == in.key) { r = hreq_intersect(V, in, r); return; }
    rs.push_back(in);
    std::sort(rs.begin(), rs.end(), [](const HReq& a, const HReq& b) { return a.key < b.key; });
  }
  bool operator==(const HReqs& o) const { return rs == o.rs; }
};

// packs device arrays into one host image; pointers are fixed up after the upload
struct Arena {
  std::vector<uint8_t> bytes;
  template <class T>
  size_t add(const std::vector<T>& v, size_t min_elems = 1) {
    size_t off = (bytes.size() + 255) & ~size_t(255);
    size_t n = std::max(v.size(), min_elems);
    bytes.resize(off + n * sizeof(T), 0);
    if (!v.empty()) std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
  size_t reserve(size_t nbytes) {
    size_t off = (bytes.size() + 255) & ~size_t(255);
    bytes.resize(off + std::max<size_t>(nbytes, 8), 0);
    return off;
  }
  // Space behind the host image that only the device touches (kernel outputs, per-claim state): it costs no host memory,
  // no zero fill and no upload.  Every add() / reserve() must come before the first reserve_tail().
  size_t tail_end = 0;
  size_t reserve_tail(size_t nbytes) {
    size_t off = (std::max(bytes.size(), tail_end) + 255) & ~size_t(255);
    tail_end = off + std::max<size_t>(nbytes, 8);
    return off;
  }
  size_t total() const { return (std::max(bytes.size(), tail_end) + 255) & ~size_t(255); }
};

// ---------------------------------------------------------------- catalog (host view + device tables)
struct CatOffering {
  uint32_t zone, ct;  // (context) value ids of the zone / capacity-type keys
  int zid;            // value id of the zone-id key or -1
  uint32_t lz = 0, lct = 0;  // catalog-local ids of the same values (CatalogTables::cvals): what masks and device tables index
  int lzid = -1;
  double price;
  bool available;
};
struct CatType {
  std::string name;
  HReqs reqs;  // keys the type mentions (DoesNotExist entries have empty vals, compl_=false)
  std::vector<std::pair<uint32_t, int64_t>> capacity, overhead;  // (resource id, milli)
  std::vector<CatOffering> offerings;
};

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

struct CatalogTables {  // offsets into `image`; rebuilt when the context's resource vocabulary grows
  uint32_t T = 0, TW = 0, R = 0;
  Arena image;
  size_t o_thr_val, o_thr_off, o_thr_n, o_thr_mask, o_cthr_val, o_cthr_off, o_cthr_n, o_cthr_mask, o_valid, o_cap, o_alloc, o_twv, o_offmask,
      o_zone_zid, o_name_rank, o_off_begin, o_off_zone, o_off_ct, o_off_zid, o_off_price, o_off_avail;
  std::vector<uint32_t> key_row_off;  // per global key: first twv row, or UINT32_MAX
  std::vector<uint32_t> key_nvals;    // per global key: rows (= values the catalog's types / offerings carry on the key)
  // catalog-local value numbering per global key: the values the catalog's types and offerings mention, in lexicographic
  // order (a property of the catalog's content alone: the same in every process that uploads the same catalog)
  std::vector<std::vector<uint32_t>> cvals;                       // catalog-local id -> context value id
  std::vector<std::unordered_map<uint32_t, uint32_t>> cmap;      // context value id -> catalog-local id
  int local(uint32_t key, uint32_t v) const {
    if (key >= cmap.size()) return -1;
    auto f = cmap[key].find(v);
    return f == cmap[key].end() ? -1 : (int)f->second;
  }
  std::vector<std::vector<uint64_t>> kU, kN;  // per global key [TW]; keys unknown to the catalog: kN = all
  std::vector<uint64_t> all_types;
  uint32_t n_zone_vals = 0, n_ct_vals = 0;
  DeviceBuffer dev;
};

}  // namespace kg

struct kg_ctx;
struct kg_catalog {
  kg_ctx* ctx;
  uint64_t generation;
  uint64_t tab_epoch = 0;  // bumped by every in-place update of the resident tables (consolidation plans are keyed on it)
  uint64_t uid = next_uid();  // never reused within the process (a pointer is, after kg_catalog_free)
  static uint64_t next_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
  std::vector<kg::CatType> types;
  kg::CatalogTables tab;
  // A merged view over several uploaded catalogs (NodePools on different EC2NodeClasses in one solve): `types` is their
  // concatenation, every type carrying a hidden requirement that names its source.  Empty for an uploaded catalog.
  std::vector<uint32_t> part_begin;  // [parts + 1] first merged type id of every source
  std::vector<uint64_t> part_key;    // (uid, generation, tab_epoch) per source: what the view was built from
  uint64_t last_use = 0;
  uint32_t part_of(uint32_t t) const { uint32_t p = 0; while (p + 2 < part_begin.size() && t >= part_begin[p + 1]) ++p; return p; }
};
