// karpgpu_desc.hpp -- header-only helpers to walk karpgpu.h descriptors (bounds-checked) and to
// fill kg_results buffers.  Pure plumbing shared by the library and by the test oracle; it holds
// no scheduling logic.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "karpgpu.h"

namespace kgdesc {

struct DescError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void check(bool ok, const char* what) {
  if (!ok) throw DescError(what);
}

inline std::string_view str(const kg_strtab& t, uint32_t id) {
  check(id < t.count, "string id out of range");
  return std::string_view(t.data + t.offsets[id], t.offsets[id + 1] - t.offsets[id]);
}

template <class T>
struct Span {
  const T* p;
  uint32_t n;
  const T* begin() const { return p; }
  const T* end() const { return p + n; }
  const T& operator[](uint32_t i) const { return p[i]; }
  uint32_t size() const { return n; }
};

template <class T>
inline Span<T> span(const T* base, uint32_t n_base, kg_range r, const char* what) {
  check((uint64_t)r.begin + r.count <= n_base, what);
  return Span<T>{base + r.begin, r.count};
}

inline Span<kg_requirement> reqs(const kg_pools& p, kg_range r) { return span(p.reqs, p.n_reqs, r, "req range"); }
inline Span<uint32_t> values(const kg_pools& p, kg_range r) { return span(p.req_values, p.n_req_values, r, "value range"); }
inline Span<kg_quantity> quantities(const kg_pools& p, kg_range r) { return span(p.quantities, p.n_quantities, r, "quantity range"); }
inline Span<kg_label> labels(const kg_pools& p, kg_range r) { return span(p.labels, p.n_labels, r, "label range"); }
inline Span<kg_taint> taints(const kg_pools& p, kg_range r) { return span(p.taints, p.n_taints, r, "taint range"); }
inline Span<kg_toleration> tolerations(const kg_pools& p, kg_range r) { return span(p.tolerations, p.n_tolerations, r, "toleration range"); }
inline Span<kg_range> ranges(const kg_pools& p, kg_range r) { return span(p.ranges, p.n_ranges, r, "range range"); }

// strconv.Atoi semantics (optional sign, decimal digits only, must fit in int64 here)
inline bool atoi_go(std::string_view s, int64_t* out) {
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') {
    neg = s[0] == '-';
    i = 1;
    if (s.size() == 1) return false;
  }
  int64_t v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    if (v > (INT64_MAX - 9) / 10) return false;
    v = v * 10 + (s[i] - '0');
  }
  *out = neg ? -v : v;
  return true;
}

// Collects output into a kg_results with capacity accounting.
struct ResultWriter {
  kg_results* r;
  bool overflow = false;
  explicit ResultWriter(kg_results* res) : r(res) {
    r->n_claims = r->needed_claims = 0;
    r->n_type_ids = r->needed_type_ids = 0;
    r->n_reqs = r->needed_reqs = 0;
    r->n_value_ids = r->needed_value_ids = 0;
    r->n_quantities = r->needed_quantities = 0;
    r->n_str_data = r->needed_str_data = 0;
    r->n_strs = r->needed_strs = 0;
  }
  uint32_t add_string(std::string_view s) {
    uint32_t id = r->needed_strs++;
    uint32_t off = r->needed_str_data;
    r->needed_str_data += (uint32_t)s.size();
    if (id + 1 < r->cap_strs && off + s.size() <= r->cap_str_data && r->str_offsets && r->str_data) {
      std::memcpy(r->str_data + off, s.data(), s.size());
      r->str_offsets[id] = off;
      r->str_offsets[id + 1] = off + (uint32_t)s.size();
      r->n_strs = id + 1;
      r->n_str_data = off + (uint32_t)s.size();
    } else {
      overflow = true;
    }
    return id;
  }
  uint32_t add_type(uint32_t t) {
    uint32_t i = r->needed_type_ids++;
    if (i < r->cap_type_ids && r->type_ids) { r->type_ids[i] = t; r->n_type_ids = i + 1; } else overflow = true;
    return i;
  }
  uint32_t add_value(uint32_t v) {
    uint32_t i = r->needed_value_ids++;
    if (i < r->cap_value_ids && r->value_ids) { r->value_ids[i] = v; r->n_value_ids = i + 1; } else overflow = true;
    return i;
  }
  uint32_t add_req(const kg_req_out& q) {
    uint32_t i = r->needed_reqs++;
    if (i < r->cap_reqs && r->reqs) { r->reqs[i] = q; r->n_reqs = i + 1; } else overflow = true;
    return i;
  }
  uint32_t add_quantity(const kg_quantity& q) {
    uint32_t i = r->needed_quantities++;
    if (i < r->cap_quantities && r->quantities) { r->quantities[i] = q; r->n_quantities = i + 1; } else overflow = true;
    return i;
  }
  uint32_t add_claim(const kg_claim_out& c) {
    uint32_t i = r->needed_claims++;
    if (i < r->cap_claims && r->claims) { r->claims[i] = c; r->n_claims = i + 1; } else overflow = true;
    return i;
  }
};

}  // namespace kgdesc
