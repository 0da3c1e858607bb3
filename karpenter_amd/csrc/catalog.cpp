// catalog.cpp -- catalog upload: cloudprovider.InstanceType list -> SoA / bitmask tables.
//
// Encodes what pkg/providers/instancetype/types.go:56-66 builds (Name, Requirements, Offerings,
// Capacity, Overhead) into:
//   alloc/cap [R][T] int64 milli        Allocatable() = Capacity - Overhead.Total()
//   thr_*   per resource: sorted distinct allocatable values + "types with alloc >= v" masks
//   cthr_*  per resource: sorted distinct capacity values + "types with cap <= v" masks
//   twv     per (key, value): types whose requirement on key contains value
//   kU/kN   per key: types that list the key as DoesNotExist / do not mention it
//   offmask per (zone, capacity-type): types with an AVAILABLE offering there (instancetype.go:289)
#include <set>
#include "ctx.hpp"

namespace kg {

static std::vector<uint32_t> vals_of(Vocab& V, const kg_pools& p, uint32_t key, kg_range r) {
  std::vector<uint32_t> out;
  for (uint32_t v : kgdesc::values(p, r)) out.push_back(V.val(key, kgdesc::str(p.strings, v)));
  return out;
}

HReq parse_req(Vocab& V, const kg_pools& p, const kg_requirement& q) {
  uint32_t key = V.key(kgdesc::str(p.strings, q.key));
  return hreq_new(V, key, q.op, vals_of(V, p, key, q.values), q.min_values);
}

HReqs parse_reqs(Vocab& V, const kg_pools& p, kg_range r) {
  HReqs out;
  for (auto& q : kgdesc::reqs(p, r)) out.add(V, parse_req(V, p, q));
  return out;
}

std::vector<std::pair<uint32_t, int64_t>> parse_rl(Vocab& V, const kg_pools& p, kg_range r) {
  std::map<uint32_t, int64_t> m;
  for (auto& q : kgdesc::quantities(p, r)) m[V.resources.id(kgdesc::str(p.strings, q.name))] += q.milli;
  return {m.begin(), m.end()};
}

void parse_catalog(kg_ctx* ctx, const kg_catalog_desc* d, kg_catalog* cat) {
  Vocab& V = ctx->V;
  const uint32_t kZone = ctx->kZone, kCT = ctx->kCT, kZid = ctx->kZid;
  for (uint32_t i = 0; i < d->n_types; ++i) {
    const kg_instance_type& t = d->types[i];
    CatType ct;
    ct.name = std::string(kgdesc::str(d->pools.strings, t.name));
    ct.reqs = parse_reqs(V, d->pools, t.requirements);
    for (auto& r : ct.reqs.rs)
      if (r.compl_) throw Unsupported("instance type requirement with a complement operator");
    ct.capacity = parse_rl(V, d->pools, t.capacity);
    ct.overhead = parse_rl(V, d->pools, t.overhead);
    for (auto& o : kgdesc::span(d->offerings, d->n_offerings, t.offerings, "offering range")) {
      CatOffering co;
      co.zone = V.val(kZone, kgdesc::str(d->pools.strings, o.zone));
      co.ct = V.val(kCT, kgdesc::str(d->pools.strings, o.capacity_type));
      co.zid = o.zone_id == KG_NONE ? -1 : (int)V.val(kZid, kgdesc::str(d->pools.strings, o.zone_id));
      co.price = o.price;
      co.available = o.available != 0;
      ct.offerings.push_back(co);
    }
    cat->types.push_back(std::move(ct));
  }
}

// (Re)build the device tables for the context's current resource vocabulary.
void build_catalog_tables(kg_ctx* ctx, kg_catalog* cat) {
  Vocab& V = ctx->V;
  CatalogTables& C = cat->tab;
  if (C.dev.ptr) { hal::dfree(C.dev.ptr); C.dev.ptr = nullptr; }
  C = CatalogTables();
  const uint32_t T = (uint32_t)cat->types.size(), TW = (T + 63) / 64, R = (uint32_t)V.resources.size();
  C.T = T; C.TW = std::max(TW, 1u); C.R = R;
  const uint32_t tw = C.TW;
  C.all_types.assign(tw, 0);
  for (uint32_t t = 0; t < T; ++t) C.all_types[t >> 6] |= 1ull << (t & 63);
  std::vector<int64_t> cap((size_t)std::max(R, 1u) * std::max(T, 1u), 0), alloc(cap.size(), 0);
  for (uint32_t t = 0; t < T; ++t) {
    for (auto& kv : cat->types[t].capacity) { cap[(size_t)kv.first * T + t] = kv.second; alloc[(size_t)kv.first * T + t] = kv.second; }
    for (auto& kv : cat->types[t].overhead)
      // resources.Subtract(Capacity, Overhead.Total()): only keys of Capacity appear in Allocatable
      for (auto& ck : cat->types[t].capacity) if (ck.first == kv.first) alloc[(size_t)kv.first * T + t] -= kv.second;
  }
  for (uint32_t r = 0; r < R; ++r)
    if (is_volume_resource(V.resources.strs[r])) for (uint32_t t = 0; t < T; ++t) alloc[(size_t)r * T + t] = INT64_MAX / 4;  // (volume limits belong to registered nodes: a NodeClaim has none)
  std::vector<uint64_t> valid(tw, 0);
  for (uint32_t t = 0; t < T; ++t) {
    bool ok = true;
    for (auto& ck : cat->types[t].capacity) if (alloc[(size_t)ck.first * T + t] < 0) ok = false;
    if (ok) valid[t >> 6] |= 1ull << (t & 63);
  }
  // threshold tables
  std::vector<int64_t> thr_val, cthr_val;
  std::vector<uint32_t> thr_off(R + 1, 0), thr_n(std::max(R, 1u), 0), cthr_off(R + 1, 0), cthr_n(std::max(R, 1u), 0);
  std::vector<uint64_t> thr_mask, cthr_mask;
  for (uint32_t r = 0; r < R; ++r) {
    std::vector<int64_t> av(alloc.begin() + (size_t)r * T, alloc.begin() + (size_t)r * T + T), cv(cap.begin() + (size_t)r * T, cap.begin() + (size_t)r * T + T);
    auto uniq = [](std::vector<int64_t> v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); return v; };
    std::vector<int64_t> ua = uniq(av), uc = uniq(cv);
    thr_off[r] = (uint32_t)thr_val.size(); thr_n[r] = (uint32_t)ua.size();
    for (int64_t x : ua) {
      thr_val.push_back(x);
      size_t base = thr_mask.size();
      thr_mask.resize(base + tw, 0);
      for (uint32_t t = 0; t < T; ++t) if (av[t] >= x) thr_mask[base + (t >> 6)] |= 1ull << (t & 63);
    }
    cthr_off[r] = (uint32_t)cthr_val.size(); cthr_n[r] = (uint32_t)uc.size();
    for (int64_t x : uc) {
      cthr_val.push_back(x);
      size_t base = cthr_mask.size();
      cthr_mask.resize(base + tw, 0);
      for (uint32_t t = 0; t < T; ++t) if (cv[t] <= x) cthr_mask[base + (t >> 6)] |= 1ull << (t & 63);
    }
  }
  // per key value->types masks, rows in the catalog's own value numbering
  const uint32_t K = (uint32_t)V.keys.size();
  C.key_row_off.assign(K, UINT32_MAX);
  C.key_nvals.assign(K, 0);
  C.kU.assign(K, std::vector<uint64_t>(tw, 0));
  C.kN.assign(K, C.all_types);
  C.cvals.assign(K, {});
  C.cmap.assign(K, {});
  std::vector<uint64_t> twv;
  std::vector<uint8_t> mentioned(K, 0);
  for (auto& t : cat->types) {
    for (auto& r : t.reqs.rs) { mentioned[r.key] = 1; C.cvals[r.key].insert(C.cvals[r.key].end(), r.vals.begin(), r.vals.end()); }
    for (auto& o : t.offerings) {
      // (offerings that are not available keep their values numbered: kg_catalog_update_offerings may bring them back)
      mentioned[ctx->kZone] = mentioned[ctx->kCT] = 1;
      C.cvals[ctx->kZone].push_back(o.zone); C.cvals[ctx->kCT].push_back(o.ct);
      if (o.zid >= 0) { mentioned[ctx->kZid] = 1; C.cvals[ctx->kZid].push_back((uint32_t)o.zid); }
    }
  }
  for (uint32_t k = 0; k < K; ++k) {
    auto& cv = C.cvals[k];
    std::sort(cv.begin(), cv.end());
    cv.erase(std::unique(cv.begin(), cv.end()), cv.end());
    std::sort(cv.begin(), cv.end(), [&](uint32_t a, uint32_t b) { return V.vals[k].strs[a] < V.vals[k].strs[b]; });
    for (uint32_t l = 0; l < cv.size(); ++l) C.cmap[k].emplace(cv[l], l);
  }
  for (uint32_t k = 0; k < K; ++k) {
    if (!mentioned[k]) continue;
    const uint32_t nv = (uint32_t)C.cvals[k].size();
    C.key_row_off[k] = (uint32_t)(twv.size() / tw);
    C.key_nvals[k] = nv;
    size_t base = twv.size();
    twv.resize(base + (size_t)nv * tw, 0);
    for (uint32_t t = 0; t < T; ++t) {
      const HReq* r = cat->types[t].reqs.find(k);
      if (!r) continue;
      C.kN[k][t >> 6] &= ~(1ull << (t & 63));
      if (r->vals.empty()) C.kU[k][t >> 6] |= 1ull << (t & 63);
      for (uint32_t v : r->vals) twv[base + (size_t)C.local(k, v) * tw + (t >> 6)] |= 1ull << (t & 63);
    }
  }
  // offerings
  const uint32_t nz = (uint32_t)C.cvals[ctx->kZone].size(), nct = (uint32_t)C.cvals[ctx->kCT].size();
  if (nz > 64 || nct > 4 || C.cvals[ctx->kZid].size() > 64) throw Unsupported("catalog with more than 64 zones, 64 zone ids or 4 capacity types");
  C.n_zone_vals = nz; C.n_ct_vals = nct;
  for (auto& t : cat->types)
    for (auto& o : t.offerings) { o.lz = (uint32_t)C.local(ctx->kZone, o.zone); o.lct = (uint32_t)C.local(ctx->kCT, o.ct); o.lzid = o.zid < 0 ? -1 : C.local(ctx->kZid, (uint32_t)o.zid); }
  std::vector<uint64_t> offmask((size_t)64 * 4 * tw, 0);
  std::vector<uint8_t> zone_zid(64, 0xFF);
  std::vector<uint32_t> off_begin(T + 1, 0);
  std::vector<uint16_t> off_zone, off_ct, off_zid;
  std::vector<double> off_price;
  std::vector<uint8_t> off_avail;
  for (uint32_t t = 0; t < T; ++t) {
    off_begin[t] = (uint32_t)off_zone.size();
    for (auto& o : cat->types[t].offerings) {
      if (o.available) offmask[((size_t)o.lz * 4 + o.lct) * tw + (t >> 6)] |= 1ull << (t & 63);
      if (o.lzid >= 0) {
        if (zone_zid[o.lz] != 0xFF && zone_zid[o.lz] != (uint8_t)o.lzid) throw Unsupported("zone with two zone ids");
        zone_zid[o.lz] = (uint8_t)o.lzid;
      }
      off_zone.push_back((uint16_t)o.lz); off_ct.push_back((uint16_t)o.lct);
      off_zid.push_back(o.lzid < 0 ? 0xFFFF : (uint16_t)o.lzid);
      off_price.push_back(o.price); off_avail.push_back(o.available);
    }
  }
  off_begin[T] = (uint32_t)off_zone.size();
  // an offering without zone id next to offerings of the same zone with one would break the zone->zid fold
  for (uint32_t t = 0; t < T; ++t)
    for (auto& o : cat->types[t].offerings)
      if (o.lzid < 0 && zone_zid[o.lz] != 0xFF) throw Unsupported("zone id present on some offerings of a zone only");
  std::vector<uint16_t> name_rank(std::max(T, 1u), 0);
  {
    std::vector<uint32_t> idx(T);
    for (uint32_t t = 0; t < T; ++t) idx[t] = t;
    // (same-named types of a merged view keep their index order: what a stable sort by name gives)
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cat->types[a].name != cat->types[b].name ? cat->types[a].name < cat->types[b].name : a < b; });
    for (uint32_t i = 0; i < T; ++i) name_rank[idx[i]] = (uint16_t)i;
  }
  Arena& A = C.image;
  C.o_thr_val = A.add(thr_val); C.o_thr_off = A.add(thr_off); C.o_thr_n = A.add(thr_n); C.o_thr_mask = A.add(thr_mask);
  C.o_cthr_val = A.add(cthr_val); C.o_cthr_off = A.add(cthr_off); C.o_cthr_n = A.add(cthr_n); C.o_cthr_mask = A.add(cthr_mask);
  C.o_valid = A.add(valid); C.o_cap = A.add(cap); C.o_alloc = A.add(alloc); C.o_twv = A.add(twv, tw);
  C.o_offmask = A.add(offmask); C.o_zone_zid = A.add(zone_zid); C.o_name_rank = A.add(name_rank);
  C.o_off_begin = A.add(off_begin); C.o_off_zone = A.add(off_zone); C.o_off_ct = A.add(off_ct); C.o_off_zid = A.add(off_zid);
  C.o_off_price = A.add(off_price); C.o_off_avail = A.add(off_avail);
  C.dev.bytes = A.bytes.size();
  C.dev.ptr = hal::dmalloc(C.dev.bytes);
  hal::h2d(C.dev.ptr, A.bytes.data(), C.dev.bytes);
  ctx->stats.device_bytes_resident += C.dev.bytes;
}

// In-place maintenance (kg_catalog_update_offerings): host model first, then the words of the device tables that differ.
void update_offerings(kg_ctx* ctx, kg_catalog* cat, const kg_strtab& strings, const kg_offering_update* ups, uint32_t n) {
  Vocab& V = ctx->V;
  CatalogTables& C = cat->tab;
  const uint32_t keys3[3] = {ctx->kZone, ctx->kCT, ctx->kZid};
  // validate everything before touching anything
  struct Hit { uint32_t t; size_t o; };
  std::vector<Hit> hits;
  for (uint32_t i = 0; i < n; ++i) {
    const kg_offering_update& u = ups[i];
    kgdesc::check(u.type_index < cat->types.size(), "offering update: type index");
    const int z = V.vals[ctx->kZone].find(std::string(kgdesc::str(strings, u.zone))), c = V.vals[ctx->kCT].find(std::string(kgdesc::str(strings, u.capacity_type)));
    size_t found = SIZE_MAX;
    const auto& offs = cat->types[u.type_index].offerings;
    for (size_t o = 0; o < offs.size(); ++o) if ((int)offs[o].zone == z && (int)offs[o].ct == c) found = o;
    kgdesc::check(found != SIZE_MAX, "offering update names an offering the catalog does not have");
    hits.push_back({u.type_index, found});
  }
  const bool resident = C.dev.ptr != nullptr;
  const uint32_t tw = C.TW;
  std::vector<size_t> dirty;  // byte offsets of 8-byte words of the image that changed
  auto set_bit = [&](size_t word_off, uint32_t t, bool on) {
    if (!resident) return;
    uint64_t* w = (uint64_t*)(C.image.bytes.data() + word_off) + (t >> 6);
    const uint64_t nv = on ? (*w | (1ull << (t & 63))) : (*w & ~(1ull << (t & 63)));
    if (nv != *w) { *w = nv; dirty.push_back(word_off + (size_t)(t >> 6) * 8); }
  };
  std::vector<uint32_t> touched;
  for (uint32_t i = 0; i < n; ++i) {
    CatOffering& o = cat->types[hits[i].t].offerings[hits[i].o];
    if (ups[i].available >= 0) o.available = ups[i].available != 0;
    if (ups[i].price >= 0) o.price = ups[i].price;
    touched.push_back(hits[i].t);
  }
  std::sort(touched.begin(), touched.end());
  touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
  for (uint32_t t : touched) {
    CatType& ty = cat->types[t];
    // offering availability per (zone, capacity type)
    std::set<std::pair<uint32_t, uint32_t>> avail;
    for (auto& o : ty.offerings) if (o.available) avail.insert({o.zone, o.ct});
    for (auto& o : ty.offerings) set_bit(C.o_offmask + ((size_t)o.lz * 4 + o.lct) * tw * 8, t, avail.count({o.zone, o.ct}) != 0);
    // computeRequirements: zone / capacity-type / zone-id In {values of the available offerings}
    for (int k3 = 0; k3 < 3; ++k3) {
      const uint32_t key = keys3[k3];
      std::vector<uint32_t> vals;
      for (auto& o : ty.offerings) {
        if (!o.available) continue;
        if (k3 == 0) vals.push_back(o.zone);
        else if (k3 == 1) vals.push_back(o.ct);
        else if (o.zid >= 0) vals.push_back((uint32_t)o.zid);
      }
      std::sort(vals.begin(), vals.end());
      vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
      HReq* cur = nullptr;
      for (auto& r : ty.reqs.rs) if (r.key == key) cur = &r;
      // the zone-id requirement exists exactly while some available offering carries a zone id (types.go:110-117); zone
      // and capacity type are always listed (types.go:80-88), with no value when nothing is available
      const bool want = k3 != 2 ? cur != nullptr : !vals.empty();
      if (want && !cur) { ty.reqs.add(V, hreq_new(V, key, KG_OP_IN, vals, -1)); for (auto& r : ty.reqs.rs) if (r.key == key) cur = &r; }
      if (!want && cur) { ty.reqs.rs.erase(ty.reqs.rs.begin() + (cur - ty.reqs.rs.data())); cur = nullptr; }
      if (cur) cur->vals = vals;
      if (key < C.key_row_off.size() && C.key_row_off[key] != UINT32_MAX) {
        for (uint32_t l = 0; l < C.key_nvals[key]; ++l)
          set_bit(C.o_twv + ((size_t)C.key_row_off[key] + l) * tw * 8, t, cur && std::binary_search(vals.begin(), vals.end(), C.cvals[key][l]));
        // a requirement left without values reads as DoesNotExist; a type without the requirement does not mention the key
        uint64_t &ku = C.kU[key][t >> 6], &kn = C.kN[key][t >> 6];
        if (cur && vals.empty()) ku |= 1ull << (t & 63); else ku &= ~(1ull << (t & 63));
        if (!cur) kn |= 1ull << (t & 63); else kn &= ~(1ull << (t & 63));
      }
    }
    // prices / availability columns of the offering table
    if (resident) {
      const uint32_t* off_begin = (const uint32_t*)(C.image.bytes.data() + C.o_off_begin);
      for (size_t oi = 0; oi < ty.offerings.size(); ++oi) {
        const size_t row = off_begin[t] + oi;
        double* pr = (double*)(C.image.bytes.data() + C.o_off_price) + row;
        uint8_t* av = C.image.bytes.data() + C.o_off_avail + row;
        if (*pr != ty.offerings[oi].price) { *pr = ty.offerings[oi].price; dirty.push_back(C.o_off_price + row * 8); }
        if (*av != (uint8_t)ty.offerings[oi].available) { *av = ty.offerings[oi].available; dirty.push_back((C.o_off_avail + row) & ~size_t(7)); }
      }
    }
  }
  std::sort(dirty.begin(), dirty.end());
  dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
  for (size_t off : dirty) hal::h2d((uint8_t*)C.dev.ptr + off, C.image.bytes.data() + off, std::min<size_t>(8, C.image.bytes.size() - off));
  ctx->stats.catalog_words_patched += dirty.size();
}

}  // namespace kg
