// solve_core.hpp -- the bin-packing solve loop as one workgroup executes it.
//
// Follows Scheduler.Solve / Scheduler.add / NodeClaim.Add / ExistingNode.Add / Topology of
// sigs.k8s.io/karpenter v0.37.1-0.20240715214311-a122e25df3de (reference go.mod:34; algorithm
// as restated in SURVEY.md Appendix A.3-A.7).  The three per-type predicates are the ones the
// reference spells out in-repo at pkg/cloudprovider/cloudprovider.go:276-281.
//
// Design (DESIGN.md has the long form):
//   * FFD is a chain of dependent placements, so ONE workgroup owns one Solve and keeps all of
//     its mutable state; parallelism is across the rows a placement must visit (thread per
//     existing node / per in-flight NodeClaim) and, for consolidation, across simulations
//     (workgroup per simulation).  Nothing crosses a CU inside a Solve, so the only barrier is
//     the workgroup barrier.
//   * Everything is a bitset over a vocabulary compiled on the host: requirement value sets per
//     label key, and -- the part that removes the per-type inner loop -- instance types
//     themselves: each NodeClaim carries a T-bit "remaining types" mask, and "compatible",
//     "fits" and "hasOffering" become ANDs with precomputed type masks (per requirement-set
//     masks from k_type_filter, per-resource threshold masks, per-(zone,capacity-type)
//     offering masks).  first-feasible-index = min over feasible rows in scan order.
#pragma once
#include "kg_hd.hpp"

namespace kg {

constexpr int KG_MAXR = 16;     // resources
constexpr int KG_MAXKA = 32;    // active label keys in one solve
constexpr int KG_MAXE = 8;      // entries (keys) per requirement set
constexpr int KG_MAXSPEC = 6;   // "special" 1-word keys: zone, capacity-type, zone-id, os + topology keys
constexpr uint32_t KG_NOIDX = 0xFFFFFFFFu;

// requirement meta word
constexpr uint32_t RM_PRESENT = 1u, RM_COMPL = 2u;
KG_HD uint32_t rm_gt(uint32_t m) { return (m >> 8) & 0xFF; }
KG_HD uint32_t rm_lt(uint32_t m) { return (m >> 16) & 0xFF; }
KG_HD uint32_t rm_make(bool compl_, uint32_t gt, uint32_t lt) { return RM_PRESENT | (compl_ ? RM_COMPL : 0) | (gt << 8) | (lt << 16); }

// operator classes
enum : int { OPC_IN = 0, OPC_NOTIN = 1, OPC_EXISTS = 2, OPC_DNE = 3 };
KG_HD int opclass(bool compl_, bool anybits) { return compl_ ? (anybits ? OPC_NOTIN : OPC_EXISTS) : (anybits ? OPC_IN : OPC_DNE); }
KG_HD bool op_neg(int opc) { return opc == OPC_NOTIN || opc == OPC_DNE; }

enum : int { TG_SPREAD = 0, TG_AFFINITY = 1, TG_ANTI = 2 };
enum : uint8_t { LF_SELF = 1, LF_INVERSE = 2 };  // list entry flags

struct Variant {
  uint32_t rs;        // requirement set: nodeSelector + heaviest preferred + first required term
  uint32_t rs_strict; // without the preferred term (topology pod domains)
  uint32_t tolset;
  uint32_t match_off, match_cnt;    // topology groups consulted by AddRequirements
  uint32_t record_off, record_cnt;  // topology groups updated by Record
  int32_t next;       // variant after one Preferences.Relax step, -1 if none
  uint32_t host_claim_ok;  // pod's hostname requirement admits a fresh placeholder hostname
  uint32_t pad;
};

struct ListEntry {
  uint16_t g;
  uint8_t flags;
  uint8_t pad;
  uint32_t poddom_lo, poddom_hi;  // strict pod requirement on the group's key as an allowed-values mask
};

struct Group {
  uint8_t type;       // TG_*
  uint8_t spec;       // special-slot index of the key, 0xFF = hostname
  uint16_t hidx;      // hostname-group index (spec == 0xFF)
  int32_t max_skew;
  int32_t min_domains;  // <0 = nil
  uint32_t filter_off, filter_cnt;  // nodeFilter requirement sets (spread only; cnt 0 = always matches)
};

// Read-only problem tables (device pointers).
struct DevProblem {
  int R, Ka, Wa, T, TW, NP, N, NW, P, S, G, GH, H, Cmax, nspec, NL;
  int slot_zone, slot_ct, slot_zid;  // special indices (sp) or -1
  uint32_t well_known_slots;         // bit per active slot
  int res_pr;                        // unused
  // active keys
  const uint16_t* aoff;   // [Ka] word offset of the slot in a requirement row
  const uint16_t* aw;     // [Ka] words
  const int8_t* slot2spec;// [Ka] special index or -1
  const uint8_t* spec_slot; // [nspec]
  const uint64_t* kU;     // [Ka][TW] types that list the key as DoesNotExist
  const uint64_t* kN;     // [Ka][TW] types that do not mention the key
  const uint32_t* ktwv_off; // [Ka] base (in TW-word rows) of the slot's value->types masks
  const uint16_t* ktwv_n;   // [Ka] number of values the catalog knows for the slot
  const uint64_t* twv;      // [rows][TW]
  const uint16_t* vrank;    // [nspec][64] lexicographic rank of each value (tie-breaks)
  // bounds (Gt / Lt) per slot
  const uint32_t* bnd_off;  // [Ka] base index into gtval/ltval (fixed stride 16 per slot)
  const int64_t* gtval;     // [Ka*16]
  const int64_t* ltval;     // [Ka*16]
  const uint64_t* gtmask;   // [Ka*16][aw] values v with int(v) > gtval
  const uint64_t* ltmask;   // [Ka*16][aw]
  const uint32_t* bndmask_off; // [Ka] word base of the slot's masks (16 * aw words each for gt and lt)
  // requirement sets
  const uint32_t* rs_off;   // [S+1]
  const uint8_t* e_slot;    // [E]
  const uint32_t* e_meta;   // [E]
  const uint32_t* e_bits;   // [E] word offset into rs_bits
  const uint64_t* rs_bits;
  const uint64_t* e_amask;  // [E][TW] k_type_filter output: types whose value satisfies the entry (| kN)
  // variants / pods
  const Variant* variants;
  const uint32_t* pod_variant0; // [P]
  const int64_t* preq;          // [R][P]
  const uint32_t* queue0;       // [P] pods in Queue order
  const ListEntry* lists;
  // templates (nodepools in weight order)
  const uint32_t* tmpl_rs;      // [NP]
  const int64_t* tmpl_req;      // [R][NP] daemon overhead
  const uint8_t* tmpl_has_limits; // [NP]
  const int64_t* tmpl_limits0;  // [R][NP] remaining after existing nodes; INT64_MAX where unlimited
  const uint8_t* tmpl_limit_res;// [R][NP] 1 where the limit names the resource
  const uint64_t* tol_tmpl;     // [TS] bit per template: tolerates its taints
  const uint32_t* tmpl_minv_off;// [NP+1] minValues entries
  const uint8_t* minv_slot;     // slot
  const uint32_t* minv_min;     // minimum
  // catalog tables
  const int64_t* thr_val; const uint32_t* thr_off; const uint32_t* thr_n; const uint64_t* thr_mask;  // alloc >= v
  const int64_t* cthr_val; const uint32_t* cthr_off; const uint32_t* cthr_n; const uint64_t* cthr_mask; // cap <= v
  const uint64_t* valid_types;  // [TW]
  const int64_t* cap;           // [R][T]
  const uint64_t* offmask;      // [64 zones][4 cts][TW] (dense over value ids of the zone / ct slots)
  const uint8_t* zone_zid;      // [64] zone value id -> zone-id value id, 0xFF none
  int n_zone_vals, n_ct_vals;
  // existing nodes (scan order)
  const int64_t* navail;        // [R][N]
  const uint64_t* nlab;         // [S][NW] label compat per requirement set (k_node_static)
  const uint64_t* ntol;         // [TS][NW]
  const uint16_t* ndom;         // [nspec][N] node's value id on each special key, 0xFFFF absent
  // topology
  const Group* groups;
  const uint32_t* filters;      // requirement-set ids
};

// Mutable state of ONE simulation (device pointers into that simulation's arena).
struct SimState {
  int64_t* nreq;      // [R][N]
  int64_t* c_req;     // [R][Cmax]
  uint64_t* c_mask;   // [TW][Cmax]
  uint64_t* c_bits;   // [Wa][Cmax]
  uint32_t* c_meta;   // [Ka][Cmax]
  uint32_t* c_npods;  // [Cmax]
  uint16_t* c_tmpl;   // [Cmax]
  uint32_t* c_stamp;  // [Cmax] step of the claim's creation / last pod: with c_npods it is the claim's place in the
                      // reference's (stable) sort.Slice order -- see claim_key()
  int64_t* remaining; // [R][NP]
  int32_t* gcount;    // [G][64]
  uint64_t* gdom;     // [G] registered domains
  uint64_t* gpos;     // [G] domains with count > 0
  int32_t* hcount;    // [GH][H]
  int32_t* hpos;      // [GH] hosts with count > 0
  uint32_t* queue;    // [P+1] ring
  uint32_t* last_len; // [P]
  uint32_t* last_epoch; // [P]
  uint32_t* pod_variant;// [P]
  uint8_t* pod_kind;  // [P]
  int32_t* pod_index; // [P]
  uint32_t* pod_reason; // [P]
  uint64_t* scratch;  // [TW][Cmax] minValues second pass
  uint32_t* header;   // [16]: 0 nclaims, 1 relaxations, 2..: counters
  uint64_t* counters; // [8]: visited_existing, visited_claims, visited_claim_types, new_claim_scans
};

struct Shared {
  uint32_t pod, variant, best, nclaims, qhead, qlen, epoch, done, placed, fresh_ok, hostname_ctr, seq, st_vis;
  unsigned long long bestkey, st_types;
  uint32_t reason;
  Variant v;
  int64_t preq[KG_MAXR];
  int npr;
  uint8_t pr_res[KG_MAXR];
};

// ------------------------------------------------------------------ requirement algebra
struct SReq {  // a requirement on a 1-word slot
  uint32_t meta;
  uint64_t word;
};

KG_HD uint32_t min_nz(uint32_t a, uint32_t b) { return a == 0 ? b : (b == 0 ? a : (a < b ? a : b)); }

// allowed-values mask of a 1-word requirement (Requirement.Has over the known vocabulary)
KG_HD uint64_t has_mask1(const DevProblem& P, int a, SReq r) {
  if (!(r.meta & RM_PRESENT)) return ~0ull;
  if (!(r.meta & RM_COMPL)) return r.word;
  uint64_t m = ~r.word;
  uint32_t gt = rm_gt(r.meta), lt = rm_lt(r.meta);
  if (gt) m &= P.gtmask[P.bndmask_off[a] + (gt - 1)];
  if (lt) m &= P.ltmask[P.bndmask_off[a] + (lt - 1)];
  return m;
}

// Requirement.Intersection on a 1-word slot; both present
KG_HD SReq intersect1(const DevProblem& P, int a, SReq x, SReq y) {
  bool xc = x.meta & RM_COMPL, yc = y.meta & RM_COMPL;
  uint32_t gt = rm_gt(x.meta) > rm_gt(y.meta) ? rm_gt(x.meta) : rm_gt(y.meta);
  uint32_t lt = min_nz(rm_lt(x.meta), rm_lt(y.meta));
  SReq r;
  if (gt && lt && P.gtval[P.bnd_off[a] + gt - 1] >= P.ltval[P.bnd_off[a] + lt - 1]) {
    r.meta = rm_make(false, 0, 0);
    r.word = 0;
    return r;
  }
  uint64_t w = (xc && yc) ? (x.word | y.word) : (xc ? (y.word & ~x.word) : (yc ? (x.word & ~y.word) : (x.word & y.word)));
  if (gt) w &= P.gtmask[P.bndmask_off[a] + (gt - 1)];
  if (lt) w &= P.ltmask[P.bndmask_off[a] + (lt - 1)];
  bool c = xc && yc;
  r.meta = c ? rm_make(true, gt, lt) : rm_make(false, 0, 0);
  r.word = w;
  return r;
}

// Generic (multi-word) intersection summary of claim slot `a` of claim c with entry e.
// Returns merged meta; *any = merged has any value bit; optionally writes merged bits.
template <bool WRITE>
KG_HD uint32_t intersect_claim_entry(const DevProblem& P, const SimState& S, int a, uint32_t c, uint32_t cm, uint32_t e, bool* any,
                                     bool* changed) {
  const int w = P.aw[a], off = P.aoff[a];
  const uint32_t pm = P.e_meta[e];
  const uint64_t* pb = P.rs_bits + P.e_bits[e];
  bool xc = cm & RM_COMPL, yc = pm & RM_COMPL;
  uint32_t gt = rm_gt(cm) > rm_gt(pm) ? rm_gt(cm) : rm_gt(pm);
  uint32_t lt = min_nz(rm_lt(cm), rm_lt(pm));
  bool dne = gt && lt && P.gtval[P.bnd_off[a] + gt - 1] >= P.ltval[P.bnd_off[a] + lt - 1];
  bool c2 = xc && yc && !dne;
  uint32_t meta = c2 ? rm_make(true, gt, lt) : rm_make(false, 0, 0);
  bool anyb = false, chg = meta != cm;
  for (int i = 0; i < w; ++i) {
    uint64_t x = S.c_bits[(size_t)(off + i) * P.Cmax + c], y = pb[i];
    uint64_t r = dne ? 0 : ((xc && yc) ? (x | y) : (xc ? (y & ~x) : (yc ? (x & ~y) : (x & y))));
    if (!dne && gt) r &= P.gtmask[P.bndmask_off[a] + (size_t)(gt - 1) * w + i];
    if (!dne && lt) r &= P.ltmask[P.bndmask_off[a] + (size_t)(lt - 1) * w + i];
    anyb |= r != 0;
    chg |= r != x;
    if (WRITE) S.c_bits[(size_t)(off + i) * P.Cmax + c] = r;
  }
  if (WRITE) S.c_meta[(size_t)a * P.Cmax + c] = meta;
  *any = anyb;
  *changed = chg;
  return meta;
}

// Requirements.Compatible(claim c, requirement set rs, AllowUndefinedWellKnownLabels) without merging.
KG_HD bool claim_compatible_rs(const DevProblem& P, const SimState& S, uint32_t c, uint32_t rs) {
  for (uint32_t e = P.rs_off[rs]; e < P.rs_off[rs + 1]; ++e) {
    int a = P.e_slot[e];
    uint32_t pm = P.e_meta[e];
    const uint64_t* pb = P.rs_bits + P.e_bits[e];
    bool panyb = false;
    for (int i = 0; i < P.aw[a]; ++i) panyb |= pb[i] != 0;
    int pop = opclass(pm & RM_COMPL, panyb);
    uint32_t cm = S.c_meta[(size_t)a * P.Cmax + c];
    if (!(cm & RM_PRESENT)) {
      if (!((P.well_known_slots >> a) & 1) && !op_neg(pop)) return false;
      continue;
    }
    bool canyb = false;
    for (int i = 0; i < P.aw[a]; ++i) canyb |= S.c_bits[(size_t)(P.aoff[a] + i) * P.Cmax + c] != 0;
    bool anyb, chg;
    uint32_t mm = intersect_claim_entry<false>(P, S, a, c, cm, e, &anyb, &chg);
    bool empty = !(mm & RM_COMPL) && !anyb;
    if (empty && !(op_neg(pop) && op_neg(opclass(cm & RM_COMPL, canyb)))) return false;
  }
  return true;
}

KG_HD int thr_lookup_ge(const int64_t* vals, int n, int64_t x) {  // first index with vals[i] >= x, n if none
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (vals[mid] >= x) hi = mid; else lo = mid + 1;
  }
  return lo;
}
KG_HD int thr_lookup_le(const int64_t* vals, int n, int64_t x) {  // last index with vals[i] <= x, -1 if none
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (vals[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

// ------------------------------------------------------------------ topology on 1-word keys
struct TopoOut {
  bool ok;
  uint64_t opts;
};

KG_HD int first_by_rank(const DevProblem& P, int sp, uint64_t m) {
  int best = -1;
  uint32_t br = 0xFFFFFFFFu;
  while (m) {
    int d = ctz64(m);
    m &= m - 1;
    uint32_t r = P.vrank[sp * 64 + d];
    if (r < br) { br = r; best = d; }
  }
  return best;
}

// TopologyGroup.Get for a key with <= 64 domains.  podDom / nodeDom are allowed-value masks.
KG_HD TopoOut topo_small(const DevProblem& P, const SimState& S, const Group& g, uint32_t gi, bool self, uint64_t podDom, uint64_t nodeDom) {
  const int sp = g.spec;
  const uint64_t gd = S.gdom[gi];
  const int32_t* cnt = S.gcount + (size_t)gi * 64;
  TopoOut out{false, 0};
  if (g.type == TG_SPREAD) {
    int32_t mn = 0x7FFFFFFF;
    int supported = 0;
    for (uint64_t m = gd & podDom; m; m &= m - 1) {
      int d = ctz64(m);
      supported++;
      if (cnt[d] < mn) mn = cnt[d];
    }
    if (g.min_domains >= 0 && supported < g.min_domains) mn = 0;
    int best = -1;
    int64_t bestc = 0;
    uint32_t bestr = 0;
    for (uint64_t m = gd & nodeDom; m; m &= m - 1) {
      int d = ctz64(m);
      int64_t c = (int64_t)cnt[d] + (self ? 1 : 0);
      if (c - mn <= g.max_skew) {
        uint32_t r = P.vrank[sp * 64 + d];
        if (best < 0 || c < bestc || (c == bestc && r < bestr)) { best = d; bestc = c; bestr = r; }
      }
    }
    if (best < 0) return out;
    out.ok = true;
    out.opts = 1ull << best;
    return out;
  }
  if (g.type == TG_AFFINITY) {
    uint64_t opts = gd & podDom & S.gpos[gi];
    if (!opts && self) {
      int a = first_by_rank(P, sp, gd & podDom & nodeDom);
      int b = first_by_rank(P, sp, gd & podDom);
      if (a >= 0) opts |= 1ull << a;
      if (b >= 0) opts |= 1ull << b;
    }
    out.ok = opts != 0;
    out.opts = opts;
    return out;
  }
  uint64_t opts = gd & ~S.gpos[gi] & podDom;
  out.ok = opts != 0;
  out.opts = opts;
  return out;
}

// hostname-keyed group for host slot h (an existing node or a NodeClaim).  Returns pass/fail.
KG_HD bool topo_host(const SimState& S, const DevProblem& P, const Group& g, bool self, uint32_t h) {
  int32_t c = S.hcount[(size_t)g.hidx * P.H + h];
  if (g.type == TG_SPREAD) return (int64_t)c + (self ? 1 : 0) <= g.max_skew;  // min is 0 for hostname
  if (g.type == TG_AFFINITY) return c > 0 || (self && S.hpos[g.hidx] == 0);
  return c == 0;
}

KG_HD void record_small(const SimState& S, uint32_t gi, int d) {
  S.gcount[(size_t)gi * 64 + d] += 1;
  S.gdom[gi] |= 1ull << d;
  S.gpos[gi] |= 1ull << d;
}
KG_HD void record_host(const SimState& S, const DevProblem& P, const Group& g, uint32_t h) {
  int32_t* p = &S.hcount[(size_t)g.hidx * P.H + h];
  if (*p == 0) S.hpos[g.hidx] += 1;
  *p += 1;
}

// Scan order of in-flight NodeClaims.  The reference re-sorts the slice by len(Pods) before every scan
// (canonical: stable).  One claim changes between two sorts, so the stable order has a closed form:
// claims with one pod stay in creation order; a claim that receives its (k+1)-th pod becomes the FIRST
// of the (k+1)-pod group.  key = pods | (1 pod ? creation step : reversed last-add step) | claim id.
KG_HD unsigned long long claim_key(uint32_t npods, uint32_t stamp, uint32_t c) {
  const uint32_t sk = npods <= 1 ? stamp : (0xFFFFFu - stamp);
  return ((unsigned long long)npods << 40) | ((unsigned long long)(sk & 0xFFFFFu) << 20) | c;
}

// ------------------------------------------------------------------ NodeClaim.Add for claim c
// fresh: the claim was just materialised from its template (all template keys still to be applied).
// Returns true when the pod fits; with COMMIT also updates the claim.
template <bool COMMIT>
KG_HD bool eval_claim(const DevProblem& P, const SimState& S, const Shared& sh, uint32_t c, bool fresh, uint32_t* reason,
                      uint32_t* types_visited) {
  const Variant& v = sh.v;
  const uint32_t tmpl = S.c_tmpl[c];
  if (!((P.tol_tmpl[v.tolset] >> tmpl) & 1)) { *reason |= 1; return false; }
  if (!v.host_claim_ok) { *reason |= 2; return false; }
  const uint32_t e0 = P.rs_off[v.rs], ne = P.rs_off[v.rs + 1] - e0;
  const uint32_t t0 = P.rs_off[P.tmpl_rs[tmpl]], nt = P.rs_off[P.tmpl_rs[tmpl] + 1] - t0;

  // merged view of the special (1-word) slots: start from the claim's own state
  SReq sp[KG_MAXSPEC];
  bool sp_changed[KG_MAXSPEC];
  bool sp_dne_ok[KG_MAXSPEC];
#pragma unroll
  for (int i = 0; i < KG_MAXSPEC; ++i) {
    sp[i].meta = 0; sp[i].word = 0; sp_changed[i] = false; sp_dne_ok[i] = false;
    if (i < P.nspec) {
      int a = P.spec_slot[i];
      sp[i].meta = S.c_meta[(size_t)a * P.Cmax + c];
      sp[i].word = (sp[i].meta & RM_PRESENT) ? S.c_bits[(size_t)P.aoff[a] * P.Cmax + c] : 0;
      sp_changed[i] = fresh && (sp[i].meta & RM_PRESENT);
    }
  }
  // step A: Requirements.Compatible(podReqs) + Add, entry by entry
  bool ent_dne_ok[KG_MAXE];  // merged operator in {NotIn, DoesNotExist}
  bool ent_changed[KG_MAXE];
#pragma unroll
  for (int j = 0; j < KG_MAXE; ++j) {
    ent_dne_ok[j] = false; ent_changed[j] = false;
    if ((uint32_t)j < ne) {
      const uint32_t e = e0 + j;
      const int a = P.e_slot[e];
      const uint32_t pm = P.e_meta[e];
      const uint64_t* pb = P.rs_bits + P.e_bits[e];
      const int w = P.aw[a];
      bool panyb = false;
      for (int i = 0; i < w; ++i) panyb |= pb[i] != 0;
      const int pop = opclass(pm & RM_COMPL, panyb);
      const uint32_t cm = S.c_meta[(size_t)a * P.Cmax + c];
      const int spi = P.slot2spec[a];
      if (!(cm & RM_PRESENT)) {
        if (!((P.well_known_slots >> a) & 1) && !op_neg(pop)) { *reason |= 2; return false; }
        ent_dne_ok[j] = op_neg(pop);
        ent_changed[j] = true;
        if (spi >= 0) {
#pragma unroll
          for (int i = 0; i < KG_MAXSPEC; ++i) if (i == spi) { sp[i].meta = pm; sp[i].word = pb[0]; sp_changed[i] = true; }
        }
        if (COMMIT) {
          for (int i = 0; i < w; ++i) S.c_bits[(size_t)(P.aoff[a] + i) * P.Cmax + c] = pb[i];
          S.c_meta[(size_t)a * P.Cmax + c] = pm;
        }
      } else {
        bool canyb = false;
        for (int i = 0; i < w; ++i) canyb |= S.c_bits[(size_t)(P.aoff[a] + i) * P.Cmax + c] != 0;
        const int cop = opclass(cm & RM_COMPL, canyb);
        bool anyb, chg;
        const uint32_t mm = intersect_claim_entry<COMMIT>(P, S, a, c, cm, e, &anyb, &chg);
        const bool empty = !(mm & RM_COMPL) && !anyb;
        if (empty && !(op_neg(pop) && op_neg(cop))) { *reason |= 2; return false; }
        ent_dne_ok[j] = op_neg(opclass(mm & RM_COMPL, anyb));
        ent_changed[j] = chg || fresh;
        if (spi >= 0) {
#pragma unroll
          for (int i = 0; i < KG_MAXSPEC; ++i)
            if (i == spi) {
              SReq x{cm, sp[i].word}, y{pm, pb[0]};
              SReq r = intersect1(P, a, x, y);
              sp_changed[i] = sp_changed[i] || r.word != sp[i].word || r.meta != sp[i].meta;
              sp[i] = r;
            }
        }
      }
    }
  }
  // step B: Topology.AddRequirements -> Compatible -> Add
  const uint32_t hslot = P.N + c;
  if (v.match_cnt) {
    uint64_t topo[KG_MAXSPEC];
    bool has_topo[KG_MAXSPEC];
#pragma unroll
    for (int i = 0; i < KG_MAXSPEC; ++i) { topo[i] = ~0ull; has_topo[i] = false; }
    for (uint32_t li = 0; li < v.match_cnt; ++li) {
      const ListEntry le = P.lists[v.match_off + li];
      const Group g = P.groups[le.g];
      const bool self = le.flags & LF_SELF;
      if (g.spec == 0xFF) {
        if (!topo_host(S, P, g, self, hslot)) { *reason |= 4; return false; }
        continue;
      }
      uint64_t podDom = ((uint64_t)le.poddom_hi << 32) | le.poddom_lo;
      uint64_t nodeDom = ~0ull;
#pragma unroll
      for (int i = 0; i < KG_MAXSPEC; ++i) if (i == g.spec) nodeDom = has_mask1(P, P.spec_slot[i], sp[i]);
      TopoOut o = topo_small(P, S, g, le.g, self, podDom, nodeDom);
      if (!o.ok) { *reason |= 4; return false; }
#pragma unroll
      for (int i = 0; i < KG_MAXSPEC; ++i) if (i == g.spec) { topo[i] &= o.opts; has_topo[i] = true; }
    }
#pragma unroll
    for (int i = 0; i < KG_MAXSPEC; ++i) {
      if (i < P.nspec && has_topo[i]) {
        const int a = P.spec_slot[i];
        SReq inc{rm_make(false, 0, 0), topo[i]};
        if (!(sp[i].meta & RM_PRESENT)) {
          // receiver lacks the key: allowed when well-known or the incoming operator is DoesNotExist
          if (!((P.well_known_slots >> a) & 1) && topo[i] != 0) { *reason |= 4; return false; }
          sp[i] = inc;
          sp_changed[i] = true;
        } else {
          SReq r = intersect1(P, a, sp[i], inc);  // incoming as AddRequirements builds it: node ∩ domains
          const bool empty = !(r.meta & RM_COMPL) && r.word == 0;
          const int eop = opclass(sp[i].meta & RM_COMPL, sp[i].word != 0);
          if (empty && !op_neg(eop)) { *reason |= 4; return false; }  // incoming is DoesNotExist-class when empty
          sp_changed[i] = sp_changed[i] || r.word != sp[i].word || r.meta != sp[i].meta;
          sp[i] = r;
        }
        if (COMMIT) {
          S.c_bits[(size_t)P.aoff[a] * P.Cmax + c] = sp[i].word;
          S.c_meta[(size_t)a * P.Cmax + c] = sp[i].meta;
        }
      }
    }
  }
  // step C: filterInstanceTypesByRequirements in type-mask space
  int64_t newreq[KG_MAXR];
  int thr_idx[KG_MAXR];
  bool thr_use[KG_MAXR];
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) {
    thr_use[r] = false; thr_idx[r] = 0; newreq[r] = 0;
    if (r < P.R) {
      const int64_t cur = S.c_req[(size_t)r * P.Cmax + c];
      const int64_t add = sh.preq[r];
      newreq[r] = cur + add;
      if ((add != 0 || fresh) && newreq[r] > 0) {
        thr_use[r] = true;
        thr_idx[r] = thr_lookup_ge(P.thr_val + P.thr_off[r], (int)P.thr_n[r], newreq[r]);
      }
    }
  }
  // offerings: recompute when zone / capacity-type / zone-id requirement moved (or fresh)
  bool off_use = fresh;
  uint64_t zmask = ~0ull, ctmask = ~0ull;
#pragma unroll
  for (int i = 0; i < KG_MAXSPEC; ++i) {
    if (i < P.nspec) {
      if (i == P.slot_zone || i == P.slot_ct || i == P.slot_zid) off_use = off_use || sp_changed[i];
      const uint64_t hm = has_mask1(P, P.spec_slot[i], sp[i]);
      if (i == P.slot_zone) zmask &= hm;
      if (i == P.slot_ct) ctmask = hm;
      if (i == P.slot_zid) {
        uint64_t zm = 0;
        for (int z = 0; z < P.n_zone_vals; ++z) {
          uint8_t zid = P.zone_zid[z];
          if (zid == 0xFF || ((hm >> zid) & 1)) zm |= 1ull << z;
        }
        zmask &= zm;
      }
      sp_dne_ok[i] = (sp[i].meta & RM_PRESENT) && op_neg(opclass(sp[i].meta & RM_COMPL, sp[i].word != 0));
    }
  }
  const bool minv = P.tmpl_minv_off[tmpl + 1] > P.tmpl_minv_off[tmpl];
  bool any = false;
  uint32_t visited = 0;
  for (int w = 0; w < P.TW; ++w) {
    uint64_t m = S.c_mask[(size_t)w * P.Cmax + c];
    visited += popc64(m);
    // pod entries on ordinary keys
#pragma unroll
    for (int j = 0; j < KG_MAXE; ++j) {
      if ((uint32_t)j < ne) {
        const uint32_t e = e0 + j;
        const int a = P.e_slot[e];
        if (P.slot2spec[a] < 0 && ent_changed[j]) {
          uint64_t f = P.e_amask[(size_t)e * P.TW + w];
          if (ent_dne_ok[j]) f |= P.kU[(size_t)a * P.TW + w];
          m &= f;
        }
      }
    }
    if (fresh) {
      // template entries on ordinary keys (merged operator class was computed if the pod shares the key)
      for (uint32_t k = 0; k < nt; ++k) {
        const uint32_t e = t0 + k;
        const int a = P.e_slot[e];
        if (P.slot2spec[a] >= 0) continue;
        uint64_t f = P.e_amask[(size_t)e * P.TW + w];
        bool shared = false, dne_ok = false;
#pragma unroll
        for (int j = 0; j < KG_MAXE; ++j)
          if ((uint32_t)j < ne && P.e_slot[e0 + j] == a) { shared = true; dne_ok = ent_dne_ok[j]; }
        if (!shared) {
          const uint64_t* tb = P.rs_bits + P.e_bits[e];
          bool anyb = false;
          for (int i = 0; i < P.aw[a]; ++i) anyb |= tb[i] != 0;
          dne_ok = op_neg(opclass(P.e_meta[e] & RM_COMPL, anyb));
        }
        if (dne_ok) f |= P.kU[(size_t)a * P.TW + w];
        m &= f;
      }
    }
    // special keys: types whose value set meets the merged allowed set
#pragma unroll
    for (int i = 0; i < KG_MAXSPEC; ++i) {
      if (i < P.nspec && sp_changed[i] && (sp[i].meta & RM_PRESENT)) {
        const int a = P.spec_slot[i];
        uint64_t hm = has_mask1(P, a, sp[i]);
        uint64_t f = P.kN[(size_t)a * P.TW + w];
        if (sp_dne_ok[i]) f |= P.kU[(size_t)a * P.TW + w];
        const int nv = P.ktwv_n[a];
        for (uint64_t mm = hm; mm; mm &= mm - 1) {
          int d = ctz64(mm);
          if (d < nv) f |= P.twv[((size_t)P.ktwv_off[a] + d) * P.TW + w];
        }
        m &= f;
      }
    }
    // fits
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) {
      if (thr_use[r]) {
        if (thr_idx[r] >= (int)P.thr_n[r]) m = 0;
        else m &= P.thr_mask[((size_t)P.thr_off[r] + thr_idx[r]) * P.TW + w];
      }
    }
    if (fresh) m &= P.valid_types[w];
    // hasOffering
    if (off_use) {
      uint64_t f = 0;
      for (int z = 0; z < P.n_zone_vals; ++z) {
        if (!((zmask >> z) & 1)) continue;
        for (int ct = 0; ct < P.n_ct_vals; ++ct)
          if ((ctmask >> ct) & 1) f |= P.offmask[((size_t)z * 4 + ct) * P.TW + w];
      }
      m &= f;
    }
    any |= m != 0;
    if (minv) S.scratch[(size_t)w * P.Cmax + c] = m;
    else if (COMMIT) S.c_mask[(size_t)w * P.Cmax + c] = m;
  }
  *types_visited = visited;
  if (any && minv) {
    // InstanceTypes.SatisfiesMinValues over the remaining set
    for (uint32_t k = P.tmpl_minv_off[tmpl]; k < P.tmpl_minv_off[tmpl + 1]; ++k) {
      const int a = P.minv_slot[k];
      uint32_t distinct = 0;
      for (int d = 0; d < P.ktwv_n[a]; ++d) {
        bool hit = false;
        for (int w = 0; w < P.TW; ++w) hit |= (P.twv[((size_t)P.ktwv_off[a] + d) * P.TW + w] & S.scratch[(size_t)w * P.Cmax + c]) != 0;
        distinct += hit;
      }
      if (distinct < P.minv_min[k]) { any = false; *reason |= 32; }
    }
    if (any && COMMIT)
      for (int w = 0; w < P.TW; ++w) S.c_mask[(size_t)w * P.Cmax + c] = S.scratch[(size_t)w * P.Cmax + c];
  }
  if (!any) { *reason |= 8; return false; }
  if (COMMIT) {
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) if (r < P.R) S.c_req[(size_t)r * P.Cmax + c] = newreq[r];
    S.c_npods[c] += 1;
    // special slots touched only through the pod entries were already written by intersect<COMMIT>;
    // Topology.Record
    for (uint32_t li = 0; li < v.record_cnt; ++li) {
      const ListEntry le = P.lists[v.record_off + li];
      const Group g = P.groups[le.g];
      if (!(le.flags & LF_INVERSE) && g.filter_cnt) {
        bool match = false;
        for (uint32_t f = 0; f < g.filter_cnt && !match; ++f) match = claim_compatible_rs(P, S, c, P.filters[g.filter_off + f]);
        if (!match) continue;
      }
      if (g.spec == 0xFF) { record_host(S, P, g, hslot); continue; }
      SReq r{0, 0};
#pragma unroll
      for (int i = 0; i < KG_MAXSPEC; ++i) if (i == g.spec) r = sp[i];
      if (!(r.meta & RM_PRESENT)) continue;
      if ((le.flags & LF_INVERSE) || g.type == TG_ANTI) {
        for (uint64_t mm = r.word; mm; mm &= mm - 1) record_small(S, le.g, ctz64(mm));
      } else if (!(r.meta & RM_COMPL) && popc64(r.word) == 1) {
        record_small(S, le.g, ctz64(r.word));
      }
    }
  }
  return true;
}

// ------------------------------------------------------------------ ExistingNode.Add for node n (scan position)
template <bool COMMIT>
KG_HD bool eval_node(const DevProblem& P, const SimState& S, const Shared& sh, uint32_t n, uint32_t* reason) {
  const Variant& v = sh.v;
  const uint32_t wi = n >> 6, bi = n & 63;
  if (!((P.ntol[(size_t)v.tolset * P.NW + wi] >> bi) & 1)) { *reason |= 1; return false; }
  // resources first (cheap reject), as upstream does
  for (int k = 0; k < sh.npr; ++k) {
    const int r = sh.pr_res[k];
    if (S.nreq[(size_t)r * P.N + n] + sh.preq[r] > P.navail[(size_t)r * P.N + n]) { *reason |= 64; return false; }
  }
  if (!((P.nlab[(size_t)v.rs * P.NW + wi] >> bi) & 1)) { *reason |= 2; return false; }
  for (uint32_t li = 0; li < v.match_cnt; ++li) {
    const ListEntry le = P.lists[v.match_off + li];
    const Group g = P.groups[le.g];
    const bool self = le.flags & LF_SELF;
    if (g.spec == 0xFF) {
      if (!topo_host(S, P, g, self, n)) { *reason |= 4; return false; }
      continue;
    }
    const uint16_t d = P.ndom[(size_t)g.spec * P.N + n];
    if (d == 0xFFFF) { *reason |= 4; return false; }  // (host compile rejects this case)
    const uint64_t podDom = ((uint64_t)le.poddom_hi << 32) | le.poddom_lo;
    TopoOut o = topo_small(P, S, g, le.g, self, podDom, 1ull << d);
    if (!o.ok || !((o.opts >> d) & 1)) { *reason |= 4; return false; }
  }
  if (COMMIT) {
    for (int k = 0; k < sh.npr; ++k) {
      const int r = sh.pr_res[k];
      S.nreq[(size_t)r * P.N + n] += sh.preq[r];
    }
    for (uint32_t li = 0; li < v.record_cnt; ++li) {
      const ListEntry le = P.lists[v.record_off + li];
      const Group g = P.groups[le.g];
      if (!(le.flags & LF_INVERSE) && g.filter_cnt) {
        bool match = false;
        for (uint32_t f = 0; f < g.filter_cnt && !match; ++f)
          match = (P.nlab[(size_t)P.filters[g.filter_off + f] * P.NW + wi] >> bi) & 1;
        if (!match) continue;
      }
      if (g.spec == 0xFF) { record_host(S, P, g, n); continue; }
      const uint16_t d = P.ndom[(size_t)g.spec * P.N + n];
      if (d != 0xFFFF) record_small(S, le.g, d);
    }
  }
  return true;
}

// ------------------------------------------------------------------ one Solve by one workgroup
// NT = threads in the workgroup.  `sh` lives in LDS.
template <int NT, bool STATS>
KG_HD void solve_block(const DevProblem& P, const SimState& S, Shared& sh, uint32_t n_queue) {
  KG_FOR_THREADS(tid, NT) {
    if (tid == 0) {
      sh.qhead = 0; sh.qlen = n_queue; sh.epoch = 1; sh.nclaims = S.header[0]; sh.done = 0; sh.hostname_ctr = 0; sh.seq = 0;
    }
  }
  KG_SYNC();
  const uint32_t QCAP = (uint32_t)P.P + 1;
  while (true) {
    // ---- Queue.Pop
    KG_FOR_THREADS(tid, NT) {
      if (tid == 0) {
        if (sh.qlen == 0) sh.done = 1;
        else {
          const uint32_t pod = S.queue[sh.qhead % QCAP];
          if (S.last_epoch[pod] == sh.epoch && S.last_len[pod] == sh.qlen) sh.done = 1;
          else {
            sh.pod = pod;
            sh.qhead = (sh.qhead + 1) % QCAP;
            sh.qlen -= 1;
            sh.variant = S.pod_variant[pod];
            sh.v = P.variants[sh.variant];
            sh.best = KG_NOIDX;
            sh.placed = 0;
            sh.reason = 0;
            int npr = 0;
            for (int r = 0; r < P.R; ++r) {
              sh.preq[r] = P.preq[(size_t)r * P.P + pod];
              if (sh.preq[r] != 0) sh.pr_res[npr++] = (uint8_t)r;
            }
            sh.npr = npr;
          }
        }
      }
    }
    KG_SYNC();
    if (sh.done) break;
    // ---- existing nodes, first feasible in scan order
    for (uint32_t base = 0; base < (uint32_t)P.N; base += NT) {
      KG_FOR_THREADS(tid, NT) {
        const uint32_t n = base + tid;
        if (n < (uint32_t)P.N) {
          uint32_t reason = 0;
          if (eval_node<false>(P, S, sh, n, &reason)) atomic_min_u32(&sh.best, n);
        }
      }
      KG_SYNC();
      if (sh.best != KG_NOIDX) break;
    }
    if (sh.best != KG_NOIDX) {
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) {
          uint32_t reason = 0;
          eval_node<true>(P, S, sh, sh.best, &reason);
          S.pod_kind[sh.pod] = 1;
          S.pod_index[sh.pod] = (int32_t)sh.best;
          if (STATS) S.counters[0] += sh.best + 1;
          sh.placed = 1;
        }
      }
      KG_SYNC();
      continue;
    }
    if (STATS) {
      KG_FOR_THREADS(tid, NT) { if (tid == 0) S.counters[0] += P.N; }
    }
    // ---- in-flight NodeClaims: first feasible in (pod count, stable) order = minimum claim_key
    KG_FOR_THREADS(tid, NT) {
      if (tid == 0) { sh.bestkey = ~0ull; sh.st_vis = 0; sh.st_types = 0; }
    }
    KG_SYNC();
    for (uint32_t base = 0; base < sh.nclaims; base += NT) {
      KG_FOR_THREADS(tid, NT) {
        const uint32_t c = base + tid;
        if (c < sh.nclaims) {
          uint32_t reason = 0, tv = 0;
          const bool ok = eval_claim<false>(P, S, sh, c, false, &reason, &tv);
          if (ok) atomic_min_u64(&sh.bestkey, claim_key(S.c_npods[c], S.c_stamp[c], c));
          if (STATS) S.scratch[(size_t)(P.TW) * P.Cmax + c] = (reason & ~(8u | 32u)) ? 0 : tv;  // reached the type filter
        }
      }
    }
    KG_SYNC();
    if (STATS) {
      // rows the reference visits: every claim up to and including the first feasible one
      for (uint32_t base = 0; base < sh.nclaims; base += NT) {
        KG_FOR_THREADS(tid, NT) {
          const uint32_t c = base + tid;
          if (c < sh.nclaims && claim_key(S.c_npods[c], S.c_stamp[c], c) <= sh.bestkey) {
            atomic_add_u32(&sh.st_vis, 1u);
            atomic_add_u64(&sh.st_types, S.scratch[(size_t)(P.TW) * P.Cmax + c]);
          }
        }
      }
      KG_SYNC();
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) { S.counters[1] += sh.st_vis; S.counters[2] += sh.st_types; }
      }
    }
    if (sh.bestkey != ~0ull) {
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) {
          const uint32_t c = (uint32_t)(sh.bestkey & 0xFFFFFu);
          uint32_t reason = 0, tv = 0;
          eval_claim<true>(P, S, sh, c, false, &reason, &tv);
          S.c_stamp[c] = ++sh.seq;
          S.pod_kind[sh.pod] = 2;
          S.pod_index[sh.pod] = (int32_t)c;
          sh.placed = 1;
        }
      }
      KG_SYNC();
      continue;
    }
    // ---- new NodeClaim from the templates, in weight order
    for (int t = 0; t < P.NP && !sh.placed; ++t) {
      const uint32_t c = sh.nclaims;
      if (c >= (uint32_t)P.Cmax) break;
      // filterByRemainingResources + NewNodeClaim: materialise claim c from template t
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) { sh.fresh_ok = 1; }
      }
      KG_SYNC();
      KG_FOR_THREADS(tid, NT) {
        for (int w = tid; w < P.TW; w += NT) {
          uint64_t m = ~0ull;
          if (w == P.TW - 1 && (P.T & 63)) m = (1ull << (P.T & 63)) - 1;
          if (P.tmpl_has_limits[t]) {
            for (int r = 0; r < P.R; ++r) {
              if (!P.tmpl_limit_res[(size_t)r * P.NP + t]) continue;
              const int idx = thr_lookup_le(P.cthr_val + P.cthr_off[r], (int)P.cthr_n[r], S.remaining[(size_t)r * P.NP + t]);
              if (idx < 0) m = 0; else m &= P.cthr_mask[((size_t)P.cthr_off[r] + idx) * P.TW + w];
            }
          }
          S.c_mask[(size_t)w * P.Cmax + c] = m;
        }
        for (int a = tid; a < P.Ka; a += NT) S.c_meta[(size_t)a * P.Cmax + c] = 0;
        for (int i = tid; i < P.Wa; i += NT) S.c_bits[(size_t)i * P.Cmax + c] = 0;
        for (int r = tid; r < P.R; r += NT) S.c_req[(size_t)r * P.Cmax + c] = P.tmpl_req[(size_t)r * P.NP + t];
        if (tid == 0) { S.c_npods[c] = 0; S.c_tmpl[c] = (uint16_t)t; }
      }
      KG_SYNC();
      KG_FOR_THREADS(tid, NT) {
        // template requirement entries -> claim state
        const uint32_t rs = P.tmpl_rs[t];
        for (uint32_t e = P.rs_off[rs] + tid; e < P.rs_off[rs + 1]; e += NT) {
          const int a = P.e_slot[e];
          S.c_meta[(size_t)a * P.Cmax + c] = P.e_meta[e];
          for (int i = 0; i < P.aw[a]; ++i) S.c_bits[(size_t)(P.aoff[a] + i) * P.Cmax + c] = P.rs_bits[P.e_bits[e] + i];
        }
      }
      KG_SYNC();
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) {
          bool anym = false;
          for (int w = 0; w < P.TW; ++w) anym |= S.c_mask[(size_t)w * P.Cmax + c] != 0;
          if (!anym) { sh.reason |= 16; }  // all available instance types exceed limits
          else {
            // hostname placeholder registered for this attempt: hcount rows of slot N+c start at zero
            for (int gh = 0; gh < P.GH; ++gh) S.hcount[(size_t)gh * P.H + P.N + c] = 0;
            uint32_t reason = 0, tv = 0;
            if (STATS) { S.counters[3] += 1; }
            const bool ok = eval_claim<true>(P, S, sh, c, true, &reason, &tv);
            if (STATS) { if (!(reason & ~(8u | 32u))) S.counters[2] += tv; }
            sh.reason |= reason;
            if (ok) {
              // subtractMax: charge the worst-case capacity of the remaining options against the limits
              if (P.tmpl_has_limits[t]) {
                for (int r = 0; r < P.R; ++r) {
                  if (!P.tmpl_limit_res[(size_t)r * P.NP + t]) continue;
                  int64_t mx = 0;
                  for (int w = 0; w < P.TW; ++w)
                    for (uint64_t mm = S.c_mask[(size_t)w * P.Cmax + c]; mm; mm &= mm - 1) {
                      const int64_t cv = P.cap[(size_t)r * P.T + (w * 64 + ctz64(mm))];
                      if (cv > mx) mx = cv;
                    }
                  S.remaining[(size_t)r * P.NP + t] -= mx;
                }
              }
              S.c_stamp[c] = ++sh.seq;
              sh.nclaims += 1;
              S.pod_kind[sh.pod] = 2;
              S.pod_index[sh.pod] = (int32_t)c;
              sh.placed = 1;
            }
          }
        }
      }
      KG_SYNC();
    }
    if (sh.placed) continue;
    // ---- Preferences.Relax + Queue.Push
    KG_FOR_THREADS(tid, NT) {
      if (tid == 0) {
        const uint32_t pod = sh.pod;
        S.pod_kind[pod] = 0;
        S.pod_index[pod] = -1;
        S.pod_reason[pod] = sh.reason ? sh.reason : 8;
        S.queue[(sh.qhead + sh.qlen) % QCAP] = pod;
        sh.qlen += 1;
        if (sh.v.next >= 0) {
          S.pod_variant[pod] = (uint32_t)sh.v.next;
          sh.epoch += 1;  // lastLen = {} on a relaxed push
          S.header[1] += 1;
        } else {
          S.last_len[pod] = sh.qlen;
          S.last_epoch[pod] = sh.epoch;
        }
      }
    }
    KG_SYNC();
  }
  KG_FOR_THREADS(tid, NT) {
    if (tid == 0) S.header[0] = sh.nclaims;
  }
  KG_SYNC();
}

}  // namespace kg

// ------------------------------------------------------------------ batch compatibility scans
// (the part of the path that is a plain data-parallel scan: requirement sets x instance types and
// requirement sets x existing nodes).  One call = one output word / one output bit.
namespace kg {

struct PrepassTables {
  // k_type_filter
  uint32_t E;
  // k_node_static
  const uint16_t* nval;        // [Ka][N] node label value id per active slot, 0xFFFF = label absent
  const uint32_t* rs_host_meta;// [S] hostname requirement meta (0 = none)
  const uint32_t* rs_host_off; // [S+1]
  const uint32_t* rs_host_nodes; // node scan positions named by the hostname requirement
  uint64_t* e_amask_out;       // [E][TW]
  uint64_t* nlab_out;          // [S][NW]
};

KG_HD bool entry_has(const DevProblem& P, uint32_t e, int a, uint32_t v) {
  const uint32_t meta = P.e_meta[e];
  const uint64_t* bits = P.rs_bits + P.e_bits[e];
  const int w = P.aw[a];
  if ((int)(v >> 6) >= w) return (meta & RM_COMPL) && !rm_gt(meta) && !rm_lt(meta);  // value unknown to this solve's rows
  const bool bit = (bits[v >> 6] >> (v & 63)) & 1;
  if (!(meta & RM_COMPL)) return bit;
  if (bit) return false;
  const uint32_t gt = rm_gt(meta), lt = rm_lt(meta);
  if (gt && !((P.gtmask[P.bndmask_off[a] + (size_t)(gt - 1) * w + (v >> 6)] >> (v & 63)) & 1)) return false;
  if (lt && !((P.ltmask[P.bndmask_off[a] + (size_t)(lt - 1) * w + (v >> 6)] >> (v & 63)) & 1)) return false;
  return true;
}

// types (word w of the T-bit mask) whose value on the entry's key satisfies the entry, plus types
// that do not mention the key at all
KG_HD uint64_t type_filter_word(const DevProblem& P, uint32_t e, int w) {
  const int a = P.e_slot[e];
  uint64_t acc = P.kN[(size_t)a * P.TW + w];
  const uint32_t nv = P.ktwv_n[a], row0 = P.ktwv_off[a];
  for (uint32_t v = 0; v < nv; ++v)
    if (entry_has(P, e, a, v)) acc |= P.twv[((size_t)row0 + v) * P.TW + w];
  return acc;
}

// Requirements.Compatible(node labels + hostname, requirement set s) with no undefined-label allowance
KG_HD bool node_static_bit(const DevProblem& P, const PrepassTables& X, uint32_t s, uint32_t n) {
  for (uint32_t e = P.rs_off[s]; e < P.rs_off[s + 1]; ++e) {
    const int a = P.e_slot[e];
    const uint16_t vid = X.nval[(size_t)a * P.N + n];
    if (vid == 0xFFFF) {
      const uint64_t* bits = P.rs_bits + P.e_bits[e];
      bool anyb = false;
      for (int i = 0; i < P.aw[a]; ++i) anyb |= bits[i] != 0;
      if (!op_neg(opclass(P.e_meta[e] & RM_COMPL, anyb))) return false;
    } else if (!entry_has(P, e, a, vid)) {
      return false;
    }
  }
  const uint32_t hm = X.rs_host_meta[s];
  if (hm & RM_PRESENT) {
    bool listed = false;
    for (uint32_t i = X.rs_host_off[s]; i < X.rs_host_off[s + 1]; ++i) listed |= X.rs_host_nodes[i] == n;
    if (hm & RM_COMPL) {
      if (listed || rm_gt(hm) || rm_lt(hm)) return false;  // hostnames are not integers: bounds never hold
    } else if (!listed) {
      return false;
    }
  }
  return true;
}

}  // namespace kg
