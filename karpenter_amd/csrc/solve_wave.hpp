// solve_wave.hpp -- Scheduler.Solve by ONE wavefront (device only).
//
// The workgroup-wide loop of solve_core.hpp pays a barrier, an LDS hand-over and the skew of four instruction streams at
// every step of a placement.  When a Solve has no topology groups, no existing nodes and no minValues NodePool (a plain
// provisioning batch: BASELINE config C2) nothing in a placement needs more than 64 lanes at a time, so one wave runs the
// whole chain with no barrier at all:
//   * prefilter: lane = scan position, 64 positions per step (headroom, option mask AND pod mask, toleration, host
//     ports, and the pod's zone / capacity-type style entries against the claim's words); the ballot is the survivor
//     list in scan order;
//   * NodeClaim.Add on the survivors in order, the claim row distributed over the lanes (wave_add<> below: the same
//     arithmetic as row_merge_entry_from / row_merge_resource / row_word_a / row_word_b of solve_core.hpp);
//   * commit straight from registers; the scan order is rotated in place.
// Same state layout, same results, same canonical order as solve_block (tests compare both with the oracle).
#pragma once
#include "solve_core.hpp"

namespace kg {

// Keeps a uniform value in a VECTOR register: the kernel is bound by scalar registers (about 100 per wave, against the
// ~90 sizes and pointers of DevProblem / SimState plus the loop state), and every scalar the compiler cannot hold is
// spilled to a VGPR lane and reloaded with v_readlane; values that only feed vector instructions cost nothing as VGPRs.
#define KG_VREG64(x) asm volatile("" : "+v"(x))

// The descriptors of a launch (DevProblem in the kernel-argument segment, the SimState of this Solve in the state image) hold
// about 130 pointers and sizes; a placement that lands on an in-flight NodeClaim needs about twenty of them.  Read through
// the by-value copies, every field is loaded at the kernel's entry and occupies scalar registers for the whole Solve (the
// compiler then parks them in VGPR lanes and pays a v_readlane per use).  The paths that run once per Solve or once per
// NodeClaim (table staging, a new NodeClaim, a false survivor, Preferences.Relax, the final columns) therefore read their
// fields through a pointer in the CONSTANT address space that is made opaque where the path begins: the scalar loads stay on
// that path, and only the two base pointers live across the placement loop.
#define KG_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const T& rare_view(const KG_CONST T* p) {
  asm volatile("" : "+s"(p));
  return *(const T*)p;
}

// has_mask1 (solve_core.hpp) with the bound masks read through the rare view
__device__ __forceinline__ uint64_t has_mask1_rare(const KG_CONST DevProblem* Pk, int a, SReq r) {
  if (!(r.meta & RM_PRESENT)) return ~0ull;
  if (!(r.meta & RM_COMPL)) return r.word;
  uint64_t m = ~r.word;
  const uint32_t gt = rm_gt(r.meta), lt = rm_lt(r.meta);
  if (__builtin_expect((gt | lt) != 0, 0)) {
    const DevProblem& Pb = rare_view(Pk);
    if (gt) m &= Pb.gtmask[Pb.bndmask_off[a] + (gt - 1)];
    if (lt) m &= Pb.ltmask[Pb.bndmask_off[a] + (lt - 1)];
  }
  return m;
}

// Step 3-6 of NodeClaim.Add with the claim row in registers: lane a holds meta[a], lane i bits[i], lanes 8r the requests of
// resource r, lane w word w of the option mask.  FRESH: the row is a NodeClaim just made from a template (every entry
// counts as changed, template entries and the valid-type mask join the filter, hasOffering is always evaluated).
// Returns "the option mask stays non-empty"; *compat = the requirement merge went through.
// what a lane needs of the small problem tables in every NodeClaim.Add: constant over the Solve, read from LDS once
struct LaneTabs {
  uint32_t spa, spoff;       // lane i < nspec: active slot of 1-word key i and its word offset in a claim row
  uint32_t tn, tbase, toff;  // lanes [8r, 8r + 8): number of fit thresholds of resource r, their base in sh.thr_val, in the catalog
};
// P: the problem as the caller reads it (FRESH: its rare view); Pk: the same struct for the branches that are rare even on
// the hot path (Gt / Lt bounds, entries whose value words did not fit the pod context, negative operators, zone ids).
template <bool FRESH>
__device__ __forceinline__ bool wave_add(const DevProblem& P, const KG_CONST DevProblem* Pk, const SharedBase& sh, const uint64_t* thr_rows, const LaneTabs& lt, const PodCtx& pc,
                                         uint32_t lane, uint32_t need, uint32_t tmpl, uint32_t& meta, uint64_t& bits, int64_t& req, uint64_t& m, bool* compat) {
  const uint32_t NR = (uint32_t)P.R, TW = (uint32_t)P.TW;
  const uint32_t rg = lane >> 3;
  uint4 pe4;
  pe4.x = pe4.y = pe4.z = pe4.w = 0;
  if (lane < KG_MAXE) pe4 = ((const uint4*)pc.vc.ent)[lane];
  const uint32_t spa = lt.spa, spoff = lt.spoff, tn = lt.tn, tbase = lt.tbase, toff = lt.toff;
  int64_t add = 0;
  if (rg < NR) add = pc.preq[rg];
  const uint32_t ne = KG_UNI(pc.vc.ne);
  bool ok = true;
  // ---- step 3: the pod's requirement entries, one after the other (distinct keys)
  uint32_t ent_changed = 0, ent_dne_ok = 0, sp_changed = 0, ent_ord = 0;
  for (uint32_t j = 0; j < ne && ok; ++j) {
    const uint32_t pm = rl32(pe4.y, j), ab = rl32(pe4.z, j), sw = rl32(pe4.w, j);
    const uint32_t aoff = ab & 0xFFFF, pbits = ab >> 16, a = sw & 0xFF, w = (sw >> 16) & 0xFF, opc = sw >> 24;
    const int spec = (int)(int8_t)((sw >> 8) & 0xFF);
    if (spec < 0) ent_ord |= 1u << j;
    const uint32_t cma = rl32(meta, a);
    const bool inr = lane >= aoff && lane < aoff + w;
    uint64_t y = 0;
    if (pbits != 0xFFFF) { if (inr) { y = pc.vc.ebits[pbits + lane - aoff]; KG_OPAQUE64(y); } }
    else { const DevProblem& Pb = rare_view(Pk); if (inr) y = Pb.rs_bits[Pb.e_bits[rl32(pe4.x, j)] + lane - aoff]; }
    if (!(cma & RM_PRESENT)) {
      if (!((P.well_known_slots >> a) & 1) && !op_neg((int)opc)) { ok = false; break; }
      if (op_neg((int)opc)) ent_dne_ok |= 1u << j;
      ent_changed |= 1u << j;
      if (spec >= 0) sp_changed |= 1u << spec;
      if (inr) bits = y;
      if (lane == a) meta = pm;
      continue;
    }
    const bool xc = cma & RM_COMPL, yc = pm & RM_COMPL;
    const uint32_t gt = rm_gt(cma) > rm_gt(pm) ? rm_gt(cma) : rm_gt(pm);
    const uint32_t lt = min_nz(rm_lt(cma), rm_lt(pm));
    bool dne = false;
    const uint64_t x = bits;
    uint64_t r = x;
    if (inr) r = (xc && yc) ? (x | y) : (xc ? (y & ~x) : (yc ? (x & ~y) : (x & y)));
    if (__builtin_expect((gt | lt) != 0, 0)) {  // Gt / Lt bounds on the key
      const DevProblem& Pb = rare_view(Pk);
      dne = gt && lt && Pb.gtval[Pb.bnd_off[a] + gt - 1] >= Pb.ltval[Pb.bnd_off[a] + lt - 1];
      if (inr) {
        if (dne) r = 0;
        if (!dne && gt) r &= Pb.gtmask[Pb.bndmask_off[a] + (size_t)(gt - 1) * w + (lane - aoff)];
        if (!dne && lt) r &= Pb.ltmask[Pb.bndmask_off[a] + (size_t)(lt - 1) * w + (lane - aoff)];
      }
    }
    const uint32_t mm = (xc && yc && !dne) ? rm_make(true, gt, lt) : rm_make(false, 0, 0);
    const bool anyb = __ballot(inr && r != 0) != 0, canyb = __ballot(inr && x != 0) != 0;
    const bool chg = mm != cma || __ballot(inr && r != x) != 0;
    if (inr) bits = r;
    if (lane == a) meta = mm;
    const int cop = opclass(xc, canyb);
    const bool empty = !(mm & RM_COMPL) && !anyb;
    if (empty && !(op_neg((int)opc) && op_neg(cop))) { ok = false; break; }
    if (op_neg(opclass(mm & RM_COMPL, anyb))) ent_dne_ok |= 1u << j;
    if (chg || FRESH) ent_changed |= 1u << j;
    if (spec >= 0 && chg) sp_changed |= 1u << spec;
  }
  *compat = ok;
  if (!ok) return false;
  // ---- step 5: requests including the pod; every fit threshold at once, 8-ary search in the resource's lanes
  const int64_t nr = req + add;
  req = nr;
  const bool ruse = rg < NR && (FRESH || ((need >> rg) & 1)) && nr > 0;
  uint32_t tidx = 0;
  {
    uint32_t lo = 0;
    const uint32_t j8 = lane & 7, sh8 = lane & 56u;
    for (uint32_t stride = KG_UNI(sh.thr_stride0); stride; stride >>= 3) {
      const uint32_t q = lo + j8 * stride;
      bool f = true;
      if (q < tn) f = sh.thr_val[tbase + q] >= nr;
      const uint32_t gb = (uint32_t)(__ballot(f) >> sh8) & 0xFFu;
      const uint32_t L = gb ? (uint32_t)__builtin_ctz(gb) : 8u;
      if (stride == 1) lo = lo + L;
      else if (L != 0) lo = lo + (L - 1) * stride + 1;
    }
    tidx = lo;
  }
  const unsigned long long useb = __ballot(ruse && (lane & 7) == 0);  // bit 8r: resource r decides
  if (__ballot(ruse && tidx >= tn) != 0) return false;                // a request above every instance type
  const uint32_t trow = toff + (tidx < tn ? tidx : 0), lrow = tbase + (tidx < tn ? tidx : 0);
  // ---- step 6: the instance-type filter on lanes [0, TW)
  const uint32_t w = lane < TW ? lane : 0;
  uint64_t tmk[KG_MAXR];
  if (thr_rows) {
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) tmk[r] = thr_rows[rl32(lrow, 8 * r) * TW + w];  // (row 0 for unused lanes)
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) KG_OPAQUE64(tmk[r]);  // (after the last load is issued: one round trip, not eight)
  } else {
    const DevProblem& Pt = rare_view(Pk);  // (the fit-threshold rows did not fit the LDS pool)
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) tmk[r] = Pt.thr_mask[(size_t)rl32(trow, 8 * r) * TW + w];  // (unconditional: one round trip)
  }
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) tmk[r] = ((useb >> (8 * r)) & 1) ? tmk[r] : ~0ull;
  for (uint32_t em = ent_changed & ent_ord; em; em &= em - 1) {
    const uint32_t j = (uint32_t)__builtin_ctz(em);
    uint64_t f = pc.eam[j][w];
    KG_OPAQUE64(f);
    if ((ent_dne_ok >> j) & 1) { const DevProblem& Pb = rare_view(Pk); f |= Pb.kU[(size_t)(rl32(pe4.w, j) & 0xFF) * TW + w]; }
    m &= f;
  }
  if (FRESH) {
    // template entries on ordinary keys (the merged operator class is the pod entry's when the pod shares the key)
    const uint32_t trs = P.tmpl_rs[tmpl];
    const uint32_t t0 = P.rs_off[trs], t1 = P.rs_off[trs + 1];
    for (uint32_t e = t0; e < t1; ++e) {
      const uint32_t a = P.e_slot[e];
      if (sh.slot2spec[a] >= 0) continue;
      uint64_t f = P.e_amask[(size_t)e * TW + w];
      bool shared = false, dne_ok = false;
      for (uint32_t j = 0; j < ne; ++j)
        if ((rl32(pe4.w, j) & 0xFF) == a) { shared = true; dne_ok = (ent_dne_ok >> j) & 1; }
      if (!shared) {
        const uint64_t* tb = P.rs_bits + P.e_bits[e];
        bool anyb = false;
        for (int i = 0; i < sh.aw[a]; ++i) anyb |= tb[i] != 0;
        dne_ok = op_neg(opclass(P.e_meta[e] & RM_COMPL, anyb));
      }
      if (dne_ok) f |= P.kU[(size_t)a * TW + w];
      m &= f;
    }
    m &= P.valid_types[w];
  }
  if (sp_changed || FRESH) {
    bool off_use = FRESH;
    uint64_t zmask = ~0ull, ctmask = ~0ull;
    for (int i = 0; i < P.nspec; ++i) {
      const uint32_t a = rl32(spa, (uint32_t)i);
      SReq sp;
      sp.meta = rl32(meta, a);
      const bool present = sp.meta & RM_PRESENT;
      sp.word = present ? rl64(bits, rl32(spoff, (uint32_t)i)) : 0;
      const bool changed = ((sp_changed >> i) & 1) || (FRESH && present);
      const uint64_t hm = has_mask1_rare(Pk, (int)a, sp);
      if (changed && present) {
        uint64_t f = P.kN[(size_t)a * TW + w];
        if (op_neg(opclass(sp.meta & RM_COMPL, sp.word != 0))) { const DevProblem& Pb = rare_view(Pk); f |= Pb.kU[(size_t)a * TW + w]; }
        {
          // (KG_SPV rows at most, every load issued before the first use)
          uint64_t g[KG_SPV];
          const uint64_t hq = hm & ((1ull << KG_SPV) - 1);
#pragma unroll
          for (int v = 0; v < (int)KG_SPV; ++v) g[v] = sh.stw[i][v][w];
#pragma unroll
          for (int v = 0; v < (int)KG_SPV; ++v) { KG_OPAQUE64(g[v]); f |= ((hq >> v) & 1) ? g[v] : 0ull; }
        }
        m &= f;
      }
      if (i == P.slot_zone || i == P.slot_ct || i == P.slot_zid) off_use = off_use || changed;
      if (i == P.slot_zone) zmask &= hm;
      if (i == P.slot_ct) ctmask = hm;
      if (i == P.slot_zid) {
        uint64_t zm = 0;
        const DevProblem& Pb = rare_view(Pk);
        for (int z = 0; z < P.n_zone_vals; ++z) {
          const uint8_t zid = Pb.zone_zid[z];
          if (zid == 0xFF || ((hm >> zid) & 1)) zm |= 1ull << z;
        }
        zmask &= zm;
      }
    }
    if (off_use) {
      uint64_t f = 0;
      for (int z = 0; z < P.n_zone_vals; ++z) {
        if (!((zmask >> z) & 1)) continue;
        for (int ct = 0; ct < P.n_ct_vals; ++ct)
          if ((ctmask >> ct) & 1) { uint64_t g = sh.offm[z * P.n_ct_vals + ct][w]; KG_OPAQUE64(g); f |= g; }
      }
      m &= f;
    }
  }
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) m &= tmk[r];
  return __ballot(lane < TW && m != 0) != 0;
}

// max over the set types of mask (word w in lane w) of tab[type], for the types' table row `tab` of T entries.
// Every load is unconditional (clamped index, selected afterwards): sixteen independent loads, one round trip.
__device__ __forceinline__ int64_t wave_max_over_mask(const int64_t* tab, uint64_t mask, uint32_t TW, uint32_t T, uint32_t lane) {
  int64_t v[KG_TWLDS];
#pragma unroll
  for (int w = 0; w < KG_TWLDS; ++w) {
    const uint32_t idx = ((uint32_t)w < TW ? (uint32_t)w : 0u) * 64 + lane;
    v[w] = tab[idx < T ? idx : T - 1];
  }
  int64_t mx = 0;
#pragma unroll
  for (int w = 0; w < KG_TWLDS; ++w) {
    const uint64_t mw = rl64(mask, (uint32_t)w);  // (zero beyond TW words)
    const int64_t x = ((mw >> lane) & 1) ? v[w] : 0;
    mx = x > mx ? x : mx;
  }
  return wave_max_i64(mx);
}

// State of a NodeClaim in the single-wave solve: ONE contiguous row per claim (SimState::c_row, 896 bytes) instead of
// the column arrays of the workgroup kernel -- the prefilter reads its first three cache lines with consecutive 16-byte
// loads, NodeClaim.Add reads / writes the rest with loads that are consecutive across the lanes, and the kernel keeps one
// pointer instead of ten.  The columns the host decodes (c_mask, c_bits, c_meta, c_req, c_npods, c_tmpl, c_stamp) are
// written from the rows once, when the Solve is over.  Words (8 bytes each):
//   [0..3]   headroom of active resources 0..3 (upper bound of what the claim can still take)
//   [4][5]   the claim's allowed values on the zone / capacity-type keys when they are plain non-empty sets, else all ones
//   [6]      template | pod count << 32          [7] order stamp | stamp of the last exact headroom << 32
//   [8..23]  option mask (zero beyond TW words)
//   [24..31] requests per active resource        [32..47] requirement meta words (32 x 4 bytes)     [48..111] value bits
// The prefilter part (words 0..23): a pod can only land on a claim with headroom >= requests, option mask AND the pod's
// own type mask, zone / capacity-type words AND the pod's, a tolerated template -- each a necessary condition of
// NodeClaim.Add.  One lane tests one claim: twelve 16-byte loads from consecutive addresses, one round trip.
constexpr int KG_ROW_WORDS = 112, KG_ROW_MASK = 8, KG_ROW_REQ = 24, KG_ROW_META = 32, KG_ROW_BITS = 48;
static_assert(KG_TWLDS == 16 && KG_MAXR == 8 && KG_MAXKA == 32 && KG_ROWW == 64, "row layout");
// allowed-values word of a 1-word requirement for the record: the set itself when it is a plain non-empty set
__device__ __forceinline__ uint64_t plain_set_or_all(uint32_t meta, uint64_t word) {
  return ((meta & (RM_PRESENT | RM_COMPL)) == RM_PRESENT && word != 0) ? word : ~0ull;
}

// The prefilter part of the first `pfn` NodeClaims lives in LDS as well: record c = words [0..5] of the row (headroom x 4,
// zone and capacity-type sets), the template, and the TW words of the option mask (rounded up to an even count: 16-byte
// records).  A lane always reads sixteen mask words: the ones beyond the record belong to the next record and are cancelled
// by the zero padding of the pod's own mask.  The records share one LDS pool with the fit-threshold rows; claims beyond `pfn`
// are tested against their global rows as before.  What this buys is small (C2 k_solve_wave 52.1 -> 50.2 ms): measured,
// a prefilter step is bound by the wave's own instruction stream, not by the round trip.  Tried and dropped: a 128-bit
// folded signature instead of the mask (8 adjacent types per bit: instance families straddle the buckets, the false
// survivors doubled the exact evaluations: 69 ms); whole rows in LDS with the per-pod results stored 64 at a time and the
// next pod's context requested behind the prefilter (no global load or store left in a placement, but half as many
// instructions again and twice the scalar-register spills: 54-55 ms); three helper waves (a 256-thread workgroup, one wave per
// SIMD) running the prefilter of the NEXT pod against every claim while this wave evaluates NodeClaim.Add, the ballots left in
// LDS as a bit vector over scan positions that is repaired after the commit (the touched claim moves and is re-tested; two
// LDS-only barriers per placement): bit-exact, but 53.2 ms against 50.8 -- with this wave ignoring the vectors the hand-over
// alone cost 6.8 ms (the helpers scan every claim where the scan here stops at the first feasible one, and their LDS
// traffic lengthens each of this wave's ~40 exposed LDS round trips), using them won back only 4.2.
constexpr uint32_t KG_PF_HEAD = 8;  // words before the mask: h0..h3, zone, ct, template, pad
constexpr uint32_t KG_POOL_WORDS = (160 * 1024 - (uint32_t)sizeof(SharedBase)) / 8 - 2;

// scan order of the single-wave solve: one array, rotated in place (LDS while it fits, global memory beyond)
constexpr uint32_t KG_WORD = 2 * KG_ORD;
// (lds: every position in use is below KG_WORD -- a uniform fact the callers pass down, so that the common case has no
// per-lane branch and no global load to wait for)
// (S: the state descriptor behind its rare view -- the global half of the order is only touched beyond 2 048 NodeClaims)
__device__ __forceinline__ unsigned long long word_get(const KG_CONST SimState* S, const SharedBase& sh, bool lds, uint32_t p) {
  if (lds || p < KG_WORD) { unsigned long long v = (&sh.ord[0][0])[lds ? (p & (KG_WORD - 1)) : p]; KG_OPAQUE64(v); return v; }
  return rare_view(S).ordg[p];
}
__device__ __forceinline__ void word_set(const KG_CONST SimState* S, SharedBase& sh, bool lds, uint32_t p, unsigned long long v) {
  if (lds || p < KG_WORD) (&sh.ord[0][0])[lds ? (p & (KG_WORD - 1)) : p] = v; else rare_view(S).ordg[p] = v;
}
// end of the k-pod group: first position whose claim holds more than k pods (64-ary search)
__device__ __forceinline__ uint32_t word_group_end(const KG_CONST SimState* S, const SharedBase& sh, bool lds, uint32_t nclaims, uint32_t k, uint32_t lane) {
  uint32_t stride = 1;
  while ((unsigned long long)stride * 64 < nclaims) stride *= 64;
  uint32_t lo = 0;
  while (true) {
    const unsigned long long q = (unsigned long long)lo + (unsigned long long)lane * stride;
    bool f = true;
    if (q < nclaims) f = (uint32_t)(word_get(S, sh, lds, (uint32_t)q) >> 32) > k;
    const unsigned long long b = __ballot(f);
    const uint32_t L = b ? (uint32_t)(__ffsll(b) - 1) : 64u;
    if (stride == 1) return lo + L;
    if (L == 0) return lo;
    lo = lo + (L - 1) * stride + 1;
    stride /= 64;
  }
}

// one Queue.Pop's worth of pod context by one wave, from the contiguous image k_pod_ctx built (one level of loads)
__device__ __forceinline__ void ctx_fetch_image(const DevProblem& P, PodCtx& nx, uint32_t lane, uint32_t pod, uint32_t variant) {
  const uint64_t* img = (const uint64_t*)(P.pctx + variant);
  uint64_t* dst = (uint64_t*)&nx;
  int64_t rq = 0;
  if (lane < (uint32_t)P.R) rq = P.preq[(size_t)pod * P.R + lane];
  const unsigned long long nb = __ballot(rq != 0);
  for (uint32_t i = lane; i < sizeof(PodCtx) / 8; i += 64) {
    uint64_t v = img[i];
    if (i < KG_MAXR) v = (uint64_t)rq;
    if (i == KG_MAXR) v = (uint64_t)(uint32_t)nb;
    dst[i] = v;
  }
}

// The whole Solve.  Preconditions (checked by the host before it picks this kernel): no topology groups, no existing
// nodes, no minValues, TW <= KG_TWLDS, every table stages into LDS, visit counters off.
// DBG: with the per-section cycle accounting (its counters would otherwise cost the production kernel two dozen scalar registers)
// P, S: the by-value copies (the placement loop's fields live in scalar registers); Pk, Sk: the same structs in memory for
// the rare paths (rare_view above).
template <bool DBG>
__device__ __forceinline__ void solve_wave(const DevProblem& P, const SimState& S, const KG_CONST DevProblem* Pk, const KG_CONST SimState* Sk, SharedBase& sh, uint64_t* pool,
                                           uint32_t n_queue) {
  const uint32_t lane = threadIdx.x & 63;
  // cycle accounting per section of a placement (stats & 8; debug): registers, written once at the end
  constexpr bool WT = DBG;
  unsigned long long wt_last = 0, wt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) wt[i] = 0;
#define KG_W(k) do { if (WT) { const unsigned long long n_ = clock64(); if (wt_last) wt[k] += n_ - wt_last; wt_last = n_; } } while (0)
  // ... and inside "new NodeClaim" (also counted there): limits + template row, NodeClaim.Add, row stores, headroom / limits, order
  unsigned long long wn_last = 0, wn[5] = {0, 0, 0, 0, 0};
#define KG_WN(k) do { if (WT) { const unsigned long long n_ = clock64(); if ((k) >= 0) wn[(k) < 0 ? 0 : (k)] += n_ - wn_last; wn_last = n_; } } while (0)
  stage_tables<64>(rare_view(Pk), sh);
  const uint32_t NR = (uint32_t)P.R, TW = (uint32_t)P.TW;
  const uint32_t rg = lane >> 3;
  // one LDS pool: the fit-threshold rows of the active resources (in the order of sh.thr_val) when they take at most
  // half of it, then the prefilter records of as many NodeClaims as fit behind them
  const uint64_t* thr_rows = nullptr;
  uint32_t pf_off = 0;
  {
    uint32_t tot = 0;
    for (uint32_t r = 0; r < NR; ++r) tot += sh.a_thr_n[r];
    tot = KG_UNI(tot);
    if (tot * TW <= KG_POOL_WORDS / 2) {
      const uint64_t* thr_mask = rare_view(Pk).thr_mask;
      for (uint32_t r = 0; r < NR; ++r) {
        const uint32_t n = KG_UNI(sh.a_thr_n[r]), b = KG_UNI(sh.thr_base[r]), o = KG_UNI(sh.a_thr_off[r]);
        for (uint32_t i = lane; i < n * TW; i += 64) pool[b * TW + i] = thr_mask[(size_t)o * TW + i];
      }
      thr_rows = pool;
      pf_off = (tot * TW + 1) & ~1u;
      __syncthreads();
    }
  }
  const uint32_t pfw = KG_PF_HEAD + ((TW + 1) & ~1u);                  // words per prefilter record
  const uint32_t pfn = (KG_POOL_WORDS - pf_off - KG_TWLDS) / pfw;     // records that fit (a lane reads KG_TWLDS mask words whatever TW is)
  uint64_t* const pf = pool + pf_off;
  LaneTabs lt;
  lt.spa = lt.spoff = lt.tn = lt.tbase = lt.toff = 0;
  if (lane < (uint32_t)P.nspec) { lt.spa = sh.spec_slot[lane]; lt.spoff = sh.aoff[lt.spa]; }
  if (rg < NR) { lt.tn = sh.a_thr_n[rg]; lt.tbase = sh.thr_base[rg]; lt.toff = sh.a_thr_off[rg]; }
  const int zslot = P.slot_zone >= 0 ? (int)sh.spec_slot[P.slot_zone] : -1, cslot = P.slot_ct >= 0 ? (int)sh.spec_slot[P.slot_ct] : -1;
  const uint32_t zoff = zslot >= 0 ? (uint32_t)sh.aoff[zslot] : 0, coff = cslot >= 0 ? (uint32_t)sh.aoff[cslot] : 0;
  const uint32_t QCAP = S.qcap;
  uint32_t qhead = 0, qlen = n_queue, epoch = 1, nclaims = 0, seq = 0, cur = 0;
  uint32_t last_c = KG_NOIDX, last_npods = 0, last_stamp = 0, last_new = 0;
  bool have_ctx = false, have_q2 = false;
  uint32_t nq_pod = 0, nq_variant = 0, nq_z = 0, nq_w = 0;
  uint4 q2;
  q2.x = q2.y = q2.z = q2.w = 0;
  while (qlen != 0) {
#ifndef KG_NO_LOOP_VIEW
    // (the descriptors through per-placement views: their fields' scalar loads stay inside the placement -- solve_core.hpp)
    const DevProblem& P = rare_view(Pk);
    const SimState& S = rare_view(Sk);
#endif
    // ---- Queue.Pop
    KG_W(7);
    // (the head's entry and context were fetched during the previous placement unless the queue was too short then)
    uint4 qv;
    if (have_ctx) { qv.x = nq_pod; qv.y = nq_variant; qv.z = nq_z; qv.w = nq_w; }
    else qv = ((const KG_GLOBAL uint4*)S.queue)[qhead];
    const uint32_t pod = KG_UNI(qv.x), variant = KG_UNI(qv.y), qz = KG_UNI(qv.z), qw = KG_UNI(qv.w);
    if (qz == epoch && qw == qlen) break;  // every remaining pod failed with the queue at this length
    qhead = qhead + 1 == QCAP ? 0 : qhead + 1;
    qlen -= 1;
    cur ^= 1;
    if (__builtin_expect(!have_ctx, 0)) ctx_fetch_image(P, sh.pcb[cur], lane, pod, variant);
    const PodCtx& pc = sh.pcb[cur];
    // what the previous placement's lanes stored is read by other lanes from here on.  (Before the loads below are
    // issued: the wait is for everything outstanding, and those loads are only needed when this placement is over.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // ---- start on the next pod: its queue entry (loaded one placement ago, or now), its context image and requests go to
    // registers now and to the other context buffer when this placement is over; the entry after it is requested as well
    const bool next_ok = qlen >= 1;
    uint32_t n_pod = 0, n_variant = 0, n_z = 0, n_w = 0;
    uint64_t iw[4] = {0, 0, 0, 0};
    int64_t n_rq = 0;
    if (next_ok) {
      uint4 ne4 = q2;
      if (!have_q2) ne4 = ((const KG_GLOBAL uint4*)S.queue)[qhead];
      n_pod = KG_UNI(ne4.x); n_variant = KG_UNI(ne4.y); n_z = KG_UNI(ne4.z); n_w = KG_UNI(ne4.w);
      const uint64_t* img = (const uint64_t*)(P.pctx + n_variant);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const uint32_t i = lane + 64 * k; iw[k] = img[i < sizeof(PodCtx) / 8 ? i : 0]; }
      if (lane < NR) n_rq = P.preq[(size_t)n_pod * NR + lane];
      have_q2 = qlen >= 2;
      if (have_q2) q2 = ((const KG_GLOBAL uint4*)S.queue)[qhead + 1 == QCAP ? 0 : qhead + 1];
    } else {
      have_q2 = false;
    }
    const uint32_t need = KG_UNI(pc.need);
    uint64_t tolw = pc.tolw, hp_conf = P.has_ports ? pc.vc.hp_conf : 0ull;  // (uniform, but only vector instructions use them)
    KG_VREG64(tolw); KG_VREG64(hp_conf);
    const bool hcok = KG_UNI(pc.vc.v.host_claim_ok) != 0;
    // the pod's side of the prefilter test (the sets were compiled on the host)
    uint64_t pzone = pc.vc.pzone, pct = pc.vc.pct;
    KG_VREG64(pzone); KG_VREG64(pct);
    last_c = KG_NOIDX; last_new = 0;
    uint32_t reason = 0;
    bool placed = false;
    const bool lds = nclaims < KG_WORD;  // the scan order (with the claim this placement may add) fits LDS
    const bool pf_all = nclaims <= pfn && lds;  // every claim's prefilter record (and the whole scan order) is in LDS
    KG_W(0);
    // the pod's side of the prefilter, in vector registers (the same value in every lane)
    int64_t prq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prq[r] = pc.preq[r];
    // ---- in-flight NodeClaims in scan order, 64 positions per step
    for (uint32_t base = 0; base < nclaims && !placed; base += 64) {
      const uint32_t pos = base + lane;
      bool ok = false;
      unsigned long long oe = 0;
      if (pos < nclaims && pf_all) {
        // every claim's prefilter record is in LDS: twelve 16-byte reads, no global round trip.  (The scan order is read
        // directly: word_get's global alternative would make the compiler wait for every outstanding memory load here.)
        oe = (&sh.ord[0][0])[pos];
        KG_OPAQUE64(oe);
        const uint32_t c = (uint32_t)oe;
        const uint4* rec = (const uint4*)(pf + (size_t)c * pfw);
        uint4 q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = rec[k];
        const int64_t h0 = (int64_t)(((uint64_t)q[0].y << 32) | q[0].x), h1 = (int64_t)(((uint64_t)q[0].w << 32) | q[0].z);
        const int64_t h2 = (int64_t)(((uint64_t)q[1].y << 32) | q[1].x), h3 = (int64_t)(((uint64_t)q[1].w << 32) | q[1].z);
        const uint64_t czone = ((uint64_t)q[2].y << 32) | q[2].x, cct = ((uint64_t)q[2].w << 32) | q[2].z;
        const uint32_t tm = q[3].x;
        uint64_t any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          any |= ((((uint64_t)q[4 + k].y << 32) | q[4 + k].x) & pc.podmask[2 * k]) | ((((uint64_t)q[4 + k].w << 32) | q[4 + k].z) & pc.podmask[2 * k + 1]);
        }
        uint32_t bad = (uint32_t)(((tolw >> tm) & 1) == 0) | (uint32_t)(any == 0) | (uint32_t)((czone & pzone) == 0) | (uint32_t)((cct & pct) == 0);
        bad |= ((uint32_t)(h0 < prq[0]) & need) | ((uint32_t)(h1 < prq[1]) & (need >> 1)) | ((uint32_t)(h2 < prq[2]) & (need >> 2)) | ((uint32_t)(h3 < prq[3]) & (need >> 3));
        bad &= 1u;
        if (P.has_ports) bad |= (uint32_t)((S.c_ports[c] & hp_conf) != 0);
        ok = hcok && bad == 0;
      } else if (pos < nclaims) {
        oe = word_get(Sk, sh, lds, pos);
        const uint32_t c = (uint32_t)oe;
        // the claim's prefilter record: twelve 16-byte loads from consecutive addresses, one round trip
        const KG_GLOBAL uint4* rec = (const KG_GLOBAL uint4*)(S.c_row + (size_t)c * KG_ROW_WORDS);
        uint4 q[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = rec[k];
        const int64_t h0 = (int64_t)(((uint64_t)q[0].y << 32) | q[0].x), h1 = (int64_t)(((uint64_t)q[0].w << 32) | q[0].z);
        const int64_t h2 = (int64_t)(((uint64_t)q[1].y << 32) | q[1].x), h3 = (int64_t)(((uint64_t)q[1].w << 32) | q[1].z);
        const uint64_t czone = ((uint64_t)q[2].y << 32) | q[2].x, cct = ((uint64_t)q[2].w << 32) | q[2].z;
        const uint32_t tm = q[3].x;
        uint64_t any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          any |= ((((uint64_t)q[4 + k].y << 32) | q[4 + k].x) & pc.podmask[2 * k]) | ((((uint64_t)q[4 + k].w << 32) | q[4 + k].z) & pc.podmask[2 * k + 1]);
        }
        uint32_t bad = (uint32_t)(((tolw >> tm) & 1) == 0) | (uint32_t)(any == 0) | (uint32_t)((czone & pzone) == 0) | (uint32_t)((cct & pct) == 0);
        bad |= ((uint32_t)(h0 < prq[0]) & need) | ((uint32_t)(h1 < prq[1]) & (need >> 1)) | ((uint32_t)(h2 < prq[2]) & (need >> 2)) | ((uint32_t)(h3 < prq[3]) & (need >> 3));
        bad &= 1u;
        if (P.has_ports) bad |= (uint32_t)((S.c_ports[c] & hp_conf) != 0);
        ok = hcok && bad == 0;
      }
      // ---- exact NodeClaim.Add on the survivors, in order
      const unsigned long long sv0 = __ballot(ok);
      KG_W(1);
      if (WT) { wt[8] += 1; wt[9] += (unsigned long long)__popcll(sv0); }
      for (unsigned long long sv = sv0; sv && !placed; sv &= sv - 1) {
        const uint32_t sl = (uint32_t)(__ffsll(sv) - 1);
        const unsigned long long woe = rl64(oe, sl);
        const uint32_t c = (uint32_t)woe, wnp = (uint32_t)(woe >> 32), wpos = base + sl;
        uint32_t meta = 0;
        uint64_t bits = 0, cm = 0;
        int64_t req = 0;
        KG_GLOBAL uint64_t* row = S.c_row + (size_t)c * KG_ROW_WORDS;
        if (lane < KG_MAXKA) meta = ((const KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane];
        bits = row[KG_ROW_BITS + lane];
        req = (int64_t)row[KG_ROW_REQ + rg];
        if (lane < KG_TWLDS) cm = row[KG_ROW_MASK + lane];
        int64_t phd = 0;  // lanes 0..3: the headroom of resources 0..3
        if (lane < 4) phd = (int64_t)row[lane];
        const uint64_t w6_v = row[6], w7_v = row[7];
        const uint32_t endk = word_group_end(Sk, sh, lds, nclaims, wnp, lane);  // (LDS work under the row's round trip)
        const uint64_t w6 = KG_UNI64(w6_v), w7 = KG_UNI64(w7_v);
        const uint32_t tm = (uint32_t)w6, stamp = (uint32_t)w7, hst = (uint32_t)(w7 >> 32);
        const int64_t req0 = req;
        uint64_t m = cm;
        bool compat = true;
        KG_W(2);
        const bool feasible = wave_add<false>(P, Pk, sh, thr_rows, lt, pc, lane, need, tm, meta, bits, req, m, &compat);
        KG_W(3);
        if (WT) wt[10] += 1;
        if (__builtin_expect(!feasible, 0)) {
          if (hst != stamp) {
            // the prefilter let an infeasible claim through: recompute its headroom (its options changed since)
            const DevProblem& Pr = rare_view(Pk);
            for (uint32_t r = 0; r < NR && r < 4; ++r) {
              const int64_t mx = wave_max_over_mask(Pr.alloc + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, cm, TW, (uint32_t)Pr.T, lane);
              const int64_t rq = (int64_t)rl64((uint64_t)req0, 8 * r);
              if (lane == 0) { row[r] = (uint64_t)(mx - rq); if (c < pfn) pf[(size_t)c * pfw + r] = (uint64_t)(mx - rq); }
            }
            if (lane == 0) row[7] = (uint64_t)stamp | ((uint64_t)stamp << 32);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the next survivor's loads may read this claim's lines)
          }
          KG_W(5);
          continue;
        }
        // ---- commit: merged row back to the claim; the claim moves to the end of its pod-count group (= the front of
        // the next one)
        if (lane < KG_MAXKA) ((KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane] = meta;
        row[KG_ROW_BITS + lane] = bits;
        if ((lane & 7) == 0) row[KG_ROW_REQ + rg] = (uint64_t)req;
        if (lane < KG_TWLDS) row[KG_ROW_MASK + lane] = m;  // (m is zero in lanes beyond TW)
        {
          // the prefilter part of the row after this pod
          const uint64_t zw = zslot >= 0 ? plain_set_or_all(rl32(meta, (uint32_t)zslot), rl64(bits, zoff)) : ~0ull;
          const uint64_t cw = cslot >= 0 ? plain_set_or_all(rl32(meta, (uint32_t)cslot), rl64(bits, coff)) : ~0ull;
          uint64_t v = (uint64_t)(phd - (lane < 4 ? pc.preq[lane < 4 ? lane : 0] : 0));
          const uint64_t n6 = (uint64_t)tm | ((uint64_t)(wnp + 1) << 32), n7 = (uint64_t)(seq + 1) | ((uint64_t)hst << 32);
          v = lane == 4 ? zw : (lane == 5 ? cw : (lane == 6 ? n6 : (lane == 7 ? n7 : v)));
          if (lane < 8) row[lane] = v;
          if (c < pfn) {  // the LDS copy of the prefilter part (template unchanged)
            uint64_t* rec = pf + (size_t)c * pfw;
            if (lane < 6) rec[lane] = v;
            if (lane < pfw - KG_PF_HEAD) rec[KG_PF_HEAD + lane] = m;  // (m is zero in lanes beyond TW)
          }
        }
        if (lane == 0) {
          S.pod_kind[pod_row(S, pod)] = 2;
          S.pod_index[pod_row(S, pod)] = (int32_t)c;
          if (P.has_ports) S.c_ports[c] = S.c_ports[c] | pc.vc.hp_use;
        }
        for (uint32_t p0 = wpos; p0 + 1 < endk; p0 += 64) {
          const uint32_t p = p0 + lane;
          unsigned long long v = 0;
          if (p + 1 < endk) v = word_get(Sk, sh, lds, p + 1);
          if (p + 1 < endk) word_set(Sk, sh, lds, p, v);
        }
        if (lane == 0) word_set(Sk, sh, lds, endk - 1, ((unsigned long long)(wnp + 1) << 32) | c);
        last_c = c; last_new = 0; last_npods = wnp; last_stamp = stamp;
        seq += 1;
        placed = true;
        KG_W(4);
      }
    }
    if (!placed) KG_W(1);
    // ---- a new NodeClaim from each template in weight order
    if (__builtin_expect(!placed, 0)) {
      const DevProblem& Pr = rare_view(Pk);
      const SimState& Sr = rare_view(Sk);
      const uint32_t NP = nclaims < (uint32_t)Pr.Cmax ? (uint32_t)Pr.NP : 0u;  // (no room for another NodeClaim: no template is tried)
      const uint32_t c = nclaims;
      for (uint32_t t = 0; t < NP && !placed; ++t) {
        KG_WN(-1);
        // filterByRemainingResources: types whose capacity fits what the NodePool's limits leave
        uint64_t cm = 0;
        if (lane < TW) {
          cm = ~0ull;
          if (lane == TW - 1 && (Pr.T & 63)) cm = (1ull << (Pr.T & 63)) - 1;
          if (Pr.tmpl_has_limits[t]) {
            for (uint32_t r = 0; r < NR; ++r) {
              if (!Pr.tmpl_limit_res[(size_t)r * NP + t]) continue;
              const int idx = thr_lookup_le(Pr.cthr_val + sh.a_cthr_off[r], (int)sh.a_cthr_n[r], Sr.remaining[(size_t)r * NP + t]);
              if (idx < 0) cm = 0; else cm &= Pr.cthr_mask[((size_t)sh.a_cthr_off[r] + idx) * TW + lane];
            }
          }
        }
        if (__ballot(cm != 0) == 0) { reason |= 16u; continue; }  // all available instance types exceed limits
        const bool tol_t = (KG_UNI64(tolw) >> t) & 1;
        if (!tol_t || !hcok) { reason |= tol_t ? 2u : 1u; continue; }
        // NewNodeClaim: the template's requirements and daemon overhead
        uint32_t meta = 0;
        uint64_t bits = 0;
        const uint32_t trs = Pr.tmpl_rs[t];
        for (uint32_t e = Pr.rs_off[trs]; e < Pr.rs_off[trs + 1]; ++e) {
          const uint32_t a = Pr.e_slot[e], off = sh.aoff[a], w = sh.aw[a];
          if (lane == a) meta = Pr.e_meta[e];
          if (lane >= off && lane < off + w) bits = Pr.rs_bits[Pr.e_bits[e] + lane - off];
        }
        int64_t req = 0;
        if (rg < NR) req = Pr.tmpl_req[(size_t)rg * NP + t];
        uint64_t m = cm;
        bool compat = true;
        KG_WN(0);
        const bool feasible = wave_add<true>(Pr, Pk, sh, thr_rows, lt, pc, lane, need, t, meta, bits, req, m, &compat);
        KG_WN(1);
        if (!feasible) { reason |= compat ? 8u : 2u; continue; }
        // ---- commit the new claim: appended to the end of the one-pod group
        KG_GLOBAL uint64_t* row = Sr.c_row + (size_t)c * KG_ROW_WORDS;
        if (lane < KG_MAXKA) ((KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane] = meta;
        row[KG_ROW_BITS + lane] = bits;
        if ((lane & 7) == 0) row[KG_ROW_REQ + rg] = (uint64_t)req;
        if (lane < KG_TWLDS) row[KG_ROW_MASK + lane] = m;
        {
          const uint64_t zw = zslot >= 0 ? plain_set_or_all(rl32(meta, (uint32_t)zslot), rl64(bits, zoff)) : ~0ull;
          const uint64_t cw = cslot >= 0 ? plain_set_or_all(rl32(meta, (uint32_t)cslot), rl64(bits, coff)) : ~0ull;
          const uint64_t n6 = (uint64_t)t | (1ull << 32), n7 = (uint64_t)(seq + 1) | ((uint64_t)(seq + 1) << 32);
          const uint64_t v = lane == 4 ? zw : (lane == 5 ? cw : (lane == 6 ? n6 : n7));
          if (lane >= NR && lane < 4) row[lane] = 0x7FFFFFFFFFFFFFFFull;  // (no such resource: never the reason to skip)
          if (lane >= 4 && lane < 8) row[lane] = v;
          if (c < pfn) {
            uint64_t* rec = pf + (size_t)c * pfw;
            if (lane >= NR && lane < 4) rec[lane] = 0x7FFFFFFFFFFFFFFFull;
            if (lane == 4 || lane == 5) rec[lane] = v;
            if (lane == 6) rec[6] = (uint64_t)t;
            if (lane < pfw - KG_PF_HEAD) rec[KG_PF_HEAD + lane] = m;
          }
        }
        KG_WN(2);
        for (uint32_t r = 0; r < NR; ++r) {
          const int64_t mx = wave_max_over_mask(Pr.alloc + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, m, TW, (uint32_t)Pr.T, lane);
          const int64_t rq = (int64_t)rl64((uint64_t)req, 8 * r);
          if (lane == 0 && r < 4) { row[r] = (uint64_t)(mx - rq); if (c < pfn) pf[(size_t)c * pfw + r] = (uint64_t)(mx - rq); }
          // subtractMax: the worst-case capacity of the remaining options is charged against the limits
          if (Pr.tmpl_has_limits[t] && Pr.tmpl_limit_res[(size_t)r * NP + t]) {
            const int64_t cx = wave_max_over_mask(Pr.cap + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, m, TW, (uint32_t)Pr.T, lane);
            if (lane == 0) Sr.remaining[(size_t)r * NP + t] -= cx;
          }
        }
        if (lane == 0) {
          Sr.pod_kind[pod_row(Sr, pod)] = 2;
          Sr.pod_index[pod_row(Sr, pod)] = (int32_t)c;
          if (Pr.has_ports) Sr.c_ports[c] = pc.vc.hp_use;
        }
        KG_WN(3);
        const uint32_t endk = word_group_end(Sk, sh, lds, nclaims, 1, lane);
        // shift [endk, nclaims) right by one, from the top
        for (uint32_t hi = nclaims; hi > endk;) {
          const uint32_t lo2 = hi - endk > 64 ? hi - 64 : endk;
          const uint32_t p = lo2 + lane;
          unsigned long long v = 0;
          if (p < hi) v = word_get(Sk, sh, lds, p);
          if (p < hi) word_set(Sk, sh, lds, p + 1, v);
          hi = lo2;
        }
        if (lane == 0) word_set(Sk, sh, lds, endk, (1ull << 32) | c);
        last_c = c; last_new = 1;
        nclaims += 1; seq += 1;
        placed = true;
        KG_WN(4);
      }
    }
    KG_W(6);
    if (__builtin_expect(!placed, 0)) {
      // ---- Preferences.Relax + Queue.Push (the push lands behind the entries already fetched: they were only fetched
      // when the queue held them)
      const SimState& Sr = rare_view(Sk);
      const int32_t vnext = (int32_t)KG_UNI((uint32_t)pc.vc.v.next);
      if (lane == 0) {
        Sr.pod_kind[pod_row(Sr, pod)] = 0;
        Sr.pod_index[pod_row(Sr, pod)] = -1;
        Sr.pod_reason[pod_row(Sr, pod)] = reason ? reason : 8;
        uint4 qn;
        qn.x = pod;
        if (vnext >= 0) { qn.y = (uint32_t)vnext; qn.z = 0; qn.w = 0; Sr.header[1] += 1; }
        else { qn.y = variant; qn.z = epoch; qn.w = qlen + 1; }
        ((KG_GLOBAL uint4*)Sr.queue)[(qhead + qlen) % QCAP] = qn;
      }
      qlen += 1;
      if (vnext >= 0) epoch += 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);  // the pushed entry may be the next pop
    }
    // ---- the next pod's context -> the other buffer
    have_ctx = next_ok;
    if (next_ok) {
      PodCtx& nx = sh.pcb[cur ^ 1];
      uint64_t* dst = (uint64_t*)&nx;
      const unsigned long long nb = __ballot(n_rq != 0);
      uint64_t w0 = iw[0];
      if (lane < KG_MAXR) w0 = (uint64_t)n_rq;          // words 0..7: the pod's requests
      if (lane == KG_MAXR) w0 = (uint64_t)(uint32_t)nb;  // word 8: need
      dst[lane] = w0;
#pragma unroll
      for (int k = 1; k < 4; ++k) { const uint32_t i = lane + 64 * k; if (i < sizeof(PodCtx) / 8) dst[i] = iw[k]; }
      nq_pod = n_pod; nq_variant = n_variant; nq_z = n_z; nq_w = n_w;
    }
  }
  // ---- the columns the host decodes, from the rows
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  const DevProblem& Pe = rare_view(Pk);
  const SimState& Se = rare_view(Sk);
  const uint32_t Ka = (uint32_t)Pe.Ka, Wa = (uint32_t)Pe.Wa;
  const size_t Cmax = (size_t)Pe.Cmax;
  for (uint32_t c = 0; c < nclaims; ++c) {
    const KG_GLOBAL uint64_t* row = Se.c_row + (size_t)c * KG_ROW_WORDS;
    if (lane < Ka) Se.c_meta[(size_t)lane * Cmax + c] = ((const KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane];
    if (lane < Wa) Se.c_bits[(size_t)lane * Cmax + c] = row[KG_ROW_BITS + lane];
    if (lane < NR) Se.c_req[(size_t)lane * Cmax + c] = (int64_t)row[KG_ROW_REQ + lane];
    if (lane < TW) Se.c_mask[(size_t)lane * Cmax + c] = row[KG_ROW_MASK + lane];
    if (lane == 0) {
      const uint64_t w6 = row[6], w7 = row[7];
      Se.c_tmpl[c] = (uint16_t)(uint32_t)w6; Se.c_npods[c] = (uint32_t)(w6 >> 32); Se.c_stamp[c] = (uint32_t)w7;
    }
  }
  if (WT && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) Se.counters[64 + i] = wt[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) Se.counters[80 + i] = wn[i];
  }
#undef KG_W
#undef KG_WN
  if (lane == 0) {
    Se.header[0] = nclaims;
    Se.header[2] = last_c; Se.header[3] = last_npods; Se.header[4] = last_stamp; Se.header[5] = last_new;
  }
}

}  // namespace kg
