// solve_spec.hpp -- Scheduler.Solve by W wavefronts of one workgroup: W pods of the Queue are placed at once, each by its
// own wave and speculatively, and retired strictly in Queue order (device only).
//
// FFD is one chain of dependent placements, and the single-wave kernel (solve_wave.hpp) runs that chain at the latency of
// one instruction stream: ~12 k cycles per placement, half of them waiting.  Almost all of a placement is READING --
// prefilter, claim row, NodeClaim.Add -- and what it reads is rarely what the W-1 placements in front of it write: the
// reference re-sorts the in-flight NodeClaims by pod count before every pod, so consecutive pods land on different claims.
// So the waves overlap the reading and serialise only the decision:
//   * ticket i = the i-th Queue.Pop; wave w takes tickets w, w + W, ...  A wave notes the number of retired tickets (e0),
//     evaluates its pod against whatever state it finds (prefilter over every NodeClaim, exact NodeClaim.Add on the
//     survivors in the reference's scan order), then waits for its turn (commit_seq == i);
//   * at its turn it looks at the log of the claims that tickets [e0, i) touched or created (at most W-1).  Feasibility of
//     NodeClaim.Add only ever shrinks as a claim takes pods (requirement sets narrow, requests grow, options and free host
//     ports shrink; the host keeps batches where that could fail -- negative operators on custom label keys -- on the
//     single-wave kernel), and a claim that takes a pod moves BACK in the scan order.  So every verdict about a claim
//     outside the log still holds; a claim in the log is tested again on its now settled state (prefilter first); if the
//     pod's own choice d is in the log the wave rescans under the token.  The first feasible claim in scan order is then
//     exactly the sequential one, and the commit happens from the registers of the speculative evaluation;
//   * the scan order needs no array: a claim's place in the reference's stable sort is the closed form claim_key(pods,
//     stamp, id), kept in the claim's prefilter record; "first feasible" = minimum key over the feasible claims.
// Torn reads are harmless by the same argument: a wave that reads a row or record while the token holder rewrites it sees,
// word by word, the old or the new value; that claim is in the reader's log window and is evaluated again at its turn.
// Claims are published (nclaims) only after their row is complete, so a row is never read before it was first written.
// A new NodeClaim and Preferences.Relax run under the token, as in the single-wave kernel.
//
// Every wait polls the error / done flags and gives up after a bounded number of polls (SpecCtl::err: the host reports
// KG_EDEVICE), so the workgroup always drains.
#pragma once
#include "solve_wave.hpp"

namespace kg {

constexpr uint32_t KG_SPEC_LOG = 16;       // ring of the claims the last tickets touched (>= waves of the workgroup)
constexpr uint32_t KG_SPEC_MAXW = 8;       // waves of the largest instantiation
constexpr uint32_t KG_SPEC_SNAP = 8;       // chunks of 64 claims whose prefilter verdicts one ticket keeps in registers
constexpr uint32_t KG_SPEC_SLOTS = 8;      // merged rows of the last tickets kept in LDS (>= the longest log window, waves - 1)
constexpr uint32_t KG_NOSLOT = 0xFFFFFFFFu;
constexpr uint32_t KG_SPEC_POLLS = 1u << 22;  // polls of one wait before the wave gives up (a wait is a few placements long)

struct SpecCtl {
  uint32_t commit_seq;    // tickets retired so far = the ticket whose turn it is
  uint32_t vis_seq;       // tickets whose global-memory stores are complete (published in ticket order, after the token was passed on)
  uint32_t nclaims_vis;   // NodeClaims whose rows are complete in global memory: what a speculative evaluation may read
  uint32_t total_pushed;  // Queue entries ever pushed (the initial Queue + re-pushes after Preferences.Relax / failures)
  uint32_t done, err;
  uint32_t nclaims, seq, epoch, relax;
  uint32_t last_c, last_npods, last_stamp, last_new;
  uint32_t log[KG_SPEC_LOG];  // log[e % KG_SPEC_LOG] = the claim ticket e placed its pod on, KG_NOIDX if none
  uint32_t lim_epoch;     // bumped whenever a new NodeClaim is charged against a NodePool's limits
  unsigned long long t_rel;  // (debug timers) when the last ticket passed the token on
  uint32_t stat[8];       // 0 commits from speculation, 1 challengers evaluated, 2 rescans under the token, 3 new NodeClaims, 4 failures, 5 exact evaluations,
                          // 6 new NodeClaims / failures evaluated again under the token
};

__device__ __forceinline__ uint32_t lds_acquire(const uint32_t* p) { return KG_UNI(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ uint32_t lds_relaxed(const uint32_t* p) { return KG_UNI(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void lds_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// minimum of a 64-bit value over the wave, in every lane: data-parallel-primitive moves on the vector ALU (no LDS traffic:
// the other waves are using it); lanes without a source in a step keep all ones
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint64_t dpp_min_step(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(uint32_t)v, CTRL, ROWMASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(uint32_t)(v >> 32), CTRL, ROWMASK, 0xF, false);
  const uint64_t o = ((uint64_t)hi << 32) | lo;
  return o < v ? o : v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  v = dpp_min_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_min_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_min_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_min_step<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row of 16 holds the row's minimum
  v = dpp_min_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_min_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
  return rl64(v, 63);
}

// the pod's side of the prefilter, the same value in every lane (vector registers: only vector instructions use them)
struct PodSide {
  uint64_t pm[KG_TWLDS];  // the pod's own type mask, zero beyond TW
  int64_t prq[4];
  uint64_t tolw, hp_conf, pzone, pct;
  uint32_t need;
  bool hcok;
};

// Prefilter of NodeClaim `c` (one lane, one claim): necessary conditions of NodeClaim.Add from the claim's prefilter words
// -- [0..3] headroom, [4][5] zone / capacity-type sets, [6] template | pods << 32, [7] order stamp, [8..8+TW) option mask.
// LDS: word j of claim c at pf[j * pfcap + c] (consecutive lanes read consecutive words: no bank conflicts); otherwise the
// first words of the claim's global row, which hold the same values.  *key = the claim's place in the scan order.
// RANGE: only claims whose key lies in (klo, khi) are wanted -- the key words are read first and the rest is skipped when
// no lane's claim is in range (a rescan between two places of the scan order touches one or two chunks of 64 claims).
template <bool LDS, bool RANGE>
__device__ __forceinline__ bool spec_prefilter(const uint64_t* pf, uint32_t pfcap, const KG_GLOBAL uint64_t* c_row, const KG_GLOBAL uint64_t* c_ports, uint32_t TW,
                                               uint32_t c, bool active, const PodCtx& pc, const PodSide& ps, uint64_t klo, uint64_t khi, uint64_t* key) {
  const uint32_t cc = active ? c : 0u;  // (every load unconditional: one round trip)
  uint64_t h[8];
  if (RANGE) {
#pragma unroll
    for (int j = 6; j < 8; ++j) {
      if (LDS) h[j] = pf[(size_t)j * pfcap + cc];
      else h[j] = c_row[(size_t)cc * KG_ROW_WORDS + j];
    }
    if (LDS) { KG_OPAQUE64(h[6]); KG_OPAQUE64(h[7]); }  // (after both loads are issued: one round trip)
    const uint64_t k0 = claim_key((uint32_t)(h[6] >> 32), (uint32_t)h[7], c);
    *key = k0;
    active = active && k0 > klo && k0 < khi;
    if (__ballot(active) == 0) return false;
  }
#pragma unroll
  for (int j = 0; j < (RANGE ? 6 : 8); ++j) {
    if (LDS) h[j] = pf[(size_t)j * pfcap + cc];
    else h[j] = c_row[(size_t)cc * KG_ROW_WORDS + j];
  }
  if (LDS) {
#pragma unroll
    for (int j = 0; j < (RANGE ? 6 : 8); ++j) KG_OPAQUE64(h[j]);
  }
  const uint32_t tm = (uint32_t)h[6], npods = (uint32_t)(h[6] >> 32), stamp = (uint32_t)h[7];
  uint32_t bad = (uint32_t)(((ps.tolw >> (tm & 63)) & 1) == 0) | (uint32_t)((h[4] & ps.pzone) == 0) | (uint32_t)((h[5] & ps.pct) == 0);
  bad |= ((uint32_t)((int64_t)h[0] < ps.prq[0]) & ps.need) | ((uint32_t)((int64_t)h[1] < ps.prq[1]) & (ps.need >> 1)) |
         ((uint32_t)((int64_t)h[2] < ps.prq[2]) & (ps.need >> 2)) | ((uint32_t)((int64_t)h[3] < ps.prq[3]) & (ps.need >> 3));
  bad &= 1u;
  if (c_ports) bad |= (uint32_t)((c_ports[cc] & ps.hp_conf) != 0);
  const bool pass1 = active && ps.hcok && bad == 0;
  uint64_t any = 0;
  if (__ballot(pass1) != 0) {  // (most claims of a late Solve are full: their option masks are never read)
    // (every load unconditional and issued before the first use: one round trip; words beyond TW meet a zero pod word)
    uint64_t mw[KG_TWLDS];
#pragma unroll
    for (int k = 0; k < KG_TWLDS; ++k) {
      const uint32_t kk = (uint32_t)k < TW ? (uint32_t)k : 0u;
      if (LDS) mw[k] = pf[(size_t)(8 + kk) * pfcap + cc];
      else mw[k] = c_row[(size_t)cc * KG_ROW_WORDS + KG_ROW_MASK + kk];
    }
#pragma unroll
    for (int k = 0; k < KG_TWLDS; ++k) {
      if (LDS) KG_OPAQUE64(mw[k]);
      any |= mw[k] & ps.pm[k];
    }
  }
  if (!RANGE) *key = claim_key(npods, stamp, c);
  return pass1 && any != 0;
}

// A new NodeClaim for the pod, evaluated ahead of its turn: the first template in weight order whose NodeClaim.Add
// succeeds, the merged row, its headroom and what subtractMax will charge.  It depends on the pod, the templates and the
// remaining limits only, so it holds at the pod's turn unless a limit has changed since (SpecCtl::lim_epoch).
struct FreshRegs {
  uint32_t meta;
  uint64_t bits, m;
  int64_t req;
  int64_t hd;  // lanes 0..3: headroom of resources 0..3 (no such resource: the largest value)
  int64_t cx;  // lane r: worst-case capacity of the options, where the template's limits name resource r (else 0)
  uint32_t t, reason, lim_epoch;  // t: the template, KG_NOIDX when none fits (reason: why, as the pod's failure reasons)
  bool valid;
};

// a claim row in registers (solve_wave.hpp: lane a meta[a], lane i bits[i], lanes 8r requests of resource r, lane w option
// mask word w), lanes 0..3 the headroom words, and the uniform words 6 / 7
struct ClaimRegs {
  uint32_t meta;
  uint64_t bits, m;
  int64_t req, phd;
  uint64_t w6, w7;
};

// DBG: per-wave cycle accounting into SimState::counters[64 + 16 * wave ..]: 0 waiting for a Queue entry, 1 speculative
// evaluation, 2 waiting for the turn, 3..7 under the token by outcome (commit from speculation, after a challenger, after
// a rescan, new NodeClaim, failure), 8 after the token (next context), 9 tickets
template <int W, bool DBG>
__device__ __forceinline__ void solve_spec(const DevProblem& P, const SimState& S, const KG_CONST DevProblem* Pk, const KG_CONST SimState* Sk, SharedBase& sh, SpecCtl& ctl,
                                           uint64_t* pool, uint32_t pool_words, uint32_t n_queue) {
  constexpr int NT = 64 * W;
  static_assert(W >= 1 && W <= (int)KG_SPEC_MAXW && W < (int)KG_SPEC_LOG, "waves of the speculative solve");
  constexpr size_t ORD_PAD = (16 - offsetof(SharedBase, ord) % 16) % 16;  // (a pod context is read with 16-byte loads)
  static_assert(alignof(SharedBase) % 16 == 0 && sizeof(PodCtx) % 16 == 0 && ORD_PAD + 8 * sizeof(PodCtx) <= sizeof(sh.ord), "pod contexts live in the scan-order memory");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = KG_UNI(tid >> 6);
  unsigned long long tk[10], tk_last = 0, ts[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tra = 0, trb = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) tk[k] = 0;
#define KG_TK(k) do { if (DBG) { const unsigned long long n_ = clock64(); if (tk_last) tk[k] += n_ - tk_last; tk_last = n_; } } while (0)
  stage_tables<NT>(rare_view(Pk), sh);
  const uint32_t NR = (uint32_t)P.R, TW = (uint32_t)P.TW;
  const uint32_t rg = lane >> 3;
  // ---- one LDS pool: fit-threshold rows, the pod contexts that do not fit the (unused) scan-order arrays, prefilter records
  const uint64_t* thr_rows = nullptr;
  uint32_t off = 0;
  {
    uint32_t tot = 0;
    for (uint32_t r = 0; r < NR; ++r) tot += sh.a_thr_n[r];
    tot = KG_UNI(tot);
    if (tot * TW <= pool_words / 2) {
      const uint64_t* thr_mask = rare_view(Pk).thr_mask;
      for (uint32_t r = 0; r < NR; ++r) {
        const uint32_t n = KG_UNI(sh.a_thr_n[r]), b = KG_UNI(sh.thr_base[r]), o = KG_UNI(sh.a_thr_off[r]);
        for (uint32_t i = tid; i < n * TW; i += NT) pool[b * TW + i] = thr_mask[(size_t)o * TW + i];
      }
      thr_rows = pool;
      off = (tot * TW + 1) & ~1u;
    }
  }
  // two pod contexts per wave (the pod being placed, the pod of the wave's next ticket): buffers 0..7 in SharedBase::ord
  PodCtx* const ctx_lo = (PodCtx*)(void*)((unsigned char*)&sh.ord[0][0] + ORD_PAD);
  PodCtx* const ctx_hi = (PodCtx*)(void*)(pool + off);
  if (2 * W > 8) off += (uint32_t)((2 * W - 8) * sizeof(PodCtx) / 8);
  // (typed as LDS pointers: picked out of two arrays at run time, a generic pointer would turn every read of the pod context
  // into a flat load -- slower, and counted on the memory counter too, so every wait for a global load would wait for them)
  typedef __attribute__((address_space(3))) PodCtx LdsPodCtx;
  LdsPodCtx* mybuf[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { const uint32_t b = 2 * wave + k; mybuf[k] = (LdsPodCtx*)(b < 8 ? ctx_lo + b : ctx_hi + (b - 8)); }
  // the merged rows of the last KG_SPEC_SLOTS tickets (slot = ticket % KG_SPEC_SLOTS): what a ticket reads at its turn when
  // it meets a claim of its log window -- the window's global stores may still be in flight
  uint64_t* const rowc = pool + off;
  off += KG_SPEC_SLOTS * KG_ROW_WORDS;
  const uint32_t pfw = 8 + TW;                                       // words per prefilter record
  const uint32_t pfcap = ((pool_words - off) / pfw) & ~63u;          // records that fit, whole chunks of 64 claims
  uint64_t* const pf = pool + off;
  if (tid == 0) {
    ctl.commit_seq = 0; ctl.vis_seq = 0; ctl.nclaims_vis = 0; ctl.total_pushed = n_queue; ctl.done = 0; ctl.err = 0;
    ctl.nclaims = 0; ctl.seq = 0; ctl.epoch = 1; ctl.relax = 0; ctl.lim_epoch = 0;
    ctl.last_c = KG_NOIDX; ctl.last_npods = 0; ctl.last_stamp = 0; ctl.last_new = 0;
    for (uint32_t k = 0; k < KG_SPEC_LOG; ++k) ctl.log[k] = KG_NOIDX;
    for (int k = 0; k < 8; ++k) ctl.stat[k] = 0;
  }
  __syncthreads();
  LaneTabs lt;
  lt.spa = lt.spoff = lt.tn = lt.tbase = lt.toff = 0;
  if (lane < (uint32_t)P.nspec) { lt.spa = sh.spec_slot[lane]; lt.spoff = sh.aoff[lt.spa]; }
  if (rg < NR) { lt.tn = sh.a_thr_n[rg]; lt.tbase = sh.thr_base[rg]; lt.toff = sh.a_thr_off[rg]; }
  const int zslot = P.slot_zone >= 0 ? (int)sh.spec_slot[P.slot_zone] : -1, cslot = P.slot_ct >= 0 ? (int)sh.spec_slot[P.slot_ct] : -1;
  const uint32_t zoff = zslot >= 0 ? (uint32_t)sh.aoff[zslot] : 0, coff = cslot >= 0 ? (uint32_t)sh.aoff[cslot] : 0;
  const uint32_t QCAP = S.qcap;
  const KG_GLOBAL uint64_t* const ports_ro = P.has_ports ? S.c_ports : nullptr;
  uint32_t st_fast = 0, st_chal = 0, st_rescan = 0, st_new = 0, st_fail = 0, st_exact = 0, st_refresh = 0;

  uint32_t cur = 0;
  bool have_next = false, have_next2 = false;
  uint4 nq, nq2;
  nq.x = nq.y = nq.z = nq.w = 0;
  nq2 = nq;
  for (uint32_t i = wave;; i += W) {
#ifndef KG_NO_LOOP_VIEW
    // (the descriptors through per-ticket views: their fields' scalar loads stay inside the ticket -- solve_core.hpp)
    const DevProblem& P = rare_view(Pk);
    const SimState& S = rare_view(Sk);
#endif
    KG_TK(8);
    // ---- wait until the Queue holds an entry for this ticket (or it is this ticket's turn and the Queue is empty)
    bool stop = false;
    if (!have_next) {
      for (uint32_t polls = 0;; ++polls) {
        if ((lds_relaxed(&ctl.done) | lds_relaxed(&ctl.err)) != 0) { stop = true; break; }
        const uint32_t cs = lds_acquire(&ctl.commit_seq);      // (before total_pushed: a retired ticket's push is then visible)
        const uint32_t tp = lds_acquire(&ctl.total_pushed);
        if (i < tp) break;
        if (cs == i) { lds_release(&ctl.done, 1u); stop = true; break; }  // every ticket before this one is retired: nothing left to pop
        if (polls > KG_SPEC_POLLS) { lds_store(&ctl.err, 1u); stop = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (stop) break;
    KG_TK(0);
    // tickets whose stores were complete when this evaluation starts: everything it reads from global memory from here on is
    // at least that new; the tickets from e0 on are its log window
    const uint32_t e0 = lds_acquire(&ctl.vis_seq);
    uint4 qv = nq;
    if (!have_next) qv = ((const KG_GLOBAL uint4*)S.queue)[i % QCAP];
    const uint32_t pod = KG_UNI(qv.x), variant = KG_UNI(qv.y), qz = KG_UNI(qv.z), qw = KG_UNI(qv.w);
    cur ^= 1;
    LdsPodCtx* const pcl = cur ? mybuf[1] : mybuf[0];
    if (!have_next) ctx_fetch_image(P, *(PodCtx*)pcl, lane, pod, variant);
    const PodCtx& pc = *(const PodCtx*)pcl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the context words other lanes stored)
    // ---- the wave's next ticket: its Queue entry, context image and requests are requested now and land in the other
    // buffer when this ticket is retired
    const uint32_t tp_seen = lds_acquire(&ctl.total_pushed);
    const bool next_ok = i + W < tp_seen, next2_ok = i + 2 * W < tp_seen;
    uint4 ne4, ne8;
    ne4.x = ne4.y = ne4.z = ne4.w = 0;
    ne8 = ne4;
    uint64_t iw[4] = {0, 0, 0, 0};
    int64_t n_rq = 0;
    // (the Queue entry after next: in registers one ticket early, so that the next ticket's context can be requested without
    // waiting for its entry first)
    if (next2_ok) ne8 = ((const KG_GLOBAL uint4*)S.queue)[(i + 2 * W) % QCAP];
    if (next_ok) {
      if (have_next2) ne4 = nq2;
      else ne4 = ((const KG_GLOBAL uint4*)S.queue)[(i + W) % QCAP];
      const uint32_t n_pod = KG_UNI(ne4.x), n_variant = KG_UNI(ne4.y);
      const uint64_t* img = (const uint64_t*)(P.pctx + n_variant);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const uint32_t q = lane + 64 * k; iw[k] = img[q < sizeof(PodCtx) / 8 ? q : 0]; }
      if (lane < NR) n_rq = P.preq[(size_t)n_pod * NR + lane];
    }
    const uint32_t need = KG_UNI(pc.need);
    PodSide ps;
    ps.tolw = pc.tolw; ps.hp_conf = P.has_ports ? pc.vc.hp_conf : 0ull; ps.pzone = pc.vc.pzone; ps.pct = pc.vc.pct;
    KG_VREG64(ps.tolw); KG_VREG64(ps.hp_conf); KG_VREG64(ps.pzone); KG_VREG64(ps.pct);
    ps.hcok = KG_UNI(pc.vc.v.host_claim_ok) != 0;
    ps.need = need;
#pragma unroll
    for (int r = 0; r < 4; ++r) ps.prq[r] = pc.preq[r];
#pragma unroll
    for (int k = 0; k < KG_TWLDS; ++k) ps.pm[k] = pc.podmask[k];
#pragma unroll
    for (int k = 0; k < KG_TWLDS; ++k) { if ((uint32_t)k >= TW) ps.pm[k] = 0; KG_VREG64(ps.pm[k]); }

    // min key in (klo, khi) over the claims [0, n) that pass the prefilter (ranged: chunks without a claim in range are skipped)
    // The newest claims hold the fewest pods and so come first in the scan order: the chunks are visited from the highest
    // claim ids down, and once a survivor is known only claims in front of it matter -- the later chunks are usually
    // dismissed on their key words alone.
    auto scan_range = [&](uint32_t n, uint64_t klo, uint64_t khi) -> uint64_t {
      uint64_t best = khi;  // (uniform)
      for (uint32_t base = (n + 63u) & ~63u; base != 0;) {
        base -= 64;
        const uint32_t c = base + lane;
        uint64_t key = 0;
        bool ok;
        if (base < pfcap) ok = spec_prefilter<true, true>(pf, pfcap, S.c_row, ports_ro, TW, c, c < n, pc, ps, klo, best, &key);
        else ok = spec_prefilter<false, true>(pf, pfcap, S.c_row, ports_ro, TW, c, c < n, pc, ps, klo, best, &key);
        if (__ballot(ok) != 0) best = wave_min_u64(ok ? key : ~0ull);  // (every survivor of this chunk is in front of the old best)
      }
      return best == khi ? ~0ull : best;
    };
    auto scan_all = [&](uint32_t n, uint64_t klo) -> uint64_t { return scan_range(n, klo, ~0ull); };
    // ... over the claims of the log window (lane l < wn: claim wc = log[e0 + l], KG_NOIDX none).  At most seven claims: the
    // words of their records are spread over the lanes (8 lanes per claim for the head words, 16 per claim for the option
    // mask) instead of one lane walking a whole record -- this runs under the token.
    auto scan_window = [&](uint32_t wc, uint32_t wn, uint64_t klo, uint64_t khi) -> uint64_t {
      const bool act = wc != KG_NOIDX;
      if (__ballot(act) == 0) return ~0ull;
      if (__builtin_expect(__ballot(act && wc >= pfcap) != 0 || ports_ro != nullptr, 0)) {
        uint64_t key = 0;
        const bool ok = spec_prefilter<false, true>(pf, pfcap, S.c_row, ports_ro, TW, wc, act, pc, ps, klo, khi, &key);
        if (__ballot(ok) == 0) return ~0ull;
        return wave_min_u64(ok ? key : ~0ull);
      }
      const uint32_t cc = act ? wc : 0u;
      uint64_t w6 = pf[(size_t)6 * pfcap + cc], w7 = pf[(size_t)7 * pfcap + cc];
      KG_OPAQUE64(w6); KG_OPAQUE64(w7);
      const uint64_t key = claim_key((uint32_t)(w6 >> 32), (uint32_t)w7, wc);
      const bool inr = act && ps.hcok && key > klo && key < khi;
      unsigned long long cand = __ballot(inr);
      if (cand == 0) return ~0ull;
      {
        // head words: lane 8l + j tests word j of window claim l
        const uint32_t l = lane >> 3, j = lane & 7;
        uint32_t c = KG_NOIDX;
        if (l < wn) c = ctl.log[(e0 + l) % KG_SPEC_LOG];
        const bool a2 = c != KG_NOIDX;
        uint64_t w = pf[(size_t)j * pfcap + (a2 ? c : 0u)];
        KG_OPAQUE64(w);
        const int64_t rq = pc.preq[j & 3];
        bool bad = false;
        if (j < 4) bad = ((need >> j) & 1) && (int64_t)w < rq;
        else if (j == 4) bad = (w & ps.pzone) == 0;
        else if (j == 5) bad = (w & ps.pct) == 0;
        else if (j == 6) bad = ((ps.tolw >> ((uint32_t)w & 63)) & 1) == 0;
        const unsigned long long bb = __ballot(a2 && bad);
        // option masks: lane 16l + k tests word k of window claim 4p + l
        unsigned long long anyb[2] = {0, 0};
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          if ((uint32_t)(4 * p2) < wn) {
            const uint32_t l4 = 4 * p2 + (lane >> 4), k = lane & 15;
            uint32_t c4 = KG_NOIDX;
            if (l4 < wn) c4 = ctl.log[(e0 + l4) % KG_SPEC_LOG];
            const bool a4 = c4 != KG_NOIDX && k < TW;
            uint64_t mw = pf[(size_t)(8 + (k < TW ? k : 0u)) * pfcap + (a4 ? c4 : 0u)];
            KG_OPAQUE64(mw);
            anyb[p2] = __ballot(a4 && (mw & pc.podmask[k]) != 0);
          }
        }
        // lane l < 8: the verdict on window claim l
        const bool badl = ((bb >> (8 * (lane & 7))) & 0xFFull) != 0;
        const bool anyl = ((anyb[(lane >> 2) & 1] >> (16 * (lane & 3))) & 0xFFFFull) != 0;
        cand = __ballot(inr && lane < 8 && !badl && anyl);
      }
      uint64_t best = ~0ull;
      for (unsigned long long q = cand; q; q &= q - 1) {
        const uint64_t k = rl64(key, (uint32_t)(__ffsll(q) - 1));
        best = k < best ? k : best;
      }
      return best;
    };
    // The window's verdicts, one claim at a time and AHEAD of the turn: while the wave waits, every ticket of its log window
    // that retires has its claim's prefilter record looked at (one LDS round trip: lane j < 8 tests head word j, lane 8 + k
    // option-mask word k); lane l keeps the claim, its key and the verdict of window entry l.  At the turn only the entry of
    // the ticket just retired is left, and a question about the window is a ballot over these lanes.  (Claims whose record
    // is not in LDS, or host ports: win_slow, the turn falls back to scan_window.)
    uint32_t wv_c = KG_NOIDX, wv_done = 0;
    uint64_t wv_key = ~0ull;
    bool wv_ok = false, win_slow = ports_ro != nullptr;
    auto win_eval = [&](uint32_t l) {
      const uint32_t c = lds_relaxed(&ctl.log[(e0 + l) % KG_SPEC_LOG]);
      if (c == KG_NOIDX) return;
      if (c >= pfcap) { win_slow = true; return; }
      const uint32_t j = lane < 8 + TW ? lane : 0u;
      uint64_t w = pf[(size_t)j * pfcap + c];
      KG_OPAQUE64(w);
      bool bad = false, hit = false;
      if (lane < 4) bad = ((need >> lane) & 1) && (int64_t)w < pc.preq[lane & 3];
      else if (lane == 4) bad = (w & ps.pzone) == 0;
      else if (lane == 5) bad = (w & ps.pct) == 0;
      else if (lane == 6) bad = ((ps.tolw >> ((uint32_t)w & 63)) & 1) == 0;
      else if (lane >= 8 && lane < 8 + TW) hit = (w & pc.podmask[(lane - 8) & (KG_TWLDS - 1)]) != 0;
      const unsigned long long bb = __ballot(bad), hb = __ballot(hit);
      const uint64_t w6 = rl64(w, 6), w7 = rl64(w, 7);
      const uint64_t key = claim_key((uint32_t)(w6 >> 32), (uint32_t)w7, c);
      if (lane == l) { wv_c = c; wv_key = key; wv_ok = ps.hcok && bb == 0 && hb != 0; }
    };
    // exact NodeClaim.Add of this pod on claim c -> R = the merged row.  An infeasible claim whose headroom words are older
    // than its options gets them recomputed (an upper bound of what it can still take whatever the token holder does to
    // the claim meanwhile: options only shrink, requests only grow)
    // (slot: the row comes from the LDS copy a ticket of the log window left, KG_NOSLOT: from global memory)
    auto exact = [&](uint32_t c, uint32_t slot, ClaimRegs& R) -> bool {
      KG_GLOBAL uint64_t* row = S.c_row + (size_t)c * KG_ROW_WORDS;
      R.meta = 0;
      uint64_t cm = 0;
      R.phd = 0;
      uint32_t hst, stamp;
      if (slot != KG_NOSLOT) {
        // (HostPortUsage conflicts are part of the prefilter only; a claim of the window that is evaluated again without
        // passing through it -- the pod's own choice after it took another pod -- has its port word looked at here)
        if (ports_ro != nullptr && (KG_UNI64(ports_ro[c]) & KG_UNI64(ps.hp_conf)) != 0) return false;
        const uint64_t* lrow = rowc + (size_t)slot * KG_ROW_WORDS;
        if (lane < KG_MAXKA) R.meta = ((const uint32_t*)(lrow + KG_ROW_META))[lane];
        R.bits = lrow[KG_ROW_BITS + lane];
        R.req = (int64_t)lrow[KG_ROW_REQ + rg];
        if (lane < KG_TWLDS) cm = lrow[KG_ROW_MASK + lane];
        if (lane < 4) R.phd = (int64_t)lrow[lane];
        const uint64_t w6_v = lrow[6], w7_v = lrow[7];
        R.w6 = KG_UNI64(w6_v); R.w7 = KG_UNI64(w7_v);
        hst = stamp = (uint32_t)R.w7;  // (its headroom words are as good as they get: the claim just took a pod)
      } else {
        if (lane < KG_MAXKA) R.meta = ((const KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane];
        R.bits = row[KG_ROW_BITS + lane];
        R.req = (int64_t)row[KG_ROW_REQ + rg];
        if (lane < KG_TWLDS) cm = row[KG_ROW_MASK + lane];
        if (lane < 4) R.phd = (int64_t)row[lane];
        const uint64_t w6_v = row[6], w7_v = row[7];
        const uint32_t hst_v = S.c_hstamp[c];
        R.w6 = KG_UNI64(w6_v); R.w7 = KG_UNI64(w7_v);
        hst = KG_UNI(hst_v); stamp = (uint32_t)R.w7;
      }
      const int64_t req0 = R.req;
      R.m = cm;
      bool compat = true;
      st_exact += 1;
      const bool feasible = wave_add<false>(P, Pk, sh, thr_rows, lt, pc, lane, need, (uint32_t)R.w6, R.meta, R.bits, R.req, R.m, &compat);
      if (__builtin_expect(!feasible && hst != stamp, 0)) {
        const DevProblem& Pr = rare_view(Pk);
        for (uint32_t r = 0; r < NR && r < 4; ++r) {
          const int64_t mx = wave_max_over_mask(Pr.alloc + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, cm, TW, (uint32_t)Pr.T, lane);
          const int64_t rq = (int64_t)rl64((uint64_t)req0, 8 * r);
          if (lane == 0) { row[r] = (uint64_t)(mx - rq); if (c < pfcap) pf[(size_t)r * pfcap + c] = (uint64_t)(mx - rq); }
        }
        if (lane == 0) S.c_hstamp[c] = stamp;
      }
      return feasible;
    };
    // NewNodeClaim + NodeClaim.Add for each template in weight order (as solve_wave.hpp), nothing written: F = the first that works
    auto fresh = [&](FreshRegs& F) {
      const DevProblem& Pr = rare_view(Pk);
      const SimState& Sr = rare_view(Sk);
      F.lim_epoch = lds_acquire(&ctl.lim_epoch);  // (before the limits are read)
      F.t = KG_NOIDX; F.reason = 0; F.valid = true;
      F.meta = 0; F.bits = F.m = 0; F.req = 0; F.hd = 0; F.cx = 0;
      const uint32_t NP = (uint32_t)Pr.NP;
      const uint64_t tolw_u = KG_UNI64(pc.tolw);
      for (uint32_t t = 0; t < NP; ++t) {
        // filterByRemainingResources: types whose capacity fits what the NodePool's limits leave
        uint64_t cm = 0;
        const bool has_lim = Pr.tmpl_has_limits[t] != 0;
        if (lane < TW) {
          cm = ~0ull;
          if (lane == TW - 1 && (Pr.T & 63)) cm = (1ull << (Pr.T & 63)) - 1;
          if (has_lim) {
            for (uint32_t r = 0; r < NR; ++r) {
              if (!Pr.tmpl_limit_res[(size_t)r * NP + t]) continue;
              const int idx = thr_lookup_le(Pr.cthr_val + sh.a_cthr_off[r], (int)sh.a_cthr_n[r], Sr.remaining[(size_t)r * NP + t]);
              if (idx < 0) cm = 0; else cm &= Pr.cthr_mask[((size_t)sh.a_cthr_off[r] + idx) * TW + lane];
            }
          }
        }
        if (__ballot(cm != 0) == 0) { F.reason |= 16u; continue; }  // all available instance types exceed limits
        const bool tol_t = (tolw_u >> t) & 1;
        if (!tol_t || !ps.hcok) { F.reason |= tol_t ? 2u : 1u; continue; }
        // NewNodeClaim: the template's requirements and daemon overhead
        uint32_t meta = 0;
        uint64_t bits = 0;
        const uint32_t trs = Pr.tmpl_rs[t];
        for (uint32_t e = Pr.rs_off[trs]; e < Pr.rs_off[trs + 1]; ++e) {
          const uint32_t a = Pr.e_slot[e], eo = sh.aoff[a], w = sh.aw[a];
          if (lane == a) meta = Pr.e_meta[e];
          if (lane >= eo && lane < eo + w) bits = Pr.rs_bits[Pr.e_bits[e] + lane - eo];
        }
        int64_t req = 0;
        if (rg < NR) req = Pr.tmpl_req[(size_t)rg * NP + t];
        uint64_t m = cm;
        bool compat = true;
        const bool feasible = wave_add<true>(Pr, Pk, sh, thr_rows, lt, pc, lane, need, t, meta, bits, req, m, &compat);
        if (!feasible) { F.reason |= compat ? 8u : 2u; continue; }
        F.meta = meta; F.bits = bits; F.req = req; F.m = m; F.t = t;
        F.hd = 0x7FFFFFFFFFFFFFFFll;  // (no such resource: never the reason to skip)
        for (uint32_t r = 0; r < NR; ++r) {
          const int64_t mx = wave_max_over_mask(Pr.alloc + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, m, TW, (uint32_t)Pr.T, lane);
          const int64_t rq = (int64_t)rl64((uint64_t)req, 8 * r);
          if (lane == r) F.hd = mx - rq;
          // subtractMax: the worst-case capacity of the remaining options is charged against the limits
          if (has_lim && Pr.tmpl_limit_res[(size_t)r * NP + t]) {
            const int64_t cx = wave_max_over_mask(Pr.cap + (size_t)KG_UNI(sh.a_res[r]) * Pr.T, m, TW, (uint32_t)Pr.T, lane);
            if (lane == r) F.cx = cx;
          }
        }
        break;
      }
    };

    // ---- speculative evaluation, on the state as this wave finds it: the first feasible claims in scan order -- as many as
    // it takes for one of them to be outside the log window at the turn whatever the tickets in flight do: one more than
    // the tickets still in flight ahead of this one, not counting candidates a retired ticket of the window is known to have
    // touched (consecutive pods of a Queue pile onto the same claims more often than not).  Every claim outside the window
    // that comes before the last candidate (every such claim at all when the list is `complete`) is then known not to take
    // the pod unless it is a candidate, and those verdicts hold at the turn.
    uint32_t d = KG_NOIDX;
    uint64_t keyd = ~0ull;
    ClaimRegs Rd;
    Rd.meta = 0; Rd.bits = Rd.m = 0; Rd.req = Rd.phd = 0; Rd.w6 = Rd.w7 = 0;
    constexpr int KC = W < 6 ? W : 6;  // candidates kept in registers
    ClaimRegs CR[KC];
    uint32_t cand_c[KC];
    uint64_t cand_k[KC];
#pragma unroll
    for (int q = 0; q < KC; ++q) { CR[q] = Rd; cand_c[q] = KG_NOIDX; cand_k[q] = ~0ull; }
    uint32_t ncand = 0;
    bool complete = false;
    FreshRegs F;
    F.valid = false; F.t = KG_NOIDX; F.reason = 0; F.lim_epoch = 0; F.meta = 0; F.bits = F.m = 0; F.req = F.hd = F.cx = 0;
    {
      const uint32_t nsnap = lds_acquire(&ctl.nclaims_vis);
      // the prefilter's survivors, once: lane l keeps the key of claim 64 b + l where it passes (records in LDS, at most
      // KG_SPEC_SNAP chunks: larger Solves scan again for every candidate)
      uint64_t sk[KG_SPEC_SNAP];
      const bool snap = nsnap <= pfcap && nsnap <= 64u * KG_SPEC_SNAP;
      if (DBG) tra = clock64();
      if (snap) {
#pragma unroll
        for (int b2 = 0; b2 < (int)KG_SPEC_SNAP; ++b2) {
          sk[b2] = ~0ull;
          if ((uint32_t)(64 * b2) < nsnap) {
            const uint32_t c = 64u * b2 + lane;
            uint64_t key = 0;
            const bool ok = spec_prefilter<true, false>(pf, pfcap, S.c_row, ports_ro, TW, c, c < nsnap, pc, ps, 0, ~0ull, &key);
            sk[b2] = ok ? key : ~0ull;
          }
        }
      }
      auto next_survivor = [&](uint64_t klo) -> uint64_t {
        if (!snap) return scan_range(nsnap, klo, ~0ull);
        uint64_t m = ~0ull;
#pragma unroll
        for (int b2 = 0; b2 < (int)KG_SPEC_SNAP; ++b2) { const uint64_t v = sk[b2]; if (v > klo && v < m) m = v; }
        if (__ballot(m != ~0ull) == 0) return ~0ull;
        return wave_min_u64(m);
      };
      uint64_t klo = 0;
      if (DBG) { trb = clock64(); tr[3] += trb - tra; }
      while (true) {
        // (the window's verdicts make progress here too; a candidate a retired ticket touched does not count)
        const uint32_t cs = lds_acquire(&ctl.commit_seq);
        while (!win_slow && e0 + wv_done < cs) { win_eval(wv_done); wv_done += 1; }
        uint32_t safe = 0;
#pragma unroll
        for (int q = 0; q < KC; ++q) if ((uint32_t)q < ncand && (win_slow || __ballot(wv_c == cand_c[q]) == 0)) safe += 1;
        if (win_slow) safe = safe > cs - e0 ? safe - (cs - e0) : 0u;
        if (ncand == (uint32_t)KC || safe >= i - cs + 1) break;
        if (DBG) tra = clock64();
        const uint64_t k = next_survivor(klo);
        if (DBG) trb = clock64();
        if (k == ~0ull) { complete = true; break; }
        const uint32_t c = (uint32_t)k & 0xFFFFFu;
        ClaimRegs Rt;
        const bool fe = exact(c, KG_NOSLOT, Rt);
        if (fe) {
#pragma unroll
          for (int q = 0; q < KC; ++q) if ((uint32_t)q == ncand) { CR[q] = Rt; cand_c[q] = c; cand_k[q] = k; }
          ncand += 1;
        }
        klo = k;
      }
      if (ncand == 0) fresh(F);  // no claim takes the pod as things stand: its new NodeClaim, ahead of its turn
    }
    KG_TK(1);
    // ---- wait for this ticket's turn
    for (uint32_t polls = 0;; ++polls) {
      const uint32_t cs = lds_acquire(&ctl.commit_seq);
      if (DBG && polls == 0 && cs == i) tr[6] += 1000;  // (late: the token was waiting for this wave)
      while (!win_slow && e0 + wv_done < cs) { win_eval(wv_done); wv_done += 1; }  // (cs <= i: entries of the window only)
      if (cs == i) break;
      if ((lds_relaxed(&ctl.done) | lds_relaxed(&ctl.err)) != 0) { stop = true; break; }
      if (polls > KG_SPEC_POLLS) { lds_store(&ctl.err, 2u); stop = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (stop) break;
    KG_TK(2);
    // ================= under the token: every ticket before this one is retired, none after it has written anything
    const uint32_t tp_now = lds_relaxed(&ctl.total_pushed), epoch = lds_relaxed(&ctl.epoch);
    const uint32_t qlen = tp_now - i;  // Queue length before this pop
    if (qz == epoch && qw == qlen) { lds_release(&ctl.done, 1u); break; }  // every remaining pod failed with the queue at this length
    const uint32_t nclaims = lds_relaxed(&ctl.nclaims), seq = lds_relaxed(&ctl.seq);
    if (DBG) { ts0 = clock64(); if (i != 0) tr[4] += ts0 - ctl.t_rel; }
    int path = 3;
    bool d_in_window = false;  // the claim the pod lands on took a pod from a ticket of the window (its global stores may be in flight)
    {
      const uint32_t wn = i - e0;  // (< W)
      uint32_t wc = KG_NOIDX;
      if (win_slow) { if (lane < wn) wc = ctl.log[(e0 + lane) % KG_SPEC_LOG]; }
      else {
        wc = wv_c;
        // a claim that took pods from two tickets of the window: the later entry saw it settled
        for (uint32_t q = 1; q < wn; ++q) { const uint32_t cq = rl32(wv_c, q); if (lane < q && wv_c == cq) wv_ok = false; }
      }
      // the LDS copy of a claim's row as the latest ticket of the window left it (KG_NOSLOT: the claim is not in the window)
      auto win_slot = [&](uint32_t c) -> uint32_t {
        const unsigned long long b = __ballot(wc == c);
        return b ? (e0 + 63u - (uint32_t)__builtin_clzll(b)) % KG_SPEC_SLOTS : KG_NOSLOT;
      };
      // (debug, DevProblem::stats bit 6: no verdict of the speculative evaluation is used -- every pod is evaluated at its turn)
      const bool serial = (P.stats & 64u) != 0;
      if (__builtin_expect(serial, 0)) {
        st_rescan += 1; path = 5;
        F.valid = false;
        uint64_t klo = 0;
        while (true) {
          const uint64_t k = scan_all(nclaims, klo);
          if (k == ~0ull) break;
          const uint32_t c = (uint32_t)k & 0xFFFFFu;
          if (exact(c, win_slot(c), Rd)) { d = c; keyd = k; break; }
          klo = k;
        }
      } else {
        // challengers: claims of the window whose key lies in (klo, khi) and that take the pod, in scan order
        auto challengers = [&](uint64_t klo, uint64_t khi) -> bool {
          while (true) {
            uint64_t k;
            if (win_slow) k = scan_window(wc, wn, klo, khi);
            else {
              k = ~0ull;
              for (unsigned long long q = __ballot(wv_ok && wv_key > klo && wv_key < khi); q; q &= q - 1) {
                const uint64_t kq = rl64(wv_key, (uint32_t)(__ffsll(q) - 1));
                k = kq < k ? kq : k;
              }
            }
            if (k == ~0ull) return false;
            st_chal += 1;
            const uint32_t c = (uint32_t)k & 0xFFFFFu;
            ClaimRegs Rc;
            if (exact(c, win_slot(c), Rc)) { d = c; keyd = k; Rd = Rc; return true; }
            klo = k;
          }
        };
        // The candidates outside the window kept their verdicts; the first of them bounds what else matters: the claims of
        // the window (the pod's own earlier choices among them, where they are NOW), evaluated on their settled rows in scan
        // order up to it.  With every candidate in the window and the list cut short, the claims behind the last candidate
        // were never looked at: scanned now.
        uint32_t jstar = KG_NOIDX;
#pragma unroll
        for (int q = KC - 1; q >= 0; --q) if ((uint32_t)q < ncand && __ballot(wc == cand_c[q]) == 0) jstar = (uint32_t)q;
        uint64_t bound = ~0ull;
#pragma unroll
        for (int q = 0; q < KC; ++q) if ((uint32_t)q == jstar) bound = cand_k[q];
        if (DBG) tra = clock64();
        const bool taken = challengers(0, bound);
        if (DBG) { trb = clock64(); tr[0] += tra - ts0; tr[1] += trb - tra; }
        if (taken && jstar == 0) path = 4;
        else if (taken) path = 5;
        else if (jstar != KG_NOIDX) {
#pragma unroll
          for (int q = 0; q < KC; ++q) if ((uint32_t)q == jstar) { d = cand_c[q]; keyd = cand_k[q]; Rd = CR[q]; }
          if (jstar == 0) st_fast += 1; else path = 5;
        } else if (ncand != 0) path = 5;
        if (jstar == KG_NOIDX && ncand != 0 && !complete) {
          // (more tickets in flight than candidates kept: what lies between the last candidate and the window claim that took
          // the pod -- or the end -- was never looked at)
          uint64_t klo = 0;
#pragma unroll
          for (int q = 0; q < KC; ++q) if ((uint32_t)q + 1 == ncand) klo = cand_k[q];
          const uint64_t khi = keyd;  // (all ones when no window claim took it)
          if (DBG) tr[7] += 1000;
          while (klo < khi) {
            const uint64_t k = scan_range(nclaims, klo, khi);
            if (k == ~0ull) break;
            const uint32_t c = (uint32_t)k & 0xFFFFFu;
            klo = k;
            ClaimRegs Rc;
            if (exact(c, win_slot(c), Rc)) { d = c; keyd = k; Rd = Rc; break; }
          }
        }
        if (path == 5) { st_rescan += 1; if (DBG) { tr[2] += clock64() - trb; tr[5] += 1000; } }
      }
      d_in_window = d != KG_NOIDX && __ballot(wc == d) != 0;
    }
    if (DBG) ts1 = clock64();
    uint32_t reason = 0;
    bool placed = false;
    uint32_t logc = KG_NOIDX;
    uint64_t pre_v = 0;  // lanes 0..7: the prefilter words of the claim after this pod
    const uint32_t slot = i % KG_SPEC_SLOTS;
    uint64_t* const lrow = rowc + (size_t)slot * KG_ROW_WORDS;
    if (d != KG_NOIDX) {
      // ---- commit, the part the next ticket may need at once: the merged row into this ticket's LDS slot, the claim's
      // prefilter record.  The global row follows when the token has been passed on.
      const uint32_t c = d, tm = (uint32_t)Rd.w6, wnp = (uint32_t)(Rd.w6 >> 32), stamp = (uint32_t)Rd.w7, hst = (uint32_t)(Rd.w7 >> 32);
      if (lane < KG_MAXKA) ((uint32_t*)(lrow + KG_ROW_META))[lane] = Rd.meta;
      lrow[KG_ROW_BITS + lane] = Rd.bits;
      if ((lane & 7) == 0) lrow[KG_ROW_REQ + rg] = (uint64_t)Rd.req;
      if (lane < KG_TWLDS) lrow[KG_ROW_MASK + lane] = Rd.m;  // (zero in lanes beyond TW)
      {
        const uint64_t zw = zslot >= 0 ? plain_set_or_all(rl32(Rd.meta, (uint32_t)zslot), rl64(Rd.bits, zoff)) : ~0ull;
        const uint64_t cw = cslot >= 0 ? plain_set_or_all(rl32(Rd.meta, (uint32_t)cslot), rl64(Rd.bits, coff)) : ~0ull;
        uint64_t v = (uint64_t)(Rd.phd - (lane < 4 ? pc.preq[lane < 4 ? lane : 0] : 0));
        const uint64_t n6 = (uint64_t)tm | ((uint64_t)(wnp + 1) << 32), n7 = (uint64_t)(seq + 1) | ((uint64_t)hst << 32);
        v = lane == 4 ? zw : (lane == 5 ? cw : (lane == 6 ? n6 : (lane == 7 ? n7 : v)));
        pre_v = v;
        if (lane < 8) lrow[lane] = v;
        if (c < pfcap) {
          if (lane < 8) pf[(size_t)lane * pfcap + c] = v;
          if (lane < TW) pf[(size_t)(8 + lane) * pfcap + c] = Rd.m;
        }
      }
      if (lane == 0) {
        if (P.has_ports) S.c_ports[c] = S.c_ports[c] | pc.vc.hp_use;
        ctl.last_c = c; ctl.last_new = 0; ctl.last_npods = wnp; ctl.last_stamp = stamp;
        ctl.seq = seq + 1;
      }
      if (P.has_ports) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the port word is read from global memory by the next ticket)
      logc = c;
      placed = true;
    } else {
      // ---- a new NodeClaim: from the evaluation made ahead of the turn unless a limit has changed since
      path = 6;
      if (!F.valid || F.lim_epoch != lds_relaxed(&ctl.lim_epoch)) { st_refresh += 1; fresh(F); }
      const DevProblem& Pr = rare_view(Pk);
      const SimState& Sr = rare_view(Sk);
      reason = F.reason;
      if (nclaims >= (uint32_t)Pr.Cmax) { F.t = KG_NOIDX; reason = 0; }  // (no room for another NodeClaim: no template is tried)
      if (F.t != KG_NOIDX) {
        const uint32_t c = nclaims, t = F.t, NP = (uint32_t)Pr.NP;
        KG_GLOBAL uint64_t* row = Sr.c_row + (size_t)c * KG_ROW_WORDS;
        if (lane < KG_MAXKA) ((KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane] = F.meta;
        row[KG_ROW_BITS + lane] = F.bits;
        if ((lane & 7) == 0) row[KG_ROW_REQ + rg] = (uint64_t)F.req;
        if (lane < KG_TWLDS) row[KG_ROW_MASK + lane] = F.m;
        // (... and into this ticket's LDS slot, as every commit: the claim is in the log window of the tickets in flight)
        if (lane < KG_MAXKA) ((uint32_t*)(lrow + KG_ROW_META))[lane] = F.meta;
        lrow[KG_ROW_BITS + lane] = F.bits;
        if ((lane & 7) == 0) lrow[KG_ROW_REQ + rg] = (uint64_t)F.req;
        if (lane < KG_TWLDS) lrow[KG_ROW_MASK + lane] = F.m;
        {
          const uint64_t zw = zslot >= 0 ? plain_set_or_all(rl32(F.meta, (uint32_t)zslot), rl64(F.bits, zoff)) : ~0ull;
          const uint64_t cw = cslot >= 0 ? plain_set_or_all(rl32(F.meta, (uint32_t)cslot), rl64(F.bits, coff)) : ~0ull;
          const uint64_t n6 = (uint64_t)t | (1ull << 32), n7 = (uint64_t)(seq + 1) | ((uint64_t)(seq + 1) << 32);
          const uint64_t v = lane < 4 ? (uint64_t)F.hd : (lane == 4 ? zw : (lane == 5 ? cw : (lane == 6 ? n6 : n7)));
          if (lane < 8) { row[lane] = v; lrow[lane] = v; }
          if (c < pfcap) {
            if (lane < 8) pf[(size_t)lane * pfcap + c] = v;
            if (lane < TW) pf[(size_t)(8 + lane) * pfcap + c] = F.m;
          }
        }
        const bool has_lim = Pr.tmpl_has_limits[t] != 0;
        if (has_lim && lane < NR && Pr.tmpl_limit_res[(size_t)lane * NP + t]) Sr.remaining[(size_t)lane * NP + t] -= F.cx;
        if (lane == 0) {
          Sr.pod_kind[pod_row(Sr, pod)] = 2;
          Sr.pod_index[pod_row(Sr, pod)] = (int32_t)c;
          if (Pr.has_ports) Sr.c_ports[c] = pc.vc.hp_use;
          Sr.c_hstamp[c] = seq + 1;
          ctl.last_c = c; ctl.last_new = 1;
          ctl.seq = seq + 1;
        }
        // the claim becomes visible to the other waves only now that its row and record are complete
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
          if (has_lim) lds_release(&ctl.lim_epoch, F.lim_epoch + 1);
          ctl.nclaims = c + 1;  // (the token holders' count; the speculative evaluations' count follows with the ticket's visibility)
        }
        logc = c;
        st_new += 1;
        placed = true;
      }
    }
    if (__builtin_expect(!placed, 0)) {
      // ---- Preferences.Relax + Queue.Push
      path = 7;
      const SimState& Sr = rare_view(Sk);
      const int32_t vnext = (int32_t)KG_UNI((uint32_t)pc.vc.v.next);
      st_fail += 1;
      if (lane == 0) {
        Sr.pod_kind[pod_row(Sr, pod)] = 0;
        Sr.pod_index[pod_row(Sr, pod)] = -1;
        Sr.pod_reason[pod_row(Sr, pod)] = reason ? reason : 8;
        ctl.last_c = KG_NOIDX; ctl.last_new = 0;  // (the reference sorted before this pod too: no claim is out of place)
        uint4 qn;
        qn.x = pod;
        if (vnext >= 0) { qn.y = (uint32_t)vnext; qn.z = 0; qn.w = 0; ctl.relax = ctl.relax + 1; ctl.epoch = epoch + 1; }
        else { qn.y = variant; qn.z = epoch; qn.w = qlen; }  // (the Queue's length after this pop and this push)
        ((KG_GLOBAL uint4*)Sr.queue)[tp_now % QCAP] = qn;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) lds_release(&ctl.total_pushed, tp_now + 1);
    }
    // ---- retire the ticket: what the next ticket's turn needs of it is in LDS (a new NodeClaim and a failure also wrote
    // global memory, and waited for it above); the release orders it before the token
    if (DBG) ts2 = clock64();
    if (lane == 0) ctl.log[i % KG_SPEC_LOG] = logc;
    if (DBG && lane == 0) ctl.t_rel = clock64();
    if (lane == 0) lds_release(&ctl.commit_seq, i + 1);
    if (DBG) { ts3 = clock64(); if (path == 3) { ts[0] += ts0 - tk_last; ts[1] += ts1 - ts0; ts[2] += ts2 - ts1; ts[3] += ts3 - ts2; } else if (path == 5) { ts[4] += ts1 - ts0; ts[5] += ts3 - ts1; } }
    // ================= token passed on
    if (DBG) { KG_TK(path); tk[9] += 1; }
    // ---- the rest of the commit: the claim's row in global memory (what speculative evaluations read) and the pod's result.
    // Two tickets that wrote the same claim must not have their row stores in flight at once: wait for the earlier ones.
    if (d != KG_NOIDX) {
      if (d_in_window) {
        for (uint32_t polls = 0;; ++polls) {
          if (lds_acquire(&ctl.vis_seq) == i) break;
          if (lds_relaxed(&ctl.err) != 0) break;
          if (polls > KG_SPEC_POLLS) { lds_store(&ctl.err, 3u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      KG_GLOBAL uint64_t* row = S.c_row + (size_t)d * KG_ROW_WORDS;
      if (lane < KG_MAXKA) ((KG_GLOBAL uint32_t*)(row + KG_ROW_META))[lane] = Rd.meta;
      row[KG_ROW_BITS + lane] = Rd.bits;
      if ((lane & 7) == 0) row[KG_ROW_REQ + rg] = (uint64_t)Rd.req;
      if (lane < KG_TWLDS) row[KG_ROW_MASK + lane] = Rd.m;
      if (lane < 8) row[lane] = pre_v;
      if (lane == 0) {
        S.pod_kind[pod_row(S, pod)] = 2;
        S.pod_index[pod_row(S, pod)] = (int32_t)d;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // visibility is published in ticket order
    for (uint32_t polls = 0;; ++polls) {
      if (lds_acquire(&ctl.vis_seq) == i) break;
      if (lds_relaxed(&ctl.err) != 0) break;
      if (polls > KG_SPEC_POLLS) { lds_store(&ctl.err, 4u); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      if (logc != KG_NOIDX && logc + 1 > ctl.nclaims_vis) ctl.nclaims_vis = logc + 1;
      lds_release(&ctl.vis_seq, i + 1);
    }
    // ---- the next ticket's context -> the other buffer
    have_next = next_ok;
    have_next2 = next2_ok;
    nq2 = ne8;
    if (next_ok) {
      __attribute__((address_space(3))) uint64_t* dst = (__attribute__((address_space(3))) uint64_t*)(cur ? mybuf[0] : mybuf[1]);
      const unsigned long long nb = __ballot(n_rq != 0);
      uint64_t w0 = iw[0];
      if (lane < KG_MAXR) w0 = (uint64_t)n_rq;          // words 0..7: the pod's requests
      if (lane == KG_MAXR) w0 = (uint64_t)(uint32_t)nb;  // word 8: need
      dst[lane] = w0;
#pragma unroll
      for (int k = 1; k < 4; ++k) { const uint32_t q = lane + 64 * k; if (q < sizeof(PodCtx) / 8) dst[q] = iw[k]; }
      nq = ne4;
    }
  }
#undef KG_TK
  // ---- every wave has left the loop: the columns the host decodes, from the rows
  if (lane == 0) {
    atomicAdd(&ctl.stat[0], st_fast); atomicAdd(&ctl.stat[1], st_chal); atomicAdd(&ctl.stat[2], st_rescan); atomicAdd(&ctl.stat[3], st_new);
    atomicAdd(&ctl.stat[4], st_fail); atomicAdd(&ctl.stat[5], st_exact); atomicAdd(&ctl.stat[6], st_refresh);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  const DevProblem& Pe = rare_view(Pk);
  const SimState& Se = rare_view(Sk);
  const uint32_t Ka = (uint32_t)Pe.Ka, Wa = (uint32_t)Pe.Wa;
  const size_t Cmax = (size_t)Pe.Cmax;
  const uint32_t nclaims = ctl.nclaims;
  for (uint32_t c = wave; c < nclaims; c += W) {
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
  if (DBG && lane == 0) {
#pragma unroll
    for (int k = 0; k < 10; ++k) Se.counters[64 + 16 * wave + k] = tk[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Se.counters[64 + 16 * wave + 10 + k] = ts[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) Se.counters[64 + 16 * KG_SPEC_MAXW + 8 * wave + k] = tr[k];
  }
  if (tid == 0) {
    Se.header[0] = nclaims; Se.header[1] = ctl.relax;
    Se.header[2] = ctl.last_c; Se.header[3] = ctl.last_npods; Se.header[4] = ctl.last_stamp; Se.header[5] = ctl.last_new;
    Se.header[15] = ctl.err;
    for (int k = 0; k < 7; ++k) Se.counters[48 + k] = ctl.stat[k];
    Se.counters[55] = pfcap; Se.counters[56] = off; Se.counters[57] = pool_words;
  }
}

}  // namespace kg
