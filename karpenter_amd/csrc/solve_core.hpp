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
//     existing node / per in-flight NodeClaim / per 64-type word of a NodeClaim's option mask)
//     and, for consolidation, across simulations (workgroup per simulation).  Nothing crosses a
//     CU inside a Solve, so the only barrier is the workgroup barrier.
//   * Everything is a bitset over a vocabulary compiled on the host: requirement value sets per
//     label key, and -- the part that removes the per-type inner loop -- instance types
//     themselves: each NodeClaim carries a T-bit "remaining types" mask, and "compatible",
//     "fits" and "hasOffering" become ANDs with precomputed type masks (per requirement-entry
//     masks from k_type_filter, per-resource threshold masks, per-(zone,capacity-type)
//     offering masks).  first-feasible-index = min over feasible rows in scan order.
//   * One placement = (1) pop + one coalesced load of the pod's precompiled VariantCtx into LDS,
//     (2) a streaming prefilter over all in-flight NodeClaims (thread per claim; conservative:
//     toleration bit, per-resource headroom, option mask AND the pod's own type mask),
//     (3) the exact NodeClaim.Add on the survivors in scan order, 16 at a time: each claim's requirement row is
//     copied to LDS, the pod's requirement entries and resources are merged into it by one thread each, the
//     instance-type filter runs one thread per 64-type word, (4) the winner's merged row is written back.
#pragma once
#include "kg_hd.hpp"

#ifndef KG_EXP
#define KG_EXP 0
#endif
namespace kg {

constexpr int KG_MAXR = 8;      // active resources in one solve (requested by a pod, daemon overhead, or limited)
constexpr int KG_MAXKA = 32;    // active label keys in one solve
constexpr int KG_MAXE = 8;      // entries (keys) per requirement set
constexpr int KG_MAXSPEC = 6;   // "special" 1-word keys: zone, capacity-type, zone-id, os + topology keys
constexpr int KG_MAXLIST = 8;   // topology groups consulted for one pod
constexpr int KG_EBITS = 32;    // requirement-value words of one pod staged in LDS
constexpr int KG_THR_LDS = 1024;  // fit-threshold values staged in LDS
constexpr int KG_ORD = 1024;      // in-flight NodeClaims whose scan order is kept in LDS (global memory beyond)
constexpr int KG_MAXGP = 24;      // NodeClaims evaluated exactly per pass
constexpr int KG_TL = 16;         // touched nodes of a simulation kept in LDS (those beyond: global list + hash)
constexpr int KG_NTMAX = 1024;    // threads of the largest workgroup solve_block is instantiated for
constexpr int KG_GP0 = 8;         // ... in the first pass over a chunk of the scan order
constexpr int KG_ROWW = 64;       // requirement-value words of one claim row
constexpr int KG_TWLDS = 16;      // option-mask words (<= 1024 instance types) for the tables staged in LDS
#if KG_DEVICE_ONLY
#define KG_CONST4 __attribute__((address_space(4)))
#else
#define KG_CONST4
#endif
constexpr int KG_NSCAN = 4;       // chunks of existing nodes one barrier of a Solve's dense scan covers
constexpr int KG_SPV = 16;        // values per special key whose value->types rows are staged in LDS
constexpr int KG_OFFROWS = 32;    // (zone, capacity-type) offering rows staged in LDS
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
  uint32_t pm_all;         // the pod's requirement set is empty: it rules out no instance type (its option mask is all ones)
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

// One pod requirement entry, precompiled (what NodeClaim.Add needs of it, without pointer chasing)
struct PodEntry {
  uint32_t e;      // global entry index (row of e_amask)
  uint32_t meta;
  uint16_t aoff;   // word offset of the slot in a claim's requirement row
  uint16_t bits;   // word offset into VariantCtx::ebits, 0xFFFF = read rs_bits[e_bits[e]]
  uint8_t slot;
  int8_t spec;     // special index or -1
  uint8_t w;       // words of the slot
  uint8_t opc;     // operator class of the entry
};
struct ListX {
  ListEntry le;
  Group g;
};
// Everything one pod variant brings to a placement; one coalesced load per Queue.Pop.
struct VariantCtx {
  Variant v;
  uint32_t ne, nlist;
  uint64_t hp_use, hp_conf;  // host ports the pod reserves / port ids that conflict with one of them (scheduling.HostPortUsage)
  uint64_t pzone, pct;       // the pod's own zone / capacity-type sets when they are plain non-empty sets, else all ones (prefilter)
  PodEntry ent[KG_MAXE];
  ListX lst[KG_MAXLIST];
  uint64_t ebits[KG_EBITS];
};
static_assert(sizeof(VariantCtx) % 8 == 0, "VariantCtx is copied as 64-bit words");

// Read-only problem tables (device pointers).
struct DevProblem {
  int R, Ka, Wa, T, TW, TWP, NP, N, NW, P, S, G, GH, H, Cmax, nspec, NL, NV;
  int slot_zone, slot_ct, slot_zid;  // special indices (sp) or -1
  uint32_t well_known_slots;         // bit per active slot
  uint32_t stats;                    // 1 = maintain the reference-visit counters (exact, slower)
  uint32_t has_ports;                // some pod of this solve (pending or bound) declares a host port
  unsigned long long tmpl_minv_mask;  // bit per template: its requirements carry minValues
  // active resources -> rows of the catalog's per-resource tables
  // (kept behind a pointer: an array inside this by-value struct would push the whole struct into scratch memory)
  const uint32_t* atab;  // [5][KG_MAXR]: thr_off, thr_n, cthr_off, cthr_n, global resource id
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
  const VariantCtx* vctx;       // [NV]
  const uint64_t* podmask;      // [NV][TW] k_pod_mask output: types the pod's own requirements can ever accept
  const struct PodCtx* pctx;    // [NV] k_pod_ctx output: everything a Queue.Pop needs of the variant, contiguous (requests left zero)
  const uint32_t* pod_variant0; // [P]
  const int64_t* preq;          // [P][R]
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
  // catalog tables (rows indexed by GLOBAL resource id; use a_* for the active ones)
  const int64_t* thr_val; const uint64_t* thr_mask;    // alloc >= v
  const int64_t* cthr_val; const uint64_t* cthr_mask;  // cap <= v
  const uint64_t* valid_types;  // [TW]
  const int64_t* cap;           // [RG][T]
  const int64_t* alloc;         // [RG][T]
  const uint64_t* offmask;      // [64 zones][4 cts][TW] (dense over value ids of the zone / ct slots)
  const uint8_t* zone_zid;      // [64] zone value id -> zone-id value id, 0xFF none
  int n_zone_vals, n_ct_vals;
  // existing nodes (scan order)
  uint32_t gp_cap;              // evaluation groups (Shared::rows) the launch has LDS for: KG_MAXGP, or fewer for k_simulate
  uint32_t sparse_nodes;        // batches of simulations: a simulation keeps what it adds to existing nodes as a short list of
                                // touched nodes (SimState::tnode ...) over the shared nfree0 / nports0 instead of its own [R][N] copy
  const int64_t* nfree0;        // [R][N] what the node can still take when the Solve starts: available - (remaining daemonset) requests
  const uint64_t* nports0;      // [N] host-port ids reserved on the node by its bound pods
  const uint64_t* nlab;         // [S][NW] label compat per requirement set (k_node_static)
  const uint64_t* ntol;         // [TS][NW]
  const uint16_t* ndom;         // [nspec][N] node's value id on each special key, 0xFFFF absent
  // topology
  const Group* groups;
  const uint32_t* filters;      // requirement-set ids
};

// Mutable state of ONE simulation (device pointers into that simulation's arena).
struct SimState {
  KG_GLOBAL int64_t* nfree;     // [R][N] (dense mode: one Solve) nfree0 minus what this Solve placed on the node
  KG_GLOBAL uint64_t* nports;   // [N] host-port ids reserved on the node (bit per entry of the solve's port vocabulary)
  // sparse mode (DevProblem::sparse_nodes): nodes this simulation has put pods on (entries KG_TL.. of the list; the first KG_TL are in LDS, Shared::tl_*)
  KG_GLOBAL uint64_t* ntouch;   // [NW] bit per node: it has an entry below
  KG_GLOBAL uint32_t* tnode;    // [Cmax] node (scan position)
  KG_GLOBAL uint64_t* thash;    // [thash_mask + 1] open addressing, (node + 1) << 32 | entry, 0 = empty: node -> its entry in O(1)
                                // (a multi-node prefix touches hundreds of nodes; every thread that meets one looks it up)
  uint32_t thash_mask;
  KG_GLOBAL int64_t* tadd;      // [R][Cmax] requests this simulation added to it
  KG_GLOBAL uint64_t* tports;   // [Cmax] its host-port word now
  KG_GLOBAL uint64_t* c_ports;  // [Cmax] ... on the NodeClaim
  KG_GLOBAL uint64_t* c_row;    // [Cmax][112] claim rows of the single-wave solve (solve_wave.hpp); device-only scratch behind the state image
  KG_GLOBAL int64_t* c_req;     // [R][Cmax]
  KG_GLOBAL int64_t* c_head;    // [R][Cmax] upper bound of what the claim can still take: max allocatable over its options - c_req
  KG_GLOBAL uint64_t* c_mask;   // [TW][Cmax]
  KG_GLOBAL uint64_t* c_bits;   // [Wa][Cmax]
  KG_GLOBAL uint32_t* c_meta;   // [Ka][Cmax]
  KG_GLOBAL uint32_t* c_npods;  // [Cmax]
  KG_GLOBAL uint16_t* c_tmpl;   // [Cmax]
  KG_GLOBAL uint32_t* c_hstamp; // [Cmax] c_stamp at the last exact recomputation of c_head
  KG_GLOBAL uint32_t* c_stamp;  // [Cmax] step of the claim's creation / last pod: with c_npods it is the claim's place in the
                      // reference's (stable) sort.Slice order -- see claim_key()
  KG_GLOBAL int64_t* remaining; // [R][NP]
  KG_GLOBAL int32_t* gcount;    // [G][64]
  KG_GLOBAL uint64_t* gdom;     // [G] registered domains
  KG_GLOBAL uint64_t* gpos;     // [G] domains with count > 0
  KG_GLOBAL int32_t* hcount;    // [GH][H]
  KG_GLOBAL int32_t* hpos;      // [GH] hosts with count > 0
  KG_GLOBAL uint32_t* queue;    // [P+1][4] ring of {pod, variant, lastLen epoch, lastLen value}: one 16-byte load per Queue.Pop
  KG_GLOBAL uint8_t* pod_kind;  // [P]
  KG_GLOBAL int32_t* pod_index; // [P]
  KG_GLOBAL uint32_t* pod_reason; // [P]
  KG_GLOBAL uint64_t* scratch;  // [TW+1][Cmax] candidate option masks of the survivors; row TW: per-claim visit statistic
  KG_GLOBAL uint32_t* header;   // [16]: 0 nclaims, 1 relaxations, 2-5 the claim touched after the last sort (id, old pods, old stamp, new)
  KG_GLOBAL unsigned long long* ordg; // [2][Cmax] scan order beyond KG_ORD claims (same packing as Shared::ord)
  const KG_GLOBAL uint64_t* excl; // [NW] existing nodes left out of this simulation (consolidation candidates)
  uint32_t n_shared, pod_base, qcap;  // pods [0,n_shared) and [pod_base, ...) take part; result rows are local indices
  KG_GLOBAL uint64_t* counters; // [32]: 8.. phase timers and scan statistics when stats&2; [0..3]: visited_existing, visited_claims, visited_claim_types, new_claim_scans
};

// One evaluation group's working copy of a NodeClaim (LDS): the claim's requirement row is loaded here, the pod is
// merged into it in place (NodeClaim.Add steps 3-5), and the winner's copy is what the commit writes back.
struct Row {
  uint32_t c, tmpl, ok, reason, pos, tv;  // pos: the claim's place in the scan order
  uint32_t npods, stamp;
  uint32_t ent_changed, ent_dne_ok, sp_changed, thr_use, has_topo, pad_;
  int32_t thr_idx[KG_MAXR];
  int64_t req[KG_MAXR];            // claim requests; after the merge: requests including the pod
  uint64_t topo[KG_MAXSPEC];       // Topology.AddRequirements result per special key
  uint64_t mw[2][KG_TWLDS];        // the two halves of the type filter when two waves share a claim
  uint32_t meta[KG_MAXKA];         // requirement meta per active key
  uint64_t bits[KG_ROWW];          // requirement value bits, Wa words
};

// Everything one Queue.Pop brings into LDS: the pod variant's precompiled record, the pod's requests, its own type mask
// (prefilter) and the type masks of its requirement entries.
struct PodCtx {
  int64_t preq[KG_MAXR];
  uint32_t need, pad_;              // bit per active resource the pod requests
  uint64_t tolw;                    // tol_tmpl word of the pod's toleration set: bit per template whose taints it tolerates
  VariantCtx vc;
  uint64_t podmask[KG_TWLDS];
  uint64_t eam[KG_MAXE][KG_TWLDS];  // e_amask rows of the pod's requirement entries
};
static_assert(sizeof(PodCtx) % 8 == 0, "PodCtx is copied as 64-bit words");
static_assert(sizeof(PodCtx) <= 4 * 64 * 8, "the single-wave kernel prefetches a pod context as four 8-byte words per lane");

struct SharedBase {
  uint32_t best2[2], reason, pf_topo;  // first feasible existing node (slot = placement parity); reasons of the pod being placed;
                                     // pf_topo: the prefilter turned a NodeClaim away for a topology group (-> reason bit 4)
  uint32_t thr_lds, tw_lds, off_lds, stw_lds;
  uint32_t fresh_lim[2];            // per template attempt (alternating): some instance type is within the limits
  uint32_t feas[2], mvf[2];         // per pass (alternating): rows whose option mask stays non-empty / rows failing minValues
  uint32_t endk, need;              // end of the winner's pod-count group in the scan order; resources the pod requests
  uint32_t thr_stride0;             // first stride of the 8-ary fit-threshold search: 8^k with 8^(k+1) >= the longest threshold list
  uint32_t ffeas[4];                // wave-local evaluation (fast path): feasible rows of pass k in slot k & 3, slot (k + 1) & 3 cleared meanwhile
  unsigned long long ball[16];      // prefilter ballots of the current chunk of scan positions, one word per wave
  unsigned long long tallow[KG_MAXLIST];  // per topology entry of the pod being placed: domains that could still take it (topo_allow)
  uint16_t wpos[16][KG_MAXGP];      // chunk positions of each wave's first survivors, in order
  uint4 qn[2];                      // the queue entry two behind the head, staged while a pod is placed (slot = placement parity)
  PodCtx pcb[2];                    // the pod being placed / the next pod of the queue, fetched while the current one is placed
                                    // (the roles swap at every Queue.Pop: no copy)
  uint64_t offm[KG_OFFROWS][KG_TWLDS];          // offering masks [zone * n_ct + ct]
  uint64_t stw[KG_MAXSPEC][KG_SPV][KG_TWLDS];   // value->types rows of the special keys
  // small problem tables
  uint16_t aoff[KG_MAXKA], aw[KG_MAXKA];
  int8_t slot2spec[KG_MAXKA];
  uint8_t spec_slot[KG_MAXSPEC + 2];
  uint32_t thr_base[KG_MAXR];
  uint32_t a_thr_off[KG_MAXR], a_thr_n[KG_MAXR], a_cthr_off[KG_MAXR], a_cthr_n[KG_MAXR], a_res[KG_MAXR];
  int64_t thr_val[KG_THR_LDS];
  // the in-flight NodeClaims in scan order (the reference's sorted slice), double-buffered: pod count << 32 | claim id
  unsigned long long ord[2][KG_ORD];
};
// ... plus the workgroup kernel's evaluation groups (the single-wave kernel keeps the claim row in registers and spends
// the LDS on its prefilter records instead)
struct Shared : SharedBase {
  // sparse mode: the first KG_TL nodes the simulation put pods on (a single-node simulation touches a handful) -- node,
  // requests added, host-port word.  Entries from KG_TL on live in SimState::tnode / tadd / tports / thash / ntouch.
  uint32_t tl_node[KG_TL];
  uint64_t tl_ports[KG_TL];
  int64_t tl_add[KG_TL][KG_MAXR];
  // evaluation groups of the current pass.  LAST member: k_simulate allocates only DevProblem::gp_cap of them (dynamic
  // LDS), which is what lets three of its workgroups share a CU.
  Row rows[KG_MAXGP];
};
// existing-node scan: the touched entry of the node each thread found feasible, read by commit_node after the barrier.
// It lives in the memory of the evaluation groups: the node phase and the NodeClaim phases of a placement never overlap.
KG_HD int32_t* scan_ntk(Shared& sh) { return (int32_t*)(void*)sh.rows; }
constexpr int KG_GPMIN = 4;        // evaluation groups a workgroup needs at least (one per wave of 256 threads; >= NT * 4 bytes for scan_ntk)
static_assert(sizeof(Row) * KG_GPMIN >= sizeof(int32_t) * KG_NTMAX, "scan_ntk fits the smallest allocation of rows");
static_assert(sizeof(Shared) <= 80 * 1024, "two workgroups of k_solve fit a CU's 160 KB of LDS (k_simulate allocates less: device.hip)");


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
  const auto* cnt = S.gcount + (size_t)gi * 64;
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

// The domains of a <= 64-domain group that could still take the pod, whatever NodeClaim it goes to: a NodeClaim whose
// requirement on the key is a plain value set disjoint from this mask fails topo_small + the merge that follows it
// (row_topology), because both only narrow the set.  ~0 = nothing can be said here (hostname groups, an affinity group
// that would bootstrap).  The prefilter of the NodeClaim scan tests it; the exact evaluation stays the judge.
KG_HD uint64_t topo_allow(const DevProblem& P, const SimState& S, const Group& g, const ListEntry& le) {
  if (g.spec == 0xFF) return ~0ull;
  const uint32_t gi = le.g;
  const bool self = le.flags & LF_SELF;
  const uint64_t podDom = ((uint64_t)le.poddom_hi << 32) | le.poddom_lo;
  const uint64_t gd = S.gdom[gi];
  if (g.type == TG_SPREAD) {
    const auto* cnt = S.gcount + (size_t)gi * 64;
    int32_t mn = 0x7FFFFFFF;
    int supported = 0;
    for (uint64_t m = gd & podDom; m; m &= m - 1) {
      const int d = ctz64(m);
      supported++;
      if (cnt[d] < mn) mn = cnt[d];
    }
    if (g.min_domains >= 0 && supported < g.min_domains) mn = 0;
    uint64_t al = 0;
    for (uint64_t m = gd; m; m &= m - 1) {
      const int d = ctz64(m);
      if ((int64_t)cnt[d] + (self ? 1 : 0) - mn <= g.max_skew) al |= 1ull << d;
    }
    return al;
  }
  if (g.type == TG_AFFINITY) {
    const uint64_t opts = gd & podDom & S.gpos[gi];
    return opts ? opts : ~0ull;
  }
  return gd & ~S.gpos[gi] & podDom;
}

KG_HD void record_small(const SimState& S, uint32_t gi, int d) {
  S.gcount[(size_t)gi * 64 + d] += 1;
  S.gdom[gi] |= 1ull << d;
  S.gpos[gi] |= 1ull << d;
}
KG_HD void record_host(const SimState& S, const DevProblem& P, const Group& g, uint32_t h) {
  auto* p = &S.hcount[(size_t)g.hidx * P.H + h];
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

// ------------------------------------------------------------------ NodeClaim.Add on a claim row held in LDS
KG_HD SReq row_special(const Shared& sh, const Row& R, int i) {
  const int a = sh.spec_slot[i];
  SReq r;
  r.meta = R.meta[a];
  r.word = (r.meta & RM_PRESENT) ? R.bits[sh.aoff[a]] : 0;
  return r;
}

// Requirement.Intersection of row slot a (w words at off) with (pm, pb), in place.  Returns the merged meta.
template <typename BitsPtr>
KG_HD uint32_t row_intersect(const DevProblem& P, Row& R, int a, int w, int off, uint32_t pm, BitsPtr pb, bool* any, bool* changed,
                             bool* cany) {
  const uint32_t cm = R.meta[a];
  const bool xc = cm & RM_COMPL, yc = pm & RM_COMPL;
  const uint32_t gt = rm_gt(cm) > rm_gt(pm) ? rm_gt(cm) : rm_gt(pm);
  const uint32_t lt = min_nz(rm_lt(cm), rm_lt(pm));
  const bool dne = gt && lt && P.gtval[P.bnd_off[a] + gt - 1] >= P.ltval[P.bnd_off[a] + lt - 1];
  const bool c2 = xc && yc && !dne;
  const uint32_t meta = c2 ? rm_make(true, gt, lt) : rm_make(false, 0, 0);
  bool anyb = false, chg = meta != cm, canyb = false;
  for (int i = 0; i < w; ++i) {
    const uint64_t x = R.bits[off + i], y = pb[i];
    canyb |= x != 0;
    uint64_t r = dne ? 0 : ((xc && yc) ? (x | y) : (xc ? (y & ~x) : (yc ? (x & ~y) : (x & y))));
    if (!dne && gt) r &= P.gtmask[P.bndmask_off[a] + (size_t)(gt - 1) * w + i];
    if (!dne && lt) r &= P.ltmask[P.bndmask_off[a] + (size_t)(lt - 1) * w + i];
    anyb |= r != 0;
    chg |= r != x;
    R.bits[off + i] = r;
  }
  R.meta[a] = meta;
  *any = anyb;
  *changed = chg;
  *cany = canyb;
  return meta;
}

// Step 3 of NodeClaim.Add for ONE pod requirement entry j: Requirements.Compatible(podReqs) + Add on that key.
// Entries of one requirement set are on distinct keys, so the entries of a pod merge in parallel.
template <typename BitsPtr>  // the entry's value words: LDS (VariantCtx::ebits) or the global requirement-set table
KG_HD void row_merge_entry_from(const DevProblem& P, const Shared& sh, const PodCtx& pc, Row& R, uint32_t j, bool fresh, const PodEntry pe, BitsPtr pb) {
  const int a = pe.slot;
  const uint32_t cm = R.meta[a];
  if (!(cm & RM_PRESENT)) {
    if (!((P.well_known_slots >> a) & 1) && !op_neg(pe.opc)) { R.ok = 0; atomic_or_u32(&R.reason, 2u); return; }
    if (op_neg(pe.opc)) atomic_or_u32(&R.ent_dne_ok, 1u << j);
    atomic_or_u32(&R.ent_changed, 1u << j);
    if (pe.spec >= 0) atomic_or_u32(&R.sp_changed, 1u << pe.spec);
    for (int i = 0; i < pe.w; ++i) R.bits[pe.aoff + i] = pb[i];
    R.meta[a] = pe.meta;
    return;
  }
  bool anyb, chg, canyb;
  const uint32_t mm = row_intersect(P, R, a, pe.w, pe.aoff, pe.meta, pb, &anyb, &chg, &canyb);
  const int cop = opclass(cm & RM_COMPL, canyb);
  const bool empty = !(mm & RM_COMPL) && !anyb;
  if (empty && !(op_neg(pe.opc) && op_neg(cop))) { R.ok = 0; atomic_or_u32(&R.reason, 2u); return; }
  if (op_neg(opclass(mm & RM_COMPL, anyb))) atomic_or_u32(&R.ent_dne_ok, 1u << j);
  if (chg || fresh) atomic_or_u32(&R.ent_changed, 1u << j);
  if (pe.spec >= 0 && chg) atomic_or_u32(&R.sp_changed, 1u << pe.spec);
}

KG_HD void row_merge_entry(const DevProblem& P, const Shared& sh, const PodCtx& pc, Row& R, uint32_t j, bool fresh) {
  const PodEntry pe = pc.vc.ent[j];
  // (two instantiations instead of one pointer select: a pointer that may be LDS or global memory compiles to flat loads)
  if (pe.bits != 0xFFFF) row_merge_entry_from(P, sh, pc, R, j, fresh, pe, pc.vc.ebits + pe.bits);
  else row_merge_entry_from(P, sh, pc, R, j, fresh, pe, P.rs_bits + P.e_bits[pe.e]);
}

// Step 5 for ONE active resource: requests including the pod, and the fit threshold they select.
KG_HD void row_merge_resource(const DevProblem& P, const Shared& sh, const PodCtx& pc, Row& R, int r, bool fresh) {
  const int64_t add = pc.preq[r];
  if (add == 0 && !fresh) return;
  const int64_t nr = R.req[r] + add;
  R.req[r] = nr;
  if (nr > 0) {
    R.thr_idx[r] = KG_UNI(sh.thr_lds) ? thr_lookup_ge(sh.thr_val + sh.thr_base[r], (int)sh.a_thr_n[r], nr)
                              : thr_lookup_ge(P.thr_val + sh.a_thr_off[r], (int)sh.a_thr_n[r], nr);
    atomic_or_u32(&R.thr_use, 1u << r);
  }
}

// Step 4: Topology.AddRequirements -> Compatible -> Add, on the special keys of the row (one thread per row).
KG_HD void row_topology(const DevProblem& P, const SimState& S, const Shared& sh, const PodCtx& pc, Row& R) {
  const uint32_t hslot = P.N + R.c;
  for (int i = 0; i < P.nspec; ++i) R.topo[i] = ~0ull;
  uint32_t has_topo = 0;
  for (uint32_t li = 0; li < pc.vc.nlist; ++li) {
    const ListEntry le = pc.vc.lst[li].le;
    const Group g = pc.vc.lst[li].g;
    const bool self = le.flags & LF_SELF;
    if (g.spec == 0xFF) {
      if (!topo_host(S, P, g, self, hslot)) { R.ok = 0; R.reason |= 4; return; }
      continue;
    }
    const uint64_t podDom = ((uint64_t)le.poddom_hi << 32) | le.poddom_lo;
    const uint64_t nodeDom = has_mask1(P, sh.spec_slot[g.spec], row_special(sh, R, g.spec));
    TopoOut o = topo_small(P, S, g, le.g, self, podDom, nodeDom);
    if (!o.ok) { R.ok = 0; R.reason |= 4; return; }
    R.topo[g.spec] &= o.opts;
    has_topo |= 1u << g.spec;
  }
  for (int i = 0; i < P.nspec; ++i) {
    if (!((has_topo >> i) & 1)) continue;
    const int a = sh.spec_slot[i];
    const SReq cur = row_special(sh, R, i);
    SReq inc{rm_make(false, 0, 0), R.topo[i]};
    SReq out;
    if (!(cur.meta & RM_PRESENT)) {
      // receiver lacks the key: allowed when well-known or the incoming operator is DoesNotExist
      if (!((P.well_known_slots >> a) & 1) && inc.word != 0) { R.ok = 0; R.reason |= 4; return; }
      out = inc;
      R.sp_changed |= 1u << i;
    } else {
      out = intersect1(P, a, cur, inc);  // incoming as AddRequirements builds it: node ∩ domains
      const bool empty = !(out.meta & RM_COMPL) && out.word == 0;
      const int eop = opclass(cur.meta & RM_COMPL, cur.word != 0);
      if (empty && !op_neg(eop)) { R.ok = 0; R.reason |= 4; return; }  // incoming is DoesNotExist-class when empty
      if (out.word != cur.word || out.meta != cur.meta) R.sp_changed |= 1u << i;
    }
    R.bits[sh.aoff[a]] = out.word;
    R.meta[a] = out.meta;
  }
}

// Step 6: filterInstanceTypesByRequirements in type-mask space: word w of the claim's remaining options after the pod.
// The filter is an AND of independent terms; it is cut in two halves of similar length so that two waves can each
// run one of them for the same claim (a wave issues one instruction per ~4 cycles: the stream's length is the cost).
// Half A: the claim's current options, the pod's entries on ordinary keys, the template's entries (fresh claim), fits.
KG_HD uint64_t row_word_a(const DevProblem& P, const SimState& S, const Shared& sh, const PodCtx& pc, const Row& R, bool fresh, int w) {
  uint64_t m = S.c_mask[(size_t)w * P.Cmax + R.c];
  const uint32_t ne = pc.vc.ne;
  // pod entries on ordinary keys
  for (uint32_t j = 0; j < ne; ++j) {
    if (!((R.ent_changed >> j) & 1)) continue;
    const PodEntry pe = pc.vc.ent[j];
    if (pe.spec >= 0) continue;
    uint64_t f;
    if (KG_UNI(sh.tw_lds)) { f = pc.eam[j][w]; KG_OPAQUE64(f); } else f = P.e_amask[(size_t)pe.e * P.TW + w];
    if ((R.ent_dne_ok >> j) & 1) f |= P.kU[(size_t)pe.slot * P.TW + w];
    m &= f;
  }
  if (fresh) {
    // template entries on ordinary keys (merged operator class was computed if the pod shares the key)
    const uint32_t t0 = P.rs_off[P.tmpl_rs[R.tmpl]], t1 = P.rs_off[P.tmpl_rs[R.tmpl] + 1];
    for (uint32_t e = t0; e < t1; ++e) {
      const int a = P.e_slot[e];
      if (sh.slot2spec[a] >= 0) continue;
      uint64_t f = P.e_amask[(size_t)e * P.TW + w];
      bool shared = false, dne_ok = false;
      for (uint32_t j = 0; j < ne; ++j)
        if (pc.vc.ent[j].slot == a) { shared = true; dne_ok = (R.ent_dne_ok >> j) & 1; }
      if (!shared) {
        const uint64_t* tb = P.rs_bits + P.e_bits[e];
        bool anyb = false;
        for (int i = 0; i < sh.aw[a]; ++i) anyb |= tb[i] != 0;
        dne_ok = op_neg(opclass(P.e_meta[e] & RM_COMPL, anyb));
      }
      if (dne_ok) f |= P.kU[(size_t)a * P.TW + w];
      m &= f;
    }
  }
  // fits
  {
    uint64_t tm[KG_MAXR];
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) {
      const bool use = r < P.R && ((R.thr_use >> r) & 1);
      const bool none = use && R.thr_idx[r] >= (int)sh.a_thr_n[r];
      uint64_t v = 0;
      if (r < P.R) v = P.thr_mask[((size_t)sh.a_thr_off[r] + ((use && !none) ? R.thr_idx[r] : 0)) * P.TW + w];  // under a scalar condition only
      tm[r] = !use ? ~0ull : (none ? 0ull : v);
    }
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) m &= tm[r];
  }
  if (fresh) m &= P.valid_types[w];
  return m;
}
// Half B: the special keys (types whose value set meets the merged allowed set) and hasOffering from zone / capacity type.
KG_HD uint64_t row_word_b(const DevProblem& P, const SimState& S, const Shared& sh, const PodCtx& pc, const Row& R, bool fresh, int w) {
  uint64_t m = ~0ull;
  bool off_use = fresh;
  uint64_t zmask = ~0ull, ctmask = ~0ull;
  for (int i = 0; i < P.nspec; ++i) {
    const int a = sh.spec_slot[i];
    const SReq sp = row_special(sh, R, i);
    const bool present = sp.meta & RM_PRESENT;
    const bool changed = ((R.sp_changed >> i) & 1) || (fresh && present);
    const uint64_t hm = has_mask1(P, a, sp);
    if (changed && present) {
      uint64_t f = P.kN[(size_t)a * P.TW + w];
      if (op_neg(opclass(sp.meta & RM_COMPL, sp.word != 0))) f |= P.kU[(size_t)a * P.TW + w];
      if (KG_UNI(sh.stw_lds)) {
        for (uint64_t mm = hm & ((1ull << KG_SPV) - 1); mm; mm &= mm - 1) f |= sh.stw[i][ctz64(mm)][w];
      } else {
        const int nv = P.ktwv_n[a];
        for (uint64_t mm = hm; mm; mm &= mm - 1) {
          const int d = ctz64(mm);
          if (d < nv) f |= P.twv[((size_t)P.ktwv_off[a] + d) * P.TW + w];
        }
      }
      m &= f;
    }
    if (i == P.slot_zone || i == P.slot_ct || i == P.slot_zid) off_use = off_use || changed;
    if (i == P.slot_zone) zmask &= hm;
    if (i == P.slot_ct) ctmask = hm;
    if (i == P.slot_zid) {
      uint64_t zm = 0;
      for (int z = 0; z < P.n_zone_vals; ++z) {
        const uint8_t zid = P.zone_zid[z];
        if (zid == 0xFF || ((hm >> zid) & 1)) zm |= 1ull << z;
      }
      zmask &= zm;
    }
  }
  // hasOffering
  if (off_use) {
    uint64_t f = 0;
    if (KG_UNI(sh.off_lds)) {
      for (int z = 0; z < P.n_zone_vals; ++z) {
        if (!((zmask >> z) & 1)) continue;
        for (int ct = 0; ct < P.n_ct_vals; ++ct)
          if ((ctmask >> ct) & 1) f |= sh.offm[z * P.n_ct_vals + ct][w];
      }
    } else {
      for (int z = 0; z < P.n_zone_vals; ++z) {
        if (!((zmask >> z) & 1)) continue;
        for (int ct = 0; ct < P.n_ct_vals; ++ct)
          if ((ctmask >> ct) & 1) f |= P.offmask[((size_t)z * 4 + ct) * P.TW + w];
      }
    }
    m &= f;
  }
  return m;
}

// InstanceTypes.SatisfiesMinValues over the candidate option mask in scratch[.][c]
KG_HD bool claim_minvalues_ok(const DevProblem& P, const SimState& S, uint32_t tmpl, uint32_t c) {
  for (uint32_t k = P.tmpl_minv_off[tmpl]; k < P.tmpl_minv_off[tmpl + 1]; ++k) {
    const int a = P.minv_slot[k];
    uint32_t distinct = 0;
    for (int d = 0; d < P.ktwv_n[a]; ++d) {
      bool hit = false;
      for (int w = 0; w < P.TW; ++w) hit |= (P.twv[((size_t)P.ktwv_off[a] + d) * P.TW + w] & S.scratch[(size_t)w * P.Cmax + c]) != 0;
      distinct += hit;
    }
    if (distinct < P.minv_min[k]) return false;
  }
  return true;
}

// Requirements.Compatible(row, requirement set rs, AllowUndefinedWellKnownLabels) without merging (Topology.Record's node filter)
KG_HD bool row_compatible_rs(const DevProblem& P, const Shared& sh, const Row& R, uint32_t rs) {
  for (uint32_t e = P.rs_off[rs]; e < P.rs_off[rs + 1]; ++e) {
    const int a = P.e_slot[e];
    const uint32_t pm = P.e_meta[e];
    const uint64_t* pb = P.rs_bits + P.e_bits[e];
    const int w = sh.aw[a], off = sh.aoff[a];
    bool panyb = false;
    for (int i = 0; i < w; ++i) panyb |= pb[i] != 0;
    const int pop = opclass(pm & RM_COMPL, panyb);
    const uint32_t cm = R.meta[a];
    if (!(cm & RM_PRESENT)) {
      if (!((P.well_known_slots >> a) & 1) && !op_neg(pop)) return false;
      continue;
    }
    const bool xc = cm & RM_COMPL, yc = pm & RM_COMPL;
    const uint32_t gt = rm_gt(cm) > rm_gt(pm) ? rm_gt(cm) : rm_gt(pm);
    const uint32_t lt = min_nz(rm_lt(cm), rm_lt(pm));
    const bool dne = gt && lt && P.gtval[P.bnd_off[a] + gt - 1] >= P.ltval[P.bnd_off[a] + lt - 1];
    bool anyb = false, canyb = false;
    for (int i = 0; i < w; ++i) {
      const uint64_t x = R.bits[off + i], y = pb[i];
      canyb |= x != 0;
      uint64_t r = dne ? 0 : ((xc && yc) ? (x | y) : (xc ? (y & ~x) : (yc ? (x & ~y) : (x & y))));
      if (!dne && gt) r &= P.gtmask[P.bndmask_off[a] + (size_t)(gt - 1) * w + i];
      if (!dne && lt) r &= P.ltmask[P.bndmask_off[a] + (size_t)(lt - 1) * w + i];
      anyb |= r != 0;
    }
    const bool empty = !(xc && yc && !dne) && !anyb;
    if (empty && !(op_neg(pop) && op_neg(opclass(xc, canyb)))) return false;
  }
  return true;
}

// Topology.Record for a pod placed on the claim of row R (after the commit of its requirements)
// (entries [first, record_cnt) in steps of `step`: the commit gives every entry of the pod's record list a lane of its own --
// the entries name different groups, and each is two or three dependent round trips to the group's counters)
KG_HD void row_record_topology(const DevProblem& P, const SimState& S, const Shared& sh, const PodCtx& pc, const Row& R, uint32_t first = 0, uint32_t step = 1) {
  const Variant& v = pc.vc.v;
  const uint32_t hslot = P.N + R.c;
  for (uint32_t li = first; li < v.record_cnt; li += step) {
    const ListEntry le = P.lists[v.record_off + li];
    const Group g = P.groups[le.g];
    if (!(le.flags & LF_INVERSE) && g.filter_cnt) {
      bool match = false;
      for (uint32_t f = 0; f < g.filter_cnt && !match; ++f) match = row_compatible_rs(P, sh, R, P.filters[g.filter_off + f]);
      if (!match) continue;
    }
    if (g.spec == 0xFF) { record_host(S, P, g, hslot); continue; }
    const SReq r = row_special(sh, R, g.spec);
    if (!(r.meta & RM_PRESENT)) continue;
    if ((le.flags & LF_INVERSE) || g.type == TG_ANTI) {
      for (uint64_t mm = r.word; mm; mm &= mm - 1) record_small(S, le.g, ctz64(mm));
    } else if (!(r.meta & RM_COMPL) && popc64(r.word) == 1) {
      record_small(S, le.g, ctz64(r.word));
    }
  }
}

// max allocatable of active resource r over the option mask of claim c, minus its requests: the prefilter's headroom.
// Computed when a claim is created and when the prefilter let an infeasible claim through.
KG_HD void claim_refresh_head(const DevProblem& P, const SimState& S, const Shared& sh, uint32_t c, int r) {
  int64_t mx = 0;
  const int64_t* al = P.alloc + (size_t)sh.a_res[r] * P.T;
  for (int w = 0; w < P.TW; ++w)
    for (uint64_t mm = S.c_mask[(size_t)w * P.Cmax + c]; mm; mm &= mm - 1) {
      const int64_t v = al[w * 64 + ctz64(mm)];
      if (v > mx) mx = v;
    }
  S.c_head[(size_t)r * P.Cmax + c] = mx - S.c_req[(size_t)r * P.Cmax + c];
}

// ------------------------------------------------------------------ ExistingNode.Add for node n (scan position)
// *tk_out: (sparse mode) the node's entry in the simulation's touched list as this evaluation found it, -1 when it has none
// ntouched: entries of that list (uniform)
KG_HD bool eval_node(const DevProblem& P, const SimState& S, const Shared& sh, const PodCtx& pc, uint32_t need, uint32_t n, uint32_t ntouched, uint32_t tl_cap, uint32_t* reason, int* tk_out) {
  const Variant& v = pc.vc.v;
  const uint32_t wi = n >> 6, bi = n & 63;
  const bool spilled = P.sparse_nodes && ntouched > tl_cap;  // some touched nodes live in global memory (tl_cap: KG_TL, or 0)
  // Every word the checks below may look at is requested before the first one is used: one round trip for the node instead
  // of one per check (each `return` below would otherwise put a wait and a branch in front of the next load).
  const uint64_t w_excl = S.excl[wi], w_tol = P.ntol[(size_t)v.tolset * P.NW + wi], w_lab = P.nlab[(size_t)v.rs * P.NW + wi];
  const uint64_t w_touch = spilled ? S.ntouch[wi] : 0ull;
  const uint32_t h0 = (n * 2654435761u) & S.thash_mask;
  const uint64_t e0 = spilled ? S.thash[h0] : 0ull;  // the node's home slot: nearly always where its entry is
  const uint64_t w_ports = !P.has_ports ? 0ull : (P.sparse_nodes ? P.nports0[n] : S.nports[n]);
  // what the node can still take of every resource the pod requests (`need` is uniform: scalar branches, no load more
  // than the pod needs)
  int64_t n_free[KG_MAXR];
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) {
    n_free[r] = 0;
    if ((need >> r) & 1) n_free[r] = P.sparse_nodes ? P.nfree0[(size_t)r * P.N + n] : S.nfree[(size_t)r * P.N + n];
  }
  // The verdicts of the static checks are combined before the first branch: a `return` after each would let the compiler
  // move the loads a later check needs behind it (a second round trip).  Upstream's order decides the reason reported.
  const bool f_excl = (w_excl >> bi) & 1;     // a candidate of this simulation: not a state node
  const bool f_tol = !((w_tol >> bi) & 1);
  const bool f_lab = !((w_lab >> bi) & 1);
  int tk = -1;  // sparse mode: the node's entry in the simulation's touched list
  int64_t have[KG_MAXR];
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) have[r] = 0;
  uint64_t ports_now = w_ports;
  if (P.sparse_nodes) {
    const uint32_t nl = ntouched < tl_cap ? ntouched : tl_cap;
    for (uint32_t i = 0; i < nl; ++i) tk = sh.tl_node[i] == n ? (int)i : tk;  // (every lane reads the same word)
    if (tk < 0 && ((w_touch >> bi) & 1)) {
      uint64_t e = e0;
      for (uint32_t h = h0;;) {
        if ((uint32_t)(e >> 32) == n + 1) { tk = (int)(uint32_t)e; break; }
        if (e == 0) break;  // (cannot happen while the bit is set)
        h = (h + 1) & S.thash_mask;
        e = S.thash[h];
      }
    }
    // what the simulation has added to the node so far (LDS for the first KG_TL touched nodes; the rest: one round trip)
    if (tk >= (int)tl_cap) {
#pragma unroll
      for (int r = 0; r < KG_MAXR; ++r) if ((need >> r) & 1) have[r] = S.tadd[(size_t)r * P.Cmax + tk];
      if (P.has_ports) ports_now = S.tports[tk];
    } else if (tk >= 0) {
#pragma unroll
      for (int r = 0; r < KG_MAXR; ++r) have[r] = sh.tl_add[tk][r];
      if (P.has_ports) ports_now = sh.tl_ports[tk];
    }
  }
  const bool f_ports = P.has_ports && (ports_now & pc.vc.hp_conf);  // HostPortUsage.Conflicts
  bool f_res = false;  // resources before labels (cheap reject), as upstream does
#pragma unroll
  for (int r = 0; r < KG_MAXR; ++r) f_res = f_res || (((need >> r) & 1) && have[r] + pc.preq[r] > n_free[r]);
  if (f_excl || f_tol || f_ports || f_res || f_lab) {
    *reason |= f_excl ? 0u : f_tol ? 1u : f_ports ? 2u : f_res ? 64u : 2u;
    return false;
  }
  for (uint32_t li = 0; li < pc.vc.nlist; ++li) {
    const ListEntry le = pc.vc.lst[li].le;
    const Group g = pc.vc.lst[li].g;
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
  *tk_out = tk;
  return true;
}

// ExistingNode.Add's state update for the node the scan chose (one thread).  Nothing is evaluated again: the state has not
// changed since the scan.  tk: the node's touched entry as its evaluation found it (-1: none yet, it becomes entry
// `ntouched`, the simulation's count so far, which the caller keeps).
KG_HD void commit_node(const DevProblem& P, const SimState& S, Shared& sh, const PodCtx& pc, uint32_t n, int tk, uint32_t ntouched, uint32_t tl_cap) {
  const Variant& v = pc.vc.v;
  const uint32_t wi = n >> 6, bi = n & 63;
  if (P.sparse_nodes) {
    if (tk < 0) {  // (the list grows by at most one entry per placed pod, Cmax in all)
      tk = (int)ntouched;
      const uint64_t w_ports = P.has_ports ? P.nports0[n] : 0ull;
      if (tk < (int)tl_cap) {
        sh.tl_node[tk] = n;
#pragma unroll
        for (int r = 0; r < KG_MAXR; ++r) sh.tl_add[tk][r] = r < P.R ? pc.preq[r] : 0;
        sh.tl_ports[tk] = w_ports | (P.has_ports ? pc.vc.hp_use : 0ull);
      } else {
        uint32_t h = (n * 2654435761u) & S.thash_mask;
        uint64_t e = S.thash[h];
        const uint64_t w_touch = S.ntouch[wi];
        while (e != 0) { h = (h + 1) & S.thash_mask; e = S.thash[h]; }
        S.thash[h] = ((uint64_t)(n + 1) << 32) | (uint32_t)tk;
        S.tnode[tk] = n;
        S.ntouch[wi] = w_touch | (1ull << bi);
        for (int r = 0; r < P.R; ++r) S.tadd[(size_t)r * P.Cmax + tk] = pc.preq[r];
        S.tports[tk] = w_ports | (P.has_ports ? pc.vc.hp_use : 0ull);
      }
    } else if (tk < (int)tl_cap) {
#pragma unroll
      for (int r = 0; r < KG_MAXR; ++r) if (r < P.R) sh.tl_add[tk][r] += pc.preq[r];
      if (P.has_ports) sh.tl_ports[tk] |= pc.vc.hp_use;
    } else {
      int64_t have[KG_MAXR];
#pragma unroll
      for (int r = 0; r < KG_MAXR; ++r) have[r] = S.tadd[(size_t)(r < P.R ? r : 0) * P.Cmax + tk];
#pragma unroll
      for (int r = 0; r < KG_MAXR; ++r) if (r < P.R && pc.preq[r] != 0) S.tadd[(size_t)r * P.Cmax + tk] = have[r] + pc.preq[r];
      if (P.has_ports) S.tports[tk] |= pc.vc.hp_use;
    }
  } else {
    for (int r = 0; r < P.R; ++r) if (pc.preq[r] != 0) S.nfree[(size_t)r * P.N + n] -= pc.preq[r];
    if (P.has_ports) S.nports[n] |= pc.vc.hp_use;
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

// row of a pod in the per-simulation result arrays
KG_HD uint32_t pod_row(const SimState& S, uint32_t pod) { return pod < S.n_shared ? pod : pod - S.pod_base + S.n_shared; }

// position of the n-th (0-based) set bit of b; b has more than n bits set
KG_HD uint32_t nth_set_bit(uint64_t b, uint32_t n) {
  uint32_t pos = 0;
#pragma unroll
  for (int s = 32; s; s >>= 1) {
    const uint32_t pc = (uint32_t)popc64(b & ((1ull << s) - 1));
    if (n >= pc) { n -= pc; b >>= s; pos += s; }
  }
  return pos;
}

// scan-order entry: pod count << 32 | claim id, in LDS while the claims fit, in global memory beyond
KG_HD unsigned long long ord_get(const SimState& S, const Shared& sh, const DevProblem& P, bool lds, uint32_t buf, uint32_t p) {
  unsigned long long v;
  if (lds) { v = sh.ord[buf][p]; KG_OPAQUE64(v); } else v = S.ordg[(size_t)buf * P.Cmax + p];
  return v;
}
KG_HD void ord_set(const SimState& S, Shared& sh, const DevProblem& P, bool lds, uint32_t buf, uint32_t p, unsigned long long v) {
  if (lds) sh.ord[buf][p] = v; else S.ordg[(size_t)buf * P.Cmax + p] = v;
}

// The contiguous image (k_pod_ctx) with at least one thread per word: the word is requested first and stored after other
// work (the existing-node scan puts every load of a placement in front of its first wait).
template <int NT>
KG_HD bool ctx_word_per_thread(const DevProblem& P) { return P.TW <= KG_TWLDS && (size_t)NT >= sizeof(PodCtx) / 8; }
KG_HD uint64_t ctx_word_load(const DevProblem& P, int tid, uint32_t pod, uint32_t variant) {
  // ONE load per thread whatever the word (a load per case would share its destination register with the others and the
  // compiler waits before overwriting a register a load is still due in); PodCtx::need is not filled: solve_block derives
  // it from the requests
  const uint32_t i = (uint32_t)tid < sizeof(PodCtx) / 8 ? (uint32_t)tid : (uint32_t)KG_MAXR;
  const uint64_t* src = ((const uint64_t*)(P.pctx + variant)) + i;
  if (i < (uint32_t)KG_MAXR) src = (const uint64_t*)(P.preq + (size_t)pod * P.R + ((int)i < P.R ? i : 0));
  return *src;  // (ctx_word_store zeroes the request words beyond the problem's resources: no use of the value here)
}
KG_HD void ctx_word_store(const DevProblem& P, PodCtx& nx, int tid, uint64_t v) {
  if (tid < KG_MAXR && tid >= P.R) v = 0;
  if ((uint32_t)tid < sizeof(PodCtx) / 8) ((uint64_t*)&nx)[tid] = v;
}

// one Queue.Pop's worth of pod context, global -> the buffer of the next pod (each thread moves its share; needs a barrier afterwards)
template <int NT>
KG_HD void ctx_fetch(const DevProblem& P, PodCtx& nx, int tid, uint32_t pod, uint32_t variant) {
  if (P.TW <= KG_TWLDS) {
    // the contiguous image k_pod_ctx built (struct PodCtx, requests left zero): one level of loads, none depends on another
    static_assert(offsetof(PodCtx, preq) == 0 && offsetof(PodCtx, need) == KG_MAXR * 8, "the words the pod (not its variant) decides come first");
    const uint64_t* img = (const uint64_t*)(P.pctx + variant);
    uint64_t* dst = (uint64_t*)&nx;
    for (uint32_t i = (uint32_t)tid; i < sizeof(PodCtx) / 8; i += NT) {
      uint64_t v;
      if (i < (uint32_t)KG_MAXR) {
        v = (int)i < P.R ? (uint64_t)P.preq[(size_t)pod * P.R + i] : 0;
      } else if (i == (uint32_t)KG_MAXR) {
        int64_t q[KG_MAXR];
#pragma unroll
        for (int r = 0; r < KG_MAXR; ++r) q[r] = P.preq[(size_t)pod * P.R + (r < P.R ? r : 0)];
        uint32_t nd = 0;
#pragma unroll
        for (int r = 0; r < KG_MAXR; ++r) if (r < P.R && q[r] != 0) nd |= 1u << r;
        v = nd;
      } else {
        v = img[i];
      }
      dst[i] = v;
    }
    return;
  }
  const uint64_t* src = (const uint64_t*)(P.vctx + variant);
  uint64_t* dst = (uint64_t*)&nx.vc;
  for (uint32_t i = tid; i < sizeof(VariantCtx) / 8; i += NT) dst[i] = src[i];
  if (tid >= NT - KG_MAXR) {
    const int r = tid - (NT - KG_MAXR);
    nx.preq[r] = r < P.R ? P.preq[(size_t)pod * P.R + r] : 0;
  }
  if (tid == NT - KG_MAXR - 2) {
    int64_t q[KG_MAXR];
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) q[r] = P.preq[(size_t)pod * P.R + (r < P.R ? r : 0)];
    uint32_t nd = 0;
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) if (r < P.R && q[r] != 0) nd |= 1u << r;
    nx.need = nd;
  }
  if (tid == NT - KG_MAXR - 1) nx.tolw = P.tol_tmpl[P.vctx[variant].v.tolset];
  if (tid >= NT / 2 && tid < NT / 2 + KG_TWLDS && P.TW <= KG_TWLDS)  // zero-padded: the prefilter ANDs all KG_TWLDS words
    nx.podmask[tid - NT / 2] = tid - NT / 2 < P.TW ? P.podmask[(size_t)variant * P.TW + (tid - NT / 2)] : 0;
  if (tid >= 96 && tid < 96 + KG_MAXE * KG_TWLDS && P.TW <= KG_TWLDS) {
    // e_amask rows of the pod's requirement entries (entry ids straight from the global record: same round trip)
    const uint32_t j = (uint32_t)(tid - 96) / KG_TWLDS, w = (uint32_t)(tid - 96) % KG_TWLDS;
    const VariantCtx* gv = P.vctx + variant;
    if (j < gv->ne && w < (uint32_t)P.TW) nx.eam[j][w] = P.e_amask[(size_t)gv->ent[j].e * P.TW + w];
  }
}


#if KG_DEVICE_ONLY
// ------------------------------------------------------------------ NodeClaim.Add by ONE wave, the claim row in registers
// The common placement (pod without topology groups, no minValues NodePool, tables staged in LDS) does not need the
// workgroup for the exact test: lane i of a wave holds word i of the claim's requirement row (lane a its meta word for
// key a, lane r its requests of resource r, lane w word w of its option mask), the pod's entries are merged one after the
// other with ballots for the per-key reductions, the fit thresholds come from a 64-wide search, and the type filter runs
// on lanes [0, TW).  Each wave evaluates a different survivor of the prefilter and no barrier separates load, merge,
// filter and verdict.  Same arithmetic as row_merge_entry_from / row_merge_resource / row_word_a / row_word_b.
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t lane) {
  return ((uint64_t)rl32((uint32_t)(v >> 32), lane) << 32) | rl32((uint32_t)v, lane);
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
  for (int s = 32; s; s >>= 1) {
    const int64_t o = __shfl_xor(v, s, 64);
    v = o > v ? o : v;
  }
  return v;
}

// first scan position whose claim holds more than k pods (= end of the k-pod group), by 64-ary search
__device__ __forceinline__ uint32_t wave_group_end(const SimState& S, const Shared& sh, const DevProblem& P, bool lds, uint32_t buf, uint32_t nclaims, uint32_t k,
                                                   uint32_t lane) {
  uint32_t stride = 1;
  while ((unsigned long long)stride * 64 < nclaims) stride *= 64;
  uint32_t lo = 0;
  while (true) {
    const unsigned long long q = (unsigned long long)lo + (unsigned long long)lane * stride;
    bool f = true;
    if (q < nclaims) f = (uint32_t)(ord_get(S, sh, P, lds, buf, (uint32_t)q) >> 32) > k;
    const unsigned long long b = __ballot(f);
    const uint32_t L = b ? (uint32_t)(__ffsll(b) - 1) : 64u;
    if (stride == 1) return lo + L;
    if (L == 0) return lo;
    lo = lo + (L - 1) * stride + 1;
    stride /= 64;
  }
}

// evaluates claim c (scan position pos, order entry oe) for the pod in sh.pc; on success the merged row goes to
// sh.rows[slot] (with the end of the claim's pod-count group in Row::pad_), the candidate option mask to S.scratch and
// bit `slot` of *feas is set
__device__ __forceinline__ void fast_eval_wave(const DevProblem& P, const SimState& S, Shared& sh, const PodCtx& pc, uint32_t lane, uint32_t slot, uint32_t c, uint32_t pos,
                                               unsigned long long oe, uint32_t need, uint32_t* feas, bool ord_lds, uint32_t ob, uint32_t nclaims,
                                               bool WT, unsigned long long* wt, unsigned long long& wt_last) {
#define KG_FT(k) do { if (WT) { const unsigned long long n_ = clock64(); if (wt_last) wt[(k) - 22] += n_ - wt_last; wt_last = n_; } } while (0)
  const uint32_t Ka = (uint32_t)P.Ka, Wa = (uint32_t)P.Wa, NR = (uint32_t)P.R, TW = (uint32_t)P.TW;
  const size_t Cmax = (size_t)P.Cmax;
  const uint32_t rg = lane >> 3;  // resources are handled by groups of 8 lanes: lanes [8r, 8r + 8) <-> active resource r
  // ---- one round trip: the claim's row
  uint32_t meta = 0;
  uint64_t bits = 0, cm = 0;
  int64_t req = 0;
  if (lane < Ka) meta = S.c_meta[(size_t)lane * Cmax + c];
  if (lane < Wa) bits = S.c_bits[(size_t)lane * Cmax + c];
  if (rg < NR) req = S.c_req[(size_t)rg * Cmax + c];
  if (lane < TW) cm = S.c_mask[(size_t)lane * Cmax + c];
  const uint32_t tm_v = S.c_tmpl[c];
  const uint32_t stamp_v = S.c_stamp[c];
  uint64_t ports_v = 0;
  if (P.has_ports) ports_v = S.c_ports[c];
  // ---- while the row is on its way: the small tables this evaluation indexes by a scalar (lane i holds entry i),
  // and the end of the claim's pod-count group (where it moves if it takes the pod)
  uint4 pe4;
  pe4.x = pe4.y = pe4.z = pe4.w = 0;
  if (lane < KG_MAXE) pe4 = ((const uint4*)pc.vc.ent)[lane];
  static_assert(sizeof(PodEntry) == 16, "PodEntry is read as one 16-byte word");
  uint32_t spa = 0, spoff = 0;
  if (lane < (uint32_t)P.nspec) { spa = sh.spec_slot[lane]; spoff = sh.aoff[spa]; }
  uint32_t tn = 0, tbase = 0, toff = 0;
  int64_t add = 0;
  if (rg < NR) { tn = sh.a_thr_n[rg]; tbase = sh.thr_base[rg]; toff = sh.a_thr_off[rg]; add = pc.preq[rg]; }
  const uint32_t ne = KG_UNI(pc.vc.ne);
  const uint64_t tolw = KG_UNI64(pc.tolw), hp_conf = P.has_ports ? KG_UNI64(pc.vc.hp_conf) : 0ull;
  const uint32_t hcok = KG_UNI(pc.vc.v.host_claim_ok);
  const uint32_t gend = wave_group_end(S, sh, P, ord_lds, ob, nclaims, (uint32_t)(oe >> 32), lane);
  KG_FT(42);
  const uint32_t tm = KG_UNI(tm_v), stamp = KG_UNI(stamp_v);
  bool ok = ((tolw >> tm) & 1) && hcok && !(KG_UNI64(ports_v) & hp_conf);
  const int64_t req0 = req;
  const uint64_t cm0 = cm;
  KG_FT(43);
  // ---- step 3: the pod's requirement entries, one after the other (distinct keys)
  uint32_t ent_changed = 0, ent_dne_ok = 0, sp_changed = 0, ent_ord = 0;
  for (uint32_t j = 0; j < ne && ok; ++j) {
    // PodEntry {e, meta, aoff:16 bits:16, slot:8 spec:8 w:8 opc:8}
    const uint32_t pm = rl32(pe4.y, j), ab = rl32(pe4.z, j), sw = rl32(pe4.w, j);
    const uint32_t aoff = ab & 0xFFFF, pbits = ab >> 16, a = sw & 0xFF, w = (sw >> 16) & 0xFF, opc = sw >> 24;
    const int spec = (int)(int8_t)((sw >> 8) & 0xFF);
    if (spec < 0) ent_ord |= 1u << j;
    const uint32_t cma = rl32(meta, a);
    const bool inr = lane >= aoff && lane < aoff + w;
    uint64_t y = 0;
    if (pbits != 0xFFFF) { if (inr) { y = pc.vc.ebits[pbits + lane - aoff]; KG_OPAQUE64(y); } }
    else if (inr) y = P.rs_bits[P.e_bits[rl32(pe4.x, j)] + lane - aoff];
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
    const bool dne = gt && lt && P.gtval[P.bnd_off[a] + gt - 1] >= P.ltval[P.bnd_off[a] + lt - 1];
    const uint32_t mm = (xc && yc && !dne) ? rm_make(true, gt, lt) : rm_make(false, 0, 0);
    const uint64_t x = bits;
    uint64_t r = x;
    if (inr) {
      r = dne ? 0 : ((xc && yc) ? (x | y) : (xc ? (y & ~x) : (yc ? (x & ~y) : (x & y))));
      if (!dne && gt) r &= P.gtmask[P.bndmask_off[a] + (size_t)(gt - 1) * w + (lane - aoff)];
      if (!dne && lt) r &= P.ltmask[P.bndmask_off[a] + (size_t)(lt - 1) * w + (lane - aoff)];
    }
    const bool anyb = __ballot(inr && r != 0) != 0, canyb = __ballot(inr && x != 0) != 0;
    const bool chg = mm != cma || __ballot(inr && r != x) != 0;
    if (inr) bits = r;
    if (lane == a) meta = mm;
    const int cop = opclass(xc, canyb);
    const bool empty = !(mm & RM_COMPL) && !anyb;
    if (empty && !(op_neg((int)opc) && op_neg(cop))) { ok = false; break; }
    if (op_neg(opclass(mm & RM_COMPL, anyb))) ent_dne_ok |= 1u << j;
    if (chg) ent_changed |= 1u << j;
    if (spec >= 0 && chg) sp_changed |= 1u << spec;
  }
  // ---- step 4: Topology.AddRequirements on the merged row (row_topology's arithmetic; every lane runs it on the same values,
  // lane i keeps the domains the groups leave for special key i), then the result merged into the row
  const uint32_t nlist = KG_UNI(pc.vc.nlist);
  if (ok && nlist != 0) {
    uint64_t topo_l = ~0ull;
    uint32_t has_topo = 0;
    const uint32_t hslot = (uint32_t)P.N + c;
    bool tok = true;
    for (uint32_t li = 0; li < nlist && tok; ++li) {
      const ListEntry le = pc.vc.lst[li].le;
      const Group g = pc.vc.lst[li].g;
      const bool self = le.flags & LF_SELF;
      const uint32_t gsp = KG_UNI((uint32_t)g.spec);
      if (gsp == 0xFF) { if (KG_UNI((uint32_t)topo_host(S, P, g, self, hslot)) == 0) tok = false; continue; }
      const uint64_t podDom = ((uint64_t)le.poddom_hi << 32) | le.poddom_lo;
      const uint32_t a = rl32(spa, gsp);
      SReq cur;
      cur.meta = rl32(meta, a);
      cur.word = (cur.meta & RM_PRESENT) ? rl64(bits, rl32(spoff, gsp)) : 0;
      const uint64_t nodeDom = has_mask1(P, (int)a, cur);
      const TopoOut o = topo_small(P, S, g, le.g, self, podDom, nodeDom);
      if (KG_UNI((uint32_t)o.ok) == 0) { tok = false; break; }
      if (lane == gsp) topo_l &= o.opts;
      has_topo |= 1u << gsp;
    }
    for (int i = 0; i < P.nspec && tok; ++i) {
      if (!((has_topo >> i) & 1)) continue;
      const uint32_t a = rl32(spa, (uint32_t)i), off = rl32(spoff, (uint32_t)i);
      SReq cur;
      cur.meta = rl32(meta, a);
      cur.word = (cur.meta & RM_PRESENT) ? rl64(bits, off) : 0;
      const SReq inc{rm_make(false, 0, 0), rl64(topo_l, (uint32_t)i)};
      SReq out;
      if (!(cur.meta & RM_PRESENT)) {
        // receiver lacks the key: allowed when well-known or the incoming operator is DoesNotExist
        if (!((P.well_known_slots >> a) & 1) && inc.word != 0) { tok = false; break; }
        out = inc;
        sp_changed |= 1u << i;
      } else {
        out = intersect1(P, (int)a, cur, inc);
        const bool empty = !(out.meta & RM_COMPL) && out.word == 0;
        const int eop = opclass(cur.meta & RM_COMPL, cur.word != 0);
        if (empty && !op_neg(eop)) { tok = false; break; }
        if (out.word != cur.word || out.meta != cur.meta) sp_changed |= 1u << i;
      }
      if (lane == off) bits = out.word;
      if (lane == a) meta = out.meta;
    }
    if (!tok) { ok = false; if (lane == 0) sh.pf_topo = 1; }
  }
  KG_FT(44);
  // ---- step 5: requests including the pod; the fit threshold of every resource at once, by 8-ary search in its lanes
  const int64_t nr = req + add;
  req = nr;
  const bool ruse = rg < NR && ((need >> rg) & 1) && nr > 0;
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
  const unsigned long long useb = __ballot(ruse && (lane & 7) == 0);   // bit 8r: resource r decides
  const bool thr_none = __ballot(ruse && tidx >= tn) != 0;
  const uint32_t trow = toff + (tidx < tn ? tidx : 0);
  KG_FT(45);
  // ---- step 6: the instance-type filter on lanes [0, TW)
  bool feasible = false;
  uint64_t m = cm;
  if (ok && !thr_none) {
    const uint32_t w = lane < TW ? lane : 0;
    // fits (independent loads first)
    uint64_t tmk[KG_MAXR];
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) {
      tmk[r] = ~0ull;
      if ((useb >> (8 * r)) & 1) tmk[r] = P.thr_mask[(size_t)rl32(trow, 8 * r) * TW + w];
    }
    // pod entries on ordinary keys
    for (uint32_t em = ent_changed & ent_ord; em; em &= em - 1) {
      const uint32_t j = (uint32_t)__builtin_ctz(em);
      uint64_t f = pc.eam[j][w];
      KG_OPAQUE64(f);
      if ((ent_dne_ok >> j) & 1) f |= P.kU[(size_t)(rl32(pe4.w, j) & 0xFF) * TW + w];
      m &= f;
    }
    // special keys and hasOffering: nothing to do unless the pod changed one of them
    if (sp_changed) {
      bool off_use = false;
      uint64_t zmask = ~0ull, ctmask = ~0ull;
      for (int i = 0; i < P.nspec; ++i) {
        const uint32_t a = rl32(spa, (uint32_t)i);
        SReq sp;
        sp.meta = rl32(meta, a);
        const bool present = sp.meta & RM_PRESENT;
        sp.word = present ? rl64(bits, rl32(spoff, (uint32_t)i)) : 0;
        const bool changed = (sp_changed >> i) & 1;
        const uint64_t hm = has_mask1(P, (int)a, sp);
        if (changed && present) {
          uint64_t f = P.kN[(size_t)a * TW + w];
          if (op_neg(opclass(sp.meta & RM_COMPL, sp.word != 0))) f |= P.kU[(size_t)a * TW + w];
          for (uint64_t q = hm & ((1ull << KG_SPV) - 1); q; q &= q - 1) { uint64_t g = sh.stw[i][ctz64(q)][w]; KG_OPAQUE64(g); f |= g; }
          m &= f;
        }
        if (i == P.slot_zone || i == P.slot_ct || i == P.slot_zid) off_use = off_use || changed;
        if (i == P.slot_zone) zmask &= hm;
        if (i == P.slot_ct) ctmask = hm;
        if (i == P.slot_zid) {
          uint64_t zm = 0;
          for (int z = 0; z < P.n_zone_vals; ++z) {
            const uint8_t zid = P.zone_zid[z];
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
    feasible = __ballot(lane < TW && m != 0) != 0;
  }
  KG_FT(46);
  if (feasible) {
    Row& R = sh.rows[slot];
    if (lane < KG_MAXKA) R.meta[lane] = meta;
    R.bits[lane] = bits;
    if ((lane & 7) == 0) R.req[rg] = req;
    if (lane < TW) S.scratch[(size_t)lane * Cmax + c] = m;
    if (lane == 63) {
      R.c = c; R.tmpl = tm; R.ok = 1; R.reason = 0; R.pos = pos; R.tv = 0; R.npods = (uint32_t)(oe >> 32); R.stamp = stamp; R.pad_ = gend;
      atomic_or_u32(feas, 1u << slot);
    }
  } else if (KG_UNI(S.c_hstamp[c]) != stamp) {
    // the prefilter let an infeasible claim through: recompute its headroom (its options changed since the last time)
    for (uint32_t r = 0; r < NR; ++r) {
      const int64_t* al = P.alloc + (size_t)KG_UNI(sh.a_res[r]) * P.T;
      int64_t mx = 0;
      for (uint32_t w = 0; w < TW; ++w) {
        const uint64_t mw = rl64(cm0, w);
        if ((mw >> lane) & 1) { const int64_t v = al[w * 64 + lane]; mx = v > mx ? v : mx; }
      }
      mx = wave_max_i64(mx);
      const int64_t rq = (int64_t)rl64((uint64_t)req0, 8 * r);
      if (lane == 0) S.c_head[(size_t)r * Cmax + c] = mx - rq;
    }
    if (lane == 0) S.c_hstamp[c] = stamp;
  }
  KG_FT(47);
#undef KG_FT
}
#endif

// ------------------------------------------------------------------ small problem tables -> LDS (start of every solve)
template <int NT>
KG_HD void stage_tables(const DevProblem& P, SharedBase& sh) {
  // ---- stage the small problem tables in LDS
  KG_FOR_THREADS(tid, NT) {
    if (tid < KG_MAXR) {
      sh.a_thr_off[tid] = P.atab[0 * KG_MAXR + tid]; sh.a_thr_n[tid] = P.atab[1 * KG_MAXR + tid]; sh.a_cthr_off[tid] = P.atab[2 * KG_MAXR + tid];
      sh.a_cthr_n[tid] = P.atab[3 * KG_MAXR + tid]; sh.a_res[tid] = P.atab[4 * KG_MAXR + tid];
    }
  }
  KG_SYNC();
  KG_FOR_THREADS(tid, NT) {
    if (tid == 0) {
      uint32_t tot = 0;
      for (int r = 0; r < P.R; ++r) { sh.thr_base[r] = tot; tot += sh.a_thr_n[r]; }
      sh.thr_lds = tot <= (uint32_t)KG_THR_LDS;
      uint32_t nmax = 1, st0 = 1;
      for (int r = 0; r < P.R; ++r) nmax = sh.a_thr_n[r] > nmax ? sh.a_thr_n[r] : nmax;
      while (st0 * 8 < nmax) st0 *= 8;
      sh.thr_stride0 = st0;
      sh.reason = 0; sh.best2[0] = sh.best2[1] = KG_NOIDX; sh.endk = 0; sh.feas[0] = sh.feas[1] = 0; sh.mvf[0] = sh.mvf[1] = 0;
      sh.ffeas[0] = sh.ffeas[1] = sh.ffeas[2] = sh.ffeas[3] = 0;
    }
    for (int a = tid; a < P.Ka; a += NT) { sh.aoff[a] = P.aoff[a]; sh.aw[a] = P.aw[a]; sh.slot2spec[a] = P.slot2spec[a]; }
    for (int i = tid; i < P.nspec; i += NT) sh.spec_slot[i] = P.spec_slot[i];
    if (tid == 0) {
      sh.tw_lds = P.TW <= KG_TWLDS;
      sh.off_lds = P.TW <= KG_TWLDS && P.n_zone_vals * P.n_ct_vals <= KG_OFFROWS;
      bool ok = P.TW <= KG_TWLDS;
      for (int i = 0; i < P.nspec; ++i) ok = ok && P.ktwv_n[P.spec_slot[i]] <= KG_SPV;
      sh.stw_lds = ok;
    }
  }
  KG_SYNC();
  KG_FOR_THREADS(tid, NT) {
    if (sh.off_lds)
      for (int i = tid; i < P.n_zone_vals * P.n_ct_vals * P.TW; i += NT) {
        const int row = i / P.TW, w = i % P.TW;
        sh.offm[row][w] = P.offmask[((size_t)(row / P.n_ct_vals) * 4 + (row % P.n_ct_vals)) * P.TW + w];
      }
    if (sh.stw_lds)
      for (int i = tid; i < P.nspec * KG_SPV * P.TW; i += NT) {
        const int sp = i / (KG_SPV * P.TW), d = (i / P.TW) % KG_SPV, w = i % P.TW;
        const int a = P.spec_slot[sp];
        sh.stw[sp][d][w] = d < (int)P.ktwv_n[a] ? P.twv[((size_t)P.ktwv_off[a] + d) * P.TW + w] : 0;
      }
  }
  KG_SYNC();
  if (KG_UNI(sh.thr_lds)) {
    KG_FOR_THREADS(tid, NT) {
      for (int r = 0; r < P.R; ++r)
        for (uint32_t i = tid; i < sh.a_thr_n[r]; i += NT) sh.thr_val[sh.thr_base[r] + i] = P.thr_val[sh.a_thr_off[r] + i];
    }
    KG_SYNC();
  }
}

// ------------------------------------------------------------------ one Solve by one workgroup
// NT = threads in the workgroup.  `sh` lives in LDS.
//
// Control state (queue ring, claim count, stage, winner) is UNIFORM: every thread keeps its own copy in registers and
// derives it from the same LDS / global words, so no phase exists only to let thread 0 decide and broadcast.  Variables
// declared outside the KG_FOR_THREADS bodies are such uniform values (the emulation keeps one copy of them).
//
// Barrier discipline: an LDS word that steers uniform control flow is only ever written in a phase that is followed by
// a barrier AND is at least one barrier after the last uniform read of it -- otherwise a fast wave overwrites it while
// a slow wave has not branched yet, and the waves' barrier counts diverge.  Words written once per pass alternate
// between two slots for that reason (feas, mvf, fresh_lim).
// DBG: with the phase timers and the per-wave cycle accounting (debug builds of the kernel only: their counters cost registers)
// TPF: with the topology groups / empty-requirement shortcut in the NodeClaim prefilter.  A batch of simulations without
// topology groups runs the instantiation without them: the extra code is never executed there, but its registers cost the
// existing-node path a third of its speed (k_simulate on C4: 0.50 -> 0.68 ms with it).
// NBD: one Solve over its own nfree (never a batch of simulations): the existing-node scan of a pod without topology groups
// looks at KG_NSCAN chunks of NT nodes per barrier (dense_scan below).
// Pk / Sk (device builds): the same two descriptors where they lie in memory (kernel-argument segment, state image), typed
// in the constant address space.  Every placement reads its fields through a view that is made opaque at the top of the
// loop, so the scalar loads stay inside the placement: about 130 pointers and sizes no longer live (parked in VGPR lanes,
// a v_readlane per use) across the whole Solve.
template <int NT, bool DBG, bool TPF = true, bool NBD = true>
KG_HD void solve_block(const DevProblem& P, const SimState& S, Shared& sh, uint32_t n_queue, const KG_CONST4 DevProblem* Pk = nullptr,
                       const KG_CONST4 SimState* Sk = nullptr) {
  const bool STATS = (P.stats & 1) != 0;
  const bool TIMERS = DBG && (P.stats & 2) != 0;  // per-phase wall-clock ticks (100 MHz) of thread 0 into counters[8..]
  unsigned long long t_mark = 0;
#define KG_T(i) do { if (TIMERS) { KG_FOR_THREADS(tid, NT) { if (tid == 0) { const unsigned long long n_ = now_ticks(); if (t_mark) S.counters[8 + (i)] += n_ - t_mark; t_mark = n_; } } } } while (0)
  // (inside a phase body: thread 0's own progress, counters[32 + k])
#define KG_ST(k) do { if (TIMERS && tid == 0) { const unsigned long long n_ = now_ticks(); if (t_mark) S.counters[32 + (k)] += n_ - t_mark; t_mark = n_; } } while (0)
  // per-wave cycle accounting (stats & 8): work / barrier wait of every phase kind, kept in registers, written at the end
#if KG_DEVICE_ONLY
  const bool WT = DBG && (P.stats & 8) != 0;
  unsigned long long wt_last = 0, wt[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) wt[i] = 0;
#define KG_W(k) do { if (WT) { const unsigned long long n_ = clock64(); if (wt_last) wt[k] += n_ - wt_last; wt_last = n_; } } while (0)
#else
#define KG_W(k) ((void)0)
#endif
#define KG_SYNCW(ph) do { KG_W(2 * (ph)); KG_SYNC(); KG_W(2 * (ph) + 1); } while (0)
  const uint32_t TWP = (uint32_t)P.TWP;      // threads per evaluation group (power of two >= max(TW, 16))
  const uint32_t GP0_ = (NT / TWP) < (uint32_t)KG_MAXGP ? (NT / TWP) : (uint32_t)KG_MAXGP;
  const uint32_t GP = GP0_ < P.gp_cap ? GP0_ : P.gp_cap;  // claims evaluated per pass (gp_cap: rows this launch has LDS for)
  static_assert(NT % 64 == 0 && NT / 64 <= 16 && NT <= KG_NTMAX, "one ballot word per wave");
  const bool STAMPS = (P.stats & 16) != 0;  // debug: the simulation leaves 100 MHz time stamps in counters[8..10]
  if (STAMPS) { KG_FOR_THREADS(tid, NT) { if (tid == 0) S.counters[8] = now_ticks(); } }
  stage_tables<NT>(P, sh);
  if (STAMPS) { KG_FOR_THREADS(tid, NT) { if (tid == 0) { S.counters[9] = now_ticks(); S.counters[11] = n_queue; } } }
  const uint32_t QCAP = S.qcap;
  // ---- uniform state of the Solve
  uint32_t qhead = 0, qlen = n_queue, epoch = 1;   // Queue ring; epoch/len implement lastLen
  uint32_t nclaims = 0, seq = 0;                   // in-flight NodeClaims, placement steps
  uint32_t ntouched = 0;                           // sparse mode: entries of the simulation's touched-node list
  // ... the first tl_cap of them in LDS.  A simulation with many pods (a multi-node prefix) touches hundreds of nodes: it
  // would pay the LDS search on every node and still go to the global list, so it keeps all of them there.
  const uint32_t tl_cap = n_queue <= 4u * KG_TL ? (uint32_t)KG_TL : 0u;
  uint32_t ob = 0;                                 // current buffer of the scan order
  bool ord_lds = true;                             // scan order still in LDS
  uint32_t last_c = KG_NOIDX, last_npods = 0, last_stamp = 0, last_new = 0;  // the one claim touched since the reference's last sort
  bool have_next = false, ctx_ready = false;       // next queue entry / its context already fetched
  uint4 qe_next;
  qe_next.x = qe_next.y = qe_next.z = qe_next.w = 0;
  bool q2_ready = false;                           // the entry two behind the head is in sh.qn (staged one placement early)
  uint32_t passno = 0;
  uint32_t cur = 0, placement = 0;                 // buffer of sh.pcb that holds the pod being placed; placements so far
  uint32_t fpass = 0, endk_fast = 0;                // wave-local passes so far; the winner's group end when such a pass found it
  bool fast_win = false;
  while (true) {
#if KG_DEVICE_ONLY && !defined(KG_NO_LOOP_VIEW)
    const KG_CONST4 DevProblem* pk_ = Pk;
    const KG_CONST4 SimState* sk_ = Sk;
    asm volatile("" : "+s"(pk_), "+s"(sk_));
    const DevProblem& P = *(const DevProblem*)pk_;  // (shadows the by-value copies inside the placement)
    const SimState& S = *(const SimState*)sk_;
#endif
    // ---- Queue.Pop
    if (qlen == 0) break;
    uint4 qe;
    if (have_next) qe = qe_next; else qe = ((const KG_GLOBAL uint4*)S.queue)[qhead];
    qe.x = KG_UNI(qe.x); qe.y = KG_UNI(qe.y); qe.z = KG_UNI(qe.z); qe.w = KG_UNI(qe.w);
    if (qe.z == epoch && qe.w == qlen) break;  // every remaining pod failed with the queue at this length: done
    const uint32_t pod = qe.x, variant = qe.y;
    qhead = qhead + 1 == QCAP ? 0 : qhead + 1;  // (a compare instead of a division: this is per placement)
    qlen -= 1;
    // ---- pod context: fetched during the previous placement into the other buffer, or now; the buffers swap roles
    if (!ctx_ready) {
      KG_FOR_THREADS(tid, NT) { ctx_fetch<NT>(P, sh.pcb[cur ^ 1], tid, pod, variant); }
      KG_SYNCW(7);
    }
    cur ^= 1;
    placement += 1;
    const PodCtx& pc = sh.pcb[cur];
    PodCtx& nx = sh.pcb[cur ^ 1];
    const uint32_t bslot = placement & 1;   // sh.best2 slot of this placement; the other one is reset meanwhile
    KG_FOR_THREADS(tid, NT) { if (tid == 0) { sh.reason = 0; sh.pf_topo = 0; } }  // (only thread 0 ever touches sh.reason)
    // the entry behind the head is settled (a push lands behind it) when at least one pod is left
    // ... and so is the one behind that when two are left: it is requested together with the other loads of this
    // placement and parked in LDS, so the next placement's context fetch does not wait for the queue first (and no load
    // stays in flight across the loop's back edge, where the compiler would wait for every outstanding one)
    if (q2_ready) { have_next = true; qe_next = sh.qn[bslot ^ 1]; }  // (staged during the previous placement)
    else {
      have_next = qlen >= 1;
      if (have_next) qe_next = ((const KG_GLOBAL uint4*)S.queue)[qhead];  // (made uniform when it is popped)
    }
    q2_ready = false;
    const bool want_q2 = qlen >= 2;                    // staged together with the context prefetch
    const uint32_t q2_pos = qhead + 1 == QCAP ? 0 : qhead + 1;
    bool want_pf = have_next;
    ctx_ready = false;
    uint32_t need = 0;  // bit per active resource the pod requests (uniform)
#pragma unroll
    for (int r = 0; r < KG_MAXR; ++r) need |= (pc.preq[r] != 0 ? 1u : 0u) << r;
    need = KG_UNI(need);
    KG_T(0);
    // ---- existing nodes, first feasible in scan order
    // One Solve, a pod without topology groups: every thread looks at KG_NSCAN nodes (one per chunk of NT) with all their
    // words requested before the first test -- one round trip and one barrier for KG_NSCAN chunks.  Late in a Solve no node
    // takes anything any more and every pod walks them all: the walk is then KG_NSCAN times shorter (C5: 2 000 nodes).
    const bool dense_scan = NBD && !P.sparse_nodes && KG_UNI(pc.vc.nlist) == 0;
    if (dense_scan) {
      const Variant& v = pc.vc.v;
      for (uint32_t base = 0; base < (uint32_t)P.N; base += NT * KG_NSCAN) {
        KG_W(16);
        const bool pf_word = want_pf && ctx_word_per_thread<NT>(P);
        KG_FOR_THREADS(tid, NT) {
          uint64_t cword = 0;
          uint4 q2;
          q2.x = q2.y = q2.z = q2.w = 0;
          if (pf_word) cword = ctx_word_load(P, tid, qe_next.x, qe_next.y);
          if (want_pf && want_q2 && tid == NT - 1) q2 = ((const KG_GLOBAL uint4*)S.queue)[q2_pos];
          uint64_t w_exc[KG_NSCAN], w_tol[KG_NSCAN], w_lab[KG_NSCAN], w_prt[KG_NSCAN];
          int64_t nf[KG_NSCAN][KG_MAXR];
#pragma unroll
          for (int j = 0; j < KG_NSCAN; ++j) {
            const uint32_t n = base + (uint32_t)j * NT + tid, nn = n < (uint32_t)P.N ? n : 0u, wi = nn >> 6;  // (every load unconditional)
            w_exc[j] = S.excl[wi];
            w_tol[j] = P.ntol[(size_t)v.tolset * P.NW + wi];
            w_lab[j] = P.nlab[(size_t)v.rs * P.NW + wi];
            w_prt[j] = P.has_ports ? S.nports[nn] : 0ull;
#pragma unroll
            for (int r = 0; r < KG_MAXR; ++r) {
              nf[j][r] = 0;
              if ((need >> r) & 1) nf[j][r] = S.nfree[(size_t)r * P.N + nn];
            }
          }
          uint32_t mine = KG_NOIDX;
#pragma unroll
          for (int j = KG_NSCAN - 1; j >= 0; --j) {
            const uint32_t n = base + (uint32_t)j * NT + tid, bi = n & 63;
            bool ok = n < (uint32_t)P.N && !((w_exc[j] >> bi) & 1) && ((w_tol[j] >> bi) & 1) && ((w_lab[j] >> bi) & 1) && !(w_prt[j] & pc.vc.hp_conf);
#pragma unroll
            for (int r = 0; r < KG_MAXR; ++r) ok = ok && (!((need >> r) & 1) || pc.preq[r] <= nf[j][r]);
            mine = ok ? n : mine;
          }
          if (mine != KG_NOIDX) { scan_ntk(sh)[tid] = -1; atomic_min_u32(&sh.best2[bslot], mine); }
          KG_W(17);
          if (tid == 0) sh.best2[bslot ^ 1] = KG_NOIDX;  // (last read one placement ago)
          if (pf_word) ctx_word_store(P, nx, tid, cword);
          else if (want_pf) ctx_fetch<NT>(P, nx, tid, qe_next.x, qe_next.y);
          if (want_pf && want_q2 && tid == NT - 1) sh.qn[bslot] = q2;
          KG_W(18);
        }
        if (want_pf) { want_pf = false; ctx_ready = true; q2_ready = want_q2; }
        KG_SYNCW(6);
        if (KG_UNI(sh.best2[bslot]) != KG_NOIDX) break;
      }
    } else
    for (uint32_t base = 0; base < (uint32_t)P.N; base += NT) {
      KG_W(16);
      const bool pf_word = want_pf && ctx_word_per_thread<NT>(P);
      KG_FOR_THREADS(tid, NT) {
        // every load of the placement is requested before the first wait: the next pod's context word, the queue entry two
        // ahead (the last thread, which has no context word), then the node's words inside eval_node
        uint64_t cword = 0;
        uint4 q2;
        q2.x = q2.y = q2.z = q2.w = 0;
        if (pf_word) cword = ctx_word_load(P, tid, qe_next.x, qe_next.y);
        if (want_pf && want_q2 && tid == NT - 1) q2 = ((const KG_GLOBAL uint4*)S.queue)[q2_pos];
        const uint32_t n = base + tid;
        if (n < (uint32_t)P.N) {
          uint32_t reason = 0;
          int tk = -1;
          if (eval_node(P, S, sh, pc, need, n, ntouched, tl_cap, &reason, &tk)) { scan_ntk(sh)[tid] = tk; atomic_min_u32(&sh.best2[bslot], n); }
        }
        KG_W(17);
        if (tid == 0) sh.best2[bslot ^ 1] = KG_NOIDX;  // (last read one placement ago)
        if (pf_word) ctx_word_store(P, nx, tid, cword);
        else if (want_pf) ctx_fetch<NT>(P, nx, tid, qe_next.x, qe_next.y);
        if (want_pf && want_q2 && tid == NT - 1) sh.qn[bslot] = q2;
        KG_W(18);
      }
      if (want_pf) { want_pf = false; ctx_ready = true; q2_ready = want_q2; }
      KG_SYNCW(6);
      if (KG_UNI(sh.best2[bslot]) != KG_NOIDX) break;
    }
    if (P.N > 0 && KG_UNI(sh.best2[bslot]) != KG_NOIDX) {
      const uint32_t bn_u = KG_UNI(sh.best2[bslot]);
      const int tk_u = (int)KG_UNI((uint32_t)scan_ntk(sh)[bn_u % NT]);  // (chunks start at multiples of NT: the thread that evaluated it)
      KG_FOR_THREADS(tid, NT) {
        if (tid == 0) {
          const uint32_t bn = bn_u;
          commit_node(P, S, sh, pc, bn, tk_u, ntouched, tl_cap);
          S.pod_kind[pod_row(S, pod)] = 1;
          S.pod_index[pod_row(S, pod)] = (int32_t)bn;
          if (STATS) S.counters[0] += bn + 1;
        }
      }
      KG_W(19);
      if (P.sparse_nodes && tk_u < 0) ntouched += 1;
      KG_SYNCW(6);
      KG_T(1);
      continue;
    }
    if (STATS) {
      KG_FOR_THREADS(tid, NT) { if (tid == 0) S.counters[0] += P.N; }
    }
    KG_T(1);
    // ---- in-flight NodeClaims (stage 0), then a new NodeClaim from each template in weight order (stage 1).
    // The claims are kept physically in the reference's scan order (pod count, then the stable order claim_key
    // describes), so stage 0 walks the order in chunks of NT positions: a conservative prefilter (thread per position,
    // ballot = survivors in order) -> exact NodeClaim.Add on the survivors, GP at a time, and the first claim that
    // accepts the pod ends the scan.  Stage 1 runs the same NodeClaim.Add on one freshly materialised claim per template.
    // Topology in the prefilter: a pod with topology groups passes almost every claim on resources and options alone and
    // then fails NodeClaim.Add on most of them (a hostname spread leaves one replica per claim; a zone spread rules out the
    // claims already pinned to a full zone).  What can be decided per claim from one or two words is decided there.
    // a pod whose own requirements rule out no instance type (no selector, no affinity terms: every replica of a plain
    // deployment) intersects every claim's options -- a claim always has some -- so the prefilter leaves the option words alone
    const bool pm_all = TPF && KG_UNI(pc.vc.v.pm_all) != 0;
    const bool topo_pf = TPF && !STATS && KG_UNI(pc.vc.nlist) != 0;
    if (topo_pf) {
      KG_FOR_THREADS(tid, NT) { if ((uint32_t)tid < pc.vc.nlist) sh.tallow[tid] = topo_allow(P, S, pc.vc.lst[tid].g, pc.vc.lst[tid].le); }
      KG_SYNCW(0);
    }
    last_c = KG_NOIDX; last_new = 0;
    uint32_t stage = 0, tnext = 0, base = 0, result = 0, win_g = 0;
    fast_win = false;
#if KG_DEVICE_ONLY
    // the wave-local NodeClaim.Add (topology groups included) applies when the tables are staged in LDS
    const bool fast_pod = !STATS && !(P.stats & 4) && P.tmpl_minv_mask == 0 && P.TW <= KG_TWLDS && P.Ka <= KG_MAXKA &&
                          P.Wa <= 64 && KG_UNI(sh.thr_lds) && KG_UNI(sh.stw_lds) && KG_UNI(sh.off_lds);
#endif
    while (stage != 2) {
      const bool fresh = stage == 1;
      uint32_t nsv = 0;
      if (!fresh) {
        if (base >= nclaims) { stage = (nclaims < (uint32_t)P.Cmax && P.NP > 0) ? 1 : 2; continue; }
        const uint32_t chunk = nclaims - base < (uint32_t)NT ? nclaims - base : (uint32_t)NT;
        KG_W(16);
        KG_FOR_THREADS(tid, NT) {
          bool ok = false;
          KG_ST(0);
          if ((uint32_t)tid < chunk) {
            ok = true;
            if (!STATS) {  // with the visit counters on, every claim is evaluated exactly
              const uint32_t c = (uint32_t)ord_get(S, sh, P, ord_lds, ob, base + tid);
              // every load below is independent, sits under scalar conditions only and is used after the last one has
              // been issued: one round trip to memory for the whole test, and no more loads than the problem needs (the
              // vector memory pipeline takes 16-34 cycles per 64-lane load: it, not latency, bounds this phase)
              const uint32_t tm = S.c_tmpl[c];
              int64_t hd[KG_MAXR];
#pragma unroll
              for (int r = 0; r < KG_MAXR; ++r) hd[r] = 0;
#pragma unroll
              for (int r = 0; r < KG_MAXR; ++r) if ((need >> r) & 1) hd[r] = S.c_head[(size_t)r * P.Cmax + c];
              uint64_t any = 0;
              if (pm_all) any = 1;
              else if (P.TW <= KG_TWLDS) {
                uint64_t mk[KG_TWLDS];
#pragma unroll
                for (int w = 0; w < KG_TWLDS; ++w) mk[w] = 0;
#pragma unroll
                for (int w = 0; w < KG_TWLDS; ++w) if (w < P.TW) mk[w] = S.c_mask[(size_t)w * P.Cmax + c];
#pragma unroll
                for (int w = 0; w < KG_TWLDS; ++w) any |= mk[w] & pc.podmask[w];
              } else {
                const uint64_t* pm = P.podmask + (size_t)variant * P.TW;
                for (int w = 0; w < P.TW; ++w) any |= S.c_mask[(size_t)w * P.Cmax + c] & pm[w];
              }
              bool tok = true;
              if (topo_pf) {
                for (uint32_t li = 0; li < pc.vc.nlist; ++li) {   // (uniform trip count and branches: the loads join the round trip above)
                  const Group g = pc.vc.lst[li].g;
                  if (g.spec == 0xFF) { tok = tok && topo_host(S, P, g, pc.vc.lst[li].le.flags & LF_SELF, (uint32_t)P.N + c); continue; }
                  const uint64_t al = sh.tallow[li];
                  const int a = sh.spec_slot[g.spec];
                  const uint32_t cm = S.c_meta[(size_t)a * P.Cmax + c];
                  const uint64_t cw = S.c_bits[(size_t)sh.aoff[a] * P.Cmax + c];
                  if ((cm & RM_PRESENT) && !(cm & RM_COMPL) && !(cw & al)) tok = false;
                }
              }
              ok = ((pc.tolw >> tm) & 1) && pc.vc.v.host_claim_ok && any != 0;
              if (P.has_ports) ok = ok && !(S.c_ports[c] & pc.vc.hp_conf);
#pragma unroll
              for (int r = 0; r < KG_MAXR; ++r) ok = ok && (!((need >> r) & 1) || hd[r] >= pc.preq[r]);
              if (ok && !tok) sh.pf_topo = 1;  // (same value from every thread)
              ok = ok && tok;
            }
          }
          KG_ST(1);
          KG_W(17);
          const uint32_t rank = wave_ballot_store(sh.ball, tid, ok);
          if (ok && rank < (uint32_t)KG_MAXGP) sh.wpos[tid >> 6][rank] = (uint16_t)tid;
          KG_W(18);
          if (want_pf) {
            ctx_fetch<NT>(P, nx, tid, qe_next.x, qe_next.y);
            if (want_q2 && tid == NT - 1) sh.qn[bslot] = ((const KG_GLOBAL uint4*)S.queue)[q2_pos];
          }
          KG_ST(2);
        }
        if (want_pf) { want_pf = false; ctx_ready = true; q2_ready = want_q2; }
        KG_SYNCW(0);
        { KG_FOR_THREADS(tid, NT) { KG_ST(3); } }
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) nsv += (uint32_t)popc64(KG_UNI64(sh.ball[i]));
        KG_T(2);
        if (TIMERS) { KG_FOR_THREADS(tid, NT) { if (tid == 0) { S.counters[20] += nsv; S.counters[21] += 1; } } }
      } else {
        // filterByRemainingResources + NewNodeClaim: materialise claim c = nclaims from template t into rows[0]
        const uint32_t c = nclaims;
        const int t = (int)tnext;
        const uint32_t fl = tnext & 1;
        KG_FOR_THREADS(tid, NT) {
          Row& R = sh.rows[0];
          if (tid == 0) {
            sh.fresh_lim[fl] = 0;
            R.c = c; R.tmpl = (uint32_t)t; R.reason = 0; R.pos = nclaims; R.tv = 0; R.ent_changed = 0; R.ent_dne_ok = 0; R.sp_changed = 0; R.thr_use = 0;
            R.npods = 0; R.stamp = 0;
            R.ok = ((pc.tolw >> t) & 1) && pc.vc.v.host_claim_ok;
            if (!R.ok) R.reason = ((pc.tolw >> t) & 1) ? 2u : 1u;
            sh.feas[passno & 1] = 0; sh.mvf[passno & 1] = 0;
          }
          for (int a = tid; a < P.Ka; a += NT) R.meta[a] = 0;
          for (int i = tid; i < P.Wa; i += NT) R.bits[i] = 0;
          for (int r = tid; r < KG_MAXR; r += NT) R.req[r] = r < P.R ? P.tmpl_req[(size_t)r * P.NP + t] : 0;
          // hostname placeholder registered for this attempt: hcount rows of slot N+c start at zero
          for (int gh = tid; gh < P.GH; gh += NT) S.hcount[(size_t)gh * P.H + P.N + c] = 0;
        }
        KG_SYNCW(4);
        KG_FOR_THREADS(tid, NT) {
          Row& R = sh.rows[0];
          for (int w = tid; w < P.TW; w += NT) {
            uint64_t m = ~0ull;
            if (w == P.TW - 1 && (P.T & 63)) m = (1ull << (P.T & 63)) - 1;
            if (P.tmpl_has_limits[t]) {
              for (int r = 0; r < P.R; ++r) {
                if (!P.tmpl_limit_res[(size_t)r * P.NP + t]) continue;
                const int idx = thr_lookup_le(P.cthr_val + sh.a_cthr_off[r], (int)sh.a_cthr_n[r], S.remaining[(size_t)r * P.NP + t]);
                if (idx < 0) m = 0; else m &= P.cthr_mask[((size_t)sh.a_cthr_off[r] + idx) * P.TW + w];
              }
            }
            S.c_mask[(size_t)w * P.Cmax + c] = m;
            if (m) sh.fresh_lim[fl] = 1;
          }
          // template requirement entries -> row
          const uint32_t rs = P.tmpl_rs[t];
          for (uint32_t e = P.rs_off[rs] + tid; e < P.rs_off[rs + 1]; e += NT) {
            const int a = P.e_slot[e];
            R.meta[a] = P.e_meta[e];
            for (int i = 0; i < sh.aw[a]; ++i) R.bits[sh.aoff[a] + i] = P.rs_bits[P.e_bits[e] + i];
          }
        }
        KG_SYNCW(4);
        nsv = KG_UNI(sh.fresh_lim[fl]) ? 1 : 0;
        KG_T(8);
      }
      // ---- exact NodeClaim.Add in scan order: up to GP claims per pass, TWP threads each
      uint32_t lo = 0;
      bool found = false;
      while (lo < nsv && !found) {
#if KG_DEVICE_ONLY
        if (fast_pod && !fresh) {
          // one survivor per wave, no barrier inside the evaluation
          const uint32_t nrow = nsv - lo < (uint32_t)(NT / 64) ? nsv - lo : (uint32_t)(NT / 64);
          const uint32_t fpar = fpass & 3;
          fpass += 1;
          {
            const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
            if (tid == 0) sh.ffeas[(fpar + 1) & 3] = 0;
            if (wave < nrow) {
              uint32_t rem = lo + wave, wv = 0;
              bool hit = false;
#pragma unroll
              for (int i = 0; i < NT / 64; ++i) {
                const uint32_t pc = (uint32_t)popc64(KG_UNI64(sh.ball[i]));
                if (!hit && rem < pc) { wv = (uint32_t)i; hit = true; }
                if (!hit) rem -= pc;
              }
              const uint32_t cpos = KG_UNI(rem < (uint32_t)KG_MAXGP ? (uint32_t)sh.wpos[wv][rem] : wv * 64 + nth_set_bit(KG_UNI64(sh.ball[wv]), rem));
              const uint32_t pos = base + cpos;
              const unsigned long long oe = KG_UNI64(ord_get(S, sh, P, ord_lds, ob, pos));
              fast_eval_wave(P, S, sh, pc, lane, wave, (uint32_t)oe, pos, oe, need, &sh.ffeas[fpar], ord_lds, ob, nclaims, WT, wt, wt_last);
            }
            KG_SYNCW(1);
          }
          const uint32_t fm = KG_UNI(sh.ffeas[fpar]);
          found = fm != 0;
          if (found) {
            win_g = (uint32_t)__builtin_ctz(fm);
            endk_fast = KG_UNI(sh.rows[win_g].pad_);
            fast_win = true;
          }
          KG_T(3);
          lo += nrow;
          continue;
        }
#endif
        // (the first survivor in order is the winner in most placements: a small first pass keeps the memory pipeline free)
        const uint32_t gcap = (lo == 0 && !STATS && GP > (uint32_t)KG_GP0) ? (uint32_t)KG_GP0 : GP;
        const uint32_t nrow = nsv - lo < gcap ? nsv - lo : gcap;
        const uint32_t par = passno & 1;
        passno += 1;
        if (!fresh) {
          // load the claims' rows: row gi <- the (lo + gi)-th survivor of the chunk
          KG_FOR_THREADS(tid, NT) {
            const uint32_t gi = (uint32_t)tid / TWP, w = (uint32_t)tid % TWP;
            if (tid == 0) { sh.feas[par] = 0; sh.mvf[par] = 0; }
            if (gi < nrow) {
              // the (lo + gi)-th survivor: which wave's ballot holds it, and its rank there
              uint32_t rem = lo + gi, wv = 0;
              bool hit = false;
#pragma unroll
              for (int i = 0; i < NT / 64; ++i) {
                const uint32_t pc = (uint32_t)popc64(KG_UNI64(sh.ball[i]));
                if (!hit && rem < pc) { wv = (uint32_t)i; hit = true; }
                if (!hit) rem -= pc;
              }
              const uint32_t cpos = rem < (uint32_t)KG_MAXGP ? (uint32_t)sh.wpos[wv][rem] : wv * 64 + nth_set_bit(sh.ball[wv], rem);
              const uint32_t pos = base + cpos;
              const unsigned long long oe = ord_get(S, sh, P, ord_lds, ob, pos);
              const uint32_t c = (uint32_t)oe;
              Row& R = sh.rows[gi];
              KG_ST(4);
              if (TWP == 16) {
                // straight-line, independent loads (Ka <= 32 metas, Wa <= 64 words, R <= 8 requests) from clamped
                // rows, selected afterwards: one round trip
                const uint32_t Ka1 = P.Ka > 0 ? (uint32_t)P.Ka - 1 : 0, Wa1 = P.Wa > 0 ? (uint32_t)P.Wa - 1 : 0, R1 = P.R > 0 ? (uint32_t)P.R - 1 : 0;
                const uint32_t m0 = S.c_meta[(size_t)(w < Ka1 ? w : Ka1) * P.Cmax + c];
                uint32_t m1 = 0;
                if (P.Ka > 16) m1 = S.c_meta[(size_t)(w + 16 < Ka1 ? w + 16 : Ka1) * P.Cmax + c];
                uint64_t b[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; ++k) if (16 * k < P.Wa) b[k] = S.c_bits[(size_t)(w + 16 * k < Wa1 ? w + 16 * k : Wa1) * P.Cmax + c];
                const int64_t rq = S.c_req[(size_t)(w < R1 ? w : R1) * P.Cmax + c];
                R.meta[w] = w < (uint32_t)P.Ka ? m0 : 0; R.meta[w + 16] = w + 16 < (uint32_t)P.Ka ? m1 : 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) R.bits[w + 16 * k] = w + 16 * k < (uint32_t)P.Wa ? b[k] : 0;
                if (w < KG_MAXR) R.req[w] = w < (uint32_t)P.R ? rq : 0;
              } else {
                for (int a = (int)w; a < P.Ka; a += (int)TWP) R.meta[a] = S.c_meta[(size_t)a * P.Cmax + c];
                for (int i = (int)w; i < P.Wa; i += (int)TWP) R.bits[i] = S.c_bits[(size_t)i * P.Cmax + c];
                if (w < KG_MAXR) R.req[w] = w < (uint32_t)P.R ? S.c_req[(size_t)w * P.Cmax + c] : 0;
              }
              if (w == TWP - 1) {
                const uint32_t tm = S.c_tmpl[c];
                R.npods = (uint32_t)(oe >> 32); R.stamp = S.c_stamp[c];
                R.c = c; R.tmpl = tm; R.reason = 0; R.pos = pos; R.tv = 0; R.ent_changed = 0; R.ent_dne_ok = 0; R.sp_changed = 0; R.thr_use = 0;
                R.ok = ((pc.tolw >> tm) & 1) && pc.vc.v.host_claim_ok;
                if (P.has_ports && (S.c_ports[c] & pc.vc.hp_conf)) { R.ok = 0; R.reason = 2; }
              }
            }
            KG_ST(5);
          }
          KG_SYNCW(5);
          KG_T(3);
        }
        // merge the pod: one thread per requirement entry (first half of the workgroup), one per resource (second
        // half) -- the two kinds of work run in different waves instead of serialising inside one
        KG_FOR_THREADS(tid, NT) {
          if (tid == NT - 1) sh.endk = 0;
          const uint32_t half = (uint32_t)tid / (NT / 2), t2 = (uint32_t)tid % (NT / 2);
          const uint32_t gi = t2 / KG_MAXE, j = t2 % KG_MAXE;
          if (gi < nrow && sh.rows[gi].ok) {
            if (half == 0) { if (j < pc.vc.ne) row_merge_entry(P, sh, pc, sh.rows[gi], j, fresh); }
            else if (j < (uint32_t)P.R) row_merge_resource(P, sh, pc, sh.rows[gi], (int)j, fresh);
          }
          KG_ST(6);
        }
        KG_SYNCW(5);
        if (KG_UNI(pc.vc.nlist)) {
          KG_FOR_THREADS(tid, NT) {
            const uint32_t gi = (uint32_t)tid / TWP, w = (uint32_t)tid % TWP;
            if (gi < nrow && w == 0 && sh.rows[gi].ok) row_topology(P, S, sh, pc, sh.rows[gi]);
          }
          KG_SYNCW(5);
        }
        KG_T(4);
        // instance-type filter: one thread per 64-type word; when the pass is small enough, two threads (in
        // different waves) per word, one for each half of the filter
        const bool split = nrow * TWP * 2 <= (uint32_t)NT && P.TW <= KG_TWLDS;
        KG_FOR_THREADS(tid, NT) {
          const uint32_t half = split ? (uint32_t)tid / (NT / 2) : 0, t2 = split ? (uint32_t)tid % (NT / 2) : (uint32_t)tid;
          const uint32_t gi = t2 / TWP, w = t2 % TWP;
          if (gi < nrow && w < (uint32_t)P.TW && sh.rows[gi].ok) {
            Row& R = sh.rows[gi];
            if (STATS && half == 0) atomic_add_u32(&R.tv, (uint32_t)popc64(S.c_mask[(size_t)w * P.Cmax + R.c]));
            if (split) {
              R.mw[half][w] = half == 0 ? row_word_a(P, S, sh, pc, R, fresh, (int)w) : row_word_b(P, S, sh, pc, R, fresh, (int)w);
            } else {
              const uint64_t m = row_word_a(P, S, sh, pc, R, fresh, (int)w) & row_word_b(P, S, sh, pc, R, fresh, (int)w);
              S.scratch[(size_t)w * P.Cmax + R.c] = m;
              if (m) atomic_or_u32(&sh.feas[par], 1u << gi);
            }
          }
          KG_ST(7);
        }
        KG_SYNCW(5);
        if (split) {
          KG_FOR_THREADS(tid, NT) {
            const uint32_t gi = (uint32_t)tid / TWP, w = (uint32_t)tid % TWP;
            if (gi < nrow && w < (uint32_t)P.TW && sh.rows[gi].ok) {
              const Row& R = sh.rows[gi];
              const uint64_t m = R.mw[0][w] & R.mw[1][w];
              S.scratch[(size_t)w * P.Cmax + R.c] = m;
              if (m) atomic_or_u32(&sh.feas[par], 1u << gi);
            }
          }
          KG_SYNCW(5);
        }
        if (P.tmpl_minv_mask != 0) {
          // InstanceTypes.SatisfiesMinValues on the candidate option masks
          KG_FOR_THREADS(tid, NT) {
            const uint32_t gi = (uint32_t)tid / TWP, w = (uint32_t)tid % TWP;
            if (gi < nrow && w == 0 && ((sh.feas[par] >> gi) & 1)) {
              const Row& R = sh.rows[gi];
              if (((P.tmpl_minv_mask >> R.tmpl) & 1) && !claim_minvalues_ok(P, S, R.tmpl, R.c)) atomic_or_u32(&sh.mvf[par], 1u << gi);
            }
          }
          KG_SYNCW(5);
        }
        KG_T(5);
        // ---- the first row in order whose option mask stays non-empty takes the pod (every thread reads the same words)
        const uint32_t fm = KG_UNI(sh.feas[par] & ~sh.mvf[par]);
        found = fm != 0;
        win_g = found ? (uint32_t)ctz64((uint64_t)fm) : 0;
        const uint32_t nlose = found ? win_g : nrow;  // rows evaluated and refused before the winner
        KG_FOR_THREADS(tid, NT) {
          const uint32_t gi = (uint32_t)tid / TWP, w = (uint32_t)tid % TWP;
          if (!fresh && !STATS && gi < nlose) {
            // the prefilter let an infeasible claim through: recompute its headroom if its options changed since
            const uint32_t c = sh.rows[gi].c;
            const uint32_t st = S.c_stamp[c];
            if (S.c_hstamp[c] != st) {
              if (w < (uint32_t)P.R) claim_refresh_head(P, S, sh, c, (int)w);
              if (w == TWP - 1) S.c_hstamp[c] = st;  // (the group's last thread: after every thread of the group has compared)
            }
          }
          if (found) {
            // end of the pod-count group the claim belongs to after this pod's arrival is decided by where the group it
            // leaves ends: positions are sorted by pod count
            const uint32_t k = fresh ? 1u : KG_UNI(sh.rows[win_g].npods);
            for (uint32_t p = tid; p < nclaims; p += NT) {
              const uint32_t np = (uint32_t)(ord_get(S, sh, P, ord_lds, ob, p) >> 32);
              const uint32_t nn = p + 1 < nclaims ? (uint32_t)(ord_get(S, sh, P, ord_lds, ob, p + 1) >> 32) : 0xFFFFFFFFu;
              if (np <= k && nn > k) sh.endk = p + 1;
            }
          }
          if (tid == 0) {
            if (fresh && !found) {
              const Row& R = sh.rows[0];
              uint32_t rr = R.reason;
              if (R.ok) rr |= ((sh.mvf[par] & 1) ? 32u : 0u) | 8u;
              sh.reason |= rr;
            }
            if (STATS) {
              // rows the reference visits: every claim up to and including the first feasible one; a visited claim's
              // options count when the pod got as far as the instance-type filter
              const uint32_t nvis = found ? win_g + 1 : nrow;
              for (uint32_t g2 = 0; g2 < nvis; ++g2) {
                const Row& R = sh.rows[g2];
                if (!fresh) S.counters[1] += 1; else S.counters[3] += 1;
                if (R.ok) S.counters[2] += R.tv;
              }
            }
            if (TIMERS) { S.counters[22] += fresh ? 0 : nlose; S.counters[23] += 1; }
          }
          KG_ST(8);
        }
        KG_SYNCW(5);
        KG_T(6);
        lo += nrow;
      }
      // ---- verdict of this chunk / attempt
      if (!fresh) {
        if (found) { result = 1; stage = 2; }
        else base += NT;
      } else {
        if (found) { result = 2; stage = 2; }
        else {
          if (nsv == 0) { KG_FOR_THREADS(tid, NT) { if (tid == 0) sh.reason |= 16u; } }  // all available instance types exceed limits
          tnext += 1;
          if (tnext >= (uint32_t)P.NP) stage = 2;
        }
      }
    }
    if (result != 0) {
      // ---- commit NodeClaim.Add: the merged row goes back to the claim, the option mask comes from scratch, the claim
      // moves to the front of the next pod-count group of the scan order (a new claim: to the end of the one-pod group)
      KG_W(26);
      const bool fr = result == 2;
      const uint32_t nnew = nclaims + (fr ? 1u : 0u);
      const bool dst_lds = ord_lds && nnew <= (uint32_t)KG_ORD;
      const uint32_t wc = KG_UNI(sh.rows[win_g].c), wpos = KG_UNI(sh.rows[win_g].pos), wnp = KG_UNI(sh.rows[win_g].npods), wst = KG_UNI(sh.rows[win_g].stamp);
      const uint32_t endk = fast_win ? endk_fast : KG_UNI(sh.endk);
      KG_FOR_THREADS(tid, NT) {
        const Row& R = sh.rows[win_g];
        const uint32_t c = wc;
        // (each small task starts in a different wave: no wave runs all of them one after the other)
        for (int a = (tid + NT - 64) % NT; a < P.Ka; a += NT) S.c_meta[(size_t)a * P.Cmax + c] = R.meta[a];
        for (int i = (tid + NT - 128) % NT; i < P.Wa; i += NT) S.c_bits[(size_t)i * P.Cmax + c] = R.bits[i];
        for (int w = (tid + NT - 192) % NT; w < P.TW; w += NT) S.c_mask[(size_t)w * P.Cmax + c] = S.scratch[(size_t)w * P.Cmax + c];
        if (tid >= NT - KG_MAXR) {
          const int r = tid - (NT - KG_MAXR);
          if (r < P.R) {
            S.c_req[(size_t)r * P.Cmax + c] = R.req[r];
            if (!fr && pc.preq[r] != 0) S.c_head[(size_t)r * P.Cmax + c] -= pc.preq[r];
          }
        }
        if (tid == NT / 2) {
          if (fr) { S.c_npods[c] = 1; S.c_tmpl[c] = (uint16_t)R.tmpl; S.c_hstamp[c] = seq + 1; }
          else S.c_npods[c] = wnp + 1;
          if (P.has_ports) S.c_ports[c] = (fr ? 0ull : S.c_ports[c]) | pc.vc.hp_use;
          S.c_stamp[c] = seq + 1;
          S.pod_kind[pod_row(S, pod)] = 2;
          S.pod_index[pod_row(S, pod)] = (int32_t)c;
        }
        if (tid >= NT / 2 && tid < NT / 2 + 64) row_record_topology(P, S, sh, pc, R, (uint32_t)(tid - NT / 2), 64u);
        // scan order -> the other buffer
        for (uint32_t p = tid; p < nclaims; p += NT) {
          unsigned long long v = ord_get(S, sh, P, ord_lds, ob, p);
          uint32_t q = p;
          if (fr) q = p < endk ? p : p + 1;
          else if (p == wpos) { q = endk - 1; v = ((unsigned long long)(wnp + 1) << 32) | c; }
          else if (p > wpos && p < endk) q = p - 1;
          ord_set(S, sh, P, dst_lds, ob ^ 1, q, v);
        }
        if (fr && tid == NT - 1) ord_set(S, sh, P, dst_lds, ob ^ 1, endk, (1ull << 32) | c);
        KG_ST(9);
      }
      KG_SYNCW(2);
      if (fr) {
        KG_FOR_THREADS(tid, NT) {
          // headroom of the new claim; subtractMax: charge the worst-case capacity of the remaining options against the limits
          const uint32_t c = wc;
          const int t = (int)KG_UNI(sh.rows[win_g].tmpl);
          if (tid < P.R) claim_refresh_head(P, S, sh, c, tid);
          if (tid >= 64 && tid < 64 + P.R && P.tmpl_has_limits[t]) {
            const int r = tid - 64;
            if (P.tmpl_limit_res[(size_t)r * P.NP + t]) {
              int64_t mx = 0;
              const int64_t* cp = P.cap + (size_t)sh.a_res[r] * P.T;
              for (int w = 0; w < P.TW; ++w)
                for (uint64_t mm = S.c_mask[(size_t)w * P.Cmax + c]; mm; mm &= mm - 1) {
                  const int64_t cv = cp[w * 64 + ctz64(mm)];
                  if (cv > mx) mx = cv;
                }
              S.remaining[(size_t)r * P.NP + t] -= mx;
            }
          }
        }
        KG_SYNCW(3);
      }
      last_c = wc; last_new = fr ? 1u : 0u;
      if (!fr) { last_npods = wnp; last_stamp = wst; }
      ob ^= 1; ord_lds = dst_lds; nclaims = nnew; seq += 1;
      KG_T(7);
      continue;
    }
    // ---- Preferences.Relax + Queue.Push
    const int32_t vnext = (int32_t)KG_UNI((uint32_t)pc.vc.v.next);
    KG_FOR_THREADS(tid, NT) {
      if (tid == 0) {
        S.pod_kind[pod_row(S, pod)] = 0;
        S.pod_index[pod_row(S, pod)] = -1;
        S.pod_reason[pod_row(S, pod)] = (sh.reason | (sh.pf_topo ? 4u : 0u)) ? (sh.reason | (sh.pf_topo ? 4u : 0u)) : 8;
        uint4 qn;
        qn.x = pod;
        if (vnext >= 0) {
          qn.y = (uint32_t)vnext;
          qn.z = 0; qn.w = 0;
          S.header[1] += 1;
        } else {
          qn.y = variant;
          qn.z = epoch; qn.w = qlen + 1;  // lastLen[pod] = len(queue) after the push
        }
        ((KG_GLOBAL uint4*)S.queue)[qhead + qlen >= QCAP ? qhead + qlen - QCAP : qhead + qlen] = qn;
      }
    }
    qlen += 1;
    if (vnext >= 0) epoch += 1;  // lastLen = {} on a relaxed push
    // the pushed entry is the head when nothing else is queued: it was not there when the next entry was fetched
    if (qlen == 1) { have_next = false; q2_ready = false; ctx_ready = false; }
    KG_SYNCW(7);
    KG_T(9);
  }
  KG_FOR_THREADS(tid, NT) {
    if (tid == 0) {
      if (STAMPS) S.counters[10] = now_ticks();
      S.header[0] = nclaims;
      S.header[2] = last_c; S.header[3] = last_npods; S.header[4] = last_stamp; S.header[5] = last_new;
    }
  }
  KG_SYNC();
#if KG_DEVICE_ONLY
  if (WT && (threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) S.counters[64 + (threadIdx.x >> 6) * 32 + i] = wt[i];
  }
#endif
#undef KG_T
#undef KG_ST
#undef KG_W
#undef KG_SYNCW
}

}  // namespace kg

// ------------------------------------------------------------------ batches of simulations (consolidation)
namespace kg {

// What the host needs back from one SimulateScheduling to run computeConsolidation's decision logic.
struct SimResult {
  uint32_t nclaims;          // new NodeClaims
  uint32_t errors;           // non-pending pods left unschedulable or placed on a node that is not initialised
  uint32_t tmpl;             // template (nodepool, weight order) of claim 0
  uint32_t n_pods;
  uint32_t zone_meta, ct_meta, zid_meta, pad;  // pad: claim 0 holds a pod that is being rescheduled (not a pending pod)
  uint64_t zone_word, ct_word, zid_word;
  uint64_t mask[KG_TWLDS];   // option mask of claim 0 (<= 1024 types)
  uint64_t visited[4];       // rows the reference visits in this simulation (visit counters on): existing nodes, NodeClaims, their options, fresh-claim attempts
};

struct SimBatch {
  uint8_t* arena;            // (nsims + 1) slots of `stride` bytes; slot 0 holds the pristine state image
  unsigned long long stride;      // bytes per slot
  unsigned long long init_bytes;  // ... of which the pristine image initialises this prefix (the rest is written before it is read)
  unsigned long long o_queue, o_excl, o_remaining;
  const uint32_t* unit_of_sim; // [nsims] the unit (row of the per-unit tables below) simulation i runs
  const uint32_t* q_off;     // [units + 1] -> q_ent
  const uint4* q_ent;        // initial queue entries {pod, variant, 0, 0} of every simulation, each in Queue order
  const uint32_t* x_off;     // [nsims + 1] -> x_node
  const uint32_t* x_node;    // scan positions of the candidate nodes of every simulation
  const int64_t* rem_adj;    // [nsims][R * NP] limits given back by the candidates (null when no NodePool has limits)
  const uint32_t* t_off;     // [nsims + 1] -> t_ent (null when the problem has no topology groups)
  const uint32_t* t_ent;     // topology counts of the simulation's own bound pods: gcount index, or 0x80000000 | hcount index
  const uint64_t* gdom_univ; // [G] domains registered without any pod
  unsigned long long o_gcount, o_gdom, o_gpos, o_hcount, o_hpos;
  uint32_t* nqueue;          // [nsims]
  SimResult* results;        // [nsims]
  const uint8_t* pod_nonpending;  // [P] 1 = bound pod being rescheduled (must schedule), 0 = pending pod
  const uint64_t* node_uninit;    // [NW] nodes that are not initialised
  int nsims;
};

// slot i+1 <- pristine image, then the simulation's queue, candidate exclusion mask and limits
KG_HD void sim_init_copy(const SimBatch& B, uint32_t i, uint32_t tid, uint32_t nthreads) {
  const uint64_t* src = (const uint64_t*)B.arena;
  uint64_t* dst = (uint64_t*)(B.arena + (size_t)(i + 1) * B.stride);
  for (size_t k = tid; k < B.init_bytes / 8; k += nthreads) dst[k] = src[k];
}
KG_HD void sim_init_patch(const DevProblem& P, const SimBatch& B, uint32_t i, uint32_t tid, uint32_t nthreads) {
  uint8_t* slot = B.arena + (size_t)(i + 1) * B.stride;
  uint4* q = (uint4*)(slot + B.o_queue);
  const uint32_t u = B.unit_of_sim[i];
  const uint32_t q0 = B.q_off[u], qn = B.q_off[u + 1] - q0;
  for (uint32_t k = tid; k < qn; k += nthreads) q[k] = B.q_ent[q0 + k];
  if (tid == 0) {
    B.nqueue[i] = qn;
    uint64_t* excl = (uint64_t*)(slot + B.o_excl);
    for (uint32_t k = B.x_off[u]; k < B.x_off[u + 1]; ++k) excl[B.x_node[k] >> 6] |= 1ull << (B.x_node[k] & 63);
    if (B.rem_adj) {
      int64_t* rem = (int64_t*)(slot + B.o_remaining);
      for (int k = 0; k < P.R * P.NP; ++k) if (rem[k] != 0x7FFFFFFFFFFFFFFFll) rem[k] += B.rem_adj[(size_t)u * P.R * P.NP + k];
    }
    if (B.t_off) {
      // Topology.excludedPods: the pods this simulation reschedules do not count in the initial domains
      int32_t* gc = (int32_t*)(slot + B.o_gcount);
      uint64_t* gdom = (uint64_t*)(slot + B.o_gdom);
      uint64_t* gpos = (uint64_t*)(slot + B.o_gpos);
      int32_t* hc = (int32_t*)(slot + B.o_hcount);
      int32_t* hpos = (int32_t*)(slot + B.o_hpos);
      for (uint32_t k = B.t_off[u]; k < B.t_off[u + 1]; ++k) {
        const uint32_t e = B.t_ent[k];
        if (e >> 31) {
          const uint32_t idx = e & 0x7FFFFFFFu;
          if (--hc[idx] == 0) hpos[idx / (uint32_t)P.H] -= 1;
        } else if (--gc[e] == 0) {
          const uint32_t g = e >> 6, d = e & 63;
          gpos[g] &= ~(1ull << d);
          if (!((B.gdom_univ[g] >> d) & 1)) gdom[g] &= ~(1ull << d);
        }
      }
    }
  }
}

// compact result of simulation i (one thread)
// the pods of simulation i, every `step`-th from `first` on (one thread: 0, 1; one wave: lane, 64): non-pending pods that
// did not schedule (or landed on a node that is not initialised), and whether one of them sits on NodeClaim 0
KG_HD void sim_gather_scan(const SimBatch& B, const SimState& S, uint32_t i, uint32_t first, uint32_t step, uint32_t* errors, uint32_t* claim0_np) {
  const uint32_t u = B.unit_of_sim[i];
  for (uint32_t k = B.q_off[u] + first; k < B.q_off[u + 1]; k += step) {
    const uint32_t pod = B.q_ent[k].x;
    if (!B.pod_nonpending[pod]) continue;
    const uint32_t row = pod_row(S, pod);
    const uint8_t kind = S.pod_kind[row];
    if (kind == 2 && S.pod_index[row] == 0) *claim0_np = 1;
    if (kind == 0) *errors += 1;
    else if (kind == 1) {
      const uint32_t n = (uint32_t)S.pod_index[row];
      if ((B.node_uninit[n >> 6] >> (n & 63)) & 1) *errors += 1;
    }
  }
}
KG_HD void sim_gather_finish(const DevProblem& P, const SimBatch& B, const SimState& S, uint32_t i, uint32_t errors, uint32_t claim0_np) {
  SimResult r;
  r.nclaims = S.header[0];
  r.errors = errors;
  r.tmpl = 0; r.n_pods = 0; r.zone_meta = r.ct_meta = r.zid_meta = 0; r.pad = claim0_np; r.zone_word = r.ct_word = r.zid_word = 0;
  for (int w = 0; w < KG_TWLDS; ++w) r.mask[w] = 0;
  if (r.nclaims == 1) {
    r.tmpl = S.c_tmpl[0];
    r.n_pods = S.c_npods[0];
    for (int w = 0; w < P.TW && w < KG_TWLDS; ++w) r.mask[w] = S.c_mask[(size_t)w * P.Cmax];
    if (P.slot_zone >= 0) { const int a = P.spec_slot[P.slot_zone]; r.zone_meta = S.c_meta[(size_t)a * P.Cmax]; r.zone_word = S.c_bits[(size_t)P.aoff[a] * P.Cmax]; }
    if (P.slot_ct >= 0) { const int a = P.spec_slot[P.slot_ct]; r.ct_meta = S.c_meta[(size_t)a * P.Cmax]; r.ct_word = S.c_bits[(size_t)P.aoff[a] * P.Cmax]; }
    if (P.slot_zid >= 0) { const int a = P.spec_slot[P.slot_zid]; r.zid_meta = S.c_meta[(size_t)a * P.Cmax]; r.zid_word = S.c_bits[(size_t)P.aoff[a] * P.Cmax]; }
  }
  for (int k = 0; k < 4; ++k) r.visited[k] = S.counters[k];
  B.results[i] = r;
}
KG_HD void sim_gather(const DevProblem& P, const SimBatch& B, const SimState& S, uint32_t i) {
  uint32_t errors = 0, claim0_np = 0;
  sim_gather_scan(B, S, i, 0, 1, &errors, &claim0_np);
  sim_gather_finish(P, B, S, i, errors, claim0_np);
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
  uint64_t* podmask_out;       // [NV][TW]
  struct PodCtx* pctx_out;     // [NV]
};

// ------------------------------------------------------------------ InstanceTypes.OrderByPrice / Truncate per new NodeClaim
// (the reference's call is pkg/providers/instance/instance.go:101 with maxInstanceTypes = 60, instance.go:54;
// comparator as re-stated by pkg/providers/instancetype/suite_test.go:414-421: price of the cheapest available offering
// compatible with the claim's requirements, ties by name).  One workgroup per NodeClaim: a thread prices its instance
// types over the catalog's resident offering table (IEEE doubles, compared as doubles), every type's rank is the number of
// option types ordered before it, and the types ranked below max_types are written in order.
constexpr uint32_t KG_TRUNC_T = 2048;  // instance types the kernel keeps prices of in LDS (catalogs beyond: ordered on the host)
struct TruncArgs {
  const uint32_t* off_begin;   // [T+1] offering rows per type
  const uint16_t* off_zone;    // catalog-local zone value id
  const uint16_t* off_ct;      // catalog-local capacity-type value id
  const double* off_price;
  const uint8_t* off_avail;
  const uint16_t* name_rank;   // [T] lexicographic rank of the type's name
  uint32_t max_types;
  // One record per NodeClaim, rec_words 32-bit words each -- everything the host decodes of a claim, so that the results
  // come back in one copy: [0] pods [1] template [2] order stamp [3] option types n, [4, 4 + max_types) the first
  // min(n, max_types) catalog type indices in (price, name) order, then Ka requirement meta words, Wa value words and R
  // requests (64-bit, low half first)
  uint32_t* rec;
  uint32_t rec_words;
};
KG_HD uint32_t trunc_rec_words(uint32_t max_types, int Ka, int Wa, int R) { return 4 + max_types + (uint32_t)Ka + 2 * (uint32_t)Wa + 2 * (uint32_t)R; }
// the columns of claim c -> its record (thread `tid` of `nt`)
KG_HD void trunc_pack(const DevProblem& P, const SimState& S, const TruncArgs& X, uint32_t c, uint32_t tid, uint32_t nt, uint32_t nset) {
  uint32_t* rec = X.rec + (size_t)c * X.rec_words;
  if (tid == 0) { rec[0] = S.c_npods[c]; rec[1] = S.c_tmpl[c]; rec[2] = S.c_stamp[c]; rec[3] = nset; }
  uint32_t* q = rec + 4 + X.max_types;
  for (uint32_t a = tid; a < (uint32_t)P.Ka; a += nt) q[a] = S.c_meta[(size_t)a * P.Cmax + c];
  q += P.Ka;
  for (uint32_t w = tid; w < (uint32_t)P.Wa; w += nt) { const uint64_t v = S.c_bits[(size_t)w * P.Cmax + c]; q[2 * w] = (uint32_t)v; q[2 * w + 1] = (uint32_t)(v >> 32); }
  q += 2 * P.Wa;
  for (uint32_t r = tid; r < (uint32_t)P.R; r += nt) { const uint64_t v = (uint64_t)S.c_req[(size_t)r * P.Cmax + c]; q[2 * r] = (uint32_t)v; q[2 * r + 1] = (uint32_t)(v >> 32); }
}
constexpr double KG_PRICE_NONE = 1.7976931348623157e308;  // math.MaxFloat64: no compatible available offering
// allowed values of the claim on the zone / capacity-type keys over the catalog's value numbering (zone ids folded into zones)
KG_HD void claim_offering_masks(const DevProblem& P, const SimState& S, uint32_t c, uint64_t* zmask, uint64_t* ctmask) {
  uint64_t hm[3] = {~0ull, ~0ull, ~0ull};
  const int sl[3] = {P.slot_zone, P.slot_ct, P.slot_zid};
  for (int i = 0; i < 3; ++i) {
    if (sl[i] < 0) continue;
    const int a = P.spec_slot[sl[i]];
    SReq r;
    r.meta = S.c_meta[(size_t)a * P.Cmax + c];
    r.word = (r.meta & RM_PRESENT) ? S.c_bits[(size_t)P.aoff[a] * P.Cmax + c] : 0;
    hm[i] = has_mask1(P, a, r);
  }
  uint64_t zm = hm[0];
  if (P.slot_zid >= 0)
    for (int z = 0; z < 64; ++z) {
      const uint8_t zid = P.zone_zid[z];
      if (zid != 0xFF && !((hm[2] >> zid) & 1)) zm &= ~(1ull << z);
    }
  *zmask = zm; *ctmask = hm[1];
}
KG_HD double type_cheapest(const TruncArgs& X, uint32_t t, uint64_t zmask, uint64_t ctmask) {
  double best = KG_PRICE_NONE;
  for (uint32_t o = X.off_begin[t]; o < X.off_begin[t + 1]; ++o) {
    if (!X.off_avail[o] || !((zmask >> X.off_zone[o]) & 1) || !((ctmask >> X.off_ct[o]) & 1)) continue;
    const double p = X.off_price[o];
    best = p < best ? p : best;
  }
  return best;
}
KG_HD bool price_before(double pa, uint32_t ra, double pb, uint32_t rb) { return pa < pb || (pa == pb && ra < rb); }

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

// word w of "types pod variant vi can ever accept": for every requirement entry of the pod, the types whose value
// satisfies it, that do not mention the key, or that list it as DoesNotExist (admissible when the merged operator is
// NotIn / DoesNotExist); and an available offering in a zone / capacity type the pod itself allows.  A NodeClaim's
// options after NodeClaim.Add are always a subset of (its options AND this mask): the prefilter's soundness.
KG_HD uint64_t pod_mask_word(const DevProblem& P, uint32_t vi, int w) {
  const VariantCtx& vc = P.vctx[vi];
  uint64_t m = ~0ull;
  uint64_t zmask = ~0ull, ctmask = ~0ull;
  for (uint32_t j = 0; j < vc.ne; ++j) {
    const PodEntry& pe = vc.ent[j];
    m &= type_filter_word(P, pe.e, w) | P.kU[(size_t)pe.slot * P.TW + w];
    if (pe.spec >= 0 && (pe.spec == P.slot_zone || pe.spec == P.slot_ct || pe.spec == P.slot_zid)) {
      SReq r{pe.meta, (P.rs_bits + P.e_bits[pe.e])[0]};
      const uint64_t hm = has_mask1(P, pe.slot, r);
      if (pe.spec == P.slot_zone) zmask &= hm;
      if (pe.spec == P.slot_ct) ctmask &= hm;
      if (pe.spec == P.slot_zid) {
        uint64_t zm = 0;
        for (int z = 0; z < P.n_zone_vals; ++z) {
          uint8_t zid = P.zone_zid[z];
          if (zid == 0xFF || ((hm >> zid) & 1)) zm |= 1ull << z;
        }
        zmask &= zm;
      }
    }
  }
  uint64_t f = 0;
  for (int z = 0; z < P.n_zone_vals; ++z) {
    if (!((zmask >> z) & 1)) continue;
    for (int ct = 0; ct < P.n_ct_vals; ++ct)
      if ((ctmask >> ct) & 1) f |= P.offmask[((size_t)z * 4 + ct) * P.TW + w];
  }
  return m & f;
}

// word i of the contiguous pod-context image of variant vi (k_pod_ctx): the variant's precompiled record, its toleration
// word, its own type mask and the type masks of its requirement entries, laid out as struct PodCtx (requests stay zero)
KG_HD uint64_t pod_ctx_word(const DevProblem& P, uint32_t vi, uint32_t i) {
  const uint32_t o_tolw = offsetof(PodCtx, tolw) / 8, o_vc = offsetof(PodCtx, vc) / 8, o_pm = offsetof(PodCtx, podmask) / 8, o_eam = offsetof(PodCtx, eam) / 8;
  const VariantCtx& vc = P.vctx[vi];
  if (i == o_tolw) return P.tol_tmpl[vc.v.tolset];
  if (i >= o_vc && i < o_vc + sizeof(VariantCtx) / 8) return ((const uint64_t*)&vc)[i - o_vc];
  if (i >= o_pm && i < o_pm + KG_TWLDS) return (int)(i - o_pm) < P.TW ? P.podmask[(size_t)vi * P.TW + (i - o_pm)] : 0;
  if (i >= o_eam && i < o_eam + KG_MAXE * KG_TWLDS) {
    const uint32_t j = (i - o_eam) / KG_TWLDS, w = (i - o_eam) % KG_TWLDS;
    return (j < vc.ne && (int)w < P.TW) ? P.e_amask[(size_t)vc.ent[j].e * P.TW + w] : 0;
  }
  return 0;
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
