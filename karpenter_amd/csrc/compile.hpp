// compile.hpp -- host compile of one scheduling problem into device tables.
//
// Mirrors what provisioner.NewScheduler / scheduling.NewScheduler / NewTopology assemble before
// Solve runs: NodePools in weight order with their
// NodeClaimTemplate requirements, daemon overhead and remaining limits; ExistingNodes in
// (initialized, name) order; pod requirement sets (NewPodRequirements / NewStrictPodRequirements);
// the Preferences.Relax chain of each pod as a list of precompiled variants; topology groups with
// their initial domain counts.
#pragma once
#include <array>
#include <future>
#include <set>

#include "ctx.hpp"

namespace kg {

struct PSel {
  bool nil = false;
  std::vector<std::pair<uint32_t, uint32_t>> ml;
  struct Ex { uint32_t key; int op; std::vector<uint32_t> vals; };
  std::vector<Ex> ex;
  bool matches(const std::vector<std::pair<uint32_t, uint32_t>>& labels) const;
  std::string canon() const;
};
struct PSpread { uint32_t key; int max_skew, min_domains, when; PSel sel; };
struct PTerm { uint32_t key; PSel sel; std::vector<uint32_t> ns; int weight; };
struct PTol { int key; uint32_t value; int op, effect; };
struct PTaint { uint32_t key, value; int effect; };

struct PodClass {
  uint32_t ns = 0;
  std::vector<std::pair<uint32_t, uint32_t>> labels;  // sorted (lab ids)
  std::vector<HReq> node_selector;
  std::vector<std::vector<HReq>> required;
  std::vector<std::pair<int, std::vector<HReq>>> preferred;
  std::vector<PTol> tols;
  std::vector<PSpread> spreads;
  std::vector<PTerm> aff_req, aff_pref, anti_req, anti_pref;
  bool host_ports = false;
  std::vector<std::array<int, 3>> ports;  // (ip label id or -1 = 0.0.0.0, port, protocol)
};

struct PodRef {
  uint32_t cls;
  uint32_t rq_off = 0, rq_n = 0;  // requests (incl. pods:1): ParsedPods::rq[rq_off .. rq_off + rq_n) -- one pool, no allocation per pod
  uint32_t uid_off = 0, uid_len = 0;  // ParsedPods::uid_pool[uid_off .. uid_off + uid_len): one pool, no allocation per pod
  int64_t creation;
  int node;
  bool reschedulable;
  int32_t priority = 0, deletion_cost = 0;
};

struct ParsedPods {
  std::vector<PodClass> classes;
  std::vector<PodRef> pods;
  std::vector<std::pair<uint32_t, int64_t>> rq;  // (resource id, milli) pairs of every pod's requests
  std::string uid_pool;                            // the pods' UIDs back to back (Queue order tie-break)
  // a set that borrows the two pools of another one instead of holding copies (the pod universe of a consolidation plan over a
  // snapshot without pending pods: its pods are the snapshot's, offsets unchanged); readers go through rqs() / uid()
  const ParsedPods* pools_of = nullptr;
  const std::vector<std::pair<uint32_t, int64_t>>& rqs() const { return pools_of ? pools_of->rq : rq; }
  std::string_view uid(const PodRef& p) const { return std::string_view(pools_of ? pools_of->uid_pool : uid_pool).substr(p.uid_off, p.uid_len); }
  // volumes (ExistingNode.Add's volume limits): the pods' volume ids, sorted; a pod's count per CSI driver rides in `rq` as a
  // request of the internal resource "karpgpu.internal/volumes/<driver>" (additive as long as no volume is shared)
  std::vector<uint32_t> vols;
  bool vols_shared = false;  // two pods of this set name the same volume
};


void parse_pods(kg_ctx* ctx, const kg_pods_desc* d, ParsedPods* out, bool add_pods_one);

struct CNode {
  std::string name;
  std::vector<std::pair<uint32_t, uint32_t>> labels;  // (global key, value id)
  std::vector<PTaint> taints;
  std::vector<std::pair<uint32_t, int64_t>> available, capacity, daemon;
  bool initialized, managed, deleting;
  uint32_t hostname_val;  // value id under the hostname key
  std::vector<uint32_t> volume_limited;  // internal volume resources this node has a limit for (its `available` carries what is left)
};

}  // namespace kg

struct kg_cluster {
  kg_ctx* ctx;
  std::vector<kg::CNode> nodes;
  kg::ParsedPods bound;
  // place of every bound pod in the Queue order among the bound pods: worked out by `rank_job` (started at the end of
  // kg_cluster_upload for large snapshots, reads `bound` only) or at first use; wait_rank() before reading it or freeing the cluster
  std::vector<uint32_t> bound_rank;
  std::future<void> rank_job;
  void wait_rank() { if (rank_job.valid()) rank_job.get(); }
  std::vector<struct kg_cons_plan*> plans;    // consolidation plans built over this snapshot (karpgpu.cpp), least recently used dropped beyond 4
  uint64_t plan_clock = 0;
};
