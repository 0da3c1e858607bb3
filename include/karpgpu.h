/*
 * karpgpu.h -- C ABI of libkarpgpu, the MI355X (gfx950) implementation of Karpenter's
 * provisioning / consolidation bin-packing hot path.
 *
 * This is the drop-in boundary: a Go controller keeps its API surface and calls these entry
 * points through cgo (binding sketch in INTEGRATION.md).  Every entry point is `extern "C"`,
 * takes plain pointers and sizes, returns an int32 status (0 = KG_OK) and keeps NO caller
 * pointer after it returns (cgo rule).  Outputs go into caller-allocated buffers with explicit
 * capacities; a `needed_*` field reports the size that would have been required.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   cloudprovider.GetInstanceTypes            pkg/cloudprovider/cloudprovider.go:172
 *   instancetype.DefaultProvider.List         pkg/providers/instancetype/instancetype.go:100-175
 *   instancetype.NewInstanceType              pkg/providers/instancetype/types.go:51-71
 *   resolveInstanceTypes (launch filter)      pkg/cloudprovider/cloudprovider.go:267-282
 *   InstanceTypes.Truncate call site          pkg/providers/instance/instance.go:101 (max 60, :54)
 *   filterExoticInstanceTypes/filterUnwantedSpot  pkg/providers/instance/instance.go:429-477
 *   Scheduler.Solve / NodeClaim.Add / ExistingNode.Add / Topology / Preferences / Queue and the
 *   disruption simulation live in sigs.k8s.io/karpenter v0.37.1-0.20240715214311-a122e25df3de
 *   (go.mod:34; source absent from the reference checkout -- SURVEY.md Appendix A is the spec).
 *
 * Conventions
 *   - strings: every name (label key/value, resource, instance type, zone ...) is a uint32 index
 *     into the descriptor's own string table (kg_strtab).  KG_NONE = "no string".
 *   - quantities: int64 milli-units (cpu "940m" = 940, memory "1Mi" = 1048576000).  Finer than
 *     milli is not representable -> KG_EUNSUPPORTED, caller uses its CPU path.
 *   - ranges: {begin,count} into the flat pool of the same descriptor.
 */
#ifndef KARPGPU_H
#define KARPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_NONE 0xFFFFFFFFu

/* status codes */
enum {
  KG_OK = 0,
  KG_EINVAL = 1,       /* malformed descriptor */
  KG_ENOMEM = 2,       /* host or device allocation failed, or an output buffer is too small */
  KG_EDEVICE = 3,      /* HIP error; kg_last_error has the text */
  KG_EUNSUPPORTED = 4, /* input not representable on the device path (caller falls back to CPU) */
  KG_ESTALE = 5        /* catalog/cluster generation mismatch */
};

/* requirement operators (corev1.NodeSelectorOperator) */
enum { KG_OP_IN = 0, KG_OP_NOT_IN = 1, KG_OP_EXISTS = 2, KG_OP_DOES_NOT_EXIST = 3, KG_OP_GT = 4, KG_OP_LT = 5 };
/* taint effects */
enum { KG_EFFECT_NONE = 0, KG_EFFECT_NO_SCHEDULE = 1, KG_EFFECT_PREFER_NO_SCHEDULE = 2, KG_EFFECT_NO_EXECUTE = 3 };
/* toleration operators */
enum { KG_TOL_EQUAL = 0, KG_TOL_EXISTS = 1 };
/* topology spread whenUnsatisfiable */
enum { KG_DO_NOT_SCHEDULE = 0, KG_SCHEDULE_ANYWAY = 1 };
/* per-pod result kinds */
enum { KG_POD_UNSCHEDULABLE = 0, KG_POD_EXISTING_NODE = 1, KG_POD_NEW_CLAIM = 2 };
/* consolidation decisions (disruption Command.Decision()) */
enum { KG_DECISION_NOOP = 0, KG_DECISION_DELETE = 1, KG_DECISION_REPLACE = 2 };

typedef struct { uint32_t begin, count; } kg_range;

typedef struct {
  const char* data;        /* concatenated bytes, no terminators needed */
  const uint32_t* offsets; /* count+1 entries; string i = data[offsets[i] .. offsets[i+1]) */
  uint32_t count;
} kg_strtab;

typedef struct {
  uint32_t key;
  uint8_t op;         /* KG_OP_* */
  int32_t min_values; /* -1 = unset (NodeSelectorRequirementWithMinValues) */
  kg_range values;    /* into pools.req_values (string ids) */
} kg_requirement;

typedef struct { uint32_t name; int64_t milli; } kg_quantity;
typedef struct { uint32_t key, value; } kg_label;
typedef struct { uint32_t key, value; uint8_t effect; } kg_taint;
typedef struct { uint32_t key /* KG_NONE = empty */; uint32_t value; uint8_t op; uint8_t effect /* KG_EFFECT_NONE = any */; } kg_toleration;

/* metav1.LabelSelector.  is_nil != 0 selects nothing; an empty non-nil selector selects all. */
typedef struct {
  uint8_t is_nil;
  kg_range match_labels; /* into pools.labels */
  kg_range match_exprs;  /* into pools.reqs, operators In/NotIn/Exists/DoesNotExist */
} kg_selector;

/* shared flat pools: every descriptor carries one */
typedef struct {
  kg_strtab strings;
  const kg_requirement* reqs;        uint32_t n_reqs;
  const uint32_t* req_values;        uint32_t n_req_values;
  const kg_quantity* quantities;     uint32_t n_quantities;
  const kg_label* labels;            uint32_t n_labels;
  const kg_taint* taints;            uint32_t n_taints;
  const kg_toleration* tolerations;  uint32_t n_tolerations;
  const kg_range* ranges;            uint32_t n_ranges; /* second-level ranges (lists of lists) */
} kg_pools;

/* ---------------------------------------------------------------- catalog */
/* cloudprovider.Offering as built at instancetype.go:290-300 */
typedef struct {
  uint32_t zone, capacity_type, zone_id /* KG_NONE when the subnet has no zone id */;
  double price;
  uint8_t available;
} kg_offering;

/* cloudprovider.InstanceType as built at types.go:56-66 */
typedef struct {
  uint32_t name;
  kg_range requirements; /* pools.reqs: 26-27 keys, DoesNotExist entries included */
  kg_range capacity;     /* pools.quantities */
  kg_range overhead;     /* pools.quantities: KubeReserved+SystemReserved+EvictionThreshold summed */
  kg_range offerings;    /* into offerings[] */
} kg_instance_type;

typedef struct {
  kg_pools pools;
  const kg_instance_type* types; uint32_t n_types;
  const kg_offering* offerings;  uint32_t n_offerings;
} kg_catalog_desc;

/* Raw EC2 facts for one type: the inputs of NewInstanceType (types.go:51).  Lets the library do
 * the whole List() miss path (instancetype.go:155-172) natively: offerings + requirements +
 * capacity + overhead. */
typedef struct {
  uint32_t name;
  uint32_t arch;          /* already mapped through AWSToKubeArchitectures (amd64/arm64) */
  uint8_t first_arch_is_arm64; /* SupportedArchitectures[0]=="arm64" (types.go:219) */
  uint32_t hypervisor;    /* KG_NONE -> "" */
  uint32_t cpu_manufacturer; /* raw ProcessorInfo.Manufacturer or KG_NONE */
  int64_t vcpus, memory_mib;
  uint8_t encryption_in_transit;
  int64_t ebs_max_bandwidth_mbps; /* -1 unless EbsOptimizedSupport==default */
  int64_t network_bandwidth_mbps; /* -1 when not in the bandwidth table */
  int64_t local_nvme_gb;     /* -1 when no NVMe instance storage (types.go:129) */
  int64_t instance_storage_total_gb; /* -1 when InstanceStorageInfo or TotalSizeInGB nil (types.go:232) */
  /* GpuInfo.Gpus, InferenceAcceleratorInfo.Accelerators */
  kg_range gpus;          /* into devices[] */
  kg_range accelerators;  /* into devices[] */
  int64_t efa_max_interfaces; /* 0 when none */
  /* NetworkInfo */
  int64_t default_card_max_interfaces; /* NetworkCards[DefaultNetworkCardIndex].MaximumNetworkInterfaces */
  int64_t ipv4_per_interface;
  /* VPC limits table (zz_generated.vpclimits.go) */
  uint8_t has_limits, trunking_compatible;
  int64_t branch_interfaces, limits_ipv4_per_interface;
  kg_range usage_classes; /* pools.req_values: "on-demand","spot","capacity-block" */
  kg_range zones_offered; /* pools.req_values: zones from DescribeInstanceTypeOfferings */
} kg_ec2_type;
typedef struct { uint32_t name, manufacturer; int64_t count, memory_mib; } kg_ec2_device;

typedef struct { uint32_t zone, zone_id /* KG_NONE */; } kg_subnet;
typedef struct { uint32_t instance_type, zone; double price; } kg_spot_price;
typedef struct { uint32_t instance_type; double price; } kg_od_price;
typedef struct { uint32_t capacity_type, instance_type, zone; } kg_unavailable; /* ICE cache key ct:it:zone */

enum { KG_AMI_AL2 = 0, KG_AMI_AL2023 = 1, KG_AMI_BOTTLEROCKET = 2, KG_AMI_UBUNTU = 3,
       KG_AMI_WINDOWS2019 = 4, KG_AMI_WINDOWS2022 = 5, KG_AMI_CUSTOM = 6 };

typedef struct {
  kg_pools pools;
  const kg_ec2_type* types;      uint32_t n_types;
  const kg_ec2_device* devices;  uint32_t n_devices;
  const kg_subnet* subnets;      uint32_t n_subnets;
  const kg_od_price* od_prices;  uint32_t n_od_prices;
  const kg_spot_price* spot_prices; uint32_t n_spot_prices;
  const kg_unavailable* unavailable; uint32_t n_unavailable;
  uint32_t region;
  uint8_t ami_family;                 /* KG_AMI_* */
  /* options.Options (pkg/operator/options/options.go:44,46) */
  double vm_memory_overhead_percent;  /* 0.075 */
  int32_t reserved_enis;
  /* KubeletConfiguration (nil-able fields: <0 = nil) */
  int32_t max_pods, pods_per_core;
  kg_range kube_reserved, system_reserved;     /* pools.quantities */
  kg_range eviction_hard, eviction_soft;       /* pools.labels: signal name -> value string ("5%", "100Mi") */
  uint8_t has_eviction_hard, has_eviction_soft;
  /* EC2NodeClass */
  uint8_t instance_store_policy_raid0;
  int64_t root_volume_milli;      /* blockDeviceMappings RootVolume size, <0 = none */
  int64_t ephemeral_device_volume_milli; /* bdm matching the AMI family's EphemeralBlockDevice, <0 = none */
  int64_t last_bdm_volume_milli;  /* Custom AMI family: last mapping's size, <0 = none */
  uint8_t has_block_device_mappings;
} kg_ec2_catalog_desc;

/* ---------------------------------------------------------------- pods */
typedef struct {
  uint32_t topology_key;
  int32_t max_skew;
  int32_t min_domains;   /* <0 = nil */
  uint8_t when_unsatisfiable;
  kg_selector selector;
} kg_spread;

typedef struct {
  uint32_t topology_key;
  kg_selector selector;
  kg_range namespaces;   /* pools.req_values; empty -> the pod's own namespace */
  int32_t weight;        /* preferred terms only */
} kg_pod_affinity_term;

typedef struct { int32_t weight; kg_range reqs; } kg_weighted_term;
typedef struct { uint32_t ip /* KG_NONE = 0.0.0.0 */; int32_t port; uint8_t protocol; } kg_host_port;

typedef struct {
  uint32_t uid, ns;
  int64_t creation_ts;          /* seconds; Queue tie-break */
  kg_range labels;              /* pools.labels */
  kg_range requests;            /* pools.quantities = resources.RequestsForPods WITHOUT the implicit pods:1 */
  kg_range node_selector;       /* pools.labels */
  kg_range required_terms;      /* pools.ranges -> each a range of pools.reqs (OR terms) */
  kg_range preferred_terms;     /* weighted_terms[] */
  kg_range tolerations;
  kg_range spreads;             /* spreads[] */
  kg_range affinity_required, affinity_preferred;       /* affinity_terms[] */
  kg_range anti_affinity_required, anti_affinity_preferred;
  kg_range host_ports;
  /* disruption cost inputs (designs/consolidation.md:25-42) */
  int32_t priority;
  int32_t deletion_cost;
  int32_t node;                 /* bound pods: index into cluster nodes; -1 for pending pods */
  uint8_t reschedulable;        /* bound pods: counted in candidate.reschedulablePods */
  kg_range volumes;             /* pools.labels: (CSI driver, volume id) of every PVC-backed volume the pod mounts -- what
                                   ExistingNode.Add checks against the node's volume limits.  A volume
                                   named by two pods of one call, or already in use on a node by another pod, makes the count
                                   a set union the device path does not model: KG_EUNSUPPORTED, the caller keeps its CPU path */
} kg_pod;

typedef struct {
  kg_pools pools;
  const kg_pod* pods;                       uint32_t n_pods;
  const kg_weighted_term* weighted_terms;   uint32_t n_weighted_terms;
  const kg_spread* spreads;                 uint32_t n_spreads;
  const kg_pod_affinity_term* affinity_terms; uint32_t n_affinity_terms;
  const kg_host_port* host_ports;           uint32_t n_host_ports;
} kg_pods_desc;

/* ---------------------------------------------------------------- cluster */
typedef struct {
  uint32_t name;            /* node name; hostname label falls back to it */
  kg_range labels;          /* pools.labels (zone, instance-type, capacity-type, nodepool, hostname ...) */
  kg_range taints;          /* StateNode.Taints() */
  kg_range available;       /* StateNode.Available(): allocatable - bound pod requests */
  kg_range capacity;        /* StateNode.Capacity(): charged against NodePool limits */
  kg_range daemon_requests; /* remaining daemonset requests, already clamped >= 0 */
  uint8_t initialized, managed, deleting;
  double price;             /* unused on input; reserved */
  kg_range volume_limits;   /* pools.quantities: CSI driver -> attachable volume count x 1000 (CSINode allocatable); drivers not
                               listed are unlimited.  Volumes in use = those of the pods bound to the node */
} kg_node;

typedef struct {
  kg_pods_desc pods;        /* bound pods; pod.node indexes nodes[] */
  const kg_node* nodes;     uint32_t n_nodes;
} kg_cluster_desc;

/* ---------------------------------------------------------------- nodepools */
typedef struct {
  uint32_t name;
  int32_t weight;
  kg_range requirements;    /* pools.reqs (spec.template.spec.requirements, minValues allowed) */
  kg_range labels;          /* pools.labels (spec.template.metadata.labels) */
  kg_range taints;          /* spec.template.spec.taints */
  uint8_t has_limits;
  kg_range limits;          /* pools.quantities */
  kg_range daemon_overhead; /* pools.quantities: requests of daemonset pods that fit the template */
  uint32_t catalog;         /* index into the catalogs[] passed alongside: the InstanceTypes of this NodePool's EC2NodeClass
                               (cloudprovider.go:172-190); NodePools may name different catalogs in one call */
  uint64_t catalog_generation; /* the List() cache key (instancetype.go:127-136) the caller resolved for this NodePool; when
                                  non-zero and different from the generation the resident catalog carries, the call returns
                                  KG_ESTALE and changes nothing (upload / kg_catalog_update_offerings first).  0 = not checked */
} kg_nodepool;

typedef struct {
  kg_pools pools;
  const kg_nodepool* nodepools; uint32_t n_nodepools;
  kg_range well_known_labels;   /* pools.req_values: karpv1.WellKnownLabels incl. provider additions (labels.go:31-53) */
  int32_t max_instance_types;   /* 60 (instance.go:54); <=0 disables truncation */
} kg_nodepools_desc;

/* ---------------------------------------------------------------- results */
typedef struct {
  uint32_t nodepool;        /* index into nodepools[] */
  uint32_t n_pods;
  kg_range types;           /* into type_ids[]: price-ordered, truncated InstanceTypeOptions (catalog-local indices) */
  kg_range requirements;    /* into reqs[] */
  kg_range requests;        /* into quantities[] */
  uint8_t truncation_failed;/* minValues broke under Truncate: claim dropped by the reference */
} kg_claim_out;

typedef struct {
  uint32_t key;             /* offset id into strings (see below) */
  uint8_t complement;
  uint8_t has_gt, has_lt;
  int64_t gt, lt;
  kg_range values;          /* into value_ids[] */
  int32_t min_values;       /* NodeSelectorRequirementWithMinValues.MinValues the claim carries for the key; -1 = none */
  uint32_t pad_;
} kg_req_out;

/* All arrays are caller-allocated; cap_* are capacities, n_* are filled by the library and
 * needed_* report the untruncated sizes.  Strings in results are offsets (string index) into
 * the result's own string blob. */
typedef struct {
  /* per pod, in input order */
  uint8_t* pod_kind;  int32_t* pod_index; uint32_t* pod_reason; uint32_t cap_pods;
  kg_claim_out* claims;   uint32_t cap_claims, n_claims, needed_claims;
  uint32_t* type_ids;     uint32_t cap_type_ids, n_type_ids, needed_type_ids;
  kg_req_out* reqs;       uint32_t cap_reqs, n_reqs, needed_reqs;
  uint32_t* value_ids;    uint32_t cap_value_ids, n_value_ids, needed_value_ids;
  kg_quantity* quantities;uint32_t cap_quantities, n_quantities, needed_quantities;
  char* str_data;         uint32_t cap_str_data, n_str_data, needed_str_data;
  uint32_t* str_offsets;  uint32_t cap_strs, n_strs, needed_strs;
  /* counters */
  uint64_t visited_existing, visited_claims, visited_claim_types, new_claim_scans, relaxations;
  double algorithmic_bytes; /* SURVEY.md 8(d) definition */
  double device_ms;         /* HIP-event time of all kernels of the call (prepass + solve) */
  double prepass_ms;        /* k_type_filter + k_node_static */
  double solve_ms;          /* k_solve (the persistent first-fit kernel) */
  double host_compile_ms;   /* descriptor parse + requirement compile + table upload */
  double host_decode_ms;    /* result download + price ordering / truncation + result encode */
} kg_results;

/* per-pod failure reason bits */
enum { KG_REASON_TAINTS = 1, KG_REASON_REQUIREMENTS = 2, KG_REASON_TOPOLOGY = 4, KG_REASON_NO_TYPE = 8,
       KG_REASON_LIMITS = 16, KG_REASON_MIN_VALUES = 32, KG_REASON_RESOURCES = 64, KG_REASON_UNINITIALIZED_NODE = 128 };

typedef struct {
  int32_t decision;          /* KG_DECISION_* */
  int32_t n_candidates;      /* candidates in the command (prefix length for multi-node) */
  int32_t first_candidate;   /* single-node: winning index in the given order; multi: 0 */
  uint32_t nodepool;         /* replacement claim's nodepool */
  uint32_t n_types;          /* replacement InstanceTypeOptions, price ordered */
  uint32_t* type_ids;        uint32_t cap_type_ids;
  uint8_t replacement_spot_only; /* capacity-type pinned to spot (computeConsolidation tail) */
  double candidate_price;
  uint64_t simulations;      /* number of Solve simulations evaluated */
  double device_ms;
} kg_decision;

/* per-candidate record (also the fixed-size record gathered across GPUs) */
/* Row sizes of the algorithmic byte count (roofline.json): what one visit of the reference algorithm to an
 * existing node / an in-flight NodeClaim / an instance type is priced at in kg_results.algorithmic_bytes and
 * kg_stats.cons_algorithmic_bytes.  They price the REFERENCE's visits, not this library's memory traffic. */
#define KG_ROW_BYTES_EXISTING_NODE 212
#define KG_ROW_BYTES_NODE_CLAIM 400
#define KG_ROW_BYTES_INSTANCE_TYPE 336

#define KG_RECORD_TYPE_WORDS 16
typedef struct {
  int32_t candidate;         /* index in candidates[] (single) or `mid` = prefix length-1 (multi) */
  int32_t decision;          /* computeConsolidation's verdict for this unit, before first-wins / binary-search selection */
  uint32_t nodepool, n_types;
  double candidate_price;    /* getCandidatePrices: sum of the cheapest offering compatible with each candidate's labels */
  uint64_t type_mask[KG_RECORD_TYPE_WORDS]; /* catalog types of the replacement after the price filters (<= 1024 types: the limit of a
                                               consolidation simulation's option mask; larger catalogs return KG_EUNSUPPORTED) */
  uint64_t zone_mask, ct_mask; /* replacement requirements on zone / capacity type as the library's value-id masks (opaque;
                                  kg_consolidate_select uses them to price-order the options) */
  uint8_t spot_only; uint8_t pad[7];
} kg_candidate_record;

typedef struct {
  const int32_t* candidates; uint32_t n_candidates; /* node indices, already in disruption-cost order */
  uint8_t spot_to_spot_enabled;                     /* FEATURE_GATES=SpotToSpotConsolidation */
  int32_t shard, n_shards;                          /* this process evaluates units i with i % n_shards == shard */
} kg_candidates_desc;

typedef struct {
  uint64_t solves, pods, kernel_launches;
  double device_ms_total, host_compile_ms_total;
  uint64_t device_bytes_resident;
  uint64_t catalog_words_patched; /* 8-byte words of resident catalogs rewritten in place by kg_catalog_update_offerings */
  uint64_t wave_solves;           /* kg_solve calls served by a barrier-free wave kernel (batches without topology groups / existing nodes / minValues) */
  /* the last kg_consolidate_single / _multi call of the context */
  uint64_t cons_simulations;      /* simulations run on the device (units with reschedulable pods) */
  double cons_algorithmic_bytes;  /* SURVEY.md 8(d) bytes summed over the simulations; 0 unless the visit counters are on */
  double cons_prepass_ms, cons_sim_init_ms, cons_simulate_ms, cons_gather_ms; /* HIP-event time per kernel group */
  uint64_t cons_state_bytes_per_simulation;
  uint64_t cons_plans_built, cons_plan_hits;  /* consolidation rounds that prepared their snapshot / found it prepared */
  /* of the wave_solves, those served by the speculative multi-wave kernel (several pods of the Queue in flight, retired in
   * Queue order), and over them: pods committed straight from their speculative evaluation, window claims evaluated again at
   * a pod's turn, pods rescanned at their turn (their claim took another pod meanwhile) */
  uint64_t spec_solves, spec_fast_commits, spec_challengers, spec_rescans;
  uint64_t spec_fallbacks;   /* speculative solves whose waves gave up waiting and that were run again on the single-wave kernel */
} kg_stats;

typedef struct kg_ctx kg_ctx;
typedef struct kg_catalog kg_catalog;
typedef struct kg_cluster kg_cluster;

/* lifecycle.  device < 0 -> current device.  Fails (KG_EDEVICE) when no GPU is present. */
int32_t kg_init(int32_t device, kg_ctx** out);
int32_t kg_device_count(int32_t* n); /* visible GPUs */
int32_t kg_destroy(kg_ctx* ctx);
int32_t kg_last_error(kg_ctx* ctx, char* buf, size_t n);
int32_t kg_stats_get(kg_ctx* ctx, kg_stats* out);
/* kg_results.visited_* / algorithmic_bytes count the rows the REFERENCE algorithm visits.  Keeping them
 * exact costs one extra pass per placement; production callers switch it off (default: on; bit 0 of `enabled`).  With
 * the counters off, a Solve without topology groups, existing nodes and minValues NodePools is served by the single-wave
 * kernel (kg_stats.wave_solves counts them), every other one by the workgroup kernel: same results.  The other bits are
 * debugging aids: 2 = phase timers to stderr, 4 / 8 = 1024 / 512 threads for the workgroup kernel, 16 = never use the
 * wave-local / single-wave paths, 32 = per-wave cycle accounting to stderr, 64 = never use the speculative multi-wave
 * kernel (such batches then run on the single-wave kernel), 128 = the speculative kernel evaluates every pod again at its
 * turn (no verdict of the speculative evaluation is used), 256 = treat every speculative solve as one whose waves gave up
 * waiting (exercises the automatic fallback to the single-wave kernel, kg_stats.spec_fallbacks).  Consolidation rounds honour 2 (every simulation
 * leaves time stamps; staging time and a per-pod fit of the batch go to stderr) and 32 (a debug instantiation of the batch
 * kernel, cycles per phase and wave summed over the batch); set them before the first round on a snapshot (plans are cached). */
int32_t kg_set_collect_stats(kg_ctx* ctx, int32_t enabled);
/* "hip-gfx950" for the product library */
const char* kg_backend(void);
/* sizeof of every element struct of this header, in declaration order (layout check for bindings); returns the count */
int32_t kg_abi_sizes(uint32_t* out, uint32_t cap);

/* catalog: replaces the List() miss path.  `generation` is the cache key of
 * instancetype.go:127-136 folded to 64 bits; re-upload when it changes. */
int32_t kg_catalog_upload(kg_ctx* ctx, const kg_catalog_desc* desc, uint64_t generation, kg_catalog** out);
/* same, starting from raw EC2 facts: does createOfferings + NewInstanceType natively. */
int32_t kg_catalog_build_ec2(kg_ctx* ctx, const kg_ec2_catalog_desc* desc, uint64_t generation, kg_catalog** out);
/* read back what the encoder computed for type i (for GetInstanceTypes parity): allocatable
 * per resource in `names` order, and offering count. */
int32_t kg_catalog_query(kg_catalog* cat, uint32_t type_index, const char* const* resource_names, uint32_t n,
                         int64_t* capacity_milli, int64_t* allocatable_milli, char* name_buf, size_t name_cap);
int32_t kg_catalog_size(kg_catalog* cat, uint32_t* n_types, uint64_t* generation);
/* read back the scheduling requirements the encoder holds for type i (computeRequirements, types.go:74-172, as maintained by
 * kg_catalog_update_offerings): one line per label key in lexicographic key order, "key<TAB>In<TAB>v1,v2,...\n" with the
 * values sorted, or "key<TAB>DoesNotExist<TAB>\n".  *needed = bytes required including the terminating 0; KG_ENOMEM when
 * cap is smaller (buf then holds a truncated, terminated prefix). */
int32_t kg_catalog_query_requirements(kg_catalog* cat, uint32_t type_index, char* buf, size_t cap, size_t* needed);
/* In-place maintenance of a resident catalog: replaces re-running DefaultProvider.List after an
 * ICE mark (pkg/cache/unavailableofferings.go:55-65 bumps SeqNum, which is part of the cache key of
 * instancetype.go:127-136) or a spot price refresh (pkg/providers/pricing/pricing.go:357-400).  Every update names an
 * EXISTING offering (type, zone, capacity type) and sets its availability and/or price, exactly what createOfferings
 * (instancetype.go:264-315) would recompute for it.  The zone / capacity-type / zone-id requirements of the touched
 * types are re-derived from their available offerings (types.go:80-88, 112-117) and only the changed words of the
 * device tables are rewritten.  `generation` becomes the catalog's new cache key. */
typedef struct {
  uint32_t type_index;
  uint32_t zone;           /* string ids into the strtab passed with the call */
  uint32_t capacity_type;
  int32_t available;       /* 0 / 1; -1 = leave as it is */
  double price;            /* < 0 = leave as it is */
} kg_offering_update;
int32_t kg_catalog_update_offerings(kg_ctx* ctx, kg_catalog* cat, const kg_strtab* strings, const kg_offering_update* updates, uint32_t n_updates,
                                    uint64_t generation);
int32_t kg_catalog_free(kg_catalog* cat);

int32_t kg_cluster_upload(kg_ctx* ctx, const kg_cluster_desc* desc, kg_cluster** out);
int32_t kg_cluster_free(kg_cluster* cl);
/* The step before the consolidation path (designs/consolidation.md:25-42, website
 * concepts/disruption.md:17-19): which state nodes are consolidation candidates, their disruption cost
 *   sum over the node's bound pods of clamp(1 + deletionCost / 2^27 + priority / 2^25, -10, 10) * lifetimeRemaining
 * and the candidate order (ascending cost; ties keep node order).  A node is a candidate when it is initialised, managed
 * by a NodePool (nodepool label present), not deleting, carries an instance-type label that `catalog` resolves (skipped
 * when catalog is NULL) and is not `blocked` (nominated / blocking PDB / do-not-disrupt pod: the controller's checks).
 * out_nodes receives node indices of the uploaded cluster, ready for kg_candidates_desc.candidates. */
typedef struct {
  const double* lifetime_remaining; /* [n_nodes] clamp(1 - age / expireAfter, 0, 1); NULL = 1 everywhere */
  const uint8_t* blocked;           /* [n_nodes]; NULL = none */
  uint32_t n_nodes;                 /* must equal the uploaded cluster's node count when either array is given */
} kg_candidate_inputs;
int32_t kg_candidates_order(kg_ctx* ctx, kg_catalog* catalog, kg_cluster* cluster, const kg_candidate_inputs* in, int32_t* out_nodes,
                            double* out_costs /* may be NULL */, uint32_t cap, uint32_t* n_out);

/* Scheduler.Solve + Results.TruncateInstanceTypes for one batch of pending pods. */
int32_t kg_solve(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster /* may be NULL */,
                 const kg_pods_desc* pods, const kg_nodepools_desc* nodepools, kg_results* out);

/* kg_solve for callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its Solve allocates freely on the Go heap).  Same inputs, outputs,
 * kernels and results as kg_solve; the per-solve device tables (problem image, solve state, per-NodeClaim rows) are carved
 * from a static 32 MiB device array inside the library (one copy per GPU, resolved at kg_init) instead of pooled hipMalloc
 * buffers, and the call makes no hipMalloc / hipFree / event creation.  A batch whose tables need more than the array
 * returns KG_ENOMEM with the size in kg_last_error (C2's 10 000 pods need about 20 MiB); a batch that names a resource or
 * label key the context has never seen returns KG_EUNSUPPORTED (the resident catalog tables would have to be rebuilt: run
 * kg_solve once).  Every call on one GPU uses the same array, which keeps the last call's contents: the library holds a
 * per-GPU mutex from carving the tables until the results are decoded, so calls that overlap on one GPU (two contexts, two
 * threads) run one after the other instead of racing; their timing events are a pair set aside at kg_init. */
int32_t kg_solve_noalloc(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster /* may be NULL */,
                         const kg_pods_desc* pods, const kg_nodepools_desc* nodepools, kg_results* out);

/* InstanceTypes.Truncate(reqs, max) / OrderByPrice for a launch (instance.go:101): orders the
 * given types by (cheapest compatible available offering price, name) and keeps max_types. */
int32_t kg_truncate(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements,
                    const uint32_t* type_ids, uint32_t n_types, int32_t max_types,
                    uint32_t* out_type_ids, uint32_t* n_out, uint8_t* min_values_failed);

/* launch-side filters of CloudProvider.Create (cloudprovider.go:267-282, instance.go:95-104):
 * resolveInstanceTypes -> (no minValues ? filterExotic + filterUnwantedSpot) -> Truncate(60). */
int32_t kg_launch_filter(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements,
                         kg_range requests, int32_t max_types,
                         uint32_t* out_type_ids, uint32_t cap, uint32_t* n_out, uint8_t* min_values_failed);

/* getCapacityType + getOverrides of a launch (pkg/providers/instance/instance.go:372-385, 323-359), the step between the
 * launch filter and CreateFleet.  *capacity_type = KG_CT_SPOT when the requirements admit spot and one of the given types has
 * an available offering compatible with them once capacity-type is pinned to spot, else KG_CT_ON_DEMAND.  out[] receives the
 * override list: for every type in type_ids order, every AVAILABLE offering in the type's offering order that is compatible
 * with the requirements (capacity-type replaced by the chosen one) and whose zone has a subnet -- subnet_zones are the keys
 * of ZonalSubnetsForLaunch (string ids into pools->strings); kg_override.subnet = index into subnet_zones, from which the
 * caller fills SubnetId / AvailabilityZone.  *n_out = entries the launch has (0: "no capacity offerings are currently
 * available given the constraints", instance.go:315-317); KG_ENOMEM when cap is smaller. */
#define KG_CT_ON_DEMAND 0
#define KG_CT_SPOT 1
typedef struct {
  uint32_t type_id;
  uint32_t subnet;
} kg_override;
int32_t kg_launch_overrides(kg_ctx* ctx, kg_catalog* catalog, const kg_pools* pools, kg_range requirements,
                            const uint32_t* type_ids, uint32_t n_types, const uint32_t* subnet_zones, uint32_t n_subnet_zones,
                            int32_t* capacity_type, kg_override* out, uint32_t cap, uint32_t* n_out);

/* consolidation: one independent SimulateScheduling per candidate (single) or per prefix (multi).
 * records[] receives one kg_candidate_record per unit evaluated by this shard (cap_records);
 * kg_consolidate_select replays the reference's selection over the gathered records. */
int32_t kg_consolidate_single(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster,
                              const kg_pods_desc* pending, const kg_nodepools_desc* nodepools,
                              const kg_candidates_desc* cands, kg_candidate_record* records, uint32_t cap_records,
                              uint32_t* n_records, double* device_ms);
int32_t kg_consolidate_multi(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster,
                             const kg_pods_desc* pending, const kg_nodepools_desc* nodepools,
                             const kg_candidates_desc* cands, kg_candidate_record* records, uint32_t cap_records,
                             uint32_t* n_records, double* device_ms);
/* first-wins (single) / binary-search replay (multi) over records gathered from all shards.  `catalogs` = the array the
 * records were computed with; the decision's type ids index the catalog of its (replacement) nodepool. */
int32_t kg_consolidate_select(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, int32_t multi, const kg_candidate_record* records,
                              uint32_t n_records, uint32_t n_candidates, kg_decision* out);
/* Validation re-simulation (Appendix A.9 "wait 15 s, recompute candidates, re-simulate, require the
 * same outcome"): the caller uploads a FRESH cluster snapshot and passes the command it is about to execute -- its
 * candidate nodes (indices into the fresh snapshot, at most 101: the longest prefix the multi-node search probes), KG_DECISION_DELETE / _REPLACE and, for a replace, the
 * instance types it would launch.  *valid = 1 when every rescheduled pod still schedules and the simulation ends in the
 * same kind of outcome: no new NodeClaim for a delete; exactly one for a replace, whose (truncated) options still
 * contain every type of the command. */
int32_t kg_consolidate_validate(kg_ctx* ctx, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* cluster, const kg_pods_desc* pending,
                                const kg_nodepools_desc* nodepools, const int32_t* candidates, uint32_t n_candidates, int32_t decision,
                                const uint32_t* type_ids, uint32_t n_types, uint8_t* valid, double* device_ms);

/* ---------------------------------------------------------------- several GPUs, one process
 * The reference's disruption controller is one Go process; a kg_group lets it use every GPU of the node without a second
 * process or a collective library: one context and one worker thread per device.  Catalogs and cluster snapshots are
 * replicated (the *_upload calls below fill an array of n_devices handles, device i's at [i]); kg_group_consolidate deals
 * the units of a round round-robin (unit k -> device k mod n_devices), runs every shard at once, gathers the records in this
 * process and replays the reference's selection (kg_consolidate_select) -- the same decision as one device, or as the
 * reference's sequential loop.  Scheduler.Solve does not shard (a chain of dependent placements): use kg_group_context(g, i)
 * with kg_solve for replicas.  One group call runs at a time; the contexts stay usable on their own from other threads. */
typedef struct kg_group kg_group;
/* devices == NULL: the first n_devices visible devices, or every visible device when n_devices == 0.  A device may be named
 * more than once (several contexts on one GPU, each with its own stream). */
int32_t kg_group_init(const int32_t* devices, uint32_t n_devices, kg_group** out);
int32_t kg_group_destroy(kg_group* g);
int32_t kg_group_size(kg_group* g, uint32_t* n_devices);
int32_t kg_group_context(kg_group* g, uint32_t i, kg_ctx** out);
int32_t kg_group_last_error(kg_group* g, char* buf, size_t n);
int32_t kg_group_catalog_upload(kg_group* g, const kg_catalog_desc* desc, uint64_t generation, kg_catalog** out /* [n_devices] */);
int32_t kg_group_catalog_update_offerings(kg_group* g, kg_catalog* const* catalogs /* [n_devices] */, const kg_strtab* strings,
                                          const kg_offering_update* updates, uint32_t n_updates, uint64_t generation);
int32_t kg_group_catalog_free(kg_group* g, kg_catalog* const* catalogs /* [n_devices] */);
int32_t kg_group_cluster_upload(kg_group* g, const kg_cluster_desc* desc, kg_cluster** out /* [n_devices] */);
int32_t kg_group_cluster_free(kg_group* g, kg_cluster* const* clusters /* [n_devices] */);
/* SingleNodeConsolidation.ComputeCommand / MultiNodeConsolidation.firstNConsolidationOption over all devices of the group.
 * catalogs = [n_devices][n_catalogs] (device-major), clusters = [n_devices]; cands->shard / n_shards are ignored.  records
 * (may be NULL) receives the gathered per-unit records; *device_ms = the slowest device's kernel time. */
int32_t kg_group_consolidate(kg_group* g, kg_catalog* const* catalogs, uint32_t n_catalogs, kg_cluster* const* clusters,
                             const kg_pods_desc* pending, const kg_nodepools_desc* nodepools, const kg_candidates_desc* cands, int32_t multi,
                             kg_candidate_record* records, uint32_t cap_records, uint32_t* n_records, kg_decision* out, double* device_ms);

#ifdef __cplusplus
}
#endif
#endif /* KARPGPU_H */
