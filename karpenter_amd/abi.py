"""ctypes / numpy mirror of include/karpgpu.h.

Element arrays are numpy structured arrays whose dtypes match the C struct layouts
(``align=True``); top-level descriptors (which hold pointers) are ctypes Structures.
``DescBuilder`` turns plain Python problem descriptions into descriptors.  This is the host
side of the boundary a Go shim would implement with cgo; it carries no
scheduling logic.
"""
import ctypes as C

import numpy as np

KG_NONE = 0xFFFFFFFF
OP_IN, OP_NOT_IN, OP_EXISTS, OP_DOES_NOT_EXIST, OP_GT, OP_LT = range(6)
OPS = {"In": OP_IN, "NotIn": OP_NOT_IN, "Exists": OP_EXISTS, "DoesNotExist": OP_DOES_NOT_EXIST, "Gt": OP_GT, "Lt": OP_LT}
EFFECTS = {None: 0, "": 0, "NoSchedule": 1, "PreferNoSchedule": 2, "NoExecute": 3}
TOL_OPS = {"Equal": 0, "": 0, None: 0, "Exists": 1}
POD_UNSCHEDULABLE, POD_EXISTING_NODE, POD_NEW_CLAIM = 0, 1, 2
DECISION_NOOP, DECISION_DELETE, DECISION_REPLACE = 0, 1, 2
STATUS = {0: "KG_OK", 1: "KG_EINVAL", 2: "KG_ENOMEM", 3: "KG_EDEVICE", 4: "KG_EUNSUPPORTED", 5: "KG_ESTALE"}

RANGE = np.dtype([("begin", "<u4"), ("count", "<u4")], align=True)
REQUIREMENT = np.dtype([("key", "<u4"), ("op", "u1"), ("min_values", "<i4"), ("values", RANGE)], align=True)
QUANTITY = np.dtype([("name", "<u4"), ("milli", "<i8")], align=True)
LABEL = np.dtype([("key", "<u4"), ("value", "<u4")], align=True)
TAINT = np.dtype([("key", "<u4"), ("value", "<u4"), ("effect", "u1")], align=True)
TOLERATION = np.dtype([("key", "<u4"), ("value", "<u4"), ("op", "u1"), ("effect", "u1")], align=True)
SELECTOR = np.dtype([("is_nil", "u1"), ("match_labels", RANGE), ("match_exprs", RANGE)], align=True)
OFFERING = np.dtype([("zone", "<u4"), ("capacity_type", "<u4"), ("zone_id", "<u4"), ("price", "<f8"), ("available", "u1")], align=True)
INSTANCE_TYPE = np.dtype([("name", "<u4"), ("requirements", RANGE), ("capacity", RANGE), ("overhead", RANGE), ("offerings", RANGE)], align=True)
SPREAD = np.dtype([("topology_key", "<u4"), ("max_skew", "<i4"), ("min_domains", "<i4"), ("when_unsatisfiable", "u1"), ("selector", SELECTOR)], align=True)
AFFINITY_TERM = np.dtype([("topology_key", "<u4"), ("selector", SELECTOR), ("namespaces", RANGE), ("weight", "<i4")], align=True)
WEIGHTED_TERM = np.dtype([("weight", "<i4"), ("reqs", RANGE)], align=True)
HOST_PORT = np.dtype([("ip", "<u4"), ("port", "<i4"), ("protocol", "u1")], align=True)
POD = np.dtype([("uid", "<u4"), ("ns", "<u4"), ("creation_ts", "<i8"), ("labels", RANGE), ("requests", RANGE), ("node_selector", RANGE),
                ("required_terms", RANGE), ("preferred_terms", RANGE), ("tolerations", RANGE), ("spreads", RANGE),
                ("affinity_required", RANGE), ("affinity_preferred", RANGE), ("anti_affinity_required", RANGE),
                ("anti_affinity_preferred", RANGE), ("host_ports", RANGE), ("priority", "<i4"), ("deletion_cost", "<i4"),
                ("node", "<i4"), ("reschedulable", "u1"), ("volumes", RANGE)], align=True)
NODE = np.dtype([("name", "<u4"), ("labels", RANGE), ("taints", RANGE), ("available", RANGE), ("capacity", RANGE),
                 ("daemon_requests", RANGE), ("initialized", "u1"), ("managed", "u1"), ("deleting", "u1"), ("price", "<f8"),
                 ("volume_limits", RANGE)], align=True)
NODEPOOL = np.dtype([("name", "<u4"), ("weight", "<i4"), ("requirements", RANGE), ("labels", RANGE), ("taints", RANGE),
                     ("has_limits", "u1"), ("limits", RANGE), ("daemon_overhead", RANGE), ("catalog", "<u4"),
                     ("catalog_generation", "<u8")], align=True)
CLAIM_OUT = np.dtype([("nodepool", "<u4"), ("n_pods", "<u4"), ("types", RANGE), ("requirements", RANGE), ("requests", RANGE),
                      ("truncation_failed", "u1")], align=True)
REQ_OUT = np.dtype([("key", "<u4"), ("complement", "u1"), ("has_gt", "u1"), ("has_lt", "u1"), ("gt", "<i8"), ("lt", "<i8"),
                    ("values", RANGE), ("min_values", "<i4"), ("pad_", "<u4")], align=True)
CANDIDATE_RECORD = np.dtype([("candidate", "<i4"), ("decision", "<i4"), ("nodepool", "<u4"), ("n_types", "<u4"),
                             ("candidate_price", "<f8"), ("type_mask", "<u8", (16,)), ("zone_mask", "<u8"), ("ct_mask", "<u8"),
                             ("spot_only", "u1"), ("pad", "u1", (7,))], align=True)
EC2_DEVICE = np.dtype([("name", "<u4"), ("manufacturer", "<u4"), ("count", "<i8"), ("memory_mib", "<i8")], align=True)
EC2_TYPE = np.dtype([("name", "<u4"), ("arch", "<u4"), ("first_arch_is_arm64", "u1"), ("hypervisor", "<u4"), ("cpu_manufacturer", "<u4"),
                     ("vcpus", "<i8"), ("memory_mib", "<i8"), ("encryption_in_transit", "u1"), ("ebs_max_bandwidth_mbps", "<i8"),
                     ("network_bandwidth_mbps", "<i8"), ("local_nvme_gb", "<i8"), ("instance_storage_total_gb", "<i8"),
                     ("gpus", RANGE), ("accelerators", RANGE), ("efa_max_interfaces", "<i8"), ("default_card_max_interfaces", "<i8"),
                     ("ipv4_per_interface", "<i8"), ("has_limits", "u1"), ("trunking_compatible", "u1"), ("branch_interfaces", "<i8"),
                     ("limits_ipv4_per_interface", "<i8"), ("usage_classes", RANGE), ("zones_offered", RANGE)], align=True)
SUBNET = np.dtype([("zone", "<u4"), ("zone_id", "<u4")], align=True)
SPOT_PRICE = np.dtype([("instance_type", "<u4"), ("zone", "<u4"), ("price", "<f8")], align=True)
OD_PRICE = np.dtype([("instance_type", "<u4"), ("price", "<f8")], align=True)
UNAVAILABLE = np.dtype([("capacity_type", "<u4"), ("instance_type", "<u4"), ("zone", "<u4")], align=True)
OFFERING_UPDATE = np.dtype([("type_index", "<u4"), ("zone", "<u4"), ("capacity_type", "<u4"), ("available", "<i4"), ("price", "<f8")], align=True)

# name -> dtype, in the order kg_abi_sizes() reports them
ABI_ELEMENTS = [("kg_range", RANGE), ("kg_requirement", REQUIREMENT), ("kg_quantity", QUANTITY), ("kg_label", LABEL), ("kg_taint", TAINT),
                ("kg_toleration", TOLERATION), ("kg_selector", SELECTOR), ("kg_offering", OFFERING), ("kg_instance_type", INSTANCE_TYPE),
                ("kg_spread", SPREAD), ("kg_pod_affinity_term", AFFINITY_TERM), ("kg_weighted_term", WEIGHTED_TERM), ("kg_host_port", HOST_PORT),
                ("kg_pod", POD), ("kg_node", NODE), ("kg_nodepool", NODEPOOL), ("kg_claim_out", CLAIM_OUT), ("kg_req_out", REQ_OUT),
                ("kg_candidate_record", CANDIDATE_RECORD), ("kg_ec2_device", EC2_DEVICE), ("kg_ec2_type", EC2_TYPE), ("kg_subnet", SUBNET),
                ("kg_spot_price", SPOT_PRICE), ("kg_od_price", OD_PRICE), ("kg_unavailable", UNAVAILABLE), ("kg_offering_update", OFFERING_UPDATE)]

P = C.c_void_p


class Range(C.Structure):
    _fields_ = [("begin", C.c_uint32), ("count", C.c_uint32)]


class StrTab(C.Structure):
    _fields_ = [("data", P), ("offsets", P), ("count", C.c_uint32)]


class Pools(C.Structure):
    _fields_ = [("strings", StrTab),
                ("reqs", P), ("n_reqs", C.c_uint32),
                ("req_values", P), ("n_req_values", C.c_uint32),
                ("quantities", P), ("n_quantities", C.c_uint32),
                ("labels", P), ("n_labels", C.c_uint32),
                ("taints", P), ("n_taints", C.c_uint32),
                ("tolerations", P), ("n_tolerations", C.c_uint32),
                ("ranges", P), ("n_ranges", C.c_uint32)]


class CatalogDesc(C.Structure):
    _fields_ = [("pools", Pools), ("types", P), ("n_types", C.c_uint32), ("offerings", P), ("n_offerings", C.c_uint32)]


class Ec2CatalogDesc(C.Structure):
    _fields_ = [("pools", Pools), ("types", P), ("n_types", C.c_uint32), ("devices", P), ("n_devices", C.c_uint32),
                ("subnets", P), ("n_subnets", C.c_uint32), ("od_prices", P), ("n_od_prices", C.c_uint32),
                ("spot_prices", P), ("n_spot_prices", C.c_uint32), ("unavailable", P), ("n_unavailable", C.c_uint32),
                ("region", C.c_uint32), ("ami_family", C.c_uint8), ("vm_memory_overhead_percent", C.c_double),
                ("reserved_enis", C.c_int32), ("max_pods", C.c_int32), ("pods_per_core", C.c_int32),
                ("kube_reserved", Range), ("system_reserved", Range), ("eviction_hard", Range), ("eviction_soft", Range),
                ("has_eviction_hard", C.c_uint8), ("has_eviction_soft", C.c_uint8), ("instance_store_policy_raid0", C.c_uint8),
                ("root_volume_milli", C.c_int64), ("ephemeral_device_volume_milli", C.c_int64), ("last_bdm_volume_milli", C.c_int64),
                ("has_block_device_mappings", C.c_uint8)]


class PodsDesc(C.Structure):
    _fields_ = [("pools", Pools), ("pods", P), ("n_pods", C.c_uint32), ("weighted_terms", P), ("n_weighted_terms", C.c_uint32),
                ("spreads", P), ("n_spreads", C.c_uint32), ("affinity_terms", P), ("n_affinity_terms", C.c_uint32),
                ("host_ports", P), ("n_host_ports", C.c_uint32)]


class ClusterDesc(C.Structure):
    _fields_ = [("pods", PodsDesc), ("nodes", P), ("n_nodes", C.c_uint32)]


class NodePoolsDesc(C.Structure):
    _fields_ = [("pools", Pools), ("nodepools", P), ("n_nodepools", C.c_uint32), ("well_known_labels", Range),
                ("max_instance_types", C.c_int32)]


class Results(C.Structure):
    _fields_ = [("pod_kind", P), ("pod_index", P), ("pod_reason", P), ("cap_pods", C.c_uint32),
                ("claims", P), ("cap_claims", C.c_uint32), ("n_claims", C.c_uint32), ("needed_claims", C.c_uint32),
                ("type_ids", P), ("cap_type_ids", C.c_uint32), ("n_type_ids", C.c_uint32), ("needed_type_ids", C.c_uint32),
                ("reqs", P), ("cap_reqs", C.c_uint32), ("n_reqs", C.c_uint32), ("needed_reqs", C.c_uint32),
                ("value_ids", P), ("cap_value_ids", C.c_uint32), ("n_value_ids", C.c_uint32), ("needed_value_ids", C.c_uint32),
                ("quantities", P), ("cap_quantities", C.c_uint32), ("n_quantities", C.c_uint32), ("needed_quantities", C.c_uint32),
                ("str_data", P), ("cap_str_data", C.c_uint32), ("n_str_data", C.c_uint32), ("needed_str_data", C.c_uint32),
                ("str_offsets", P), ("cap_strs", C.c_uint32), ("n_strs", C.c_uint32), ("needed_strs", C.c_uint32),
                ("visited_existing", C.c_uint64), ("visited_claims", C.c_uint64), ("visited_claim_types", C.c_uint64),
                ("new_claim_scans", C.c_uint64), ("relaxations", C.c_uint64),
                ("algorithmic_bytes", C.c_double), ("device_ms", C.c_double), ("prepass_ms", C.c_double), ("solve_ms", C.c_double),
                ("host_compile_ms", C.c_double), ("host_decode_ms", C.c_double)]


class Decision(C.Structure):
    _fields_ = [("decision", C.c_int32), ("n_candidates", C.c_int32), ("first_candidate", C.c_int32), ("nodepool", C.c_uint32),
                ("n_types", C.c_uint32), ("type_ids", P), ("cap_type_ids", C.c_uint32), ("replacement_spot_only", C.c_uint8),
                ("candidate_price", C.c_double), ("simulations", C.c_uint64), ("device_ms", C.c_double)]


class CandidatesDesc(C.Structure):
    _fields_ = [("candidates", P), ("n_candidates", C.c_uint32), ("spot_to_spot_enabled", C.c_uint8),
                ("shard", C.c_int32), ("n_shards", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("solves", C.c_uint64), ("pods", C.c_uint64), ("kernel_launches", C.c_uint64), ("device_ms_total", C.c_double),
                ("host_compile_ms_total", C.c_double), ("device_bytes_resident", C.c_uint64), ("catalog_words_patched", C.c_uint64), ("wave_solves", C.c_uint64),
                ("cons_simulations", C.c_uint64), ("cons_algorithmic_bytes", C.c_double), ("cons_prepass_ms", C.c_double), ("cons_sim_init_ms", C.c_double),
                ("cons_simulate_ms", C.c_double), ("cons_gather_ms", C.c_double), ("cons_state_bytes_per_simulation", C.c_uint64),
                ("cons_plans_built", C.c_uint64), ("cons_plan_hits", C.c_uint64),
                ("spec_solves", C.c_uint64), ("spec_fast_commits", C.c_uint64), ("spec_challengers", C.c_uint64), ("spec_rescans", C.c_uint64),
                ("spec_fallbacks", C.c_uint64)]


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class _PoolBuilder:
    """Accumulates the flat pools of one descriptor."""

    def __init__(self):
        self.strs = {}
        self.str_list = []
        self.reqs, self.req_values, self.quantities = [], [], []
        self.labels, self.taints, self.tolerations, self.ranges = [], [], [], []
        self._keep = []

    def s(self, v):
        if v is None:
            return KG_NONE
        i = self.strs.get(v)
        if i is None:
            i = len(self.str_list)
            self.strs[v] = i
            self.str_list.append(v)
        return i

    def values(self, vals):
        b = len(self.req_values)
        self.req_values.extend(self.s(str(v)) for v in vals)
        return (b, len(self.req_values) - b)

    def req_list(self, reqs):
        """reqs: iterable of (key, op, values[, min_values])."""
        rows = []
        for r in reqs or ():
            key, op, vals = r[0], r[1], r[2]
            mv = r[3] if len(r) > 3 and r[3] is not None else -1
            rows.append((self.s(key), OPS[op] if isinstance(op, str) else op, mv, self.values(vals)))
        b = len(self.reqs)
        self.reqs.extend(rows)
        return (b, len(rows))

    def quantity_list(self, d):
        b = len(self.quantities)
        for k, v in (d or {}).items():
            self.quantities.append((self.s(k), int(v)))
        return (b, len(self.quantities) - b)

    def label_list(self, d):
        b = len(self.labels)
        for k, v in (d or {}).items():
            self.labels.append((self.s(k), self.s(str(v))))
        return (b, len(self.labels) - b)

    def pair_list(self, pairs):
        """(key, value) pairs, repeats allowed (a pod's (CSI driver, volume id) list)."""
        b = len(self.labels)
        for k, v in pairs or ():
            self.labels.append((self.s(k), self.s(str(v))))
        return (b, len(self.labels) - b)

    def taint_list(self, ts):
        b = len(self.taints)
        for k, v, e in ts or ():
            self.taints.append((self.s(k), self.s(v if v is not None else ""), EFFECTS[e] if not isinstance(e, int) else e))
        return (b, len(self.taints) - b)

    def toleration_list(self, ts):
        b = len(self.tolerations)
        for k, op, v, e in ts or ():
            self.tolerations.append((self.s(k), self.s(v) if v is not None else KG_NONE,
                                     TOL_OPS[op] if not isinstance(op, int) else op, EFFECTS[e] if not isinstance(e, int) else e))
        return (b, len(self.tolerations) - b)

    def range_list(self, rs):
        b = len(self.ranges)
        self.ranges.extend(rs)
        return (b, len(rs))

    def selector(self, sel):
        if sel is None:
            return (1, (0, 0), (0, 0))
        return (0, self.label_list(sel.get("match_labels")), self.req_list(sel.get("match_exprs")))

    def finish(self):
        blob = "".join(self.str_list).encode("utf-8")
        # offsets in BYTES
        offs = np.zeros(len(self.str_list) + 1, dtype="<u4")
        pos = 0
        for i, s in enumerate(self.str_list):
            pos += len(s.encode("utf-8"))
            offs[i + 1] = pos
        data = np.frombuffer(blob, dtype="u1").copy() if blob else np.zeros(1, dtype="u1")
        arrs = dict(
            reqs=np.array(self.reqs, dtype=REQUIREMENT) if self.reqs else np.zeros(0, REQUIREMENT),
            req_values=np.array(self.req_values, dtype="<u4"),
            quantities=np.array(self.quantities, dtype=QUANTITY) if self.quantities else np.zeros(0, QUANTITY),
            labels=np.array(self.labels, dtype=LABEL) if self.labels else np.zeros(0, LABEL),
            taints=np.array(self.taints, dtype=TAINT) if self.taints else np.zeros(0, TAINT),
            tolerations=np.array(self.tolerations, dtype=TOLERATION) if self.tolerations else np.zeros(0, TOLERATION),
            ranges=np.array(self.ranges, dtype=RANGE) if self.ranges else np.zeros(0, RANGE),
        )
        pools = Pools()
        pools.strings = StrTab(_ptr(data), _ptr(offs), len(self.str_list))
        for name, a in arrs.items():
            setattr(pools, name, _ptr(a))
            setattr(pools, "n_" + name, a.shape[0])
        self._keep = [data, offs, arrs]
        return pools


class Built:
    """A descriptor plus everything that must stay alive while it is used."""

    def __init__(self, desc, keep):
        self.desc = desc
        self.keep = keep

    @property
    def ref(self):
        return C.byref(self.desc)


def build_catalog(types):
    """types: list of dict(name, requirements, capacity, overhead, offerings=[(zone, ct, zone_id, price, available)])."""
    pb = _PoolBuilder()
    rows, offs = [], []
    for t in types:
        ob = len(offs)
        for z, ct, zid, price, avail in t["offerings"]:
            offs.append((pb.s(z), pb.s(ct), pb.s(zid), float(price), 1 if avail else 0))
        rows.append((pb.s(t["name"]), pb.req_list(t["requirements"]), pb.quantity_list(t["capacity"]),
                     pb.quantity_list(t["overhead"]), (ob, len(offs) - ob)))
    types_a = np.array(rows, dtype=INSTANCE_TYPE) if rows else np.zeros(0, INSTANCE_TYPE)
    offs_a = np.array(offs, dtype=OFFERING) if offs else np.zeros(0, OFFERING)
    d = CatalogDesc()
    d.pools = pb.finish()
    d.types, d.n_types = _ptr(types_a), len(rows)
    d.offerings, d.n_offerings = _ptr(offs_a), len(offs)
    return Built(d, [pb, types_a, offs_a])


def _fill_pods(pb, pods):
    rows, wterms, spreads, aterms, hports = [], [], [], [], []

    def terms(ts):
        b = len(aterms)
        for t in ts or ():
            aterms.append((pb.s(t["key"]), pb.selector(t.get("selector")), pb.values(t.get("namespaces") or ()), int(t.get("weight", 0))))
        return (b, len(aterms) - b)

    for p in pods:
        req_terms = pb.range_list([pb.req_list(term) for term in p.get("required_terms") or ()])
        wb = len(wterms)
        for w, term in p.get("preferred_terms") or ():
            wterms.append((int(w), pb.req_list(term)))
        sb = len(spreads)
        for s in p.get("spreads") or ():
            md = s.get("min_domains")
            spreads.append((pb.s(s["key"]), int(s.get("max_skew", 1)), -1 if md is None else int(md),
                            1 if s.get("when", "DoNotSchedule") == "ScheduleAnyway" else 0, pb.selector(s.get("selector"))))
        hb = len(hports)
        for ip, port, proto in p.get("host_ports") or ():
            hports.append((pb.s(ip), int(port), int(proto)))
        rows.append((pb.s(p["uid"]), pb.s(p.get("ns", "default")), int(p.get("creation_ts", 0)),
                     pb.label_list(p.get("labels")), pb.quantity_list(p.get("requests")), pb.label_list(p.get("node_selector")),
                     req_terms, (wb, len(wterms) - wb), pb.toleration_list(p.get("tolerations")), (sb, len(spreads) - sb),
                     terms(p.get("affinity_required")), terms(p.get("affinity_preferred")),
                     terms(p.get("anti_affinity_required")), terms(p.get("anti_affinity_preferred")),
                     (hb, len(hports) - hb), int(p.get("priority", 0)), int(p.get("deletion_cost", 0)),
                     int(p.get("node", -1)), 1 if p.get("reschedulable", True) else 0, pb.pair_list(p.get("volumes"))))
    arrs = dict(pods=np.array(rows, dtype=POD) if rows else np.zeros(0, POD),
                weighted_terms=np.array(wterms, dtype=WEIGHTED_TERM) if wterms else np.zeros(0, WEIGHTED_TERM),
                spreads=np.array(spreads, dtype=SPREAD) if spreads else np.zeros(0, SPREAD),
                affinity_terms=np.array(aterms, dtype=AFFINITY_TERM) if aterms else np.zeros(0, AFFINITY_TERM),
                host_ports=np.array(hports, dtype=HOST_PORT) if hports else np.zeros(0, HOST_PORT))
    return arrs


def _pods_desc(pb, arrs):
    d = PodsDesc()
    for name, a in arrs.items():
        setattr(d, name, _ptr(a))
        setattr(d, "n_" + name, a.shape[0])
    d.pools = pb.finish()
    return d


def build_pods(pods):
    pb = _PoolBuilder()
    arrs = _fill_pods(pb, pods)
    return Built(_pods_desc(pb, arrs), [pb, arrs])


def build_cluster(nodes, bound_pods):
    pb = _PoolBuilder()
    arrs = _fill_pods(pb, bound_pods)
    rows = []
    for n in nodes:
        rows.append((pb.s(n["name"]), pb.label_list(n.get("labels")), pb.taint_list(n.get("taints")),
                     pb.quantity_list(n.get("available")), pb.quantity_list(n.get("capacity")),
                     pb.quantity_list(n.get("daemon_requests")), 1 if n.get("initialized", True) else 0,
                     1 if n.get("managed", True) else 0, 1 if n.get("deleting", False) else 0, 0.0,
                     pb.quantity_list({k: int(v) * 1000 for k, v in (n.get("volume_limits") or {}).items()})))
    nodes_a = np.array(rows, dtype=NODE) if rows else np.zeros(0, NODE)
    d = ClusterDesc()
    d.pods = _pods_desc(pb, arrs)
    d.nodes, d.n_nodes = _ptr(nodes_a), len(rows)
    return Built(d, [pb, arrs, nodes_a])


def build_nodepools(nodepools, well_known, max_instance_types=60):
    pb = _PoolBuilder()
    rows = []
    for np_ in nodepools:
        lim = np_.get("limits")
        rows.append((pb.s(np_["name"]), int(np_.get("weight", 0)), pb.req_list(np_.get("requirements")),
                     pb.label_list(np_.get("labels")), pb.taint_list(np_.get("taints")), 1 if lim else 0,
                     pb.quantity_list(lim), pb.quantity_list(np_.get("daemon_overhead")), int(np_.get("catalog", 0)),
                     int(np_.get("catalog_generation", 0))))
    wk = pb.values(well_known)
    a = np.array(rows, dtype=NODEPOOL) if rows else np.zeros(0, NODEPOOL)
    d = NodePoolsDesc()
    d.nodepools, d.n_nodepools = _ptr(a), len(rows)
    d.well_known_labels = Range(*wk)
    d.max_instance_types = max_instance_types
    d.pools = pb.finish()
    return Built(d, [pb, a])


class ResultBuffers:
    """Caller-allocated output buffers for kg_solve / ko_solve, with decoding helpers."""

    def __init__(self, n_pods, cap_claims=None, cap_types=None):
        cap_claims = cap_claims if cap_claims is not None else max(n_pods, 1)
        cap_types = cap_types if cap_types is not None else max(cap_claims * 64, 1024)
        self.pod_kind = np.zeros(max(n_pods, 1), "u1")
        self.pod_index = np.zeros(max(n_pods, 1), "<i4")
        self.pod_reason = np.zeros(max(n_pods, 1), "<u4")
        self.claims = np.zeros(cap_claims, CLAIM_OUT)
        self.type_ids = np.zeros(cap_types, "<u4")
        self.reqs = np.zeros(cap_claims * 12 + 64, REQ_OUT)
        self.value_ids = np.zeros(cap_claims * 24 + 256, "<u4")
        self.quantities = np.zeros(cap_claims * 12 + 64, QUANTITY)
        self.str_offsets = np.zeros(cap_claims * 48 + 1024, "<u4")
        self.str_data = np.zeros(cap_claims * 48 * 24 + 65536, "u1")
        r = Results()
        r.pod_kind, r.pod_index, r.pod_reason, r.cap_pods = _ptr(self.pod_kind), _ptr(self.pod_index), _ptr(self.pod_reason), n_pods
        r.claims, r.cap_claims = _ptr(self.claims), cap_claims
        r.type_ids, r.cap_type_ids = _ptr(self.type_ids), cap_types
        r.reqs, r.cap_reqs = _ptr(self.reqs), self.reqs.shape[0]
        r.value_ids, r.cap_value_ids = _ptr(self.value_ids), self.value_ids.shape[0]
        r.quantities, r.cap_quantities = _ptr(self.quantities), self.quantities.shape[0]
        r.str_data, r.cap_str_data = _ptr(self.str_data), self.str_data.shape[0]
        r.str_offsets, r.cap_strs = _ptr(self.str_offsets), self.str_offsets.shape[0]
        self.res = r
        self.n_pods = n_pods

    def string(self, i):
        return bytes(self.str_data[self.str_offsets[i]:self.str_offsets[i + 1]]).decode("utf-8")

    def decode(self, with_requirements=True):
        """-> dict(pods=[(kind, index)], claims=[dict(nodepool, n_pods, types, requirements, requests, truncation_failed)])."""
        r = self.res
        out = {"pods": list(zip(self.pod_kind[:self.n_pods].tolist(), self.pod_index[:self.n_pods].tolist())), "claims": []}
        for c in self.claims[:r.n_claims]:
            tb, tc = int(c["types"]["begin"]), int(c["types"]["count"])
            claim = {"nodepool": int(c["nodepool"]), "n_pods": int(c["n_pods"]),
                     "types": self.type_ids[tb:tb + tc].tolist(), "truncation_failed": int(c["truncation_failed"])}
            qb, qc = int(c["requests"]["begin"]), int(c["requests"]["count"])
            claim["requests"] = {self.string(int(q["name"])): int(q["milli"]) for q in self.quantities[qb:qb + qc]}
            if with_requirements:
                rb, rc = int(c["requirements"]["begin"]), int(c["requirements"]["count"])
                reqs = {}
                for q in self.reqs[rb:rb + rc]:
                    vb, vc = int(q["values"]["begin"]), int(q["values"]["count"])
                    vals = tuple(sorted(self.string(int(v)) for v in self.value_ids[vb:vb + vc]))
                    reqs[self.string(int(q["key"]))] = (bool(q["complement"]), vals,
                                                        int(q["gt"]) if q["has_gt"] else None, int(q["lt"]) if q["has_lt"] else None)
                    if int(q["min_values"]) >= 0:
                        claim.setdefault("min_values", {})[self.string(int(q["key"]))] = int(q["min_values"])
                claim["requirements"] = reqs
            out["claims"].append(claim)
        return out


def build_offering_updates(updates):
    """updates: iterable of (type_index, zone, capacity_type, available or None, price or None) -> (Built(StrTab), array)."""
    pb = _PoolBuilder()
    rows = [(int(t), pb.s(z), pb.s(ct), -1 if av is None else int(bool(av)), -1.0 if pr is None else float(pr)) for t, z, ct, av, pr in updates]
    pools = pb.finish()
    a = np.array(rows, dtype=OFFERING_UPDATE) if rows else np.zeros(0, OFFERING_UPDATE)
    return Built(pools.strings, [pb, pools]), a


def build_candidates(candidates, spot_to_spot=False, shard=0, n_shards=1):
    a = np.array(list(candidates), dtype="<i4")
    d = CandidatesDesc()
    d.candidates, d.n_candidates = _ptr(a), a.shape[0]
    d.spot_to_spot_enabled = 1 if spot_to_spot else 0
    d.shard, d.n_shards = shard, n_shards
    return Built(d, [a])


class DecisionBuffer:
    def __init__(self, cap_types=1024):
        self.type_ids = np.zeros(cap_types, "<u4")
        self.dec = Decision()
        self.dec.type_ids, self.dec.cap_type_ids = _ptr(self.type_ids), cap_types

    def decode(self):
        d = self.dec
        return {"decision": d.decision, "n_candidates": d.n_candidates, "first_candidate": d.first_candidate, "nodepool": d.nodepool,
                "types": self.type_ids[:min(d.n_types, d.cap_type_ids)].tolist(), "spot_only": int(d.replacement_spot_only),
                "candidate_price": d.candidate_price, "simulations": int(d.simulations)}


def record_tuple(r):
    """The comparable part of a kg_candidate_record (zone_mask / ct_mask are library-internal)."""
    return (int(r["candidate"]), int(r["decision"]), int(r["nodepool"]), int(r["n_types"]), float(r["candidate_price"]),
            tuple(int(x) for x in r["type_mask"]), int(r["spot_only"]))


def build_launch(requirements, requests, strings=()):
    """A NodeClaim's launch inputs: (Pools, requirements range, requests range) for kg_launch_filter / kg_truncate /
    kg_launch_overrides; with `strings`, their ids in the pools' string table come back as a fourth element."""
    pb = _PoolBuilder()
    pb.s("nodeclaim")
    rr = pb.req_list(requirements)
    qq = pb.quantity_list(requests)
    ids = [pb.s(s) for s in strings]
    pools = pb.finish()
    if strings:
        return Built(pools, [pb]), Range(*rr), Range(*qq), ids
    return Built(pools, [pb]), Range(*rr), Range(*qq)


OVERRIDE = np.dtype([("type_id", "<u4"), ("subnet", "<u4")])


AMI_FAMILIES = {"AL2": 0, "AL2023": 1, "Bottlerocket": 2, "Ubuntu": 3, "Windows2019": 4, "Windows2022": 5, "Custom": 6}
AWS_TO_KUBE_ARCH = {"x86_64": "amd64", "arm64": "arm64"}


def build_ec2_catalog(infos, ctx, limits, bandwidth, offering_zones):
    """Raw EC2 facts (ec2.InstanceTypeInfo-shaped dicts) + provider state -> kg_ec2_catalog_desc for kg_catalog_build_ec2.
    Pure data marshalling: every derived number is computed by the library."""
    pb = _PoolBuilder()
    rows, devs = [], []

    def dev_range(items, mem_key):
        b = len(devs)
        for g in items:
            devs.append((pb.s(g["Name"]), pb.s(g["Manufacturer"]), int(g["Count"]), int((g.get(mem_key) or {}).get("SizeInMiB", 0)) if mem_key else 0))
        return (b, len(devs) - b)

    for info in infos:
        name = info["InstanceType"]
        archs = info["ProcessorInfo"]["SupportedArchitectures"]
        arch = next((AWS_TO_KUBE_ARCH[a] for a in archs if a in AWS_TO_KUBE_ARCH), "[" + " ".join(archs) + "]")
        ni = info.get("NetworkInfo") or {}
        cards = ni.get("NetworkCards") or [{"MaximumNetworkInterfaces": 0}]
        card = cards[ni.get("DefaultNetworkCardIndex", 0)]
        st = info.get("InstanceStorageInfo")
        ebs = info.get("EbsInfo")
        lim = limits.get(name)
        man = info["ProcessorInfo"].get("Manufacturer")
        efa = (ni.get("EfaInfo") or {}).get("MaximumEfaInterfaces") or 0
        rows.append((
            pb.s(name), pb.s(arch), 1 if archs and archs[0] == "arm64" else 0, pb.s(info.get("Hypervisor") or None) if info.get("Hypervisor") else KG_NONE,
            pb.s(man) if man is not None else KG_NONE, int(info["VCpuInfo"]["DefaultVCpus"]), int(info["MemoryInfo"]["SizeInMiB"]),
            1 if ni.get("EncryptionInTransitSupported") else 0,
            int(ebs["EbsOptimizedInfo"]["MaximumBandwidthInMbps"]) if ebs is not None and ebs.get("EbsOptimizedSupport") == "default" else -1,
            int(bandwidth[name]) if name in bandwidth else -1,
            int(st.get("TotalSizeInGB") or 0) if st is not None and st.get("NvmeSupport") != "unsupported" else -1,
            int(st["TotalSizeInGB"]) if st is not None and st.get("TotalSizeInGB") is not None else -1,
            dev_range((info.get("GpuInfo") or {}).get("Gpus") or [], "MemoryInfo"),
            dev_range((info.get("InferenceAcceleratorInfo") or {}).get("Accelerators") or [], None),
            int(efa), int(card["MaximumNetworkInterfaces"]), int(ni.get("Ipv4AddressesPerInterface", 0)),
            1 if lim else 0, 1 if lim and lim["trunking"] else 0, int(lim["branch_interface"]) if lim else 0, int(lim["ipv4_per_interface"]) if lim else 0,
            pb.values(info.get("SupportedUsageClasses") or []), pb.values(offering_zones.get(name, []))))
    types_a = np.array(rows, dtype=EC2_TYPE) if rows else np.zeros(0, EC2_TYPE)
    devs_a = np.array(devs, dtype=EC2_DEVICE) if devs else np.zeros(0, EC2_DEVICE)
    subnets_a = np.array([(pb.s(s["zone"]), pb.s(s.get("zone_id"))) for s in ctx["subnets"]], dtype=SUBNET) if ctx["subnets"] else np.zeros(0, SUBNET)
    od_a = np.array([(pb.s(k), float(v)) for k, v in ctx["od_prices"].items()], dtype=OD_PRICE) if ctx["od_prices"] else np.zeros(0, OD_PRICE)
    sp = ctx.get("spot_prices") or {}
    spot_a = np.array([(pb.s(k[0]), pb.s(k[1]), float(v)) for k, v in sp.items()], dtype=SPOT_PRICE) if sp else np.zeros(0, SPOT_PRICE)
    un = ctx.get("unavailable") or ()
    un_a = np.array([(pb.s(c), pb.s(i), pb.s(z)) for c, i, z in un], dtype=UNAVAILABLE) if un else np.zeros(0, UNAVAILABLE)
    d = Ec2CatalogDesc()
    d.types, d.n_types = _ptr(types_a), len(rows)
    d.devices, d.n_devices = _ptr(devs_a), len(devs)
    d.subnets, d.n_subnets = _ptr(subnets_a), subnets_a.shape[0]
    d.od_prices, d.n_od_prices = _ptr(od_a), od_a.shape[0]
    d.spot_prices, d.n_spot_prices = _ptr(spot_a), spot_a.shape[0]
    d.unavailable, d.n_unavailable = _ptr(un_a), un_a.shape[0]
    d.region = pb.s(ctx.get("region", "us-east-1"))
    d.ami_family = AMI_FAMILIES[ctx.get("ami_family", "AL2023")]
    d.vm_memory_overhead_percent = float(ctx.get("vm_memory_overhead_percent", 0.075))
    d.reserved_enis = int(ctx.get("reserved_enis", 0))
    d.max_pods = -1 if ctx.get("max_pods") is None else int(ctx["max_pods"])
    d.pods_per_core = -1 if ctx.get("pods_per_core") is None else int(ctx["pods_per_core"])
    parse_quantity = _parse_quantity
    d.kube_reserved = Range(*pb.quantity_list({k: parse_quantity(v) for k, v in (ctx.get("kube_reserved") or {}).items()}))
    d.system_reserved = Range(*pb.quantity_list({k: parse_quantity(v) for k, v in (ctx.get("system_reserved") or {}).items()}))
    d.eviction_hard = Range(*pb.label_list(ctx.get("eviction_hard") or {}))
    d.eviction_soft = Range(*pb.label_list(ctx.get("eviction_soft") or {}))
    d.has_eviction_hard = 1 if ctx.get("eviction_hard") is not None else 0
    d.has_eviction_soft = 1 if ctx.get("eviction_soft") is not None else 0
    d.instance_store_policy_raid0 = 1 if ctx.get("instance_store_policy") == "RAID0" else 0
    bdms = ctx.get("block_device_mappings") or []
    d.has_block_device_mappings = 1 if bdms else 0
    root = [b["volume_size"] for b in bdms if b.get("root_volume") and b.get("volume_size") is not None]
    d.root_volume_milli = int(root[0]) if root else -1
    d.last_bdm_volume_milli = int(bdms[-1]["volume_size"]) if bdms and bdms[-1].get("volume_size") is not None else -1
    d.ephemeral_device_volume_milli = int(ctx["ephemeral_device_volume_milli"]) if ctx.get("ephemeral_device_volume_milli") is not None else -1
    d.pools = pb.finish()
    return Built(d, [pb, types_a, devs_a, subnets_a, od_a, spot_a, un_a])


def _parse_quantity(v):
    """'100Mi' / '1' / '500m' -> int milli-units (kubelet reserved-resource strings); ints pass through as milli."""
    if isinstance(v, int):
        return v
    import re
    from fractions import Fraction
    m = re.fullmatch(r"([0-9.]+)(m|Ki|Mi|Gi|Ti|k|M|G|T|)", str(v).strip())
    if not m:
        raise ValueError("bad quantity %r" % (v,))
    mult = {"m": Fraction(1), "": Fraction(1000), "Ki": 1000 * 1024, "Mi": 1000 * 1024 ** 2, "Gi": 1000 * 1024 ** 3, "Ti": 1000 * 1024 ** 4,
            "k": 10 ** 6, "M": 10 ** 9, "G": 10 ** 12, "T": 10 ** 15}[m.group(2)]
    x = Fraction(m.group(1)) * mult
    return int(-(-x.numerator // x.denominator))
