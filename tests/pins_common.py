"""Known answers the reference's own tests hold for the catalog (List() miss path) and for Solve over it, restated as data
and checked on BOTH the oracle and the C ABI (kg_catalog_build_ec2 / kg_solve).  Sources, all under
pkg/providers/instancetype/suite_test.go unless another file is named:

  :603-690   pod-eni / PrivateIPv4Address land on a compatible type; a type absent from the VPC limits is never chosen
  :790-851   trn1.2xlarge neuron pod and dl1.24xlarge EFA pods: one node each
  :1020-1036 a pod selecting the local zone schedules
  :1053-1160 system-reserved / kube-reserved defaults and overrides on m5.xlarge
  :1165-1525 eviction thresholds (hard / soft / percent / 100% / max-of / Bottlerocket) with VMMemoryOverheadPercent = 0
  :1527-1810 max-pods / pods-per-core variants
  :1812-1898 predicted allocatable <= what kubelet reports on t4g.small / t4g.medium / t4g.xlarge / m5.large
  :2186-2279 ephemeral-storage defaults per AMI family when the block device mapping carries no volume size

The inputs come from the reference's fixtures (tests/golden/fake_ec2_16.json = pkg/fake/zz_generated.describe_instance_types.go
+ the offerings of pkg/fake/ec2api.go:555-674).  A backend is an object with
  catalog(infos, ctx, offering_zones) -> list of (name, capacity, allocatable) per type, dicts resource -> milli-units
  solve(infos, ctx, offering_zones, pods, nodepools) -> (type names, decoded result)
Exact pod -> claim assignment stays PARITY UNPINNED: no reference test asserts it."""
import json
import os

from karpenter_amd import abi, workloads as wl

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MI, GI, KI = 1024 ** 2, 1024 ** 3, 1024
CPU, MEM, EPH, PODS = "cpu", "memory", "ephemeral-storage", "pods"
POD_ENI, PRIVATE_IPV4 = "vpc.amazonaws.com/pod-eni", "vpc.amazonaws.com/PrivateIPv4Address"
NEURON, EFA = "aws.amazon.com/neuron", "vpc.amazonaws.com/efa"
RESOURCES = [CPU, MEM, EPH, PODS, POD_ENI, PRIVATE_IPV4, NEURON, EFA, "nvidia.com/gpu", "amd.com/gpu", "habana.ai/gaudi"]


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fake16(**ctx_kw):
    """(infos, ctx, offering_zones) of the provider's default test environment: AL2 node class over subnets 1a/1b/1c
    (suite_test.go:130-143), spot price = on-demand / 2."""
    f = _load("fake_ec2_16.json")
    od = _load("pricing_od.json")["on_demand"]
    zones = [s["zone"] for s in f["subnets"][:3]]
    infos = [i for i in f["instance_types"] if i["InstanceType"] in f["offering_zones"]]
    ctx = {"ami_family": "AL2", "region": "us-east-1", "vm_memory_overhead_percent": 0.075, "reserved_enis": 0,
           "subnets": f["subnets"][:3], "od_prices": od,
           "spot_prices": {(i["InstanceType"], z): round(od.get(i["InstanceType"], 1.0) * 0.5, 3) for i in infos for z in zones},
           "unavailable": set(), "limits": _load("vpclimits.json"), "bandwidth": _load("bandwidth.json")}
    ctx.update(ctx_kw)
    return infos, ctx, f["offering_zones"]


def _one(backend, name, **ctx_kw):
    infos, ctx, oz = fake16(**ctx_kw)
    for n, cap, alloc in backend.catalog(infos, ctx, oz):
        if n == name:
            return cap, alloc
    raise KeyError(name)


def _overhead(backend, name, **ctx_kw):
    cap, alloc = _one(backend, name, **ctx_kw)
    return cap, {r: cap[r] - alloc[r] for r in (CPU, MEM, EPH)}


RESERVED_30GI = dict(system_reserved={MEM: "20Gi"}, kube_reserved={MEM: "10Gi"})


def check_reserved_resources(B):
    """:1053-1160.  Overhead = kubeReserved + systemReserved + evictionThreshold; the defaults of each part are pinned by
    their own case (system 0/0/0, kube 80m/893Mi/1Gi, eviction 0/100Mi/10% of the 20Gi volume), so the totals are exact."""
    cap, ov = _overhead(B, "m5.xlarge")
    assert cap[CPU] == 4000 and cap[EPH] == 20 * GI * 1000
    assert ov == {CPU: 80, MEM: (893 + 100) * MI * 1000, EPH: (1 + 2) * GI * 1000}
    # system-reserved override 2 cpu / 20Gi / 10Gi (:1067-1093) on top of the kube defaults
    _, ov = _overhead(B, "m5.xlarge", system_reserved={CPU: "2", MEM: "20Gi", EPH: "10Gi"})
    assert ov == {CPU: 2000 + 80, MEM: (20 * GI + (893 + 100) * MI) * 1000, EPH: (10 + 1 + 2) * GI * 1000}
    # kube-reserved override 2 cpu / 10Gi / 2Gi (:1125-1158) with system-reserved memory 20Gi
    _, ov = _overhead(B, "m5.xlarge", system_reserved={MEM: "20Gi"}, kube_reserved={CPU: "2", MEM: "10Gi", EPH: "2Gi"})
    assert ov == {CPU: 2000, MEM: (30 * GI + 100 * MI) * 1000, EPH: (2 + 2) * GI * 1000}


def check_eviction_thresholds(B):
    """:1165-1525, all with VMMemoryOverheadPercent = 0 on m5.xlarge (16384 MiB), system 20Gi + kube 10Gi memory reserved:
    the eviction part of the overhead is what is left of the memory overhead after those 30Gi."""
    def evict_mem(**kw):
        cap, ov = _overhead(B, "m5.xlarge", vm_memory_overhead_percent=0.0, **dict(RESERVED_30GI, **kw))
        assert cap[MEM] == 16384 * MI * 1000
        return cap[MEM] // 1000, ov[MEM] // 1000 - 30 * GI

    capm = 16384 * MI
    assert evict_mem(eviction_hard={"memory.available": "500Mi"})[1] == 500 * MI                      # :1165-1193
    assert abs(evict_mem(eviction_hard={"memory.available": "10%"})[1] - capm * 0.1) <= 10             # :1194-1222
    assert evict_mem(eviction_hard={"memory.available": "100%"})[1] == 0                               # :1223-1251
    assert evict_mem(eviction_soft={"memory.available": "50Mi"})[1] == 50 * MI                         # :1252-1280
    assert evict_mem(eviction_soft={"memory.available": "500Mi"})[1] == 500 * MI                       # :1283-1311
    assert abs(evict_mem(eviction_hard={"memory.available": "5%"}, eviction_soft={"memory.available": "10%"})[1] - capm * 0.1) <= 10  # :1312-1343
    assert evict_mem(eviction_soft={"memory.available": "100%"})[1] == 0                               # :1344-1372
    # Bottlerocket ignores evictionSoft (:1373-1405)
    assert evict_mem(ami_family="Bottlerocket", eviction_hard={"memory.available": "1Gi"}, eviction_soft={"memory.available": "10Gi"})[1] == 1 * GI
    assert evict_mem(eviction_soft={"memory.available": "3Gi"}, eviction_hard={"memory.available": "1Gi"})[1] == 3 * GI     # :1428-1459
    assert abs(evict_mem(eviction_soft={"memory.available": "2%"}, eviction_hard={"memory.available": "5%"})[1] - capm * 0.05) <= 10  # :1460-1491
    assert abs(evict_mem(eviction_soft={"memory.available": "10%"}, eviction_hard={"memory.available": "1Gi"})[1] - capm * 0.1) <= 10  # :1492-1523
    # :1407-1427 no kubelet configuration at all: cpu 0, memory 100Mi, storage ~2Gi (and kube-reserved 80m on 4 vCPUs)
    cap, ov = _overhead(B, "m5.xlarge", vm_memory_overhead_percent=0.0)
    assert ov[CPU] == 80 and ov[MEM] == (893 + 100) * MI * 1000 and ov[EPH] == 3 * GI * 1000


def check_pod_density(B):
    """:1527-1810."""
    def pods(name, **kw):
        return _one(B, name, **kw)[0][PODS]

    assert pods("t3.large") == 35000 and pods("m6idn.32xlarge") == 394000                              # :1525-1567 ENI-limited defaults
    infos = fake16()[0]
    vcpus = {i["InstanceType"]: i["VCpuInfo"]["DefaultVCpus"] for i in infos}
    eni = {n: c[PODS] // 1000 for n, c, _ in B.catalog(*_args())}
    for n, cap, _ in B.catalog(*_args(max_pods=10)):                                                    # :1568-1617 every type
        assert cap[PODS] == 10000, n
    for n, cap, _ in B.catalog(*_args(pods_per_core=1)):                                                # :1689-1713 == vCPUs, every type
        assert cap[PODS] == vcpus[n] * 1000, n
    for n, cap, _ in B.catalog(*_args(pods_per_core=4, max_pods=20)):                                   # :1714-1739
        assert cap[PODS] == min(20, 4 * vcpus[n]) * 1000, n
    for n, cap, _ in B.catalog(*_args(ami_family="Bottlerocket", pods_per_core=1)):                     # :1740-1766 Bottlerocket ignores podsPerCore
        assert cap[PODS] == eni[n] * 1000, n
    assert pods("t3.large", pods_per_core=0) == 35000 and pods("m6idn.32xlarge", pods_per_core=0) == 394000  # :1767-1810
    assert pods("t3.large", reserved_enis=1) == 24000 and pods("t3.large", reserved_enis=1_000_000) == 0     # :1618-1688


def _args(**kw):
    return fake16(**kw)


def check_reported_allocatable(B):
    """:1812-1898: the prediction never exceeds what kubelet reports on a launched node."""
    reported = {"t4g.small": (1408312 * KI, 1930), "t4g.medium": (3377496 * KI, 1930), "t4g.xlarge": (15136012 * KI, 3920), "m5.large": (7220184 * KI, 1930)}
    seen = 0
    for n, _, alloc in B.catalog(*_args()):
        if n in reported:
            assert alloc[MEM] <= reported[n][0] * 1000, (n, alloc[MEM])
            assert alloc[CPU] <= reported[n][1], (n, alloc[CPU])
            seen += 1
    assert seen >= 1
    # t4g.* have no offering in the default environment: the reference installs its own offerings for this case
    infos = [i for i in _load("fake_ec2_16.json")["instance_types"] if i["InstanceType"] in reported]
    _, ctx, _ = fake16(subnets=[{"zone": "us-west-2a", "zone_id": None}])
    got = {n: a for n, _, a in B.catalog(infos, ctx, {n: ["us-west-2a"] for n in reported})}
    assert set(got) == set(reported)
    for n, (mem, cpu) in reported.items():
        assert got[n][MEM] <= mem * 1000 and got[n][CPU] <= cpu, (n, got[n])
        assert got[n][MEM] > 0.9 * mem * 1000 and got[n][CPU] >= cpu - 10     # ... and is not a trivial lower bound


def check_ephemeral_storage_defaults(B):
    """:2186-2279: one block device mapping (a snapshot, no volumeSize) on the family's device: 20Gi for every family."""
    for fam in ("Custom", "AL2", "AL2023", "Bottlerocket", "Ubuntu"):
        for n, cap, _ in B.catalog(*_args(ami_family=fam, block_device_mappings=[{"device_name": "ignored", "volume_size": None}])):
            assert cap[EPH] == 20 * GI * 1000, (fam, n)


def _pod(i, **requests):
    return {"uid": "pod-%02d" % i, "ns": "default", "creation_ts": 1700000000, "labels": {}, "requests": requests}


def check_special_resources(B):
    """:603-690, :790-851, :1020-1036 through Solve."""
    limits = _load("vpclimits.json")
    od = [wl.default_nodepool(capacity_types=("on-demand",))]
    # pod-eni: the node that comes up (the cheapest option) and every other option is trunking compatible
    names, res = B.solve(*_args(), [_pod(0, **{POD_ENI: 1000})], od)
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM
    assert res["claims"][0]["types"] and all(limits[names[t]]["trunking"] for t in res["claims"][0]["types"])
    # PrivateIPv4Address on Windows 2022
    names, res = B.solve(*_args(ami_family="Windows2022"), [_pod(0, **{PRIVATE_IPV4: 1000})], od)
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM
    assert all(limits[names[t]]["ipv4_per_interface"] != 0 for t in res["claims"][0]["types"])
    # a type the VPC resource controller does not know ("test", :636-690) is never launched for PrivateIPv4Address
    infos, ctx, _ = _args(ami_family="Windows2022")
    test_type = {"InstanceType": "test", "ProcessorInfo": {"SupportedArchitectures": ["x86_64"]}, "VCpuInfo": {"DefaultCores": 1, "DefaultVCpus": 2},
                 "MemoryInfo": {"SizeInMiB": 8192}, "NetworkInfo": {"Ipv4AddressesPerInterface": 10, "DefaultNetworkCardIndex": 0,
                                                                     "NetworkCards": [{"NetworkCardIndex": 0, "MaximumNetworkInterfaces": 3}]},
                 "SupportedUsageClasses": ["on-demand", "spot"]}
    ctx["od_prices"] = dict(ctx["od_prices"], test=0.1)
    pool = [wl.default_nodepool(capacity_types=("on-demand",), extra_reqs=[(wl.L_IT, "In", ["test"])])]
    names, res = B.solve([test_type], ctx, {"test": ["test-zone-1a"]}, [_pod(0, **{PRIVATE_IPV4: 1000})], pool)
    assert res["pods"][0][0] == abi.POD_UNSCHEDULABLE and not res["claims"]
    names, res = B.solve([test_type], ctx, {"test": ["test-zone-1a"]}, [_pod(0, cpu=100)], pool)   # ... the type itself is launchable
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM and [names[t] for t in res["claims"][0]["types"]] == ["test"]
    # :790-820 trn1.2xlarge: one neuron
    p = _pod(0, **{NEURON: 1000})
    p["node_selector"] = {wl.L_IT: "trn1.2xlarge"}
    names, res = B.solve(*_args(), [p], od)
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM and [names[t] for t in res["claims"][0]["types"]] == ["trn1.2xlarge"]
    # :821-851 dl1.24xlarge: EFA requests 1 and 2 share one node
    pods = []
    for i, c in enumerate((1, 2)):
        p = _pod(i, **{EFA: c * 1000})
        p["node_selector"] = {wl.L_IT: "dl1.24xlarge"}
        pods.append(p)
    names, res = B.solve(*_args(), pods, od)
    assert [k for k, _ in res["pods"]] == [abi.POD_NEW_CLAIM] * 2 and len({i for _, i in res["pods"]}) == 1
    # :1020-1036 local zone: the node class selects all four subnets, the pod's affinity names the local zone
    f = _load("fake_ec2_16.json")
    p = _pod(0, cpu=100)
    p["node_selector"] = {wl.L_ZONE: "test-zone-1a-local"}
    names, res = B.solve(*_args(subnets=f["subnets"]), [p], od)
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM
    assert [names[t] for t in res["claims"][0]["types"]] == ["m5.large"]     # the only type offered there (ec2api.go:555-674)
    names, res = B.solve(*_args(), [p], od)                                  # without the local subnet nothing is available there
    assert res["pods"][0][0] == abi.POD_UNSCHEDULABLE


def _all_checks():
    return [check_reserved_resources, check_eviction_thresholds, check_pod_density, check_reported_allocatable,
            check_ephemeral_storage_defaults, check_special_resources, check_consolidation_e2e]


N_CHECKS = 7


class OracleBackend:
    """oracle/catalog_oracle.py + oracle/oracle.cpp (Solve)."""

    @staticmethod
    def _types(infos, ctx, oz):
        from oracle import catalog_oracle as co
        return co.list_instance_types(infos, ctx, ctx["limits"], ctx["bandwidth"], oz)

    @classmethod
    def catalog(cls, infos, ctx, oz):
        from oracle import catalog_oracle as co
        out = []
        for t in cls._types(infos, ctx, oz):
            alloc = co.allocatable(t)
            out.append((t["name"], {r: t["capacity"].get(r, 0) for r in RESOURCES}, {r: alloc.get(r, 0) for r in RESOURCES}))
        return out

    @staticmethod
    def types776(zones):
        from oracle import loader
        return loader.catalog_types(wl.ec2_facts_776(), wl.ec2_context(zones))

    @staticmethod
    def solve_types(types, pods, nodepools):
        from oracle import loader
        st, rb = loader.solve([abi.build_catalog(types)], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods))
        assert st == 0
        return rb.decode()

    @staticmethod
    def decide(types, nodes, bound, nodepools, order, multi, spot_to_spot):
        from oracle import loader
        st, dec = loader.consolidate_decide([abi.build_catalog(types)], abi.build_cluster(nodes, bound), abi.build_pods([]),
                                            abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS), abi.build_candidates(order, spot_to_spot=spot_to_spot), multi)
        assert st == 0
        return dec

    @classmethod
    def solve(cls, infos, ctx, oz, pods, nodepools):
        from oracle import loader
        types = cls._types(infos, ctx, oz)
        st, rb = loader.solve([abi.build_catalog(types)], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods))
        assert st == 0
        return [t["name"] for t in types], rb.decode()


class LibraryBackend:
    """The C ABI: kg_catalog_build_ec2 + kg_catalog_query + kg_solve on the catalog the library built itself."""

    def __init__(self, L):
        self.L = L

    types776 = staticmethod(lambda zones: OracleBackend.types776(zones))   # the launch side of the replay reads prices from these dicts

    def solve_types(self, types, pods, nodepools):
        h = self.L.catalog_upload(abi.build_catalog(types))
        try:
            return self.L.solve([h], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods)).decode()
        finally:
            self.L.dll.kg_catalog_free(h)

    def decide(self, types, nodes, bound, nodepools, order, multi, spot_to_spot):
        hcat, hcl = self.L.catalog_upload(abi.build_catalog(types)), self.L.cluster_upload(abi.build_cluster(nodes, bound))
        try:
            recs, _ = self.L.consolidate_records([hcat], hcl, abi.build_pods([]), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS),
                                                 abi.build_candidates(order, spot_to_spot=spot_to_spot), multi)
            return self.L.consolidate_select(hcat, recs, len(order), multi)
        finally:
            self.L.dll.kg_cluster_free(hcl)
            self.L.dll.kg_catalog_free(hcat)

    def _build(self, infos, ctx, oz):
        return self.L.catalog_build_ec2(abi.build_ec2_catalog(infos, ctx, ctx["limits"], ctx["bandwidth"], oz))

    def catalog(self, infos, ctx, oz):
        h = self._build(infos, ctx, oz)
        try:
            out = []
            for i in range(self.L.catalog_size(h)[0]):
                name, cap, alloc = self.L.catalog_query(h, i, RESOURCES)
                out.append((name, cap, alloc))
            return out
        finally:
            self.L.dll.kg_catalog_free(h)

    def solve(self, infos, ctx, oz, pods, nodepools):
        h = self._build(infos, ctx, oz)
        try:
            names = [self.L.catalog_query(h, i, [CPU])[0] for i in range(self.L.catalog_size(h)[0])]
            rb = self.L.solve([h], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods))
            return names, rb.decode()
        finally:
            self.L.dll.kg_catalog_free(h)


# ---------------------------------------------------------------------------------------------------------------------
# test/suites/consolidation/suite_test.go: the replace scenarios, replayed as provision -> launch -> consolidate rounds
# ---------------------------------------------------------------------------------------------------------------------
BURSTABLE = ["t2", "t3", "c1", "t3a", "t4g", "a1"]
G = "karpenter.k8s.aws"


def _spread_pod(uid, app, cpu):
    return {"uid": uid, "ns": "default", "creation_ts": 1700000000, "labels": {"app": app}, "requests": {"cpu": cpu},
            "spreads": [{"key": wl.L_HOSTNAME, "max_skew": 1, "when": "DoNotSchedule", "selector": {"match_labels": {"app": app}}}]}


def _allowed(req, value):
    if req is None:
        return True
    complement, vals = req[0], req[1]
    return (value not in vals) if complement else (value in vals)


def _launch(types, type_ids, zone_req=None, ct_req=None):
    """What the fake CreateFleet does with a NodeClaim: the cheapest available offering over the claim's types."""
    best = None
    for t in type_ids:
        for z, ct, _zid, price, avail in types[t]["offerings"]:
            if avail and _allowed(zone_req, z) and _allowed(ct_req, ct) and (best is None or (price, types[t]["name"]) < best[0]):
                best = ((price, types[t]["name"]), t, z, ct)
    assert best is not None
    return best[1], best[2], best[3]


def _node(types, t, zone, ct, name, pods):
    ty = types[t]
    labels = {k: v[0] for k, op, v in ty["requirements"] if op == "In" and len(v) == 1}
    labels.update({wl.L_ZONE: zone, wl.L_CT: ct, wl.L_NODEPOOL: "default", wl.L_HOSTNAME: name})
    alloc = {k: v - ty["overhead"].get(k, 0) for k, v in ty["capacity"].items()}
    used = {"pods": 1000 * len(pods)}
    for p in pods:
        for k, v in p["requests"].items():
            used[k] = used.get(k, 0) + v
    return {"name": name, "labels": labels, "taints": [], "available": {k: v - used.get(k, 0) for k, v in alloc.items()}, "capacity": dict(ty["capacity"]),
            "daemon_requests": {}, "initialized": True, "managed": True, "deleting": False}


def _provision(B, types, pods, pool):
    """Solve + launch: -> [(type index, zone, capacity type, [pods])] per NodeClaim."""
    res = B.solve_types(types, pods, [pool])
    assert all(k == abi.POD_NEW_CLAIM for k, _ in res["pods"])
    out = []
    for ci, c in enumerate(res["claims"]):
        t, z, ct = _launch(types, c["types"], c["requirements"].get(wl.L_ZONE), c["requirements"].get(wl.L_CT))
        out.append((t, z, ct, [p for p, (_, i) in zip(pods, res["pods"]) if i == ci]))
    return out


def _consolidate_until_quiet(B, types, fleet, pool, spot_to_spot):
    """Disruption rounds (multi-node first, then single-node: the controller's method order) until neither finds a command.
    fleet: [(type, zone, ct, pods)].  Every command is applied the way the controller does: launch the replacement, move the pods."""
    commands = []
    for rnd in range(12):
        nodes = [_node(types, t, z, ct, "n%d-%d" % (rnd, i), pods) for i, (t, z, ct, pods) in enumerate(fleet)]
        bound = [dict(p, node=i, reschedulable=True) for i, (_, _, _, pods) in enumerate(fleet) for p in pods]
        order = wl.disruption_order(nodes, bound)
        dec = None
        for multi in (True, False):
            if multi and len(order) < 2:
                continue
            d = B.decide(types, nodes, bound, [pool], order, multi, spot_to_spot)
            if d["decision"] != abi.DECISION_NOOP:
                dec = d
                break
        if dec is None:
            return commands, fleet
        gone = order[:dec["n_candidates"]] if multi else [dec["first_candidate"]]
        moved = [p for i in gone for p in fleet[i][3]]
        fleet = [f for i, f in enumerate(fleet) if i not in gone]
        if dec["decision"] == abi.DECISION_REPLACE:
            t, z, ct = _launch(types, dec["types"], None, (False, ("spot",)) if dec["spot_only"] else None)
            fleet.append((t, z, ct, moved))
        else:
            raise AssertionError("a delete is not expected in these scenarios (the pods cannot share a node): %r" % (dec,))
        commands.append((dec["decision"], len(gone), dec["spot_only"]))
    raise AssertionError("consolidation did not settle: %r" % (commands,))


def check_consolidation_e2e(B):
    """test/suites/consolidation/suite_test.go:482-637 ("should consolidate nodes (replace)", on-demand and spot entries) and
    :639-768 ("should consolidate on-demand nodes to spot (replace)"), over the 776-type catalog with spot = on-demand / 2."""
    zones = wl.zones_n(3)
    types = B.types776(zones)
    size = lambda t: types[t]["name"].split(".")[1]   # noqa: E731
    for spot in (False, True):
        pool = wl.default_nodepool(capacity_types=("spot",) if spot else ("on-demand",),
                                   extra_reqs=[(G + "/instance-size", "In", ["large", "2xlarge"]), (G + "/instance-family", "NotIn", BURSTABLE),
                                               ("kubernetes.io/os", "In", ["linux"])])
        pods = [_spread_pod("large-%d" % i, "large-app", 4000) for i in range(3)] + [_spread_pod("small-%d" % i, "small-app", 1800) for i in range(3)]
        fleet = _provision(B, types, pods, pool)
        assert len(fleet) == 3 and all(size(t) == "2xlarge" for t, _, _, _ in fleet)                  # "3 nodes due to the anti-affinity rules"
        assert all(sorted(p["labels"]["app"] for p in ps) == ["large-app", "small-app"] for _, _, _, ps in fleet)
        fleet = [(t, z, ct, [p for p in ps if p["labels"]["app"] == "small-app"]) for t, z, ct, ps in fleet]   # largeDep scaled to 0
        cmds, fleet = _consolidate_until_quiet(B, types, fleet, pool, spot_to_spot=spot)
        # "we now must replace each node in turn": three single-node replacements, nothing else ...
        assert [c[:2] for c in cmds] == [(abi.DECISION_REPLACE, 1)] * 3, cmds
        # ... "all of the 2xlarge nodes should have been replaced with large instance types and we should have no other nodes"
        assert len(fleet) == 3 and all(size(t) == "large" and len(ps) == 1 for t, _, _, ps in fleet), fleet
    # on-demand -> spot
    od_pool = wl.default_nodepool(capacity_types=("on-demand",), extra_reqs=[(G + "/instance-size", "In", ["large"]), (G + "/instance-family", "NotIn", BURSTABLE)])
    fleet = _provision(B, types, [_spread_pod("small-%d" % i, "small-app", 1800) for i in range(2)], od_pool)
    assert len(fleet) == 2 and all(ct == "on-demand" and size(t) == "large" for t, _, ct, _ in fleet)
    any_pool = dict(wl.default_nodepool(extra_reqs=[(G + "/instance-size", "In", ["large"])]))
    any_pool["requirements"] = [(wl.L_CT, "Exists", [])] + any_pool["requirements"][1:]
    cmds, fleet = _consolidate_until_quiet(B, types, fleet, any_pool, spot_to_spot=False)
    assert cmds == [(abi.DECISION_REPLACE, 1, 1)] * 2, cmds                                              # each on-demand node -> a spot node
    assert len(fleet) == 2 and all(ct == "spot" and size(t) == "large" for t, _, ct, _ in fleet), fleet

