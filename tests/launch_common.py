"""Launch-side selection (CloudProvider.Create): kg_launch_filter / kg_truncate vs the oracle, on the catalogs and the
cases of pkg/providers/instancetype/suite_test.go:390-602."""
from karpenter_amd import abi, workloads as wl
from oracle import loader
from tests.helpers import catalog776
from tests.test_oracle_pins import fake16_types, make_instances_types

G = "karpenter.k8s.aws"
CASES = [
    # (name, requirements, requests)
    ("od_only_1cpu", [(wl.L_CT, "In", ["on-demand"])], {"cpu": 1000, "pods": 1000}),
    ("mixed_1cpu", [(wl.L_CT, "In", ["on-demand", "spot"])], {"cpu": 1000, "pods": 1000}),
    ("spot_zone_a", [(wl.L_CT, "In", ["spot"]), (wl.L_ZONE, "In", ["us-east-1a"])], {"cpu": 4000, "memory": 8 * wl.GI * 1000, "pods": 3000}),
    ("family_notin", [(wl.L_CT, "In", ["on-demand", "spot"]), (G + "/instance-family", "NotIn", ["m5", "c5", "t3"]), (G + "/instance-generation", "Gt", ["4"])],
     {"cpu": 2000, "pods": 1000}),
    ("gpu_request", [(wl.L_CT, "In", ["on-demand"])], {"cpu": 1000, "nvidia.com/gpu": 1000, "pods": 1000}),
    ("metal_selected", [(wl.L_CT, "In", ["on-demand"]), (G + "/instance-size", "In", ["metal"])], {"cpu": 1000, "pods": 1000}),
    ("min_values_family", [(wl.L_CT, "In", ["spot"]), (G + "/instance-family", "Exists", [], 5)], {"cpu": 1000, "pods": 1000}),
    ("min_values_unsat", [(wl.L_CT, "In", ["spot"]), (G + "/instance-family", "In", ["m5", "c5"], 3)], {"cpu": 1000, "pods": 1000}),
    ("nothing_fits", [(wl.L_CT, "In", ["on-demand"])], {"cpu": 10_000_000, "pods": 1000}),
    ("arm64", [(wl.L_ARCH, "In", ["arm64"]), (wl.L_CT, "In", ["spot", "on-demand"])], {"cpu": 500, "memory": wl.GI * 1000, "pods": 2000}),
]


def check_launch(L):
    zones, types = catalog776()
    cat = abi.build_catalog(types)
    hcat = L.catalog_upload(cat)
    names = [t["name"] for t in types]
    seen_nonempty = 0
    try:
        for name, reqs, requests in CASES:
            exp = loader.launch_filter(cat, reqs, requests, wl.WELL_KNOWN_LABELS)
            got = L.launch_filter(hcat, reqs, requests)
            assert got == exp, (name, [names[t] for t in got[0]][:8], [names[t] for t in exp[0]][:8], got[1], exp[1])
            seen_nonempty += bool(exp[0])
            # Truncate alone, over the types the filter kept (any list works: take every third catalog type too)
            for ids in (exp[0][::-1], list(range(0, len(types), 3))):
                assert L.truncate(hcat, reqs, ids, 60) == loader.truncate(cat, reqs, ids, 60), name
        assert seen_nonempty >= 7
    finally:
        L.dll.kg_catalog_free(hcat)


def check_reference_pins(L):
    """instancetype/suite_test.go:390-434 (exactly 60 overrides, all within the 100 cheapest by (price, name)), :435-506
    (with spot and on-demand allowed every launched spot type is cheaper than the cheapest on-demand), :507-602 (metal and
    accelerator types are left out unless nothing else fits or they are selected explicitly)."""
    import json, os
    od = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pricing_od.json")))["on_demand"]
    types = make_instances_types()
    hcat = L.catalog_upload(abi.build_catalog(types))
    try:
        got, mv = L.launch_filter(hcat, [(wl.L_CT, "In", ["on-demand"])], {"cpu": 1000, "pods": 1000})
        ranked = sorted(od, key=lambda n: (od[n], n))
        # (MakeInstances gives every type the same shape, metal ones included: the exotic filter drops *.metal names)
        assert len(got) == 60 and not mv
        assert {types[t]["name"] for t in got} <= set(ranked[:100])
        assert [types[t]["name"] for t in got] == [n for n in ranked if not n.endswith("metal")][:60]
    finally:
        L.dll.kg_catalog_free(hcat)
    t16 = fake16_types()
    hcat = L.catalog_upload(abi.build_catalog(t16))
    try:
        names = [t["name"] for t in t16]
        got, _ = L.launch_filter(hcat, [(wl.L_CT, "In", ["on-demand"])], {"cpu": 1000, "pods": 1000})
        assert got and not any(names[t] in ("m5.metal", "p3.8xlarge", "g4dn.8xlarge", "inf1.2xlarge", "dl1.24xlarge") for t in got)
        got, _ = L.launch_filter(hcat, [(wl.L_CT, "In", ["on-demand"]), (wl.L_IT, "In", ["m5.metal"])], {"cpu": 1000, "pods": 1000})
        assert [names[t] for t in got] == ["m5.metal"]
        got, _ = L.launch_filter(hcat, [(wl.L_CT, "In", ["on-demand"])], {"nvidia.com/gpu": 1000, "pods": 1000})
        assert sorted(names[t] for t in got) == ["g4dn.8xlarge", "p3.8xlarge"]
        got, _ = L.launch_filter(hcat, [(wl.L_CT, "In", ["on-demand", "spot"])], {"cpu": 1000, "pods": 1000})
        cheapest_od = min(o[3] for t in got for o in t16[t]["offerings"] if o[4] and o[1] == "on-demand")
        assert all(min(o[3] for o in t16[t]["offerings"] if o[4]) <= cheapest_od for t in got)
    finally:
        L.dll.kg_catalog_free(hcat)


# ---------------------------------------------------------------------------------------------------------------------
# getCapacityType + getOverrides (instance.go:372-385, 323-359)
def overrides_by_hand(types, reqs, type_ids, subnet_zones):
    """The two functions stated directly for requirements that constrain only capacity type / zone with In."""
    want = {k: set(v) for k, op, v in reqs if op == "In" and k in (wl.L_CT, wl.L_ZONE)}

    def compatible(o, ct):
        return o[4] and o[1] == ct and (wl.L_ZONE not in want or o[0] in want[wl.L_ZONE])

    ct = "on-demand"
    if wl.L_CT not in want or "spot" in want[wl.L_CT]:
        if any(compatible(o, "spot") for t in type_ids for o in types[t]["offerings"]):
            ct = "spot"
    return ct, [(t, o[0]) for t in type_ids for o in types[t]["offerings"] if compatible(o, ct) and o[0] in subnet_zones]


def check_overrides(L):
    from tests.update_common import apply_updates, scenario_updates
    zones, types = catalog776()
    types = apply_updates(types, scenario_updates(types, zones))  # ICE'd offerings in the catalog: left out of the overrides
    cat = abi.build_catalog(types)
    hcat = L.catalog_upload(cat)
    names = [t["name"] for t in types]
    seen = set()
    try:
        for name, reqs, requests in CASES + [("zone_b_od", [(wl.L_CT, "In", ["on-demand"]), (wl.L_ZONE, "In", [zones[1]])], {"cpu": 1000, "pods": 1000}),
                                             ("no_ct_key", [(wl.L_ZONE, "In", [zones[0], zones[2]])], {"cpu": 1000, "pods": 1000}),
                                             ("ct_notin_od", [(wl.L_CT, "NotIn", ["on-demand"])], {"cpu": 1000, "pods": 1000}),
                                             ("zone_id", [(wl.L_CT, "In", ["spot", "on-demand"]), ("topology.k8s.aws/zone-id", "In", ["use1-az2"])], {"cpu": 1000, "pods": 1000})]:
            ids = loader.launch_filter(cat, reqs, requests, wl.WELL_KNOWN_LABELS)[0]
            for sub in (zones, zones[:1], [zones[2], zones[0]], [], ["eu-north-9z"] + zones[1:]):
                exp = loader.launch_overrides(cat, reqs, ids, sub, wl.WELL_KNOWN_LABELS)
                got = L.launch_overrides(hcat, reqs, ids, sub)
                assert got == exp, (name, sub, got[0], exp[0], [(names[t], z) for t, z in got[1]][:6], [(names[t], z) for t, z in exp[1]][:6])
                if all(op == "In" and k in (wl.L_CT, wl.L_ZONE) for k, op, *_ in reqs):
                    assert exp == overrides_by_hand(types, reqs, ids, sub), (name, sub)
                seen.add((exp[0], bool(exp[1])))
        # c6i.large has no available offering left: it contributes nothing, whatever is asked
        gone = names.index("c6i.large")
        assert L.launch_overrides(hcat, [(wl.L_CT, "In", ["spot", "on-demand"])], [gone], zones) == ("on-demand", [])
        assert seen >= {("spot", True), ("on-demand", True), ("on-demand", False)}
        # a buffer too small reports the size needed
        import ctypes as C
        import numpy as np
        pools, rr, _, zid = abi.build_launch([(wl.L_CT, "In", ["on-demand"])], {}, zones)
        ids = np.arange(8, dtype="<u4")
        zs = np.array(zid, dtype="<u4")
        out = np.zeros(2, abi.OVERRIDE)
        n, ct = C.c_uint32(0), C.c_int32(-1)
        st = L.dll.kg_launch_overrides(L.ctx, hcat, C.addressof(pools.desc), rr, ids.ctypes.data, 8, zs.ctypes.data, 3, C.byref(ct), out.ctypes.data, 2, C.byref(n))
        assert abi.STATUS[st] == "KG_ENOMEM" and n.value == len(L.launch_overrides(hcat, [(wl.L_CT, "In", ["on-demand"])], range(8), zones)[1])
    finally:
        L.dll.kg_catalog_free(hcat)


# ---------------------------------------------------------------------------------------------------------------------
# A provisioning attempt as the reference's suites drive it: Solve -> per NodeClaim CloudProvider.Create (launch filter,
# capacity type, overrides) -> the fake CreateFleet -> ICE marks -> the catalog the next attempt sees.
def claim_requirements(decoded, min_values=None):
    """A decoded NodeClaim's requirements (+ the minValues it carries) -> requirement tuples for the launch calls."""
    out = []
    for key, (complement, vals, gt, lt) in sorted(decoded.items()):
        mv = (min_values or {}).get(key, -1)
        if not complement:
            out.append((key, "In", list(vals), mv) if vals else (key, "DoesNotExist", [], mv))
            continue
        if vals:
            out.append((key, "NotIn", list(vals), mv))
        if gt is not None:
            out.append((key, "Gt", [str(gt)], mv))
        if lt is not None:
            out.append((key, "Lt", [str(lt)], mv))
        if not vals and gt is None and lt is None:
            out.append((key, "Exists", [], mv))
    return out


class OracleBackend:
    """Every call on the oracle, the catalog rebuilt from the type dicts (createOfferings + computeRequirements in Python)."""

    def __init__(self, types):
        self.types = types

    def solve(self, pods, nodepools):
        cat = abi.build_catalog(self.types)
        st, rb = loader.solve([cat], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods))
        assert st == 0
        return rb.decode()

    def launch_filter(self, reqs, requests):
        return loader.launch_filter(abi.build_catalog(self.types), reqs, requests, wl.WELL_KNOWN_LABELS)

    def launch_overrides(self, reqs, ids, zones):
        return loader.launch_overrides(abi.build_catalog(self.types), reqs, ids, zones, wl.WELL_KNOWN_LABELS)

    def mark_unavailable(self, marks):
        from tests.update_common import apply_updates
        self.types = apply_updates(self.types, [(t, z, ct, False, None) for t, z, ct in marks])

    def available(self, t):
        return [o for o in self.types[t]["offerings"] if o[4]]

    def close(self):
        pass


class LibraryBackend(OracleBackend):
    """Every call through the C ABI on ONE resident catalog, ICE marks applied in place by kg_catalog_update_offerings."""

    def __init__(self, L, types):
        self.L, self.types, self.gen = L, types, 1
        self.h = L.catalog_upload(abi.build_catalog(types))

    def solve(self, pods, nodepools):
        return self.L.solve([self.h], None, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods)).decode()

    def launch_filter(self, reqs, requests):
        return self.L.launch_filter(self.h, reqs, requests)

    def launch_overrides(self, reqs, ids, zones):
        return self.L.launch_overrides(self.h, reqs, ids, zones)

    def mark_unavailable(self, marks):
        OracleBackend.mark_unavailable(self, marks)  # (the Python copy: only for available())
        self.gen += 1
        self.L.catalog_update_offerings(self.h, [(t, z, ct, False, None) for t, z, ct in marks], generation=self.gen)

    def close(self):
        self.L.dll.kg_catalog_free(self.h)


def provision(B, pods, nodepools, ice_pools, subnet_zones):
    """One ExpectProvisioned: -> per pod the node it was bound to ({type, zone, capacity_type}) or None.  `ice_pools` =
    the fake's InsufficientCapacityPools {(type name, zone, capacity type)}.  The fake fleet (pkg/fake/ec2api.go:112-197)
    skips the overrides naming an ICE'd pool, reports them as errors (-> marked unavailable, instance.go:361-367), and, if
    any override is left, launches ONE instance that it reports with the FIRST override's type and zone."""
    names = [t["name"] for t in B.types]
    res = B.solve(pods, nodepools)
    nodes = []
    for claim in res["claims"]:
        reqs = claim_requirements(claim["requirements"], claim.get("min_values"))
        ids, mv_failed = B.launch_filter(reqs, claim["requests"])
        node = None
        if ids and not mv_failed:  # (instance.go:101-104: Truncate breaking minValues fails the launch)
            ct, ov = B.launch_overrides(reqs, ids, subnet_zones)
            skipped = [(t, z) for t, z in ov if (names[t], z, ct) in ice_pools]
            if skipped:
                B.mark_unavailable(sorted({(t, z, ct) for t, z in skipped}))
            if len(skipped) < len(ov):
                node = {"type": names[ov[0][0]], "zone": ov[0][1], "capacity_type": ct, "n_overrides": len(ov), "override_types": sorted({names[t] for t, _ in ov})}
        nodes.append(node)
    return [nodes[i] if k == abi.POD_NEW_CLAIM else None for k, i in res["pods"]], res


def check_ice_and_capacity_type(make_backend):
    """pkg/providers/instancetype/suite_test.go:1901-2114 (Insufficient Capacity Error Cache) and :2117-2184 (CapacityType),
    replayed on `make_backend(types)`."""
    from tests.test_oracle_pins import OD_POOL, pod
    zones3 = ["test-zone-1a", "test-zone-1b", "test-zone-1c"]
    base = fake16_types()

    def labels_of(B, node):
        t = [t for t in B.types if t["name"] == node["type"]][0]
        return {k: v[0] for k, op, v in t["requirements"] if op == "In" and len(v) == 1}

    # :1901-1933 two 1-neuron pods in 1a, inf1.6xlarge on-demand ICE'd there: first attempt fails, second lands on two
    # inferentia nodes
    B = make_backend(base)
    try:
        pods = [dict(pod(i, **{"aws.amazon.com/neuron": 1000}), node_selector={wl.L_ZONE: "test-zone-1a"}) for i in range(2)]
        ice = {("inf1.6xlarge", "test-zone-1a", "on-demand")}
        got, _ = provision(B, pods, OD_POOL, ice, zones3)
        assert got == [None, None]
        got, res = provision(B, pods, OD_POOL, ice, zones3)
        assert all(got) and res["pods"][0][1] != res["pods"][1][1]
        assert all(labels_of(B, n)["karpenter.k8s.aws/instance-accelerator-name"] == "inferentia" for n in got)
    finally:
        B.close()
    # :1934-1960 and :2014-2040 a type pinned by selector with a PREFERRED zone 1a, ICE'd there: second attempt lands in 1b
    for tname, resource in (("p3.8xlarge", "nvidia.com/gpu"), ("dl1.24xlarge", "habana.ai/gaudi")):
        B = make_backend(base)
        try:
            p = dict(pod(0, **{resource: 1000}), node_selector={wl.L_IT: tname}, preferred_terms=[(1, [(wl.L_ZONE, "In", ["test-zone-1a"])])])
            ice = {(tname, "test-zone-1a", "on-demand")}
            assert provision(B, [p], OD_POOL, ice, zones3)[0] == [None]
            node = provision(B, [p], OD_POOL, ice, zones3)[0][0]
            assert node and node["type"] == tname and node["zone"] == "test-zone-1b"
        finally:
            B.close()
    # :1961-1994 m5.xlarge ICE'd in 1a, NodePool limited to m5.large / m5.xlarge: two 1-CPU pods end up on two m5.large
    B = make_backend(base)
    try:
        pool = [wl.default_nodepool(capacity_types=("on-demand",), extra_reqs=[(wl.L_IT, "In", ["m5.large", "m5.xlarge"])])]
        pods = [dict(pod(i, cpu=1000), node_selector={wl.L_ZONE: "test-zone-1a"}) for i in range(2)]
        ice = {("m5.xlarge", "test-zone-1a", "on-demand")}
        assert provision(B, pods, pool, ice, zones3)[0] == [None, None]
        got, _ = provision(B, pods, pool, ice, zones3)
        assert [n and n["type"] for n in got] == ["m5.large", "m5.large"]
    finally:
        B.close()
    # :1995-2013 the shortage ends and the cache entry is deleted: the same pod schedules on the type again
    B = make_backend(base)
    try:
        p = dict(pod(0, **{"aws.amazon.com/neuron": 2000}), node_selector={wl.L_IT: "inf1.6xlarge"})
        assert provision(B, [p], OD_POOL, {("inf1.6xlarge", z, "on-demand") for z in zones3}, zones3)[0] == [None]
    finally:
        B.close()
    B = make_backend(base)  # (UnavailableOfferingsCache.Delete + empty pools = the catalog as it first was)
    try:
        node = provision(B, [p], OD_POOL, set(), zones3)[0][0]
        assert node and node["type"] == "inf1.6xlarge"
    finally:
        B.close()
    # :2041-2062 spot ICE'd for every type in 1a, NodePool flexible to both and pinned to 1a: spot is tried and fails, the
    # next attempt falls back to on-demand
    B = make_backend(base)
    try:
        pool = [wl.default_nodepool(capacity_types=("spot", "on-demand"), extra_reqs=[(wl.L_ZONE, "In", ["test-zone-1a"])])]
        ice = {(t["name"], "test-zone-1a", "spot") for t in base}
        assert provision(B, [pod(0, cpu=100)], pool, ice, zones3)[0] == [None]
        node = provision(B, [pod(0, cpu=100)], pool, ice, zones3)[0][0]
        assert node and node["capacity_type"] == "on-demand"
    finally:
        B.close()
    # :2063-2114 every offering of m5.xlarge in 1a / 1b ICE'd: the type is still listed, with no available offering
    B = make_backend(base)
    try:
        pool = [wl.default_nodepool(capacity_types=("spot", "on-demand"), extra_reqs=[(wl.L_IT, "In", ["m5.xlarge"])])]
        ice = {("m5.xlarge", z, ct) for z in zones3[:2] for ct in ("spot", "on-demand")}
        for ct in ("on-demand", "spot"):
            for z in zones3[:2]:
                p = dict(pod(0, cpu=1000), node_selector={wl.L_CT: ct, wl.L_ZONE: z})
                provision(B, [p], pool, ice, zones3[:2])
        t = [i for i, ty in enumerate(B.types) if ty["name"] == "m5.xlarge"][0]
        assert [o for o in B.available(t) if o[0] in zones3[:2]] == []
        if hasattr(B, "L"):
            rq = B.L.catalog_requirements(B.h, t)
            assert not set(rq[wl.L_ZONE]) & set(zones3[:2])
    finally:
        B.close()
    # :2117-2132 CapacityType: default NodePool launches on-demand; flexible to both launches spot
    B = make_backend(base)
    try:
        assert provision(B, [pod(0, cpu=100)], OD_POOL, set(), zones3)[0][0]["capacity_type"] == "on-demand"
        assert provision(B, [pod(0, cpu=100)], [wl.default_nodepool(capacity_types=("spot", "on-demand"))], set(), zones3)[0][0]["capacity_type"] == "spot"
    finally:
        B.close()
    # :2133-2184 spot price known for m5.large in 1a only: a spot NodePool pinned to 1b cannot launch; unpinned it can
    od = {t["name"]: 1.0 for t in base}
    spot_only_1a = fake16_types(spot_prices={("m5.large", "test-zone-1a"): 0.004})
    B = make_backend(spot_only_1a)
    try:
        pool = [wl.default_nodepool(capacity_types=("spot",), extra_reqs=[(wl.L_IT, "In", ["m5.large"]), (wl.L_ZONE, "In", ["test-zone-1b"])])]
        assert provision(B, [pod(0, cpu=100)], pool, set(), zones3)[0] == [None]
        pool = [wl.default_nodepool(capacity_types=("spot",), extra_reqs=[(wl.L_IT, "In", ["m5.large"])])]
        node = provision(B, [pod(0, cpu=100)], pool, set(), zones3)[0][0]
        assert node and node["zone"] == "test-zone-1a" and node["capacity_type"] == "spot" and node["n_overrides"] == 1
    finally:
        B.close()


def check_min_values_launches(make_backend):
    """pkg/cloudprovider/suite_test.go:285-590 (MinValues): two 0.9-CPU pods; the types are the first of 2 (3) distinct
    families of MakeInstances with 1 / 8 (1 / 4 / 8) vCPUs and spot prices 0.002 / 0.003 (/ 0.004) in test-zone-1a.  minValues
    keeps the 1-vCPU type among every NodeClaim's options, so the pods land on two nodes, and each CreateFleet request
    names at least the minimum number of distinct types (families)."""
    import json, os
    from tests.test_oracle_pins import pod
    od = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pricing_od.json")))["on_demand"]

    def unique_families(k):  # fake.MakeUniqueInstancesAndFamilies
        fams, names = set(), []
        for n in sorted(od):
            if n.split(".")[0] not in fams:
                fams.add(n.split(".")[0])
                names.append(n)
            if len(names) == k:
                return names

    def pool(reqs):
        return [{"name": "default", "weight": 0, "requirements": reqs, "labels": {}, "taints": [], "limits": None, "daemon_overhead": {}, "catalog": 0}]

    def run(names, vcpus, reqs):
        types = make_instances_types(names, vcpus=dict(zip(names, vcpus)), spot={(n, "test-zone-1a"): 0.002 + 0.001 * i for i, n in enumerate(names)})
        B = make_backend(types)
        try:
            got, res = provision(B, [pod(0, cpu=900), pod(1, cpu=900)], pool(reqs), set(), ["test-zone-1a"])
        finally:
            B.close()
        assert all(got) and res["pods"][0][1] != res["pods"][1][1] and len(res["claims"]) == 2
        return got

    two, three = unique_families(2), unique_families(3)
    # :285-385 In + minValues 2 on instance-type, spot only
    for n in run(two, (1, 8), [(wl.L_CT, "In", ["spot"]), (wl.L_IT, "In", two, 2)]):
        assert len(n["override_types"]) >= 2 and n["capacity_type"] == "spot"
    # :386-483 Exists minValues 2 and In minValues 1 on the same key (the larger one holds), no capacity-type requirement
    for n in run(two, (1, 8), [(wl.L_IT, "Exists", [], 2), (wl.L_IT, "In", two, 1)]):
        assert len(n["override_types"]) >= 2
    # :484-590 minValues on two keys: 2 instance types and 3 families -> all three types in every request
    for n in run(three, (1, 4, 8), [(wl.L_IT, "In", three, 2), ("karpenter.k8s.aws/instance-family", "In", [x.split(".")[0] for x in three], 3)]):
        assert n["override_types"] == sorted(three) and len({x.split(".")[0] for x in n["override_types"]}) == 3
