"""Every input the device path refuses (KG_EUNSUPPORTED: "not representable on the device path, the caller keeps its CPU
path", include/karpgpu.h) provoked once through the C ABI: the status, the message, and a normal Solve right after on the
same context (a refusal leaves nothing behind).  Shared by the emulation (CPU) and the GPU test files."""
import pytest

from karpenter_amd import abi, lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families

EUNSUPPORTED = 4


def _pod(uid, **kw):
    p = {"uid": uid, "ns": "default", "creation_ts": 1700000000, "labels": {"app": "x"}, "requests": {"cpu": 250, "memory": wl.GI * 1000}}
    p.update(kw)
    return p


def _spread(key, app="x", **kw):
    return dict({"key": key, "max_skew": 1, "when": "DoNotSchedule", "selector": {"match_labels": {"app": app}}}, **kw)


def _small_types(n, zones, complement=False, two_zone_ids=False):
    """n synthetic instance types (uploaded through kg_catalog_upload: no EC2 facts needed)."""
    out = []
    for i in range(n):
        name = "syn%04d.large" % i
        reqs = [(wl.L_IT, "In", [name]), (wl.L_ARCH, "In", ["amd64"]), ("kubernetes.io/os", "In", ["linux"]), (wl.L_ZONE, "In", list(zones)), (wl.L_CT, "In", ["on-demand"])]
        if complement and i == 0:
            reqs.append(("example.com/odd", "NotIn", ["a"]))
        offs = [(z, "on-demand", ("zid-%d" % (zi + (1 if two_zone_ids and i == 1 else 0))) if two_zone_ids else None, 0.1 + 0.001 * i, True) for zi, z in enumerate(zones)]
        out.append({"name": name, "requirements": reqs, "capacity": {"cpu": 2000, "memory": 8 * wl.GI * 1000, "pods": 29000},
                    "overhead": {"cpu": 70, "memory": 600 * wl.MI * 1000}, "offerings": offs})
    return out


def _expect(fn, needle):
    with pytest.raises(lib.KgError) as e:
        fn()
    assert e.value.status == EUNSUPPORTED, (needle, str(e.value))
    assert needle in str(e.value), (needle, str(e.value))


def check_refusals(L):
    zones, types = catalog776()
    fams = families(types)
    pool = [wl.default_nodepool()]
    hcat = L.catalog_upload(abi.build_catalog(types))
    healthy = Problem(types, wl.mixed_pods(120, 5, zones, fams), pool)
    want = healthy.run_oracle().decode()
    nodes, bound = wl.existing_cluster(12, 3, types, zones)
    keyless = [dict(n, labels={k: v for k, v in n["labels"].items() if k != wl.L_ZONE or i % 2}) for i, n in enumerate(nodes)]

    def solve(pods, nodepools=pool, nodes_=None, bound_=None, cats=None, noalloc=False):
        hcl = L.cluster_upload(abi.build_cluster(nodes_, bound_ or [])) if nodes_ is not None else None
        try:
            return L.solve(cats or [hcat], hcl, abi.build_pods(pods), abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, 60), len(pods), noalloc=noalloc)
        finally:
            if hcl is not None:
                L.dll.kg_cluster_free(hcl)

    def still_usable():
        d = diff_results(want, healthy.run_library(L, hcat).decode())
        assert d == "", d

    cases = []
    # --- Solve: vocabulary limits of the bitset rows
    cases.append(("more than 32 active label keys", lambda: solve([_pod("p%d" % i, node_selector={"example.com/k%02d" % i: "v"}) for i in range(33)])))
    cases.append(("requirement set with more than 8 keys", lambda: solve([_pod("p", node_selector={"example.com/k%d" % i: "v" for i in range(9)})])))
    cases.append(("more than 8 distinct resources", lambda: solve([_pod("p", requests={"example.com/r%d" % i: 1000 for i in range(9)})])))
    cases.append(("more than 16 distinct Gt/Lt bounds", lambda: solve([_pod("p%d" % i, required_terms=[[(wl.G + "/instance-cpu", "Gt", [str(i)])]]) for i in range(17)])))
    cases.append(("more than 64 distinct host ports", lambda: solve([_pod("p%d" % i, host_ports=[("0.0.0.0", 8000 + i, 0)]) for i in range(65)])))
    cases.append(("relaxation chain longer than 64", lambda: solve([_pod("p", preferred_terms=[(i + 1, [(wl.L_ZONE, "In", [zones[i % 3]])]) for i in range(66)])])))
    cases.append(("need 1..64 nodepools", lambda: solve([_pod("p")], nodepools=[wl.default_nodepool("np%d" % i) for i in range(65)])))
    # --- Solve: topology
    cases.append(("more than 6 special keys", lambda: solve([_pod("p", spreads=[_spread("example.com/t%d" % i) for i in range(7)])])))
    cases.append(("pod consults more than 8 topology groups", lambda: solve([_pod("p", spreads=[_spread(wl.L_ZONE, max_skew=1 + i) for i in range(9)])])))
    cases.append(("hostname topology with a hostname node selector", lambda: solve([_pod("p", node_selector={wl.L_HOSTNAME: "node-00001"}, spreads=[_spread(wl.L_HOSTNAME)])])))
    cases.append(("more than 64 values", lambda: solve([_pod("p%d" % i, node_selector={"example.com/rack": "r%d" % i}, spreads=[_spread("example.com/rack")]) for i in range(66)])))
    # --- Solve: reference quirks on existing nodes that lack a label
    cases.append(("NotIn/DoesNotExist and In/Exists pods on a label some existing node lacks",
                  lambda: solve([_pod("a", required_terms=[[(wl.L_ZONE, "NotIn", [zones[0]])]]), _pod("b", node_selector={wl.L_ZONE: zones[1]})], nodes_=keyless, bound_=bound)))
    cases.append(("NotIn/DoesNotExist pods on a topology key some existing node lacks",
                  lambda: solve([_pod("a", required_terms=[[(wl.L_ZONE, "NotIn", [zones[0]])]], spreads=[_spread(wl.L_ZONE)])], nodes_=keyless, bound_=bound)))
    # --- Solve: volumes, catalogs
    cases.append(("a volume is mounted by more than one pod", lambda: solve([_pod("a", volumes=[("ebs.csi.aws.com", "vol-1")]), _pod("b", volumes=[("ebs.csi.aws.com", "vol-1")])])))
    cases.append(("more than 60 catalogs", lambda: solve([_pod("p")], cats=[hcat] * 61)))
    cases.append(("kg_solve_noalloc: one catalog per solve", lambda: solve([_pod("p")], nodepools=[wl.default_nodepool("a"), dict(wl.default_nodepool("b"), catalog=1)], cats=[hcat, hcat], noalloc=True)))
    # --- catalog upload
    cases.append(("instance type requirement with a complement operator", lambda: L.catalog_upload(abi.build_catalog(_small_types(3, zones, complement=True)))))
    cases.append(("more than 64 zones", lambda: L.catalog_upload(abi.build_catalog(_small_types(2, ["z%02d" % i for i in range(65)])))))
    cases.append(("zone with two zone ids", lambda: L.catalog_upload(abi.build_catalog(_small_types(3, zones, two_zone_ids=True)))))

    # --- consolidation over more than 1024 instance types
    def cons_1025():
        big = _small_types(1025, zones)
        h = L.catalog_upload(abi.build_catalog(big))
        nd, bd = [], []
        for i in range(3):
            labels = {k: v[0] for k, op, v in big[i]["requirements"] if len(v) == 1}
            labels.update({wl.L_ZONE: zones[0], wl.L_NODEPOOL: "default", wl.L_HOSTNAME: "n%d" % i})
            nd.append({"name": "n%d" % i, "labels": labels, "taints": [], "available": {"cpu": 1500, "memory": 6 * wl.GI * 1000, "pods": 28000},
                       "capacity": dict(big[i]["capacity"]), "daemon_requests": {}, "initialized": True, "managed": True, "deleting": False})
            bd.append(dict(_pod("b%d" % i), node=i, reschedulable=True))
        hcl = L.cluster_upload(abi.build_cluster(nd, bd))
        try:
            L.consolidate_records([h], hcl, abi.build_pods([]), abi.build_nodepools(pool, wl.WELL_KNOWN_LABELS), abi.build_candidates([0, 1, 2]), False)
        finally:
            L.dll.kg_cluster_free(hcl)
            L.dll.kg_catalog_free(h)
    cases.append(("more than 1024 instance types", cons_1025))

    try:
        for needle, fn in cases:
            _expect(fn, needle)
            still_usable()
    finally:
        L.dll.kg_catalog_free(hcat)
    return len(cases)
