"""More than one catalog per call (NodePools on different EC2NodeClasses: cloudprovider.go:172-190, 254-265) and the
catalog generation check (KG_ESTALE), shared by the emulation (CPU) and GPU tests."""
import copy

import numpy as np
import pytest

from karpenter_amd import abi, workloads as wl
from oracle import loader
from tests.helpers import catalog776, diff_results, families


def two_catalogs(lo=300, hi=500):
    """Two overlapping cuts of the 776-type catalog: A = types[:hi], B = types[lo:] with every price 10 % lower, 300m more
    cpu overhead and no offering in the first zone -- the same type NAME means different things in the two."""
    zones, types = catalog776()
    a = copy.deepcopy(types[:hi])
    b = copy.deepcopy(types[lo:])
    for t in b:
        t["offerings"] = [(z, c, zid, round(p * 0.9, 6), av) for z, c, zid, p, av in t["offerings"] if z != zones[0]]
        t["overhead"] = dict(t["overhead"], cpu=t["overhead"].get("cpu", 0) + 300)
        t["requirements"] = [(k, op, [x for x in v if x != zones[0]] if k == wl.L_ZONE else v) for k, op, v in t["requirements"]]
        t["requirements"] = [(k, op, [x for x in v if x in {o[2] for o in t["offerings"]}] if k == "topology.k8s.aws/zone-id" else v) for k, op, v in t["requirements"]]
    return zones, a, b


def pools_ab(limits_a=None):
    return [dict(wl.default_nodepool(name="pool-a", weight=10, limits=limits_a), catalog=0),
            dict(wl.default_nodepool(name="pool-b", weight=1), catalog=1)]


def check_multi_catalog_solve(L):
    zones, a, b = two_catalogs()
    ca, cb = abi.build_catalog(a), abi.build_catalog(b)
    ha, hb = L.catalog_upload(ca), L.catalog_upload(cb)
    fams = families(a + b)
    try:
        st0 = L.stats()
        for seed, n, limits in ((3, 400, {"cpu": 150000}), (4, 300, None), (5, 500, {"cpu": 20000, "memory": 64 * wl.GI * 1000})):
            pods = wl.mixed_pods(n, seed, zones, fams)
            pd, nps = abi.build_pods(pods), abi.build_nodepools(pools_ab(limits), wl.WELL_KNOWN_LABELS)
            st, ro = loader.solve([ca, cb], None, pd, nps, len(pods))
            assert st == 0
            want = ro.decode()
            for stats in (True, False):
                L.set_collect_stats(stats)
                try:
                    got = L.solve([ha, hb], None, pd, nps, len(pods)).decode()
                finally:
                    L.set_collect_stats(True)
                assert diff_results(got, want) == "", (seed, stats, diff_results(got, want))
            used = {c["nodepool"] for c in want["claims"]}
            if limits:
                assert used == {0, 1}  # pool-a runs into its limits, pool-b takes the rest
            # type ids are those of the claim's own catalog
            for c in want["claims"]:
                assert all(t < (len(a) if c["nodepool"] == 0 else len(b)) for t in c["types"])
            # the catalogs the other way round, NodePools re-pointed: the same claims with the same per-catalog type ids
            nps_r = abi.build_nodepools([dict(p, catalog=1 - p["catalog"]) for p in pools_ab(limits)], wl.WELL_KNOWN_LABELS)
            assert diff_results(L.solve([hb, ha], None, pd, nps_r, len(pods)).decode(), want) == ""
        # both NodePools on one of two catalogs passed == that catalog alone
        pods = wl.mixed_pods(200, 9, zones, fams)
        pd = abi.build_pods(pods)
        one = [dict(wl.default_nodepool(name="pool-a", weight=10), catalog=1), dict(wl.default_nodepool(name="pool-b", weight=1, capacity_types=("on-demand",)), catalog=1)]
        st, ro = loader.solve([ca, cb], None, pd, abi.build_nodepools(one, wl.WELL_KNOWN_LABELS), len(pods))
        assert st == 0
        assert diff_results(L.solve([ha, hb], None, pd, abi.build_nodepools(one, wl.WELL_KNOWN_LABELS), len(pods)).decode(), ro.decode()) == ""
        assert diff_results(L.solve([hb], None, pd, abi.build_nodepools([dict(p, catalog=0) for p in one], wl.WELL_KNOWN_LABELS), len(pods)).decode(), ro.decode()) == ""
        # an in-place update of one of the catalogs is seen by the next solve over both
        ups = [(i, z, "spot", False, None) for i, t in enumerate(a) for z in zones if t["name"].split(".")[0] in ("c5", "m5", "c6i") and any(o[0] == z and o[1] == "spot" for o in t["offerings"])]
        from tests.update_common import apply_updates
        a2 = apply_updates(a, ups)
        L.catalog_update_offerings(ha, ups, generation=2)
        pods = wl.mixed_pods(300, 12, zones, fams)
        pd, nps = abi.build_pods(pods), abi.build_nodepools(pools_ab(), wl.WELL_KNOWN_LABELS)
        st, ro = loader.solve([abi.build_catalog(a2), cb], None, pd, nps, len(pods))
        assert st == 0
        assert diff_results(L.solve([ha, hb], None, pd, nps, len(pods)).decode(), ro.decode()) == ""
        # a nodepool naming a catalog that was not passed
        bad = abi.build_nodepools([dict(wl.default_nodepool(), catalog=2)], wl.WELL_KNOWN_LABELS)
        with pytest.raises(Exception, match="EINVAL"):
            L.solve([ha, hb], None, pd, bad, len(pods))
    finally:
        L.dll.kg_catalog_free(ha)
        L.dll.kg_catalog_free(hb)


def check_multi_catalog_consolidation(L, n=80):
    zones, a, b = two_catalogs(300, 776)  # (A = the whole catalog, B = its upper part, cheaper and without the first zone: 1252 types would be
    a = a[250:700]                        #  too many for a consolidation record -- cut A down to 450, B stays 476)
    a_names, b_names = {t["name"] for t in a}, {t["name"] for t in b}
    both = [t for t in a if t["name"] in b_names]
    nodes, bound = wl.dense_cluster(n, 23, both, zones, lo=0.5, hi=0.95)
    for i, nd in enumerate(nodes):
        if i % 2:
            nd["labels"][wl.L_NODEPOOL] = "pool-b"
            if nd["labels"][wl.L_ZONE] == zones[0]:
                nd["labels"][wl.L_ZONE] = zones[1]
        else:
            nd["labels"][wl.L_NODEPOOL] = "pool-a"
    order = wl.disruption_order(nodes, bound)
    ca, cb, cl = abi.build_catalog(a), abi.build_catalog(b), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools(pools_ab(), wl.WELL_KNOWN_LABELS)
    ha, hb, hcl = L.catalog_upload(ca), L.catalog_upload(cb), L.cluster_upload(cl)
    try:
        seen = set()
        for multi in (False, True):
            cd = abi.build_candidates(order, spot_to_spot=True)
            st, ro, _ = loader.consolidate_records([ca, cb], cl, pend, nps, cd, multi)
            assert st == 0
            rg, _ = L.consolidate_records([ha, hb], hcl, pend, nps, cd, multi)
            # (the records' type masks are over the call's merged type numbering: A's types, then B's -- the oracle's too)
            assert [abi.record_tuple(r) for r in ro] == [abi.record_tuple(r) for r in rg], multi
            seen.update(int(r["decision"]) for r in rg)
            st, do = loader.consolidate_decide([ca, cb], cl, pend, nps, cd, multi)
            assert st == 0
            dg = L.consolidate_select([ha, hb], rg, len(order), multi)
            for k in ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price"):
                assert do[k] == dg[k], (multi, k, do[k], dg[k])
        assert 2 in seen and 1 in seen
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(ha)
        L.dll.kg_catalog_free(hb)


def check_stale_generation(L):
    """instancetype.go:127-136: the List() cache key.  A call whose NodePool expects another generation than the resident
    catalog carries is refused with KG_ESTALE (and changes nothing); after the update the same call goes through."""
    zones, types = catalog776()
    h = L.catalog_upload(abi.build_catalog(types), generation=41)
    pods = wl.mixed_pods(50, 2, zones, families(types))
    pd = abi.build_pods(pods)
    nodes, bound = wl.dense_cluster(30, 3, types, zones)
    hcl = L.cluster_upload(abi.build_cluster(nodes, bound))
    try:
        ok = abi.build_nodepools([dict(wl.default_nodepool(), catalog_generation=41)], wl.WELL_KNOWN_LABELS)
        unchecked = abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
        newer = abi.build_nodepools([dict(wl.default_nodepool(), catalog_generation=42)], wl.WELL_KNOWN_LABELS)
        base = L.solve([h], None, pd, ok, len(pods)).decode()
        assert diff_results(L.solve([h], None, pd, unchecked, len(pods)).decode(), base) == ""
        with pytest.raises(Exception, match="ESTALE"):
            L.solve([h], None, pd, newer, len(pods))
        with pytest.raises(Exception, match="ESTALE"):
            L.consolidate_records([h], hcl, abi.build_pods([]), newer, abi.build_candidates(list(range(10))), False)
        with pytest.raises(Exception, match="ESTALE"):
            L.consolidate_validate([h], hcl, abi.build_pods([]), newer, [0], 1, [])
        L.catalog_update_offerings(h, [(0, zones[0], "spot", False, None)], generation=42)
        assert L.solve([h], None, pd, newer, len(pods)).decode()["pods"]
        with pytest.raises(Exception, match="ESTALE"):
            L.solve([h], None, pd, ok, len(pods))
        assert len(L.consolidate_records([h], hcl, abi.build_pods([]), newer, abi.build_candidates(list(range(10))), False)[0]) == 10
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(h)
