"""Consolidation parity plumbing shared by the emulation (CPU) and GPU tests: kg_consolidate_single / _multi records and
kg_consolidate_select decisions against the oracle's computeConsolidation per unit and its sequential selection."""
import numpy as np

from karpenter_amd import abi, workloads as wl
from oracle import loader
from tests.helpers import catalog776, families


def cluster(name, n):
    zones, types = catalog776()
    if name == "dense":
        nodes, bound = wl.dense_cluster(n, 11, types, zones, lo=0.7, hi=0.99)
        return types, nodes, bound, []
    if name == "sparse":
        nodes, bound = wl.existing_cluster(n, 11, types, zones)
        return types, nodes, bound, []
    if name == "dense_pending":
        nodes, bound = wl.dense_cluster(n, 5, types, zones, lo=0.8, hi=0.99)
        return types, nodes, bound, wl.mixed_pods(6, 9, zones, families(types))
    if name == "pending_ties":
        # pending pods that tie with bound pods (and with each other) on cpu, memory and creation time: the Queue order among
        # them is the UID's, across the pending / bound boundary of the simulations' pod universe
        nodes, bound = wl.dense_cluster(n, 13, types, zones, lo=0.6, hi=0.99)
        pending = []
        for k, p in enumerate(bound[::max(1, len(bound) // 12)][:12]):
            for tag in ("a", "zz"):   # one UID before every "node-..." UID, one after
                pending.append({"uid": "%s-tie-%02d" % (tag, k), "ns": "default", "creation_ts": p["creation_ts"], "labels": {"app": "tie"},
                                "requests": dict(p["requests"])})
        return types, nodes, bound, pending
    if name == "uninitialized":
        nodes, bound = wl.dense_cluster(n, 3, types, zones, lo=0.5, hi=0.9)
        for i, nd in enumerate(nodes):
            if i % 3 == 0:
                nd["initialized"] = False
        return types, nodes, bound, []
    if name == "topology":
        # bound pods carry the constraints their deployments would: zone spread, hostname spread, required hostname
        # anti-affinity, required zone affinity -- by app label, a quarter of the apps each
        nodes, bound = wl.dense_cluster(n, 17, types, zones, lo=0.4, hi=0.9)
        for p in bound:
            app = p["labels"]["app"]
            sel = {"match_labels": {"app": app}}
            kind = int(app[-2:]) % 4
            if kind == 0:
                p["spreads"] = [{"key": wl.L_ZONE, "max_skew": 2, "when": "DoNotSchedule", "selector": sel}]
            elif kind == 1:
                p["spreads"] = [{"key": wl.L_HOSTNAME, "max_skew": 3, "when": "DoNotSchedule", "selector": sel}]
            elif kind == 2:
                p["anti_affinity_required"] = [{"key": wl.L_HOSTNAME, "selector": sel}]
            else:
                p["affinity_required"] = [{"key": wl.L_ZONE, "selector": sel}]
        return types, nodes, bound, []
    if name == "host_ports":
        # a third of the bound pods hold host ports: a candidate's pods cannot move to a node (or share a replacement)
        # where a conflicting port is taken
        from tests.helpers import _with_host_ports
        nodes, bound = wl.dense_cluster(n, 19, types, zones, lo=0.4, hi=0.9)
        return types, nodes, _with_host_ports(bound, 8), []
    if name == "volumes":
        # every bound pod of a third of the apps holds an EBS volume; nodes carry CSINode counts that leave room for 0..3 more:
        # a candidate's pods can only move where their volumes still attach (or to a replacement)
        nodes, bound = wl.dense_cluster(n, 29, types, zones, lo=0.3, hi=0.8)
        per_node = {}
        for i, p in enumerate(bound):
            if int(p["labels"]["app"][-2:]) % 3 == 0:
                p["volumes"] = [("ebs.csi.aws.com", "vol-%05d" % i)]
                per_node[p["node"]] = per_node.get(p["node"], 0) + 1
        for i, nd in enumerate(nodes):
            if i % 5 != 4:
                nd["volume_limits"] = {"ebs.csi.aws.com": per_node.get(i, 0) + i % 4}
        return types, nodes, bound, []
    raise KeyError(name)


def nodepool_for(name):
    """NodePool of a named consolidation scenario: minValues on the instance family, loose / binding / beyond the 60-type cut."""
    G = "karpenter.k8s.aws"
    if name == "minvalues_loose":
        return wl.default_nodepool(extra_reqs=[(G + "/instance-family", "Exists", [], 3)])
    if name == "minvalues_tight":
        return wl.default_nodepool(extra_reqs=[(G + "/instance-family", "Exists", [], 12)])
    if name in ("minvalues_trunc", "minvalues_trunc_pending"):
        return wl.default_nodepool(extra_reqs=[("node.kubernetes.io/instance-type", "Exists", [], 70)])
    return wl.default_nodepool()


def check_consolidation(L, name, n, spot_to_spot, n_shards=3):
    types, nodes, bound, pending = cluster(("dense_pending" if name.endswith("_pending") else "dense") if name.startswith("minvalues") else name, n)
    order = wl.disruption_order(nodes, bound)
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods(pending), abi.build_nodepools([nodepool_for(name)], wl.WELL_KNOWN_LABELS)
    hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
    seen = set()
    try:
        for multi in (False, True):
            cd = abi.build_candidates(order, spot_to_spot=spot_to_spot)
            st, ro, _ = loader.consolidate_records([cat], cl, pend, nps, cd, multi)
            assert st == 0
            rg, _ = L.consolidate_records([hcat], hcl, pend, nps, cd, multi)
            a, b = [abi.record_tuple(r) for r in ro], [abi.record_tuple(r) for r in rg]
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert x == y, (name, multi, x, y)
            # (above: visit counters on = every simulation scans exactly; the production setting takes the prefilter path)
            L.set_collect_stats(False)
            try:
                rp, _ = L.consolidate_records([hcat], hcl, pend, nps, cd, multi)
            finally:
                L.set_collect_stats(True)
            assert [abi.record_tuple(r) for r in rp] == b, (name, multi, "prefilter path")
            seen.update(x[1] for x in a)
            # the shards' records, gathered, give the reference's sequential decision
            parts = [L.consolidate_records([hcat], hcl, pend, nps, abi.build_candidates(order, spot_to_spot=spot_to_spot, shard=s, n_shards=n_shards), multi)[0]
                     for s in range(n_shards)]
            assert sum(len(p) for p in parts) == len(a)
            dg = L.consolidate_select(hcat, np.concatenate(parts), len(order), multi)
            st, do = loader.consolidate_decide([cat], cl, pend, nps, cd, multi)
            assert st == 0
            for k in ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price"):
                assert do[k] == dg[k], (name, multi, k, do[k], dg[k])
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(hcat)
    return seen


def check_candidates_order(L):
    """kg_candidates_order against the oracle's restatement and, where every node is eligible, the
    Python statement of the cost formula: eligibility filters, lifetimeRemaining weights, blocked nodes, stable ties."""
    zones, types = catalog776()
    nodes, bound = wl.dense_cluster(300, 5, types, zones)
    r = wl.SplitMix64(3)
    for p in bound:
        p["deletion_cost"] = int(r.below(2000)) * 100000 - 100000000
    for i, nd in enumerate(nodes):
        if i % 11 == 0:
            nd["initialized"] = False
        if i % 13 == 0:
            nd["deleting"] = True
        if i % 17 == 0:
            nd["managed"] = False
        if i % 19 == 0:
            nd["labels"]["node.kubernetes.io/instance-type"] = "x9.mystery"
        if i % 23 == 0:
            del nd["labels"][wl.L_NODEPOOL]
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
    try:
        lifetime = np.array([1.0 - (i % 7) / 10.0 for i in range(len(nodes))])
        blocked = np.array([1 if i % 29 == 0 else 0 for i in range(len(nodes))], "u1")
        sizes = []
        for lt, bl, with_cat in ((None, None, True), (lifetime, blocked, True), (lifetime, None, False)):
            st, o_nodes, o_costs = loader.candidates_order(cat if with_cat else None, cl, len(nodes), lt, bl)
            assert st == 0
            g_nodes, g_costs = L.candidates_order(hcat if with_cat else None, hcl, len(nodes), lt, bl)
            assert o_nodes == g_nodes and o_costs == g_costs
            assert g_costs == sorted(g_costs)
            sizes.append(len(g_nodes))
        assert sizes[0] > sizes[1] and sizes[2] > sizes[1] and sizes[0] < len(nodes)
        nodes2, bound2 = wl.dense_cluster(200, 6, types, zones)
        hcl2 = L.cluster_upload(abi.build_cluster(nodes2, bound2))
        try:
            assert L.candidates_order(hcat, hcl2, len(nodes2))[0] == wl.disruption_order(nodes2, bound2)
        finally:
            L.dll.kg_cluster_free(hcl2)
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(hcat)


def check_validate(L, n=90):
    """kg_consolidate_validate against the oracle: commands computed on a cluster are re-checked on
    the same snapshot (valid), on a snapshot where the other nodes have filled up (deletes turn invalid), and with the wrong
    kind of command; single- and multi-candidate commands."""
    types, nodes, bound, pending = cluster("dense", n)
    order = wl.disruption_order(nodes, bound)
    cat = abi.build_catalog(types)
    pend, nps = abi.build_pods(pending), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    cl = abi.build_cluster(nodes, bound)
    # the later snapshot: every node but the first 5 candidates has lost its free cpu
    nodes_b = [dict(nd, available=dict(nd["available"])) for nd in nodes]
    for i, nd in enumerate(nodes_b):
        if i not in order[:5]:
            nd["available"]["cpu"] = 0
    cl_b = abi.build_cluster(nodes_b, bound)
    hcat, hcl, hcl_b = L.catalog_upload(cat), L.cluster_upload(cl), L.cluster_upload(cl_b)
    try:
        recs, _ = L.consolidate_records([hcat], hcl, pend, nps, abi.build_candidates(order[:40], spot_to_spot=True), False)
        outcomes = set()
        n_cmd = 0
        for r in recs:
            dec = int(r["decision"])
            if dec == 0:
                continue
            n_cmd += 1
            cands = [order[int(r["candidate"])]]
            tids = [w * 64 + b for w in range(16) for b in range(64) if (int(r["type_mask"][w]) >> b) & 1]
            for snap, hsnap in ((cl, hcl), (cl_b, hcl_b)):
                for d, t in ((dec, tids), (3 - dec, tids if dec == 1 else []), (2, [0, 1, 2])):
                    if d == 2 and not t:
                        continue
                    st, vo = loader.consolidate_validate([cat], snap, pend, nps, cands, d, t if d == 2 else [])
                    assert st == 0
                    vg = L.consolidate_validate([hcat], hsnap, pend, nps, cands, d, t if d == 2 else [])
                    assert vo == vg, (cands, d, snap is cl)
                    outcomes.add((snap is cl, d == dec, vg))
            assert L.consolidate_validate([hcat], hcl, pend, nps, cands, dec, tids if dec == 2 else [])  # unchanged snapshot: still valid
        assert n_cmd >= 5
        assert (True, True, True) in outcomes and (False, True, False) in outcomes and (True, False, False) in outcomes
        # a multi-node command
        dm = L.consolidate_select(hcat, L.consolidate_records([hcat], hcl, pend, nps, abi.build_candidates(order), True)[0], len(order), True)
        if dm["decision"] != 0:
            cands = order[:dm["n_candidates"]]
            for snap, hsnap in ((cl, hcl), (cl_b, hcl_b)):
                st, vo = loader.consolidate_validate([cat], snap, pend, nps, cands, dm["decision"], dm["types"])
                assert st == 0 and vo == L.consolidate_validate([hcat], hsnap, pend, nps, cands, dm["decision"], dm["types"])
            assert L.consolidate_validate([hcat], hcl, pend, nps, cands, dm["decision"], dm["types"])
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_cluster_free(hcl_b)
        L.dll.kg_catalog_free(hcat)


def upstream_first_n(decisions, n):
    """MultiNodeConsolidation.firstNConsolidationOption replayed by hand over per-prefix verdicts (decisions[mid] = the
    command for candidates[0:mid+1] is not a no-op): max = Clamp(len, 0, 100); if len <= max: max = len - 1; binary search
    over mid in [1, max] -> number of candidates of the last saved command (0 = none)."""
    if n < 2:
        return 0
    lo, hi = 1, min(max(n, 0), 100)
    if n <= hi:
        hi = n - 1
    last = 0
    while lo <= hi:
        mid = (lo + hi) // 2
        if decisions[mid]:
            last, lo = mid + 1, mid + 1
        else:
            hi = mid - 1
    return last


def check_multi_over_100(L, n=140):
    """More than 100 candidates: the binary search runs over mid in [1, 100], so the longest command consolidates
    candidates[0:101] -- 101 nodes (upstream passes max = 100 and only lowers it to len - 1 when len <= 100).  A sparse cluster
    whose every prefix can be deleted must come back with 101 candidates; the records must hold prefix units 1..100; a
    101-candidate command validates and a 102-candidate one is rejected as malformed."""
    zones, types = catalog776()
    nodes, bound = wl.existing_cluster(n, 11, types, zones, pods_lambda=2.0, util=0.1)
    order = wl.disruption_order(nodes, bound)
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
    try:
        cd = abi.build_candidates(order)
        recs, _ = L.consolidate_records([hcat], hcl, pend, nps, cd, True)
        assert [int(r["candidate"]) for r in recs] == list(range(1, 101))
        st, ro, _ = loader.consolidate_records([cat], cl, pend, nps, cd, True)
        assert st == 0 and [abi.record_tuple(r) for r in ro] == [abi.record_tuple(r) for r in recs]
        verdict = {int(r["candidate"]): int(r["decision"]) != 0 for r in recs}
        want = upstream_first_n(verdict, len(order))
        dg = L.consolidate_select(hcat, recs, len(order), True)
        st, do = loader.consolidate_decide([cat], cl, pend, nps, cd, True)
        assert st == 0 and do["n_candidates"] == dg["n_candidates"] == want and do["decision"] == dg["decision"]
        assert want == 101 and dg["decision"] == abi.DECISION_DELETE  # every prefix of this cluster fits on the remaining nodes
        assert L.consolidate_validate([hcat], hcl, pend, nps, order[:101], dg["decision"], [])
        with __import__("pytest").raises(Exception):
            L.consolidate_validate([hcat], hcl, pend, nps, order[:102], dg["decision"], [])
        # exactly 100 and 101 candidates: max = len - 1 = 99, and max = 100
        for k, longest in ((100, 100), (101, 101)):
            cdk = abi.build_candidates(order[:k])
            rk, _ = L.consolidate_records([hcat], hcl, pend, nps, cdk, True)
            assert [int(r["candidate"]) for r in rk] == list(range(1, longest))
            dk = L.consolidate_select(hcat, rk, k, True)
            st, ok = loader.consolidate_decide([cat], cl, pend, nps, cdk, True)
            assert st == 0 and dk["n_candidates"] == ok["n_candidates"] == upstream_first_n({int(r["candidate"]): int(r["decision"]) != 0 for r in rk}, k)
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(hcat)


def check_plan_reuse(L, name="dense", n=90):
    """Rounds over one snapshot reuse what the first prepared (kg_stats.cons_plan_hits) and still give the oracle's records:
    the same round again, the shards of a 3-way split in turn (twice), single / multi alternating; and whatever the plan
    was built from changing -- an in-place catalog update, another candidate order, other pending pods, the visit-counter
    setting -- prepares anew.  More plans than a snapshot keeps (4) evict the least recently used one."""
    from tests.update_common import apply_updates
    types, nodes, bound, _ = cluster(name, n)
    zones, _t = catalog776()
    order = wl.disruption_order(nodes, bound)
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
    tup = lambda recs: [abi.record_tuple(r) for r in recs]

    def oracle(cat_, pend_, cd, multi):
        st, ro, _ = loader.consolidate_records([cat_], cl, pend_, nps, cd, multi)
        assert st == 0
        return tup(ro)

    def counters():
        s = L.stats()
        return int(s.cons_plans_built), int(s.cons_plan_hits)

    try:
        cd = abi.build_candidates(order, spot_to_spot=True)
        want = {m: oracle(cat, pend, cd, m) for m in (False, True)}
        b0, h0 = counters()
        for rnd in range(3):
            for multi in (False, True):
                assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, multi)[0]) == want[multi], (rnd, multi)
        assert counters() == (b0 + 2, h0 + 4)
        for rnd in range(2):  # the shards of a 3-way split share the plan; their records interleave to the whole
            parts = [tup(L.consolidate_records([hcat], hcl, pend, nps, abi.build_candidates(order, spot_to_spot=True, shard=s, n_shards=3), False)[0]) for s in range(3)]
            assert sorted(sum(parts, []), key=lambda r: r[0]) == want[False]
        assert counters() == (b0 + 2, h0 + 10)
        # spot-to-spot is decided per call on the cached simulations
        cd_off = abi.build_candidates(order, spot_to_spot=False)
        assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd_off, False)[0]) == oracle(cat, pend, cd_off, False)
        assert counters() == (b0 + 2, h0 + 11)
        # an in-place catalog update: the c / m / r families of generations 5 and 6 sold out everywhere, t3 spot nearly free
        ups = []
        for i, t in enumerate(types):
            fam = t["name"].split(".")[0]
            if fam[0] in "cmr" and fam[1] in "56":
                ups += [(i, z, ct, False, None) for z in zones for ct in ("spot", "on-demand") if any(o[0] == z and o[1] == ct for o in t["offerings"])]
            if fam == "t3":
                ups += [(i, z, "spot", None, 0.0003) for z in zones if any(o[0] == z and o[1] == "spot" for o in t["offerings"])]
        types_b = apply_updates(types, ups)
        L.catalog_update_offerings(hcat, ups, generation=2)
        cat_b = abi.build_catalog(types_b)
        want_b = oracle(cat_b, pend, cd, False)
        assert want_b != want[False] or name != "dense"
        assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, False)[0]) == want_b
        assert counters() == (b0 + 3, h0 + 11)
        # another candidate order, other pending pods, the production setting (prefilter path)
        rev = abi.build_candidates(order[::-1], spot_to_spot=True)
        assert tup(L.consolidate_records([hcat], hcl, pend, nps, rev, False)[0]) == oracle(cat_b, pend, rev, False)
        pend2 = abi.build_pods(wl.mixed_pods(5, 4, zones, families(types)))
        assert tup(L.consolidate_records([hcat], hcl, pend2, nps, cd, False)[0]) == oracle(cat_b, pend2, cd, False)
        L.set_collect_stats(False)
        try:
            assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, False)[0]) == want_b
            assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, True)[0]) == oracle(cat_b, pend, cd, True)
        finally:
            L.set_collect_stats(True)
        assert counters() == (b0 + 7, h0 + 11)
        # (7 plans built over a snapshot that keeps 4) the evicted first ones are prepared again, the kept ones are found
        assert tup(L.consolidate_records([hcat], hcl, pend2, nps, cd, False)[0]) == oracle(cat_b, pend2, cd, False)
        assert counters() == (b0 + 7, h0 + 12)
        assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, False)[0]) == want_b
        assert counters()[0] == b0 + 8
        # a validated command does not disturb the plans
        rec = [r for r in L.consolidate_records([hcat], hcl, pend, nps, cd, False)[0] if int(r["decision"]) == 1]
        if rec:
            assert L.consolidate_validate([hcat], hcl, pend, nps, [order[int(rec[0]["candidate"])]], 1, [])
        assert tup(L.consolidate_records([hcat], hcl, pend, nps, cd, False)[0]) == want_b
    finally:
        L.dll.kg_cluster_free(hcl)
        L.dll.kg_catalog_free(hcat)
