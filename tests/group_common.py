"""kg_group: several devices behind the C ABI in one process.  A group over however many devices the box
has -- and over one device named several times (several contexts, one GPU) -- must give the oracle's sequential decision."""
from karpenter_amd import abi, lib, workloads as wl
from oracle import loader
from tests.cons_common import cluster, nodepool_for


def check_group(path, device_lists, name="dense", n=120):
    types, nodes, bound, pending = cluster(name, n)
    order = wl.disruption_order(nodes, bound)
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods(pending), abi.build_nodepools([nodepool_for(name)], wl.WELL_KNOWN_LABELS)
    want, want_recs = {}, {}
    for multi in (False, True):
        cd = abi.build_candidates(order, spot_to_spot=True)
        st, want[multi] = loader.consolidate_decide([cat], cl, pend, nps, cd, multi)
        assert st == 0
        st, ro, _ = loader.consolidate_records([cat], cl, pend, nps, cd, multi)
        assert st == 0
        want_recs[multi] = sorted(abi.record_tuple(r) for r in ro)
    sizes = []
    for devices in device_lists:
        G = lib.Group(path, devices=devices)
        try:
            sizes.append(G.n)
            hc, hl = G.catalog_upload(cat), G.cluster_upload(cl)
            try:
                for rnd in range(2):  # (the second round finds every device's plan prepared)
                    for multi in (False, True):
                        dec, recs, ms = G.consolidate([hc], hl, pend, nps, abi.build_candidates(order, spot_to_spot=True), multi)
                        for k in ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price"):
                            assert dec[k] == want[multi][k], (devices, multi, k, dec[k], want[multi][k])
                        assert sorted(abi.record_tuple(r) for r in recs) == want_recs[multi], (devices, multi)
                        assert ms > 0
                if G.n > 1:
                    # every device ran its share of the simulations, and prepared its snapshot once per mode
                    sims = [int(G.stats(i).cons_simulations) for i in range(G.n)]
                    assert all(s > 0 for s in sims), sims
                    assert [int(G.stats(i).cons_plans_built) for i in range(G.n)] == [2] * G.n
                    assert [int(G.stats(i).cons_plan_hits) for i in range(G.n)] == [2] * G.n
                # an in-place update reaches every device's catalog: all replicas keep deciding alike
                zones = sorted({o[0] for t in types for o in t["offerings"]})
                ups = [(i, z, "spot", False, None) for i, t in enumerate(types) for z in zones
                       if t["name"].split(".")[0] in ("c5", "m5", "r5", "c6i", "m6i") and any(o[0] == z and o[1] == "spot" for o in t["offerings"])]
                from tests.update_common import apply_updates
                G.catalog_update_offerings(hc, ups, generation=2)
                cat2 = abi.build_catalog(apply_updates(types, ups))
                st, want2 = loader.consolidate_decide([cat2], cl, pend, nps, abi.build_candidates(order, spot_to_spot=True), False)
                assert st == 0
                dec, _, _ = G.consolidate([hc], hl, pend, nps, abi.build_candidates(order, spot_to_spot=True), False, want_records=False)
                for k in ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price"):
                    assert dec[k] == want2[k], (devices, "after update", k)
                # a failing device call surfaces with its text
                bad = abi.build_nodepools([dict(nodepool_for(name), catalog=3)], wl.WELL_KNOWN_LABELS)
                try:
                    G.consolidate([hc], hl, pend, bad, abi.build_candidates(order), False)
                    raise AssertionError("expected KG_EINVAL")
                except lib.KgError as e:
                    assert "EINVAL" in str(e) and "catalog index" in str(e), e
            finally:
                G.cluster_free(hl)
                G.catalog_free(hc)
        finally:
            G.close()
    return sizes
