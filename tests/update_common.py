"""kg_catalog_update_offerings parity plumbing, shared by the emulation (CPU) and GPU tests: a
resident catalog patched in place must behave exactly like a fresh upload of the catalog the provider would rebuild
(createOfferings with the new ICE marks / spot prices, computeRequirements over the available offerings)."""
import copy

from karpenter_amd import abi, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families

L_ZID = "topology.k8s.aws/zone-id"


def apply_updates(types, updates):
    """The provider's rebuild in Python: new availability / price per offering, then the three offering-derived keys."""
    out = copy.deepcopy(types)
    for t, zone, ct, avail, price in updates:
        offs = out[t]["offerings"]
        for i, (z, c, zid, p, a) in enumerate(offs):
            if z == zone and c == ct:
                offs[i] = (z, c, zid, p if price is None else price, a if avail is None else bool(avail))
    for t in {u[0] for u in updates}:
        ty = out[t]
        live = [o for o in ty["offerings"] if o[4]]
        derived = {wl.L_ZONE: sorted({o[0] for o in live}), wl.L_CT: sorted({o[1] for o in live}), L_ZID: sorted({o[2] for o in live if o[2]})}
        reqs = [(k, op, derived[k]) if k in derived else (k, op, v) for k, op, v in ty["requirements"] if k != L_ZID]
        if derived[L_ZID]:  # types.go:110-117: the zone-id requirement exists only while an available offering carries a zone id
            reqs.append((L_ZID, "In", derived[L_ZID]))
        ty["requirements"] = reqs
    return out


def type_requirements(t):
    """{key: sorted values} of a catalog type dict, the shape Library.catalog_requirements returns."""
    return {k: sorted(str(x) for x in v) for k, op, v in t["requirements"]}


def scenario_updates(types, zones):
    """ICE marks and spot price moves: whole families out of one zone, single offerings, one type with nothing left,
    one type flipped back, spot prices above on-demand and far below."""
    ups = []
    for i, t in enumerate(types):
        fam = t["name"].split(".")[0]
        if fam in ("c5", "m5", "r5"):
            ups.append((i, zones[0], "spot", False, None))
        if fam == "m6i":
            ups.append((i, zones[1], "on-demand", False, None))
            ups.append((i, zones[1], "spot", False, None))
        if t["name"] == "c6i.large":
            for z in zones:
                for ct in ("on-demand", "spot"):
                    ups.append((i, z, ct, False, None))
        if t["name"] == "c6i.xlarge":
            ups.append((i, zones[2], "spot", False, None))
            ups.append((i, zones[2], "spot", True, 0.001))
        if fam == "t3":
            ups.append((i, zones[0], "spot", None, 9.99))
        if fam == "c6a":
            ups.append((i, zones[2], "spot", None, 0.0005))
    return ups


def check_updates(L, n_pods=600):
    zones, types = catalog776()
    ups = scenario_updates(types, zones)
    types_b = apply_updates(types, ups)
    pods = wl.mixed_pods(n_pods, 21, zones, families(types))
    prob_a, prob_b = Problem(types, pods, [wl.default_nodepool()]), Problem(types_b, pods, [wl.default_nodepool()])
    h = L.catalog_upload(prob_a.cat)
    try:
        before = prob_a.run_library(L, h).decode()
        L.catalog_update_offerings(h, ups, generation=7)
        assert L.catalog_size(h)[1] == 7
        after = prob_a.run_library(L, h).decode()     # patched resident catalog
        # the requirements the encoder holds for every touched type == computeRequirements over the rebuilt offerings: the
        # all-ICE'd type lists no zone / capacity type and has lost its zone-id requirement
        for t in sorted({u[0] for u in ups}):
            assert L.catalog_requirements(h, t) == type_requirements(types_b[t]), types[t]["name"]
        gone = [i for i, t in enumerate(types) if t["name"] == "c6i.large"][0]
        assert L_ZID not in L.catalog_requirements(h, gone) and L.catalog_requirements(h, gone)[wl.L_ZONE] == []
        fresh = prob_b.run_library(L).decode()        # fresh upload of the rebuilt catalog
        oracle = prob_b.run_oracle().decode()
        assert diff_results(fresh, oracle) == ""
        assert diff_results(after, fresh) == "", diff_results(after, fresh)
        assert diff_results(before, after) != ""      # the scenario does change the outcome
        # undo: the catalog returns to its first behaviour
        undo = [(t, z, ct, True if av is not None else None, [o for o in types[t]["offerings"] if o[0] == z and o[1] == ct][0][3]) for t, z, ct, av, pr in ups]
        L.catalog_update_offerings(h, undo, generation=8)
        assert diff_results(before, prob_a.run_library(L, h).decode()) == ""
        for t in sorted({u[0] for u in ups}):  # ... and to its first requirements: the zone-id requirement is back
            assert L.catalog_requirements(h, t) == type_requirements(types[t]), types[t]["name"]
        st = L.stats()
        assert st.catalog_words_patched > 0
    finally:
        L.dll.kg_catalog_free(h)
    # an offering the catalog does not have is an input error, and nothing is changed by the call
    h = L.catalog_upload(prob_a.cat)
    try:
        try:
            L.catalog_update_offerings(h, [(0, zones[0], "spot", False, None), (0, "nowhere-1z", "spot", False, None)], generation=9)
            raise AssertionError("expected KG_EINVAL")
        except Exception as e:  # noqa: BLE001
            assert "EINVAL" in str(e), e
        assert diff_results(before, prob_a.run_library(L, h).decode()) == ""
    finally:
        L.dll.kg_catalog_free(h)
