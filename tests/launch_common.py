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
