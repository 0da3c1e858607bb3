"""kg_catalog_build_ec2 (the List() miss path done by the library) against the catalog oracle, which is itself pinned on
the reference's 776 documented rows (tests/test_catalog_oracle.py): same capacity / allocatable per resource for every
type, and the same behaviour as an uploaded catalog in Solve and in the launch filter."""
from karpenter_amd import abi, workloads as wl
from oracle import catalog_oracle as co, loader
from tests.helpers import Problem, diff_results, families

RES = [co.RES_CPU, co.RES_MEM, co.RES_EPH, co.RES_PODS, co.RES_POD_ENI, co.RES_NVIDIA, co.RES_AMD, co.RES_NEURON, co.RES_GAUDI, co.RES_EFA,
       co.RES_PRIVATE_IPV4]


def _check_tables(L, infos, ctx, oz, doc=None):
    types = co.list_instance_types(infos, ctx, ctx["limits"], ctx["bandwidth"], oz)
    h = L.catalog_build_ec2(abi.build_ec2_catalog(infos, ctx, ctx["limits"], ctx["bandwidth"], oz))
    try:
        assert L.catalog_size(h)[0] == len(types)
        for i, t in enumerate(types):
            name, cap, alloc = L.catalog_query(h, i, RES)
            assert name == t["name"]
            exp_alloc = co.allocatable(t)
            for r in RES:
                assert cap[r] == t["capacity"].get(r, 0), (name, r, cap[r], t["capacity"].get(r, 0))
                assert alloc[r] == exp_alloc.get(r, 0), (name, r, alloc[r], exp_alloc.get(r, 0))
            # computeRequirements (types.go:74-172), read back from the encoder: every key, every value, DoesNotExist entries
            got = L.catalog_requirements(h, i)
            assert got == {k: sorted(str(x) for x in v) for k, op, v in t["requirements"]}, (name, got)
            # ... and directly against the labels the reference documents for the type (instance-types.md)
            if doc is not None and name in doc:
                for k, v in doc[name].items():
                    assert got.get(k) == [v], (name, k, got.get(k), v)
    finally:
        L.dll.kg_catalog_free(h)
    return types


def check_ec2_build(L):
    zones = wl.zones_n(3)
    infos = wl.ec2_facts_776()
    base = wl.ec2_context(zones)
    oz = {i["InstanceType"]: zones for i in infos}
    oz[infos[5]["InstanceType"]] = zones[:1]          # a type offered in one zone only
    variants = [
        dict(base),
        dict(base, ami_family="Bottlerocket", pods_per_core=2, reserved_enis=1),
        dict(base, ami_family="AL2", max_pods=58, kube_reserved={"cpu": "250m", "memory": "1Gi"}, system_reserved={"memory": "200Mi"},
             eviction_hard={"memory.available": "5%", "nodefs.available": "15%"}, eviction_soft={"memory.available": "750Mi"}),
        dict(base, ami_family="Windows2022", pods_per_core=4),
        # quantities with fractions and exponents (parsed exactly), still whole milli-units
        dict(base, ami_family="AL2", eviction_hard={"memory.available": "1e9", "nodefs.available": "10%"}, eviction_soft={"memory.available": "1.5Gi", "nodefs.available": "100.5Mi"}),
        dict(base, ami_family="AL2023", instance_store_policy="RAID0", unavailable={("spot", infos[0]["InstanceType"], zones[0]), ("on-demand", infos[9]["InstanceType"], zones[2])}),
    ]
    # the single-valued labels of the 776 documented rows (AL2023, the doc generator's node class = variant 0)
    doc = {r["name"]: {k: v for k, v in r["labels"].items() if k != "karpenter.k8s.aws/instance-network-bandwidth" or True} for r in wl._load("instance_types_doc.json")}
    for vi, ctx in enumerate(variants):
        types = _check_tables(L, infos, ctx, oz, doc if vi == 0 else None)
    # a quantity finer than one milli-unit is not representable: refused, not rounded
    # ... and so is one whose milli-units overflow: a long digit string with a large suffix must not wrap around 128 bits
    # on its way to the range check, nor a long fraction with a negative exponent
    for bad in ("0.0001", "1.0000001Mi", "5u", "99999999999999999999Ei", "999999999999999999999999999999e24", "1.000000000000000000000000000001e-24"):
        ctx = dict(base, eviction_hard={"memory.available": bad})
        try:
            L.dll.kg_catalog_free(L.catalog_build_ec2(abi.build_ec2_catalog(infos[:4], ctx, ctx["limits"], ctx["bandwidth"], oz)))
            raise AssertionError("expected KG_EUNSUPPORTED for %r" % bad)
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            assert "EUNSUPPORTED" in str(e) and "milli" in str(e), (bad, e)
    # behaviour: a Solve over the natively built catalog == a Solve over the uploaded oracle catalog
    ctx = variants[-1]
    types = co.list_instance_types(infos, ctx, ctx["limits"], ctx["bandwidth"], oz)
    prob = Problem(types, wl.mixed_pods(300, 1, zones, families(types)), [wl.default_nodepool()])
    h = L.catalog_build_ec2(abi.build_ec2_catalog(infos, ctx, ctx["limits"], ctx["bandwidth"], oz))
    try:
        ro, rg = prob.run_oracle(), prob.run_library(L, h)
        d = diff_results(ro.decode(), rg.decode())
        assert d == "", d
        cat = abi.build_catalog(types)
        for reqs, requests in (([(wl.L_CT, "In", ["on-demand", "spot"])], {"cpu": 1000, "pods": 1000}),
                               ([(wl.L_CT, "In", ["spot"]), (wl.L_ZONE, "In", [zones[0]])], {"cpu": 500, "pods": 1000})):
            assert L.launch_filter(h, reqs, requests) == loader.launch_filter(cat, reqs, requests, wl.WELL_KNOWN_LABELS)
    finally:
        L.dll.kg_catalog_free(h)
