"""Pins the Solve oracle (oracle/oracle.cpp) on the set-level outcomes the reference's own tests assert when they
drive the real core scheduler through the AWS catalog code.  The reference tests cannot
run here (no Go toolchain, core module absent); their inputs are rebuilt from the fixtures they use
(tests/golden/fake_ec2_16.json = pkg/fake/zz_generated.describe_instance_types.go + ec2api.go:555-674 offerings,
fake.MakeInstances() = uniform 2 vCPU / 8 GiB types over the static price table keys) and the asserted outcomes
are checked on the oracle.  Exact pod->claim assignment is NOT asserted by any reference test: PARITY UNPINNED
beyond these outcomes.
"""
import json
import os

import pytest

from karpenter_amd import abi, workloads as wl
from oracle import catalog_oracle as co, loader

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MI, GI = co.MI, co.GI


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fake16_types(**ctx_kw):
    """The provider's default test catalog (pkg/test/environment.go + pkg/fake/ec2api.go defaults), AL2 node class."""
    f = _load("fake_ec2_16.json")
    od = _load("pricing_od.json")["on_demand"]
    zones = [s["zone"] for s in f["subnets"][:3]]  # the test node class selects subnets 1a/1b/1c (suite_test.go:130-143)
    ctx = {"ami_family": "AL2", "region": "us-east-1", "vm_memory_overhead_percent": 0.075, "reserved_enis": 0,
           "subnets": f["subnets"][:3], "od_prices": od,
           "spot_prices": {(i["InstanceType"], z): round(od.get(i["InstanceType"], 1.0) * 0.5, 3) for i in f["instance_types"] for z in zones},
           "unavailable": set()}
    ctx.update(ctx_kw)
    infos = [i for i in f["instance_types"] if i["InstanceType"] in f["offering_zones"]]
    return co.list_instance_types(infos, ctx, _load("vpclimits.json"), _load("bandwidth.json"), f["offering_zones"])


def make_instances_types(names=None, vcpus=None, spot=None):
    """fake.MakeInstances(): uniform x86_64 / 2 vCPU / 8 GiB / 3 ENIs x 10 IPs types, one zone (pkg/fake/utils.go:164-229)."""
    od = _load("pricing_od.json")["on_demand"]
    names = names or sorted(od)
    infos = []
    for n in names:
        infos.append({"InstanceType": n, "ProcessorInfo": {"SupportedArchitectures": ["x86_64"]},
                      "VCpuInfo": {"DefaultVCpus": (vcpus or {}).get(n, 2)}, "MemoryInfo": {"SizeInMiB": 8192},
                      "NetworkInfo": {"Ipv4AddressesPerInterface": 10, "DefaultNetworkCardIndex": 0,
                                      "NetworkCards": [{"NetworkCardIndex": 0, "MaximumNetworkInterfaces": 3}]},
                      "SupportedUsageClasses": ["on-demand", "spot"]})
    ctx = {"ami_family": "AL2", "region": "us-east-1", "vm_memory_overhead_percent": 0.075, "reserved_enis": 0,
           "subnets": [{"zone": "test-zone-1a", "zone_id": "tstz1-1a"}], "od_prices": od,
           "spot_prices": spot if spot is not None else {(n, "test-zone-1a"): od[n] for n in names}, "unavailable": set()}
    return co.list_instance_types(infos, ctx, _load("vpclimits.json"), _load("bandwidth.json"), {n: ["test-zone-1a"] for n in names})


def solve(types, pods, nodepools, max_types=60):
    cat = abi.build_catalog(types)
    pd = abi.build_pods(pods)
    nps = abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, max_types)
    st, rb = loader.solve([cat], None, pd, nps, len(pods))
    assert st == 0
    return rb.decode()


def pod(i, **requests):
    return {"uid": "pod-%02d" % i, "ns": "default", "creation_ts": 1700000000, "labels": {}, "requests": requests}


OD_POOL = [wl.default_nodepool(capacity_types=("on-demand",))]


@pytest.mark.parametrize("resource,type_name,n_nodes", [
    ("nvidia.com/gpu", "p3.8xlarge", 2),          # instancetype/suite_test.go:693-725
    ("habana.ai/gaudi", "dl1.24xlarge", 1),       # :726-756
    ("aws.amazon.com/neuron", "inf1.6xlarge", 2),  # :757-789
    ("amd.com/gpu", "g4ad.16xlarge", 2),          # :852-884
])
def test_accelerator_packing(resource, type_name, n_nodes):
    types = fake16_types()
    names = [t["name"] for t in types]
    for order in ((1, 2, 4), (4, 2, 1), (2, 4, 1)):  # the reference's queue tie-break is the (random) UID: any order must agree
        res = solve(types, [pod(i, **{resource: c * 1000}) for i, c in enumerate(order)], OD_POOL)
        assert all(k == abi.POD_NEW_CLAIM for k, _ in res["pods"])
        assert len({idx for _, idx in res["pods"]}) == n_nodes
        for _, idx in res["pods"]:
            # the node that comes up is of this type: it is the cheapest launchable option of every claim
            assert names[res["claims"][idx]["types"][0]] == type_name


def test_pod_eni_on_t3_large_not_scheduled():
    """instancetype/suite_test.go:376-389: t3.large is not trunking compatible -> a pod-eni request cannot be met."""
    p = pod(0, **{"vpc.amazonaws.com/pod-eni": 1000})
    p["node_selector"] = {wl.L_IT: "t3.large"}
    res = solve(fake16_types(), [p], OD_POOL)
    assert res["pods"][0][0] == abi.POD_UNSCHEDULABLE and not res["claims"]


def test_ephemeral_storage_5000gi():
    """instancetype/suite_test.go:885-907: 5000Gi ephemeral storage does not fit any default 20Gi volume; with
    instanceStorePolicy RAID0 it lands on m6idn.32xlarge."""
    p = pod(0, **{"ephemeral-storage": 5000 * GI * 1000})
    assert solve(fake16_types(), [p], OD_POOL)["pods"][0][0] == abi.POD_UNSCHEDULABLE
    types = fake16_types(instance_store_policy="RAID0")
    res = solve(types, [p], OD_POOL)
    assert res["pods"][0][0] == abi.POD_NEW_CLAIM
    assert [types[t]["name"] for t in res["claims"][0]["types"]] == ["m6idn.32xlarge"]


def test_cheapest_60_of_make_instances():
    """instancetype/suite_test.go:390-434: 1-CPU pod over MakeInstances(): exactly 60 types are sent to CreateFleet,
    all among the 100 cheapest by (price, name)."""
    types = make_instances_types()
    od = _load("pricing_od.json")["on_demand"]
    res = solve(types, [pod(0, cpu=1000)], OD_POOL)
    got = [types[t]["name"] for t in res["claims"][0]["types"]]
    ranked = sorted(od, key=lambda n: (od[n], n))
    assert len(got) == 60
    assert set(got) <= set(ranked[:100])
    assert got == ranked[:60]


def test_min_values_two_nodes():
    """cloudprovider/suite_test.go:285-384: two 0.9-CPU pods, two types (1 vCPU and 8 vCPU), minValues=2 on
    instance-type: the 1-vCPU type must stay among the options, so the pods land on two separate NodeClaims, each
    carrying both types."""
    od = _load("pricing_od.json")["on_demand"]
    fams, names = set(), []
    for n in sorted(od):  # fake.MakeUniqueInstancesAndFamilies(instances, 2)
        if n.split(".")[0] not in fams:
            fams.add(n.split(".")[0])
            names.append(n)
        if len(names) == 2:
            break
    types = make_instances_types(names, vcpus={names[0]: 1, names[1]: 8},
                                 spot={(names[0], "test-zone-1a"): 0.002, (names[1], "test-zone-1a"): 0.003})
    npool = wl.default_nodepool(capacity_types=("spot",), extra_reqs=[(wl.L_IT, "In", names, 2)])
    # daemon overhead is zero here; kube-reserved of the 1-vCPU type is 60m, so 0.9 CPU fits once (940m allocatable)
    res = solve(types, [pod(0, cpu=900), pod(1, cpu=900)], [npool])
    assert [k for k, _ in res["pods"]] == [abi.POD_NEW_CLAIM] * 2
    assert res["pods"][0][1] != res["pods"][1][1]
    for c in res["claims"]:
        assert len(c["types"]) == 2 and not c["truncation_failed"]


def test_min_values_unsatisfiable():
    """Same catalog, minValues=3 on two types: nothing can be scheduled (cloudprovider/suite_test.go minValues family)."""
    od = _load("pricing_od.json")["on_demand"]
    names = [sorted(od)[0], [n for n in sorted(od) if n.split(".")[0] != sorted(od)[0].split(".")[0]][0]]
    types = make_instances_types(names)
    npool = wl.default_nodepool(capacity_types=("spot",), extra_reqs=[(wl.L_IT, "In", names, 3)])
    res = solve(types, [pod(0, cpu=900)], [npool])
    assert res["pods"][0][0] == abi.POD_UNSCHEDULABLE


def test_well_known_label_selectors_schedule():
    """instancetype/suite_test.go:216-270: a pod selecting each well-known label value of g4dn.8xlarge schedules."""
    types = fake16_types()
    g = [t for t in types if t["name"] == "g4dn.8xlarge"][0]
    labels = {k: v[0] for k, op, v in g["requirements"] if op == "In" and len(v) == 1}
    labels[wl.L_ZONE] = "test-zone-1a"
    labels[wl.L_CT] = "on-demand"
    for k, v in labels.items():
        p = pod(0, cpu=100)
        p["node_selector"] = {k: v}
        res = solve(types, [p], OD_POOL)
        assert res["pods"][0][0] == abi.POD_NEW_CLAIM, (k, v)
    p = pod(0, cpu=100)
    p["node_selector"] = {wl.L_IT: "g4dn.8xlarge"}
    res = solve(types, [p], OD_POOL)
    assert [types[t]["name"] for t in res["claims"][0]["types"]] == ["g4dn.8xlarge"]


def test_ice_cache_and_capacity_type_outcomes():
    """instancetype/suite_test.go:1901-2114, 2117-2184 on the oracle (Solve, launch filter, getCapacityType / getOverrides)
    with the fake CreateFleet's behaviour restated in tests/launch_common.py::provision."""
    from tests.launch_common import OracleBackend, check_ice_and_capacity_type
    check_ice_and_capacity_type(OracleBackend)


def test_min_values_createfleet_requests():
    """cloudprovider/suite_test.go:285-590 on the oracle."""
    from tests.launch_common import OracleBackend, check_min_values_launches
    check_min_values_launches(OracleBackend)


@pytest.mark.parametrize("check", range(7))
def test_reference_known_answers(check):
    """instancetype/suite_test.go:603-690, 790-851, 1020-1036, 1053-1160, 1165-1525, 1527-1810, 1812-1898, 2186-2279 on the oracle
    (the same table runs through the C ABI in tests/test_host_logic.py and tests/test_gpu_solve.py)."""
    from tests.pins_common import OracleBackend, _all_checks
    _all_checks()[check](OracleBackend)
