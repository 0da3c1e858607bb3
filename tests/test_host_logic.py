"""Host compile + solve-core control flow, debugged without a GPU: the TEST-ONLY emulation build
(tests/emu/libkarpgpu_emu.so = the product sources compiled with -DKG_EMU, a workgroup run as a host loop) against
the CPU oracle, bit for bit.  The GPU parity tests proper are tests/test_gpu_*.py (-m gpu)."""
import os

import pytest

from karpenter_amd import lib, workloads as wl
from tests.helpers import EMU_PATH, Problem, catalog776, diff_results, families

pytestmark = pytest.mark.skipif(not os.path.exists(EMU_PATH), reason="emulation build missing: run __graft_entry__.build()")


@pytest.fixture(scope="module")
def L():
    l = lib.Library(EMU_PATH)
    assert l.backend() == "emu"
    yield l
    l.close()


def check(L, prob):
    ro, rg = prob.run_oracle(), prob.run_library(L)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d
    assert (ro.res.visited_claims, ro.res.visited_claim_types, ro.res.new_claim_scans, ro.res.visited_existing, ro.res.relaxations) == \
           (rg.res.visited_claims, rg.res.visited_claim_types, rg.res.new_claim_scans, rg.res.visited_existing, rg.res.relaxations)
    return ro, rg


def test_c1_homogeneous(L):
    zones, types = catalog776()
    ro, _ = check(L, Problem(types, wl.homogeneous_pods(1000), [wl.default_nodepool(capacity_types=("on-demand",))]))
    assert all(k == 2 for k, _ in ro.decode()["pods"])


@pytest.mark.parametrize("n,seed", [(0, 1), (1, 1), (300, 1), (1500, 42)])
def test_c2_mixed(L, n, seed):
    zones, types = catalog776()
    check(L, Problem(types, wl.mixed_pods(n, seed, zones, families(types)), [wl.default_nodepool()]))


def test_small_batches_sweep(L):
    """Every batch size 1..24 (x3 seeds): covers 'the last add() appended / touched a claim after the last sort'
    (Results.NewNodeClaims order), single-pod and two-claim corner cases."""
    zones, types = catalog776()
    fams = families(types)
    for seed in (1, 2, 3):
        for n in range(1, 25):
            check(L, Problem(types, wl.mixed_pods(n, seed, zones, fams), [wl.default_nodepool()]))


@pytest.mark.parametrize("n,seed", [(40, 3), (700, 5), (1500, 42)])
def test_prefilter_path_same_placements(L, n, seed):
    """With the visit counters off the claim scan goes through the conservative prefilter: placements must not change."""
    zones, types = catalog776()
    prob = Problem(types, wl.mixed_pods(n, seed, zones, families(types)), [wl.default_nodepool()])
    ro = prob.run_oracle()
    L.set_collect_stats(False)
    try:
        rg = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d


@pytest.mark.parametrize("name", ["c3_small", "c3_mid", "existing_small", "existing_mid", "c5_small", "host_ports", "host_ports_existing"])
def test_scenarios(L, name):
    """C3 (zone + hostname spread, hostname anti-affinity), existing nodes first (ExistingNode.Add), C5 (weighted
    NodePools with limits, 6 zones, existing nodes charged against the limits)."""
    from tests.helpers import scenario
    ro, _ = check(L, scenario(name))
    if name.startswith("existing") or name.startswith("c5"):
        assert any(k == 1 for k, _ in ro.decode()["pods"])


@pytest.mark.parametrize("name", ["c3_small", "c3_mid", "existing_small", "c5_small", "host_ports", "host_ports_existing"])
def test_scenarios_prefilter_path(L, name):
    """The same scenarios with the visit counters off (the production setting): prefilter + small first pass."""
    from tests.helpers import scenario
    prob = scenario(name)
    ro = prob.run_oracle()
    L.set_collect_stats(False)
    try:
        rg = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d


@pytest.mark.parametrize("name", __import__("tests.helpers", fromlist=["x"]).FEATURE_SCENARIOS)
def test_feature_scenarios(L, name):
    """Scheduler features beyond the BASELINE shapes: every Preferences.Relax step (OR-ed required terms, preferred pod
    (anti-)affinity, preferred node affinity, ScheduleAnyway spreads, PreferNoSchedule toleration), minDomains, daemon
    overhead / daemon requests, tainted nodes with Equal / Exists tolerations, capacity-type spread, required pod affinity,
    NotIn / DoesNotExist / Exists / Gt / Lt selectors -- exact scan (visit counters equal the oracle's) and prefilter path."""
    from tests.helpers import scenario
    prob = scenario(name)
    ro, _ = check(L, prob)
    L.set_collect_stats(False)
    try:
        rg = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d
    assert ro.res.relaxations == rg.res.relaxations
    if name in ("schedule_anyway", "prefer_no_schedule", "preferred_pod_affinity", "two_required_terms", "relax_everything"):
        assert ro.res.relaxations > 0


def test_aged_context_keeps_the_wave_kernel(L):
    from tests.helpers import check_context_age
    check_context_age(L)


def test_more_claims_than_the_lds_scan_order(L):
    """> 2048 NodeClaims in flight: the scan order moves from LDS to global memory mid-solve; both counter settings."""
    from tests.helpers import scenario
    prob = scenario("many_claims")
    ro, rg = check(L, prob)
    assert rg.res.n_claims > 2048
    L.set_collect_stats(False)
    try:
        rg2 = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg2.decode())
    assert d == "", d


def test_launch_filter_and_truncate(L):
    from tests.launch_common import check_launch, check_reference_pins
    check_launch(L)
    check_reference_pins(L)


def test_launch_overrides(L):
    from tests.launch_common import check_overrides
    check_overrides(L)


def test_ice_cache_and_capacity_type_outcomes(L):
    """The reference's ICE-cache and capacity-type scenarios end to end through the C ABI: kg_solve -> kg_launch_filter ->
    kg_launch_overrides -> (fake fleet) -> kg_catalog_update_offerings -> kg_solve."""
    from tests.launch_common import LibraryBackend, check_ice_and_capacity_type
    check_ice_and_capacity_type(lambda types: LibraryBackend(L, types))


def test_min_values_createfleet_requests(L):
    from tests.launch_common import LibraryBackend, check_min_values_launches
    check_min_values_launches(lambda types: LibraryBackend(L, types))


def test_catalog_build_ec2(L):
    from tests.catalog_common import check_ec2_build
    check_ec2_build(L)


@pytest.mark.parametrize("check", range(7))
def test_reference_known_answers(L, check):
    """The reference's own catalog / scheduling known answers (tests/pins_common.py) through the C ABI."""
    from tests.pins_common import LibraryBackend, _all_checks
    _all_checks()[check](LibraryBackend(L))


def test_every_refusal_is_reported_and_leaves_the_context_usable(L):
    from tests.refusal_common import check_refusals
    assert check_refusals(L) >= 20


def test_catalog_update_offerings_in_place(L):
    """ICE marks / spot price moves patched into the resident catalog == fresh upload of the rebuilt catalog == oracle."""
    from tests.update_common import check_updates
    check_updates(L)


def test_multi_catalog_solve(L):
    from tests.multicat_common import check_multi_catalog_solve
    check_multi_catalog_solve(L)


def test_multi_catalog_consolidation(L):
    from tests.multicat_common import check_multi_catalog_consolidation
    check_multi_catalog_consolidation(L)


def test_stale_catalog_generation(L):
    from tests.multicat_common import check_stale_generation
    check_stale_generation(L)


def test_volume_limits(L):
    from tests.volume_common import check_volume_limits
    check_volume_limits(L)


def test_threaded_descriptor_parsing():
    """Pod descriptors of 4 096 pods and more are parsed on several host threads (host_threads.hpp) and a large snapshot's
    Queue ranks on a thread of their own: the same tests in a child process whose environment makes a handful of pods
    'large' (KG_HOST_CHUNK_MIN=2, four threads) must still agree with the oracle bit for bit."""
    import subprocess
    import sys
    env = dict(os.environ, KG_HOST_CHUNK_MIN="2", KG_HOST_THREADS="4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        "tests/test_host_logic.py::test_feature_scenarios", "tests/test_host_logic.py::test_volume_limits",
                        "tests/test_host_logic.py::test_every_refusal_is_reported_and_leaves_the_context_usable",
                        "tests/test_consolidation_host.py::test_consolidation", "tests/test_consolidation_host.py::test_rounds_reuse_the_prepared_snapshot"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
