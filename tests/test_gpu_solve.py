"""GPU parity: kg_solve through the C ABI of libkarpgpu.so vs the CPU oracle, bit for bit
(pod -> node/NodeClaim assignment, price-ordered truncated type lists, claim requirements and
requests, visit counters)."""
import pytest

from karpenter_amd import lib, workloads as wl
from tests.helpers import FEATURE_SCENARIOS, Problem, catalog776, diff_results, families, scenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    l = lib.Library()
    assert l.backend() == "hip-gfx950"
    yield l
    l.close()


def check(L, prob):
    ro, rg = prob.run_oracle(), prob.run_library(L)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d
    assert (ro.res.visited_claims, ro.res.visited_claim_types, ro.res.new_claim_scans, ro.res.visited_existing) == \
           (rg.res.visited_claims, rg.res.visited_claim_types, rg.res.new_claim_scans, rg.res.visited_existing)
    return ro, rg


def test_c1_homogeneous(L):
    zones, types = catalog776()
    check(L, Problem(types, wl.homogeneous_pods(1000), [wl.default_nodepool(capacity_types=("on-demand",))]))


@pytest.mark.parametrize("n,seed", [(300, 1), (2000, 42)])
def test_c2_mixed(L, n, seed):
    zones, types = catalog776()
    check(L, Problem(types, wl.mixed_pods(n, seed, zones, families(types)), [wl.default_nodepool()]))


def test_small_batches_sweep(L):
    zones, types = catalog776()
    fams = families(types)
    for seed in (1, 2):
        for n in (1, 2, 3, 4, 5, 6, 7, 9, 13, 21, 34, 55):
            check(L, Problem(types, wl.mixed_pods(n, seed, zones, fams), [wl.default_nodepool()]))


def test_stats_off_same_result(L):
    """The visit counters are optional: placements must not depend on them."""
    zones, types = catalog776()
    prob = Problem(types, wl.mixed_pods(700, 5, zones, families(types)), [wl.default_nodepool()])
    ro = prob.run_oracle()
    L.set_collect_stats(False)
    try:
        rg = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d


def test_c2_full(L):
    zones, types = catalog776()
    ro, rg = check(L, Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()]))
    print("C2 full: oracle %.1f ms, device %.2f ms" % (ro.res.device_ms, rg.res.device_ms))


@pytest.mark.parametrize("name", ["c3_small", "c3_mid", "c3_large", "existing_small", "existing_mid", "existing_large", "c5_small", "c5_mid", "host_ports", "host_ports_existing"])
def test_scenarios(L, name):
    """C3 topology spread / anti-affinity, existing nodes, C5 weighted NodePools with limits -- bit for bit vs the oracle."""
    check(L, scenario(name))


@pytest.mark.parametrize("name", ["c3_small", "c3_mid", "c3_large", "existing_small", "existing_mid", "existing_large", "c5_small", "c5_mid", "host_ports", "host_ports_existing"])
def test_scenarios_prefilter_path(L, name):
    """The same scenarios with the visit counters off (the production setting): prefilter + small first pass."""
    prob = scenario(name)
    ro = prob.run_oracle()
    L.set_collect_stats(False)
    try:
        rg = prob.run_library(L)
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d


@pytest.mark.parametrize("name", FEATURE_SCENARIOS)
def test_feature_scenarios(L, name):
    """Every Preferences.Relax step, minDomains, daemon overhead / requests, node taints and tolerations, capacity-type
    spread, required pod affinity, OR-ed node-affinity terms, negative operators (tests/helpers.feature_scenario): exact scan
    with the visit counters compared, then the production setting (prefilter; the single-wave kernel where it applies)."""
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


def test_more_claims_than_the_lds_scan_order(L):
    """> 2048 NodeClaims in flight: the scan order moves from LDS to global memory mid-solve; both counter settings."""
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


@pytest.fixture()
def LW(L):
    """The wave-kernel tests run on the module's long-lived context: kernel choice depends on the batch and the catalog
    (solve-local value numbering), not on the label values earlier tests fed the context."""
    return L


def test_aged_context_keeps_the_wave_kernel(L):
    from tests.helpers import check_context_age
    check_context_age(L)


def _check_wave(L, prob, expect_wave=True, expect_spec=None):
    """Visit counters off = the production setting; batches without topology groups / existing nodes / minValues are
    served by a barrier-free wave kernel -- the speculative multi-wave one (k_solve_spec: several pods of the Queue in
    flight, retired in Queue order) unless a requirement set holds a negative operator on a custom label key, and the
    single-wave one (k_solve_wave; forced here with debug bit 64) -- everything else by the workgroup kernel: the same
    results all three ways."""
    ro = prob.run_oracle()
    rg = None
    for mode in (0, 64):
        L.set_collect_stats(mode)
        try:
            w0, s0 = L.stats().wave_solves, L.stats().spec_solves
            rg = prob.run_library(L)
            took_wave, took_spec = L.stats().wave_solves > w0, L.stats().spec_solves > s0
        finally:
            L.set_collect_stats(True)
        d = diff_results(ro.decode(), rg.decode())
        assert d == "", ("mode %d" % mode, d)
        assert took_wave == expect_wave
        assert not (took_spec and mode == 64)
        if expect_spec is not None and mode == 0:
            assert took_spec == expect_spec
        assert ro.res.relaxations == rg.res.relaxations
    return ro, rg


def test_wave_kernel_c1_c2(LW):
    zones, types = catalog776()
    _check_wave(LW, Problem(types, wl.homogeneous_pods(1000), [wl.default_nodepool(capacity_types=("on-demand",))]), expect_spec=True)
    fams = families(types)
    for n, seed in ((1, 1), (2, 1), (3, 4), (7, 2), (8, 3), (9, 3), (55, 2), (300, 1), (2000, 42)):
        _check_wave(LW, Problem(types, wl.mixed_pods(n, seed, zones, fams), [wl.default_nodepool()]), expect_spec=True)


def test_wave_kernel_c2_full(LW):
    zones, types = catalog776()
    ro, rg = _check_wave(LW, Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()]), expect_spec=True)
    print("C2 full, single-wave kernel: device %.2f ms" % rg.res.device_ms)


def test_wave_kernel_weighted_nodepools_with_limits(LW):
    """Fresh NodeClaims from four weighted NodePools, two of them with cpu limits that run out (subtractMax), 6 zones."""
    z6, t6 = catalog776(6)
    from tests.helpers import c5_nodepools
    _check_wave(LW, Problem(t6, wl.mixed_pods(6000, 99, z6, families(t6)), c5_nodepools()), expect_spec=True)


def test_wave_kernel_host_ports(LW):
    _check_wave(LW, scenario("host_ports"))


def test_wave_kernel_unschedulable_and_relaxation(LW):
    """Pods nothing can hold (relax / re-push / lastLen stop), preferred terms that relax away, a tainted NodePool."""
    zones, types = catalog776()
    pods = wl.mixed_pods(400, 31, zones, families(types))
    for i, p in enumerate(pods):
        if i % 37 == 0:
            p["requests"]["cpu"] = 10_000_000  # larger than any instance type
        if i % 11 == 0:
            p["preferred_terms"] = [(10, [("karpenter.k8s.aws/instance-family", "In", ["nosuchfamily"])]), (5, [(wl.L_ARCH, "In", ["arm64"])])]
    tainted = wl.default_nodepool("tainted", weight=10)
    tainted["taints"] = [("dedicated", "x", "NoSchedule")]
    _check_wave(LW, Problem(types, pods, [tainted, wl.default_nodepool()]))


@pytest.mark.parametrize("name", ["prefer_no_schedule", "two_required_terms", "negative_selectors"])
def test_wave_kernel_feature_scenarios(LW, name):
    """The feature scenarios without topology groups / existing nodes run on the single-wave kernel: PreferNoSchedule
    toleration relax, OR-ed required terms, NotIn / DoesNotExist / Exists / Gt / Lt."""
    _check_wave(LW, scenario(name))


def test_spec_kernel_negative_operators_and_relaxation(LW):
    """The speculative multi-wave kernel on NotIn / DoesNotExist / Exists / Gt / Lt over well-known keys, and on a batch
    whose pods fail, relax and are pushed back while other pods of the Queue are in flight."""
    _check_wave(LW, scenario("negative_selectors_wk"), expect_spec=True)
    _check_wave(LW, scenario("prefer_no_schedule"), expect_spec=True)
    _check_wave(LW, scenario("two_required_terms"), expect_spec=True)


def test_wave_kernel_scan_order_beyond_lds(LW):
    """> 4096 NodeClaims in flight (one 400-vCPU pod each), then small pods that land on them: the in-place scan order
    spills from LDS into global memory."""
    zones, types = catalog776()
    big = [{"uid": "big-%05d" % i, "ns": "default", "creation_ts": 1700000000, "labels": {"app": "big"},
            "requests": {"cpu": 400000, "memory": 64 * wl.GI * 1000}} for i in range(4300)]
    small = wl.mixed_pods(600, 8, zones, families(types), prefix="small")
    ro, rg = _check_wave(LW, Problem(types, big + small, [wl.default_nodepool()]))
    assert rg.res.n_claims > 4096


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


def test_threaded_abi_on_hip_streams():
    """tests/native/abi_threads.cpp (four solve threads on one context, consolidation rounds on a second, catalog readers, an
    in-place catalog writer, a three-context device group -- every result equal to the serial run) linked against the PRODUCT
    library: real HIP streams and events instead of the emulation HAL the sanitizer builds use."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "_hip", "abi_threads_hip")
    assert os.path.exists(exe), "tests/native/_hip/abi_threads_hip missing: run __graft_entry__.build()"
    r = subprocess.run([exe, "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and "abi_threads ok" in out, out[-4000:]


def test_speculative_kernel_falls_back(L):
    """A speculative solve whose waves gave up waiting is run again on the single-wave kernel inside the same call (debug
    bit 256 pretends every speculative solve ended that way): same results, kg_stats.spec_fallbacks counts it."""
    zones, types = catalog776()
    prob = Problem(types, wl.mixed_pods(1500, 5, zones, families(types)), [wl.default_nodepool()])
    ro = prob.run_oracle()
    try:
        L.set_collect_stats(256)
        f0, s0 = L.stats().spec_fallbacks, L.stats().spec_solves
        rg = prob.run_library(L)
        assert L.stats().spec_solves == s0 + 1 and L.stats().spec_fallbacks == f0 + 1
    finally:
        L.set_collect_stats(True)
    assert diff_results(ro.decode(), rg.decode()) == ""
