"""kg_solve_noalloc (include/karpgpu.h): the Solve whose device tables live in the library's static scratch array instead of
pooled hipMalloc buffers.  Called through the C ABI and compared with kg_solve on the same inputs -- bit for bit, pod
assignment, type lists, requirements, requests -- on two calls in a row, the second with a smaller input (the array keeps
the first call's contents), then a batch too large for the array (KG_ENOMEM, nothing launched) and once more after it.
Collected last (pytest -x: a failure here hides nothing behind it)."""
import pytest

from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families, scenario

pytestmark = pytest.mark.gpu


def _both(L, prob, hcat):
    hcl = L.cluster_upload(prob.cluster) if prob.cluster is not None else None
    try:
        a = L.solve([hcat], hcl, prob.pods, prob.nodepools, prob.n_pods)
        b = L.solve([hcat], hcl, prob.pods, prob.nodepools, prob.n_pods, noalloc=True)
    finally:
        if hcl is not None:
            L.dll.kg_cluster_free(hcl)
    d = diff_results(a.decode(), b.decode())
    assert d == "", d
    d = diff_results(prob.run_oracle().decode(), b.decode())  # ... and with the CPU oracle, not only with the other entry point
    assert d == "", d
    assert (a.res.visited_claims, a.res.visited_claim_types, a.res.new_claim_scans, a.res.visited_existing, a.res.relaxations) == \
           (b.res.visited_claims, b.res.visited_claim_types, b.res.new_claim_scans, b.res.visited_existing, b.res.relaxations)
    return a, b


def test_solve_noalloc_matches_solve():
    L = lib.Library()
    assert L.backend() == "hip-gfx950"
    zones, types = catalog776()
    fams = families(types)
    big = Problem(types, wl.mixed_pods(3000, 42, zones, fams), [wl.default_nodepool()])
    small = Problem(types, wl.mixed_pods(300, 7, zones, fams), [wl.default_nodepool()])
    hcat = L.catalog_upload(big.cat)
    try:
        for stats in (True, False):  # exact scan (workgroup kernel), then the production setting (single-wave kernel)
            L.set_collect_stats(stats)
            w0 = L.stats().wave_solves
            a, _ = _both(L, big, hcat)
            assert a.res.n_claims > 50
            _both(L, small, hcat)      # second call, smaller input, same array
            assert (L.stats().wave_solves - w0 == 4) == (not stats)
        # topology groups + existing nodes through the same array
        L.set_collect_stats(False)
        _both(L, scenario("c3_small"), hcat)
        _both(L, scenario("existing_small"), hcat)
        # a batch whose tables do not fit the 32 MiB array is refused, and the array is still good afterwards
        huge = Problem(types, wl.homogeneous_pods(60000), [wl.default_nodepool()])
        with pytest.raises(lib.KgError) as e:
            L.solve([hcat], None, huge.pods, huge.nodepools, huge.n_pods, cap_claims=64, noalloc=True)
        assert e.value.status == 2 and "static device scratch" in str(e.value)
        _both(L, small, hcat)
    finally:
        L.set_collect_stats(True)
        L.dll.kg_catalog_free(hcat)
        L.close()


def test_two_contexts_share_one_gpus_array():
    """Two contexts on ONE GPU call kg_solve_noalloc at the same time (each under its own context lock): the library
    serialises them on the device's scratch mutex, so neither overwrites the other's tables mid-kernel."""
    import threading
    zones, types = catalog776()
    fams = families(types)
    probs = [Problem(types, wl.mixed_pods(1500, 11, zones, fams), [wl.default_nodepool()]),
             Problem(types, wl.mixed_pods(1200, 23, zones, fams), [wl.default_nodepool()])]
    want = [p.run_oracle().decode() for p in probs]
    libs = [lib.Library(device=0), lib.Library(device=0)]
    cats = [L.catalog_upload(p.cat) for L, p in zip(libs, probs)]
    errors = []

    def worker(k):
        try:
            L, prob = libs[k], probs[k]
            L.set_collect_stats(False)
            for _ in range(6):
                r = L.solve([cats[k]], None, prob.pods, prob.nodepools, prob.n_pods, noalloc=True)
                d = diff_results(want[k], r.decode())
                assert d == "", d
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)[:500]))

    try:
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    finally:
        for L, h in zip(libs, cats):
            L.dll.kg_catalog_free(h)
            L.close()
