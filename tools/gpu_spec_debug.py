"""Speculative multi-wave kernel vs the oracle on small batches: wave counts, with and without the serial debug mode; then
the C2 batch with the per-wave cycle accounting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families

zones, types = catalog776()
fams = families(types)
full = Problem(types, wl.mixed_pods(10000, 42, zones, fams), [wl.default_nodepool()])
ro_full = full.run_oracle()
for waves in (2, 4, 8):
    os.environ["KG_SPEC_WAVES"] = str(waves)
    L = lib.Library()
    for n, seed in ((5, 1), (40, 2), (700, 5), (3000, 42)):
        prob = Problem(types, wl.mixed_pods(n, seed, zones, fams), [wl.default_nodepool()])
        ro = prob.run_oracle()
        for mode in (128, 0):
            L.set_collect_stats(mode)
            rg = prob.run_library(L)
            d = diff_results(ro.decode(), rg.decode())
            print("waves %d pods %d mode %d: %s  (%.2f ms, claims %d)" % (waves, n, mode, "OK" if d == "" else "DIFF " + d[:300], rg.res.solve_ms, rg.res.n_claims), flush=True)
    for mode in (0, 0, 32, 64):
        L.set_collect_stats(mode)
        rg = full.run_library(L)
        d = diff_results(ro_full.decode(), rg.decode())
        print("waves %d C2 full mode %d: %s  (%.2f ms, claims %d)" % (waves, mode, "OK" if d == "" else "DIFF " + d[:300], rg.res.solve_ms, rg.res.n_claims), flush=True)
    L.close()
