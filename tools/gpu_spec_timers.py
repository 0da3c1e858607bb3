"""C2 through the speculative multi-wave kernel with the per-wave cycle accounting (kg_set_collect_stats 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families

zones, types = catalog776()
full = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
ro = full.run_oracle()
for waves in [int(x) for x in (sys.argv[1:] or ["4", "8"])]:
    os.environ["KG_SPEC_WAVES"] = str(waves)
    L = lib.Library()
    for mode in (0, 0, 32, 64):
        L.set_collect_stats(mode)
        rg = full.run_library(L)
        d = diff_results(ro.decode(), rg.decode())
        print("waves %d C2 full mode %d: %s  (%.2f ms, claims %d)" % (waves, mode, "OK" if d == "" else "DIFF " + d[:300], rg.res.solve_ms, rg.res.n_claims), flush=True)
    L.close()
