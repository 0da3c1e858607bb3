"""C2 through the single-wave kernel (debug bit 64 forces it) and the default path: k_solve ms and parity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families
zones, types = catalog776()
full = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
ro = full.run_oracle()
L = lib.Library()
for mode in (64, 0):
    L.set_collect_stats(mode)
    ms = []
    for _ in range(4):
        rg = full.run_library(L)
        ms.append(rg.res.solve_ms)
    d = diff_results(ro.decode(), rg.decode())
    print("mode %d (%s): %s  k_solve ms %s" % (mode, "single wave" if mode else "speculative", "OK" if d == "" else "DIFF " + d[:200], " ".join("%.2f" % m for m in ms)), flush=True)
L.close()
