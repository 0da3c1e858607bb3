import sys
sys.path.insert(0, '.')
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families
L = lib.Library()
zones, types = catalog776()
for stats in (True, False):
    L.set_collect_stats(stats)
    for n in (1, 2, 3, 5, 8, 12, 20, 40, 80, 300):
        prob = Problem(types, wl.mixed_pods(n, 1, zones, families(types)), [wl.default_nodepool()])
        ro, rg = prob.run_oracle(), prob.run_library(L)
        d = diff_results(ro.decode(), rg.decode())
        print("stats", stats, "n", n, "OK" if d == "" else "DIFF " + d[:300].replace("\n", " | "), flush=True)
