"""C3 / C5-small through the workgroup kernel with the phase timers (kg_set_collect_stats 2) and per-wave accounting (32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, problem, workloads as wl

L = lib.Library()
zones, hcat, types = problem.product_catalog(L, 3)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    prob = problem.Problem(types, wl.spread_pods(200, 250, 7), [wl.default_nodepool()])
else:
    nodes, bound = wl.existing_cluster(600, 11, types, zones)
    prob = problem.Problem(types, wl.mixed_pods(12000, 99, zones, problem.families(types)), [wl.default_nodepool()], nodes=nodes, bound=bound)
for mode in (0, 0, 2, 32):
    L.set_collect_stats(mode)
    sys.stderr.write("---- mode %d\n" % mode)
    t = time.perf_counter()
    rg = prob.run_library(L, hcat)
    print("%s mode %d: solve %.2f ms wall %.1f ms claims %d" % (which, mode, rg.res.solve_ms, (time.perf_counter() - t) * 1e3, rg.res.n_claims), flush=True)
