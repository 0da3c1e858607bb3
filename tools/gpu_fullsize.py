"""Time the full-size configs C3 (50k pods, spreads) and C5 (100k pods, 4 NodePools, 2000 nodes) on the GPU."""
import sys, time
sys.path.insert(0, '.')
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families, c5_nodepools
L = lib.Library()
zones, types = catalog776()
which = sys.argv[1:] or ["c3", "c5"]
if "c3" in which:
    t0 = time.time()
    prob = Problem(types, wl.spread_pods(200, 250, 7), [wl.default_nodepool()])
    print("c3 build %.1fs" % (time.time() - t0), flush=True)
    for stats in (False, True):
        L.set_collect_stats(stats)
        t0 = time.time()
        rb = prob.run_library(L)
        print("c3 stats", stats, "wall %.2fs solve_ms %.1f claims %d" % (time.time() - t0, rb.res.solve_ms, rb.res.n_claims), flush=True)
if "c5" in which:
    z6, t6 = catalog776(6)
    t0 = time.time()
    nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
    prob = Problem(t6, wl.mixed_pods(100000, 99, z6, families(t6)), c5_nodepools(), nodes=nodes, bound=bound)
    print("c5 build %.1fs" % (time.time() - t0), flush=True)
    for stats in (False, True) if "c5stats" in which else (False,):
        L.set_collect_stats(stats)
        t0 = time.time()
        rb = prob.run_library(L)
        print("c5 stats", stats, "wall %.2fs solve_ms %.1f claims %d" % (time.time() - t0, rb.res.solve_ms, rb.res.n_claims), flush=True)
