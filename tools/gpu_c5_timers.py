"""C5 at full size through the workgroup kernel with the phase timers (kg_set_collect_stats 2) and per-wave accounting (32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, problem as pb, workloads as wl
L = lib.Library()
z6, hcat6, t6 = pb.product_catalog(L, 6)
nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
p5 = pb.Problem(None, wl.mixed_pods(100000, 99, z6, pb.families(t6)), pb.c5_nodepools(), nodes=nodes, bound=bound)
for mode in (0, 2, 32):
    L.set_collect_stats(mode)
    sys.stderr.write("---- mode %d\n" % mode)
    t = time.perf_counter()
    rg = p5.run_library(L, hcat6)
    print("c5 mode %d: solve %.2f ms wall %.1f ms claims %d" % (mode, rg.res.solve_ms, (time.perf_counter() - t) * 1e3, rg.res.n_claims), flush=True)
