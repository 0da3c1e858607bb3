import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, lib, problem as pb, workloads as wl
L = lib.Library()
z6, hcat6, t6 = pb.product_catalog(L, 6)
nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
p5 = pb.Problem(None, wl.mixed_pods(100000, 99, z6, pb.families(t6)), pb.c5_nodepools(), nodes=nodes, bound=bound)
L.set_collect_stats(True)
t = time.perf_counter()
rb = p5.run_library(L, hcat6)
print("C5 with visit counters: %.1f s wall, k_solve %.1f ms, alg bytes %.0f" % (time.perf_counter() - t, rb.res.solve_ms, rb.res.algorithmic_bytes), flush=True)
