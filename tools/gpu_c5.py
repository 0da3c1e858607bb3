"""C5 (and C3) at full size through kg_solve: k_solve ms.  python tools/gpu_c5.py [c3] [c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, lib, problem as pb, workloads as wl
L = lib.Library(path=os.environ.get("KG_LIB"))
L.set_collect_stats(False)
which = sys.argv[1:] or ["c3", "c5"]
if "c3" in which:
    zones, hcat, types = pb.product_catalog(L, 3)
    p3 = pb.Problem(None, wl.spread_pods(200, 250, 7), [wl.default_nodepool()])
    for _ in range(3):
        rb = p3.run_library(L, hcat)
        print("C3 k_solve %.1f ms, host compile %.1f, decode %.1f, claims %d" % (rb.res.solve_ms, rb.res.host_compile_ms, rb.res.host_decode_ms, rb.res.n_claims), flush=True)
if "c5" in which:
    z6, hcat6, t6 = pb.product_catalog(L, 6)
    nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
    p5 = pb.Problem(None, wl.mixed_pods(100000, 99, z6, pb.families(t6)), pb.c5_nodepools(), nodes=nodes, bound=bound)
    for _ in range(2):
        rb = p5.run_library(L, hcat6)
        print("C5 k_solve %.1f ms, host compile %.1f, decode %.1f, claims %d" % (rb.res.solve_ms, rb.res.host_compile_ms, rb.res.host_decode_ms, rb.res.n_claims), flush=True)
