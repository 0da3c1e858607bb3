#!/usr/bin/env python3
"""Host-side profile of a COLD consolidation round on the GPU box (KG_HOST_PROF=1): kg_cluster_upload + plan + round of the
C4 cluster, a few repetitions.  python tools/gpu_cold_prof.py [dense]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KG_HOST_PROF"] = "1"
from karpenter_amd import abi, lib, problem, workloads as wl  # noqa: E402

L = lib.Library()
zones, hcat, types = problem.product_catalog(L, 3)
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
nodes, bound = wl.dense_cluster(5000, 17, types, zones, lo=0.9, hi=1.0) if dense else wl.existing_cluster(5000, 11, types, zones)
order = wl.disruption_order(nodes, bound)
L.set_collect_stats(False)
clb = abi.build_cluster(nodes, bound)
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
cd = abi.build_candidates(order)
for i in range(4):
    sys.stderr.write("---- cold round %d\n" % i)
    t = time.perf_counter()
    hcl = L.cluster_upload(clb)
    t1 = time.perf_counter()
    L.consolidate_records([hcat], hcl, pend, nps, cd, False)
    t2 = time.perf_counter()
    L.consolidate_records([hcat], hcl, pend, nps, cd, False)
    t3 = time.perf_counter()
    sys.stderr.write("upload %.2f ms, cold round %.2f ms, warm round %.2f ms\n" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    L.dll.kg_cluster_free(hcl)
