#!/usr/bin/env python3
"""kg_cluster_upload of the C4 snapshot (5 000 nodes, ~60 000 bound pods) on the GPU box's host cores: wall time per call for
KG_HOST_THREADS given in the environment, and KG_HOST_PROF sections of the last call.  python tools/gpu_upload_prof.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, lib, problem, workloads as wl  # noqa: E402

L = lib.Library()
zones, hcat, types = problem.product_catalog(L, 3)
nodes, bound = wl.existing_cluster(5000, 11, types, zones)
clb = abi.build_cluster(nodes, bound)
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
cd = abi.build_candidates(wl.disruption_order(nodes, bound))
ts = []
for i in range(10):
    t = time.perf_counter()
    hcl = L.cluster_upload(clb)
    t1 = time.perf_counter()
    L.consolidate_records([hcat], hcl, pend, nps, cd, False)
    t2 = time.perf_counter()
    L.dll.kg_cluster_free(hcl)
    t3 = time.perf_counter()
    ts.append("%.2f/%.2f/%.2f" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("threads %s: upload/cold round/free ms: %s" % (os.environ.get("KG_HOST_THREADS", "default"), " ".join(ts)), flush=True)
