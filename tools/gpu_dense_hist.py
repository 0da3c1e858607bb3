"""Verdict histogram of single- / multi-node consolidation on dense clusters of several fill ranges (picks the bench's dense leg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, distributed, lib, problem as pb, workloads as wl
L = lib.Library()
zones, hcat, types = pb.product_catalog(L, 3)
L.set_collect_stats(False)
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
for lo, hi in ((0.9, 1.0), (0.95, 1.0), (0.97, 1.0), (0.985, 1.0)):
    nodes, bound = wl.dense_cluster(5000, 17, types, zones, lo=lo, hi=hi)
    order = wl.disruption_order(nodes, bound)
    hcl = L.cluster_upload(abi.build_cluster(nodes, bound))
    for multi in (False, True):
        dec, info = distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, want_info=True)
        print("lo %.3f hi %.3f multi %d: pods %d verdicts [noop, delete, replace] %s decision %s device %.2f ms" % (
            lo, hi, multi, len(bound), info["decisions"], {k: dec[k] for k in ("decision", "n_candidates", "first_candidate")}, info["device_ms"]), flush=True)
    L.dll.kg_cluster_free(hcl)
