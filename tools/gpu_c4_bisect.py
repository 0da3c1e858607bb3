"""C4 consolidation rounds (warm plan) through one library build: KG_LIB=path python tools/gpu_c4_bisect.py [catalog: product|oracle]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, lib, problem as pb, workloads as wl
L = lib.Library(path=os.environ.get("KG_LIB"))
if (sys.argv[1:] or ["product"])[0] == "product":
    zones, hcat, types = pb.product_catalog(L, 3)
else:
    from tests.helpers import catalog776
    zones, types = catalog776()
    hcat = L.catalog_upload(abi.build_catalog(types))
nodes, bound = wl.existing_cluster(5000, 11, types, zones)
order = wl.disruption_order(nodes, bound)
cl = abi.build_cluster(nodes, bound)
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
hcl = L.cluster_upload(cl)
cd = abi.build_candidates(order)
L.set_collect_stats(False)
for multi in (False, True):
    ks = []
    for it in range(6):
        recs, dms = L.consolidate_records([hcat], hcl, pend, nps, cd, multi)
        ks.append(L.stats().cons_simulate_ms)
    print(os.environ.get("KG_LIB", "HEAD"), "multi" if multi else "single", "k_simulate ms:", " ".join("%.3f" % k for k in ks), flush=True)
