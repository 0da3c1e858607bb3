import sys, time, collections
sys.path.insert(0, '.')
import numpy as np
from karpenter_amd import abi, lib, workloads as wl
from oracle import loader
from tests.helpers import catalog776
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
L = lib.Library()
zones, types = catalog776()
t0 = time.time()
nodes, bound = wl.existing_cluster(n, 11, types, zones)
order = wl.disruption_order(nodes, bound)
cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
print("setup %.1fs, %d nodes, %d bound pods" % (time.time() - t0, len(nodes), len(bound)), flush=True)
cd = abi.build_candidates(order)
for multi in (False, True):
    for it in range(3):
        t0 = time.time()
        recs, dms = L.consolidate_records([hcat], hcl, pend, nps, cd, multi)
        dt = time.time() - t0
        print("multi" if multi else "single", "units", len(recs), "wall %.1f ms device %.1f ms -> %.0f units/s" % (dt * 1e3, dms, len(recs) / dt),
              collections.Counter(int(r["decision"]) for r in recs), flush=True)
# oracle on a bounded sample: the first K candidates
K = 30
cds = abi.build_candidates(order[:K])
t0 = time.time(); st, ro, ms = loader.consolidate_records([cat], cl, pend, nps, cds, False); dt = time.time() - t0
print("oracle single: %d candidates in %.2fs (%.1f ms each, excl. parse %.1f ms)" % (K, dt, dt / K * 1e3, ms / K), flush=True)
rg, _ = L.consolidate_records([hcat], hcl, pend, nps, cds, False)
print("sample parity:", all(abi.record_tuple(a) == abi.record_tuple(b) for a, b in zip(ro, rg)))
