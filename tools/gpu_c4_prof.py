"""Host-side sections of one C4 consolidation round (KG_HOST_PROF=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KG_HOST_PROF"] = "1"
from karpenter_amd import abi, lib, distributed, workloads as wl
from tests.helpers import catalog776
L = lib.Library()
zones, types = catalog776()
hcat = L.catalog_upload(abi.build_catalog(types))
nodes, bound = wl.existing_cluster(5000, 11, types, zones)
order = wl.disruption_order(nodes, bound)
hcl = L.cluster_upload(abi.build_cluster(nodes, bound))
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
for i in range(3):
    sys.stderr.write("---- round %d\n" % i); sys.stderr.flush()
    t0 = time.perf_counter()
    dec, n_local, ms = distributed.consolidate(L, hcat, hcl, pend, nps, order, False, device="cpu")
    print("round %.2f ms, device %.2f ms" % ((time.perf_counter() - t0) * 1e3, ms), flush=True)
