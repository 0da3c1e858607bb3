"""k_simulate time against the number of single-node simulations in the batch (the first k candidates of the C4 cluster):
the latency of one wave of workgroups, and what co-residency costs."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import abi, lib, workloads as wl
from tests.helpers import catalog776
zones, types = catalog776()
nodes, bound = wl.existing_cluster(5000, 11, types, zones)
order = wl.disruption_order(nodes, bound)
L = lib.Library()
if len(sys.argv) > 1:
    L.set_collect_stats(int(sys.argv[1]))  # 2: the simulations leave time stamps, summarised on stderr
hcat = L.catalog_upload(abi.build_catalog(types))
hcl = L.cluster_upload(abi.build_cluster(nodes, bound))
pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
per_node = {}
for p in bound:
    per_node[p["node"]] = per_node.get(p["node"], 0) + 1
cnt = sorted(per_node.values())
print("pods per node: min %d median %d p90 %d max %d" % (cnt[0], cnt[len(cnt) // 2], cnt[len(cnt) * 9 // 10], cnt[-1]), flush=True)
import random
rnd = random.Random(5)
shuf = list(order)
rnd.shuffle(shuf)  # (the disruption order puts the nodes with few pods first: a prefix of it is not a sample)
ks_ = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 8, 64, 128, 256, 512, 1024, 2048, 5000)
for k in ks_:
    cands = abi.build_candidates(shuf[:k])
    npods = sum(per_node.get(n, 0) for n in shuf[:k])
    ks = []
    for _ in range(4):
        L.consolidate_records([hcat], hcl, pend, nps, cands, False)
        ks.append(L.stats().cons_simulate_ms)
    print("k %5d pods %6d k_simulate ms %s" % (k, npods, " ".join("%.4f" % x for x in ks)), flush=True)
