import sys
sys.path.insert(0, '.')
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families
L = lib.Library()
zones, types = catalog776()
prob = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
hcat = L.catalog_upload(prob.cat)
for mode in (0, 2, 4, 6):
    L.set_collect_stats(mode)
    rb = prob.run_library(L, hcat)
    print("mode", mode, "solve_ms %.2f claims %d" % (rb.res.solve_ms, rb.res.n_claims), flush=True)
