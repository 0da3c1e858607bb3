import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families
L = lib.Library()
zones, types = catalog776()
prob = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
hcat = L.catalog_upload(prob.cat)
L.set_collect_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rb = prob.run_library(L, hcat)
print("solve_ms %.2f claims %d" % (rb.res.solve_ms, rb.res.n_claims), flush=True)
