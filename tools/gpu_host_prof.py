"""Host-side sections of kg_solve on C2 (KG_HOST_PROF=1 prints compile_problem's sections to stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KG_HOST_PROF"] = "1"
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families
L = lib.Library()
zones, types = catalog776()
prob = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
hcat = L.catalog_upload(prob.cat)
L.set_collect_stats(False)
for i in range(3):
    sys.stderr.write("---- run %d\n" % i)
    rb = prob.run_library(L, hcat)
    sys.stderr.flush()
    print("host_compile_ms %.2f decode %.2f solve %.2f" % (rb.res.host_compile_ms, rb.res.host_decode_ms, rb.res.solve_ms), flush=True)
