"""Where a single-node simulation's time goes: one Solve shaped like one (5000 existing nodes, a dozen pods) on the
workgroup kernel with the debug phase timers (stderr of the library)."""
import sys
sys.path.insert(0, '.')
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families
zones, types = catalog776()
nodes, bound = wl.existing_cluster(5000, 11, types, zones)
pods = [dict(p, uid="re-%03d" % i, node=-1) for i, p in enumerate(bound[:12])]
for p in pods:
    p.pop("node", None)
prob = Problem(types, pods, [wl.default_nodepool()], nodes=nodes, bound=bound[12:])
L = lib.Library()
hcat = L.catalog_upload(prob.cat)
for mode in (0, 0, 2, 32):
    L.set_collect_stats(mode)
    rb = prob.run_library(L, hcat)
    sys.stderr.flush()
    print("mode", mode, "solve_ms %.4f prepass %.4f compile %.2f claims %d" % (rb.res.solve_ms, rb.res.prepass_ms, rb.res.host_compile_ms, rb.res.n_claims), flush=True)
