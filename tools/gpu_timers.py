"""Per-phase timers of k_solve on C2 (stderr of the library).  usage: gpu_timers.py [lib.so ...] [--modes 2,8]"""
import sys
sys.path.insert(0, '.')
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, families
paths = [a for a in sys.argv[1:] if not a.startswith('--')] or [None]
modes = [0, 2]
for a in sys.argv[1:]:
    if a.startswith('--modes='):
        modes = [int(x) for x in a[8:].split(',')]
zones, types = catalog776()
prob = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
for path in paths:
    print("== lib", path, flush=True)
    L = lib.Library(path)
    hcat = L.catalog_upload(prob.cat)
    for mode in modes:
        L.set_collect_stats(mode)
        rb = prob.run_library(L, hcat)
        sys.stderr.flush()
        print("mode", mode, "solve_ms %.2f claims %d" % (rb.res.solve_ms, rb.res.n_claims), flush=True)
    L.close()
