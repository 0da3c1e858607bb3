"""C2 through the speculative kernel with W waves (KG_SPEC_WAVES): k_solve ms and parity.  python tools/gpu_spec_waves.py 4 5 6"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import lib, workloads as wl
from tests.helpers import Problem, catalog776, diff_results, families
zones, types = catalog776()
full = Problem(types, wl.mixed_pods(10000, 42, zones, families(types)), [wl.default_nodepool()])
ro = full.run_oracle()
for waves in [int(x) for x in (sys.argv[1:] or ["4", "5", "6"])]:
    os.environ["KG_SPEC_WAVES"] = str(waves)
    L = lib.Library()
    L.set_collect_stats(0)
    ms = []
    for _ in range(4):
        rg = full.run_library(L)
        ms.append(rg.res.solve_ms)
    d = diff_results(ro.decode(), rg.decode())
    print("waves %d: %s  k_solve ms %s" % (waves, "OK" if d == "" else "DIFF " + d[:200], " ".join("%.2f" % m for m in ms)), flush=True)
    L.close()
