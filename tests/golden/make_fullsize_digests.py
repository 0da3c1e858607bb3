#!/usr/bin/env python3
"""Generates tests/golden/fullsize_digests.json: the CPU oracle's results on BASELINE.json configs 3 and 5 at FULL size
(50 000 spread pods; 100 000 mixed pods x 4 NodePools x 2000 nodes), as sha256 digests of the decoded results plus a few
plain counts.  The oracle needs ~20 s and ~6.5 min for them, which is why the GPU tests compare against these digests
instead of running it (python tests/golden/make_fullsize_digests.py, from the repo root)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import full_size_problem, result_digest  # noqa: E402

out = {}
for name in ("c3_full", "c5_full"):
    prob = full_size_problem(name)
    t0 = time.time()
    ro = prob.run_oracle()
    d = ro.decode()
    out[name] = {"digest": result_digest(d), "n_pods": len(d["pods"]), "n_claims": len(d["claims"]),
                 "on_existing_nodes": sum(1 for k, _ in d["pods"] if k == 1), "unschedulable": sum(1 for k, _ in d["pods"] if k == 0),
                 "oracle_seconds": round(time.time() - t0, 1)}
    print(name, out[name], flush=True)
with open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
