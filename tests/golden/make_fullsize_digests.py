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
from tests.helpers import c4_full, full_size_problem, records_digest, result_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
out = json.load(open(OUT)) if os.path.exists(OUT) else {}
names = sys.argv[1:] or ["c3_full", "c5_full", "c4_full", "c4_dense_full"]
for c4name in [n for n in names if n in ("c4_full", "c4_dense_full")]:
    # config 4: every single-node candidate of the 5 000-node cluster + the multi-node prefixes, records and decisions
    from karpenter_amd import abi, workloads as wl
    from oracle import loader
    types, nodes, bound, order = c4_full(dense=c4name == "c4_dense_full")
    cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    ent = {"n_nodes": len(nodes), "n_bound_pods": len(bound)}
    t0 = time.time()
    for multi in (False, True):
        cd = abi.build_candidates(order)
        st, recs, _ = loader.consolidate_records([cat], cl, pend, nps, cd, multi)
        assert st == 0
        # (the selection over the records: first candidate with a command / the binary search -- what ko_consolidate_decide
        # computes sequentially; evaluated here from the records so the 5 000 simulations run once)
        key = "multi" if multi else "single"
        ent[key + "_records"] = len(recs)
        ent[key + "_digest"] = records_digest(recs)
        ent[key + "_decisions"] = [int((recs["decision"] == k).sum()) for k in (0, 1, 2)]
        if multi:
            st, dec = loader.consolidate_decide([cat], cl, pend, nps, cd, True)
            assert st == 0
            ent["multi_decision"] = {k: dec[k] for k in ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price")}
        else:
            win = [r for r in recs if int(r["decision"]) != 0]
            ent["single_first_command"] = int(win[0]["candidate"]) if win else -1
    ent["oracle_seconds"] = round(time.time() - t0, 1)
    out[c4name] = ent
    print(c4name, ent, flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
for name in [n for n in names if n not in ("c4_full", "c4_dense_full")]:
    prob = full_size_problem(name)
    t0 = time.time()
    ro = prob.run_oracle()
    d = ro.decode()
    out[name] = {"digest": result_digest(d), "n_pods": len(d["pods"]), "n_claims": len(d["claims"]),
                 "on_existing_nodes": sum(1 for k, _ in d["pods"] if k == 1), "unschedulable": sum(1 for k, _ in d["pods"] if k == 0),
                 "oracle_seconds": round(time.time() - t0, 1)}
    print(name, out[name], flush=True)
with open(OUT, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
