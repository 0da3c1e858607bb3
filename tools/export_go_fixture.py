#!/usr/bin/env python3
"""The C2 batch of bench.py and its 776-type catalog as JSON for bench/go_ref.
Usage: python tools/export_go_fixture.py OUT.json [n_pods]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_amd import problem as pb, workloads as wl  # noqa: E402
from tests.helpers import catalog776  # noqa: E402  (the catalog oracle's encoding of the same 776 types: plain dicts)


def qty(res, milli):
    """milli-quantity -> a Kubernetes quantity string (memory-like resources are milli-bytes)."""
    return "%dm" % milli


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    zones, types = catalog776(3)
    pods = wl.mixed_pods(n, 42, zones, pb.families(types))
    fx = {"types": [], "pods": []}
    for t in types:
        fx["types"].append({
            "name": t["name"],
            "requirements": [{"key": k, "op": op, "values": list(vals)} for (k, op, vals) in t["requirements"]],
            "offerings": [{"zone": z, "capacity_type": ct, "price": price, "available": bool(avail)} for (z, ct, _zid, price, avail) in t["offerings"]],
            "capacity": {r: qty(r, v) for r, v in t["capacity"].items() if v},
            "kube_reserved": {r: qty(r, v) for r, v in t["overhead_parts"]["kube"].items()},
            "system_reserved": {r: qty(r, v) for r, v in t["overhead_parts"]["system"].items()},
            "eviction_threshold": {r: qty(r, v) for r, v in t["overhead_parts"]["eviction"].items()},
        })
    for p in pods:
        fx["pods"].append({
            "uid": p["uid"], "creation_ts": p["creation_ts"], "requests": {r: qty(r, v) for r, v in p["requests"].items()},
            "node_selector": p.get("node_selector", {}),
            "required_terms": [[{"key": k, "op": op, "values": list(vals)} for (k, op, vals) in term] for term in p.get("required_terms", [])],
        })
    with open(out, "w") as f:
        json.dump(fx, f)
    print("%s: %d types, %d pods" % (out, len(fx["types"]), len(fx["pods"])))


if __name__ == "__main__":
    main()
