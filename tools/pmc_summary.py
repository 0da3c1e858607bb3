#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each its own run, counter_collection.csv).
Usage: pmc_summary.py FETCH.csv WRITE.csv "command that was profiled" > profiles/NAME_pmc.json
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the
bytes of wide coalesced reads, so it is doubled (an upper bound for narrow accesses)."""
import collections
import csv
import json
import sys

out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + sys.argv[3],
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)", "kernels": {}}
for path, key in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == key:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        d = out["kernels"].setdefault(k, {})
        d[key + "_KiB_avg"] = sum(v) / len(v)
        d["dispatches_" + key] = len(v)
for k, d in out["kernels"].items():
    d["hbm_bytes_per_launch_corrected"] = int(2 * 1024 * d.get("FETCH_SIZE_KiB_avg", 0) + 1024 * d.get("WRITE_SIZE_KiB_avg", 0))
solve = [k for k in out["kernels"] if "k_solve" in k]
if solve:
    wave = [k for k in solve if "k_solve_wave" in k]
    out["k_solve_traffic_bytes"] = max(out["kernels"][k]["hbm_bytes_per_launch_corrected"] for k in (wave or solve))
print(json.dumps(out, indent=1))
