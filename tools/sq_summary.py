#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass (counter_collection.csv): one row per (kernel, counter).
Usage: sq_summary.py COUNTERS.csv "command that was profiled" > profiles/NAME_sq_counters.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(list)
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
print("# rocprofv3 --pmc (one pass, no trace flags) -- %s; values are averages per dispatch, summed over all waves of the dispatch" % sys.argv[2])
print("Kernel,Counter,Dispatches,AveragePerDispatch,WorkgroupSize,LDS,VGPR,AccumVGPR,SGPR,Scratch")
for (k, c), v in sorted(acc.items()):
    if not k.startswith("kg::") and "kg::" not in k:
        continue
    print('"%s",%s,%d,%.1f,%s' % (k, c, len(v), sum(v) / len(v), ",".join(meta[k])))
