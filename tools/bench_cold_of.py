#!/usr/bin/env python3
"""The cold-round figures of bench.py JSON lines, one file per line: python tools/bench_cold_of.py BENCH.json ..."""
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    c=d['consolidation']
    print(f, "single cold %.2f (upload %.2f) multi cold %.2f dense cold %.2f / %.2f" % (c['single_node']['cold']['ms_per_round'], c['single_node']['cold']['cluster_upload_ms_rank0'], c['multi_node']['cold']['ms_per_round'], c['dense']['single_node']['cold']['ms_per_round'], c['dense']['multi_node']['cold']['ms_per_round']))
