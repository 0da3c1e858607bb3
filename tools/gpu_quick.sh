#!/bin/bash
# Quick GPU check of the Solve path: the solve parity tests, then the C2 bench leg alone (tag = $1).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-q}
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_solve.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 150 --timeout-method thread > $O/${TAG}_tests.log 2>&1 || { tail -15 $O/${TAG}_tests.log; exit 1; }
tail -2 $O/${TAG}_tests.log
KG_SPEC_DEBUG=1 timeout -k 10 300 python -u bench.py --no-consolidation --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'breakdown', d['breakdown_ms'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'speedup', d.get('speedup_vs_cpu_baseline'))
"
