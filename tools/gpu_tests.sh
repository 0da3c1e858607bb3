#!/bin/bash
# Full GPU parity suite, then the default bench line (tag = $1).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-t}
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 200 --timeout-method thread > $O/${TAG}_pytest.log 2>&1 || { tail -30 $O/${TAG}_pytest.log; exit 1; }
tail -3 $O/${TAG}_pytest.log
timeout -k 10 500 python -u bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
tail -c 3000 $O/${TAG}_bench.json
