#!/bin/bash
# One GPU session of a round: parity suite, bench line, rocprofv3 kernel-trace summary, HBM traffic PMC passes (each its own run).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r01f}
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -5 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 400 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG $O/pmc_fetch $O/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_$TAG.log 2>&1 || { tail -5 $O/prof_$TAG.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-consolidation > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-consolidation > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
find $O/prof_$TAG $O/pmc_fetch $O/pmc_write -name "*.csv" | head -20
