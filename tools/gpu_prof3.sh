#!/bin/bash
# Profile passes of a round (tag = $1): rocprofv3 kernel-trace summary of the default bench command, then counter passes
# (each its own run, no trace flags): FETCH_SIZE, WRITE_SIZE (HBM traffic per launch) and one pass of 8 SQ counters, over a
# bench command whose every launch of a kernel is the same work (C2 Solve + C4 single-node consolidation only).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/pmc_sq_$TAG
PCMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cons single --no-alg-bytes --no-c3c5 --no-dense"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_$TAG.log 2>&1 || { tail -5 $O/prof_$TAG.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$TAG -o f --output-format csv -- $PCMD > $O/pmc_fetch_$TAG.log 2>&1 || { tail -5 $O/pmc_fetch_$TAG.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$TAG -o w --output-format csv -- $PCMD > $O/pmc_write_$TAG.log 2>&1 || { tail -5 $O/pmc_write_$TAG.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq_$TAG -o s --output-format csv -- $PCMD > $O/pmc_sq_$TAG.log 2>&1 || { tail -5 $O/pmc_sq_$TAG.log; exit 1; }
find $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/pmc_sq_$TAG -name "*.csv" | head -20
tail -c 600 $O/prof_$TAG.log
