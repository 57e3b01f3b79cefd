#!/bin/bash
# Runs on the GPU box: kernel trace of scripts/ptbench.py (one timed render). Output gpurun_out/ptprof_<tag>/
set -u
TAG=${1:-pt}
SPP=${2:-32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ptprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/scripts/ptbench.py $SPP 1 > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
tail -2 $OUT/kt.log
find $OUT -name "*kernel_trace.csv" | head
