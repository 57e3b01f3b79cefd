#!/bin/bash
# Runs on the GPU box: kernel trace of scripts/ptbench.py (path tracer, one timed render after a warm-up). Output gpurun_out/ptprof_<tag>/
set -u
TAG=${1:-pt}
SPP=${2:-32}
CFG=${3:-Config_Glossy1024.xml}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ptprof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/scripts/ptbench.py $SPP 1 $CFG > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
tail -n 1 $OUT/kt.log
f=$(ls -t $OUT/kt/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) < 0.3: continue
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("mirt::", "")[:60]
    print("%-62s calls %5s total %9.3f ms avg %10.1f us %6.2f%%" % (n, r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
