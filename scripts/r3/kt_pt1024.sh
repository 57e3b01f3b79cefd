#!/bin/bash
# kernel trace of a long chained render (1024^2 x 1024 spp: eight wavefronts of 128 Mi paths)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_pt1024_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/scripts/ptbench.py 1024 1 $2 0 pt > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
f=$(ls -t $OUT/kt/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) < 0.2: continue
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("mirt::", "")[:60]
    print("%-62s calls %5s total %9.3f ms avg %10.1f us %6.2f%%" % (n, r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
