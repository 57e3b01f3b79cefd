#!/bin/bash
# kernel trace --stats of a path-tracing render: $1 tag, $2 config, $3 spp, $4 mode (pt|pttest)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/scripts/ptbench.py ${3:-256} 1 ${2:-Config_Glossy1024.xml} 0 ${4:-pt} > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
python3 - <<PY
import csv,glob,re
for f in sorted(glob.glob("$OUT/kt/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"])<0.3: continue
        n=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("mirt::","")[:60]
        print("%-62s calls %6s total %9.3f ms avg %10.3f us %6.2f%%"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
