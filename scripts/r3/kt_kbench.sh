#!/bin/bash
# kernel trace --stats of scripts/kbench.py per scene ($1 = scenes) and library variant ($2 = variants, main = libmirt.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${2:-main}; do
lib=$ROOT/renderer_amd/variants/libmirt_$v.so; [ "$v" = "main" ] && lib=$ROOT/renderer_amd/libmirt.so
export MIRT_LIB=$lib
for sc in ${1:-cornell mesh1m}; do
  OUT=$ROOT/gpurun_out/ktk_${sc}_$v
  rm -rf $OUT; mkdir -p $OUT
  SCENE=$sc timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/scripts/kbench.py 10 > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
  echo "== $sc $v"
  python3 - <<PY
import csv,glob,re
for f in sorted(glob.glob("$OUT/kt/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("mirt::","")[:60]
        if "k_" not in n or float(r["Percentage"])<0.2: continue
        print("%-62s calls %6s avg %10.3f us min %10.3f %6.2f%%"%(n,r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["Percentage"])))
PY
done
done
