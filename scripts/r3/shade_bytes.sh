#!/bin/bash
# HBM-side bytes of every k_shade launch of one 128 Mi-path wavefront (1024^2 x 128 spp in one wavefront)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/shade_bytes; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/ptbench.py 128 0 Config_Glossy1024.xml 128"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
d = sys.argv[1]
def rows(sub):
    f = sorted(glob.glob(d + "/" + sub + "/*/*_counter_collection.csv"))[-1]
    r = [x for x in csv.DictReader(open(f)) if "k_shade" in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return [float(x["Counter_Value"]) for x in r]
kt = sorted(glob.glob(d + "/kt/*/*_kernel_trace.csv"))[-1]
t = [x for x in csv.DictReader(open(kt)) if "k_shade" in x["Kernel_Name"]]
t.sort(key=lambda x: int(x["Start_Timestamp"]))
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in t]
fe, wr = rows("fetch"), rows("write")
for k in range(min(8, len(dur))):
    print("k_shade launch %d: %8.1f us  FETCH raw %8.1f MB  WRITE %8.1f MB" % (k, dur[k], fe[k] / 1024, wr[k] / 1024))
PY
