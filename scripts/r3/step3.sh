#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_step3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FF=$ROOT/scripts/experiments/fetch_factor
timeout -k 10 120 $FF > $OUT/fetch_factor_plain.txt 2>&1 || { echo "fetch_factor failed"; cat $OUT/fetch_factor_plain.txt; exit 1; }
cat $OUT/fetch_factor_plain.txt
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/ff_rdreq -- $FF > $OUT/ff_rdreq.log 2>&1 || { echo "ff rdreq failed"; tail -5 $OUT/ff_rdreq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/ff_tcc -- $FF > $OUT/ff_tcc.log 2>&1 || { echo "ff tcc failed"; tail -5 $OUT/ff_tcc.log; }
python3 - <<'PY'
import csv,glob,collections,os
d=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_step3"
for sub in ("ff_rdreq","ff_tcc"):
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/"+sub+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(out):
        if "k_" in k: print(sub,k,{c:round(sum(v)/len(v)) for c,v in out[k].items()})
PY
