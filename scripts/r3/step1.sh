#!/bin/bash
# round 3, first GPU call: the test suite with the far-origin guard, the FETCH_SIZE calibration for 64-B gathers,
# and baseline kernel timings of the three bench batches. Every GPU step under its own time limit, chained by &&.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_step1
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1 || { echo "tests failed"; tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
FF=$ROOT/scripts/experiments/fetch_factor
timeout -k 10 120 $FF > $OUT/fetch_factor_plain.txt 2>&1 || { echo "fetch_factor failed"; cat $OUT/fetch_factor_plain.txt; exit 1; }
cat $OUT/fetch_factor_plain.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ff_kt -- $FF > $OUT/ff_kt.log 2>&1 || { echo "ff kt failed"; tail -5 $OUT/ff_kt.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ff_fetch -- $FF > $OUT/ff_fetch.log 2>&1 || { echo "ff fetch failed"; tail -5 $OUT/ff_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/ff_rdreq -- $FF > $OUT/ff_rdreq.log 2>&1 || { echo "ff rdreq failed (counter names?)"; tail -5 $OUT/ff_rdreq.log; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ff_tcc -- $FF > $OUT/ff_tcc.log 2>&1 || { echo "ff tcc failed"; tail -5 $OUT/ff_tcc.log; }
timeout -s KILL 60 rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1 || echo "list-avail failed"
grep -i -E "TCC_(EA0_RDREQ|BUBBLE|EA0_WRREQ)" $OUT/list_avail.txt | head -20
cd $ROOT
for sc in cornell mesh1m fog; do
  SCENE=$sc timeout -k 10 200 python3 scripts/kbench.py 10 2>&1 | grep -v amdgpu.ids >> $OUT/kbench.log || { echo "kbench $sc failed"; tail -5 $OUT/kbench.log; exit 1; }
done
cat $OUT/kbench.log
