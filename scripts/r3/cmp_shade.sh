#!/bin/bash
# k_shade<VOL> under two library variants: kernel trace + VALU counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  OUT=$ROOT/gpurun_out/cmp_$v; rm -rf $OUT; mkdir -p $OUT
  export MIRT_LIB=$ROOT/renderer_amd/variants/libmirt_$v.so
  CMD="python3 $ROOT/scripts/ptbench.py 64 1 Config_Fog1024.xml 0"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1 || exit 1
done
