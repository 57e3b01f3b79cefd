#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of scripts/kbench.py (the two
# closest-hit launches of the bench batch only). Results under gpurun_out/kprof_<tag>/.
set -u
TAG=${1:-k}
REPS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/kbench.py $REPS"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "kernel trace done"
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pmc pass $name done"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass ta TA_BUSY_avr TA_TA_BUSY_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
