#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of a short path-tracing render
# (scripts/ptbench.py 64 spp). Results under gpurun_out/ptprof_<tag>/; summarise with scripts/summarize_profile.py.
set -u
TAG=${1:-pt}
CFG=${2:-Config_Glossy1024.xml}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ptprof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/ptbench.py 64 1 $CFG 0"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pmc pass $name done"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
