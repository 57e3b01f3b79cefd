#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + stats, then PMC passes (each its own run),
# of the default bench.py command. Results under gpurun_out/prof_<tag>/; summarise with
# scripts/summarize_profile.py and commit the summary under profiles/.
set -u
TAG=${1:-r1}
STEPS=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=${3:-}
CMD="python3 $ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu --pt-spp 8 $EXTRA"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "kernel trace done"
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pmc pass $name done"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.csv" | head -40
