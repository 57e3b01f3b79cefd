#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_step2
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_tiles.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_pt.py -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/tests_new.log 2>&1 || { echo "new tests failed"; tail -40 $OUT/tests_new.log; exit 1; }
tail -3 $OUT/tests_new.log
timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/tests.log 2>&1 || { echo "tests failed"; tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json | head -c 6000
