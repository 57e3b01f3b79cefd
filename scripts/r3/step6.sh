#!/bin/bash
# full GPU suite, then kernel + path-tracing timings of the main library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r3_tests.log; exit 1; }
tail -2 gpurun_out/r3_tests.log
bash scripts/r3/step4.sh "main" "cornell mesh1m fog" pt
