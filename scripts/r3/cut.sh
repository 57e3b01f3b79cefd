#!/bin/bash
# chain cut depth sweep at the adaptive wavefront width: 1024^2 x 1024 spp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cfg in Config_Glossy1024.xml Config_Fog1024.xml; do
  for cut in 4 5 6 7 8; do
    echo -n "cut $cut " >> gpurun_out/cut.log
    MIRT_CHAIN_CUT=$cut timeout -k 10 200 python3 scripts/ptbench.py 1024 1 $cfg 0 pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/cut.log || exit 1
  done
done
cat gpurun_out/cut.log
