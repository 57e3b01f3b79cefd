#!/bin/bash
# wavefront width sweep: 1024^2 x 1024 spp, spp_per_launch 32 / 64 / 128 (32 / 64 / 128 Mi paths in flight)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cfg in Config_Glossy1024.xml Config_Fog1024.xml; do
  for spl in 32 64 128; do
    timeout -k 10 200 python3 scripts/ptbench.py 1024 1 $cfg $spl pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/width.log || exit 1
  done
done
cat gpurun_out/width.log
