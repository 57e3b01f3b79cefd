#!/bin/bash
# path-tracing A/B: $1 variants, $2 config, $3 spp, $4 mode
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in $1; do
  lib=$ROOT/renderer_amd/variants/libmirt_$v.so; [ "$v" = "main" ] && lib=$ROOT/renderer_amd/libmirt.so
  MIRT_LIB=$lib timeout -k 10 200 python3 scripts/ptbench.py ${3:-256} 2 ${2:-Config_Fog1024.xml} 0 ${4:-pt} 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log || exit 1
done
tail -n 8 gpurun_out/ab.log
