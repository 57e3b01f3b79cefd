#!/bin/bash
# A/B of kernel variants: $1 = variant names (main = renderer_amd/libmirt.so), $2 = scenes, $3 = "pt" to add path-tracing timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
echo "=== $(date +%H:%M:%S) variants: $1" >> gpurun_out/ab.log
bash scripts/ab.sh "$1" "${2:-cornell mesh1m fog}" || { echo "ab failed"; tail -5 gpurun_out/ab.log; exit 1; }
if [ "${3:-}" = "pt" ]; then
  for v in $1; do
    lib=$ROOT/renderer_amd/variants/libmirt_$v.so; [ "$v" = "main" ] && lib=$ROOT/renderer_amd/libmirt.so
    MIRT_LIB=$lib timeout -k 10 200 python3 scripts/ptbench.py 256 2 Config_Glossy1024.xml 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log || exit 1
    MIRT_LIB=$lib timeout -k 10 200 python3 scripts/ptbench.py 256 2 Config_Fog1024.xml 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log || exit 1
  done
fi
tail -n 30 gpurun_out/ab.log
