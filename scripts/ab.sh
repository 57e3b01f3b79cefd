#!/bin/bash
# A/B timing of libmirt variants on the GPU box: usage ab.sh "variant names" [scenes]   (results appended to gpurun_out/ab.log)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $1; do
  for sc in ${2:-cornell mesh1m}; do
    lib=$ROOT/renderer_amd/variants/libmirt_$v.so
    [ "$v" = "main" ] && lib=$ROOT/renderer_amd/libmirt.so
    echo -n "$sc " >> $ROOT/gpurun_out/ab.log
    SCENE=$sc MIRT_LIB=$lib timeout -k 10 200 python3 $ROOT/scripts/kbench.py 10 2>&1 | grep -v amdgpu.ids >> $ROOT/gpurun_out/ab.log || exit 1
  done
done
