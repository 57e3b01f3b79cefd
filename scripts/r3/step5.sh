#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for sc in cornell mesh1m fog; do
  SCENE=$sc AUDIT=1 MIRT_AUDIT_FAST=1 timeout -k 10 200 python3 scripts/kbench.py 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/audit_r3.log || exit 1
done
cat gpurun_out/audit_r3.log
