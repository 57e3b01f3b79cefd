#!/bin/bash
# builds A/B variants of libmirt.so into renderer_amd/variants/: usage build_variants.sh name "-DX=1" name2 "-DY=2" ...
set -e
cd "$(dirname "$0")/../renderer_amd"
mkdir -p variants
make -s -j8 >/dev/null
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Wno-unused-result $defs -c -o variants/$name.o csrc/mirt.hip 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmirt_$name.so variants/$name.o host/*.o
  rm variants/$name.o
  echo built $name
done
