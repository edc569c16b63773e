#!/bin/bash
# dev: builds of the library with extra -D flags on rc_traverse.hip (the experiments of rc_traverse_core.h), for A/B runs on one GPU box:
#   tools/probes/build_variants.sh name1 "-DRC_EXP_VALU=12" name2 "-DRC_EXP_NOP=4" ...   -> tools/probes/libs/<name>.so
set -e
cd "$(dirname "$0")/../../raycore.jl_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
make -s >/dev/null 2>&1
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c rc_traverse.hip -o /tmp/rc_traverse_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libs/$name.so rc_capi.o rc_build.o /tmp/rc_traverse_$name.o rc_drivers.o rc_bvh4.o rc_collision.o rc_multi.o -ldl -lpthread
  echo "built $name ($defs)"
done
