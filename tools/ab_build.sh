#!/bin/bash
# Dev: builds a second copy of the library with extra compiler flags into tools/ab/<name>.so for A/B measurements inside ONE gpurun call
# (box-to-box differences are larger than most kernel changes).   tools/ab_build.sh noprefetch -DRC_NO_RAY_PREFETCH
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/rc_ab_$NAME
rm -rf $B && mkdir -p $B/raycore.jl_amd $B/include
cp -r $ROOT/raycore.jl_amd/csrc $B/raycore.jl_amd/csrc
cp $ROOT/include/*.h $B/include/
rm -f $B/raycore.jl_amd/csrc/*.o
make -C $B/raycore.jl_amd/csrc -j6 FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result $*" > $B/build.log 2>&1
mkdir -p $ROOT/tools/ab
cp $B/raycore.jl_amd/libraycore_mi355x.so $ROOT/tools/ab/$NAME.so
echo built tools/ab/$NAME.so
