#!/bin/bash
# dev: A/B builds of the library (tools/probes/libs/*.so, tools/probes/build_variants.sh) on one GPU box, base between every two variants:
#   tools/probes/ab_variants.sh "v1 v2 ..." [perf_probe args...]      (restores the product library at the end)
VS=$1; shift
L=raycore.jl_amd/libraycore_mi355x.so
cp $L /tmp/product.so
for round in 1 2; do
  for v in $VS; do
    cp tools/probes/libs/$v.so $L; echo "== $v (round $round)"; RC_PROBE_REPS=${RC_PROBE_REPS:-8} python3 tools/perf_probe.py "$@" 2>&1 | grep "Mrays"
  done
done
cp /tmp/product.so $L
