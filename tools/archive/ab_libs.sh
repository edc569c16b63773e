#!/bin/bash
# dev: A/B two builds of the library on one GPU box: tools/ab_libs.sh <variant.so> [perf_probe args...]
# runs tools/perf_probe.py with the product library, then with the variant copied over it, twice each (ABAB), and restores the product.
V=$1; shift
L=raycore.jl_amd/libraycore_mi355x.so
cp $L /tmp/base.so
for round in 1 2; do
  cp /tmp/base.so $L; echo "== base (round $round)"; python tools/perf_probe.py "$@" 2>&1 | grep "Mrays"
  cp $V $L; echo "== variant (round $round)"; python tools/perf_probe.py "$@" 2>&1 | grep "Mrays"
done
cp /tmp/base.so $L
