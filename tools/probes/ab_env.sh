#!/bin/bash
# dev: one library, several environments (e.g. RC_EVENT_MODE), interleaved on one GPU box:
#   tools/probes/ab_env.sh "RC_EVENT_MODE=0 RC_EVENT_MODE=2 RC_EVENT_MODE=3" [perf_probe args...]
ES=$1; shift
for round in 1 2; do
  for e in $ES; do
    echo "== $e (round $round)"; env $e RC_PROBE_REPS=${RC_PROBE_REPS:-8} python3 tools/perf_probe.py "$@" 2>&1 | grep "Mrays"
  done
done
