#!/bin/bash
# dev: what a launch costs beyond its kernel (VERDICT r4 #2b / #6) -> $1 (a text file).  40 launches back to back between two events
# (RC_PROBE_LOOP, tools/perf_probe.py), best of 6, library variants interleaved twice on one box:
#   old  = the library before the claim order moved into the launch (k_order_select + count + scatter in front of every launch, an event of
#          its own behind every per-stream resource the launch used: 3 dispatches + 4 event records per launch)
#   new2 = order_select / order_commit inside the launch, rebuild pair only when a recording waits, resources follow the launch's closing event
#   new5 = new2 + the opening event without a system-scope fence (the product)
# then the product under RC_EVENT_MODE (rc_traverse.hip rc_event_mode): 4 = system fence on t0 as well, 0 = product, 1 = no system fence on t1
# either, 2 = no t0 at all, 3 = no events at all
OUT=$1
export RC_PROBE_LOOP=40 RC_PROBE_REPS=6
{
  echo "# tools/probes/launch_fixed_costs.sh  (ms per launch over 40 back-to-back launches, best of 6; see the script's header for the variants)"
  tools/probes/ab_variants.sh "old new2 new5" --variants "cost_order=1" --workloads c2,c3,shadow,c4,r1m 2>&1 | grep -E "==|Mrays"
  tools/probes/ab_env.sh "RC_EVENT_MODE=4 RC_EVENT_MODE=0 RC_EVENT_MODE=1 RC_EVENT_MODE=2 RC_EVENT_MODE=3" --variants "cost_order=1" --workloads c2,c3,shadow,r1m 2>&1 | grep -E "==|Mrays"
  echo "# kernel alone (rc_last_kernel_ms of single launches, best of 12)"
  RC_PROBE_LOOP=0 RC_PROBE_REPS=12 python3 tools/perf_probe.py --variants "cost_order=1" --workloads c2,c3,shadow,c4,r1m 2>&1 | grep Mrays
} > $OUT
