#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
for m in 0 1 2 3; do RC_EVENT_MODE=$m timeout 300 python3 tools/event_probe.py 2>&1 | grep RC_EVENT; done | tee $O/event_probe.txt
