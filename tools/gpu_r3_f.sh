#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python3 tools/timeline_r3.py > $O/timelines.txt 2>&1; cat $O/timelines.txt
