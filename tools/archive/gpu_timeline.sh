#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02timeline; mkdir -p $O
timeout 600 python3 tools/timeline_probe.py > $O/timeline.log 2>&1; cat $O/timeline.log
