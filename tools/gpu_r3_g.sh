#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 1200 python3 tools/lpt_probe.py > $O/lpt_probe.txt 2>&1; cat $O/lpt_probe.txt
