#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02streams; mkdir -p $O
timeout 600 python3 tools/stream_overlap_probe.py > $O/streams.log 2>&1; cat $O/streams.log
