#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 600 python3 tools/illum_probe.py > $O/illum.txt 2>&1; grep -E "cost_order|rror" $O/illum.txt
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "illumination or grid or drivers" 2>&1 | tail -3
