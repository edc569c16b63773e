#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python3 tools/order_debug.py > $O/order_debug.txt 2>&1; grep -E "rep|rror" $O/order_debug.txt | cut -c1-400
