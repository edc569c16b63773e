#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python3 tools/order_debug.py > $O/order_debug.txt 2>&1; grep -E "rep|rror" $O/order_debug.txt | cut -c1-300
