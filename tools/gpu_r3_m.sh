#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 900 python3 tools/big_tlas_work.py > $O/big_tlas_work.txt 2>&1; grep -E "instances|rror" $O/big_tlas_work.txt
