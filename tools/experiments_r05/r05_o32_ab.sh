#!/bin/bash
# fp32 OSNet: ab_tmp/old.so vs in-tree
cd $GRAFT_REPO_ROOT
for i in 1 2; do
SS_LIB_PATH=$PWD/ab_tmp/old.so python tools/osnet32_time.py 10 1024 2>&1 | tail -1 | cut -c1-200 | sed 's/^/old /'
python tools/osnet32_time.py 10 1024 2>&1 | tail -1 | cut -c1-200 | sed 's/^/new /'
done
timeout 900 python -m pytest tests/test_gpu_nets32.py -x -q -m gpu 2>&1 | tail -2
