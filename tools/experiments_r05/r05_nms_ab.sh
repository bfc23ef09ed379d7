#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
SS_LIB_PATH=$PWD/ab_tmp/old.so python tools/nms_time.py 2>&1 | tail -1 | sed 's/^/old /'
python tools/nms_time.py 2>&1 | tail -1 | sed 's/^/new /'
done
timeout 600 python -m pytest tests/test_gpu_front.py -x -q -m gpu 2>&1 | tail -2
