#!/bin/bash
# A/B of the f16 OSNet tail (shortcut weights + biases staged through LDS): ab_tmp/old.so = the library before the change
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  SS_LIB_PATH=$PWD/ab_tmp/old.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/old /'
  python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/new /'
done
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -3
