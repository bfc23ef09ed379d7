#!/bin/bash
# A/B of an f16 OSNet kernel change: ab_tmp/old.so = the library before it
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  SS_LIB_PATH=$PWD/ab_tmp/old.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/old /'
  python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/new /'
done
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -3
