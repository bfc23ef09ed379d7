#!/bin/bash
# detector A/B by ss_op_set_option: 128-wide K chunks (pw_k128 = smallest K that takes them, 0 = off), bottleneck tile rule
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  SS_OP_OPTS=pw_k128=0 python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/k64        /'
  python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/default    /'
  SS_OP_OPTS=bneck_big_min=1024 python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/bneck 1024 /'
  SS_OP_OPTS=bneck_big_min=4096 python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/bneck 4096 /'
done
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -2
