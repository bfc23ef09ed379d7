#!/bin/bash
# A/B of the detect head's grouped launches: ab_tmp/old.so = the library before (split-K body in every group kernel),
# ab_tmp/w0.so = split-K-free group kernels, in-tree = the same + amdgpu_waves_per_eu(6) (80 VGPRs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  SS_LIB_PATH=$PWD/ab_tmp/old.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/old      /'
  SS_LIB_PATH=$PWD/ab_tmp/w0.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/w0  /'
  python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/w6  /'
done
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -3
