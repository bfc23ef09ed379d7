#!/bin/bash
# A/B of amdgpu_waves_per_eu on the latency-bound convolution kernels: ab_tmp/base.so = none (besides the grouped head launches),
# ab_tmp/A.so = k_pw family, in-tree = k_pw family + k_osnet_tail + k_bneck<16>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  SS_LIB_PATH=$PWD/ab_tmp/base.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/base /'
  SS_LIB_PATH=$PWD/ab_tmp/A.so python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/A    /'
  python tools/osnet_time.py 40 32 2>&1 | tail -1 | sed 's/^/B    /'
done
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -3
