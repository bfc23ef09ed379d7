#!/bin/bash
# k_crop_hwc8: base.so = 174 VGPRs (2 waves per SIMD); in-tree = tall boxes staged per 16 output rows, global-load fallback rolled;
# B4.so = the same + waves_per_eu(4) (19-35 spilled registers)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  SS_LIB_PATH=$PWD/ab_tmp/base.so python tools/crop_time.py 2>&1 | tail -1 | sed 's/^/base /'
  python tools/crop_time.py 2>&1 | tail -1 | sed 's/^/new  /'
  SS_LIB_PATH=$PWD/ab_tmp/B4.so python tools/crop_time.py 2>&1 | tail -1 | sed 's/^/new4 /'
done
timeout 900 python -m pytest tests/test_gpu_front.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3
