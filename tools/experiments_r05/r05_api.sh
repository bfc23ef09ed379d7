#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_boundary.py -q -x > gpurun_out/tbound.log 2>&1; tail -3 gpurun_out/tbound.log
python tools/api_profile.py 2>&1 | grep -v amdgpu.ids
python tools/api_profile2.py 2>&1 | grep -v amdgpu.ids
