#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_boundary.py -q -x > gpurun_out/tbound.log 2>&1; tail -4 gpurun_out/tbound.log
timeout 300 python tools/api_rates.py c2 2 2>/dev/null | tee gpurun_out/api_rates.json
