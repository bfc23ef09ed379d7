#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sequence.py tests/test_gpu_cmc.py tests/test_gpu_stages.py -q -x > gpurun_out/tseq.log 2>&1; tail -4 gpurun_out/tseq.log
timeout 120 python tools/tracker_only.py pred_ahead=1 2>/dev/null
timeout 120 python tools/tracker_only.py pred_ahead=0 2>/dev/null
