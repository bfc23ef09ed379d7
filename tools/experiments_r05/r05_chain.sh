#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sequence.py -q -x > gpurun_out/tseq.log 2>&1; tail -3 gpurun_out/tseq.log
timeout 120 python tools/tracker_only.py 2>/dev/null
