#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sequence.py -q -x > gpurun_out/tseq2.log 2>&1; tail -6 gpurun_out/tseq2.log
timeout 120 python tools/tracker_only.py assoc_pack=1 2>/dev/null
timeout 120 python tools/tracker_only.py assoc_pack=0 2>/dev/null
