#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sequence.py -q -x > gpurun_out/tseq.log 2>&1; tail -4 gpurun_out/tseq.log
python tools/tracker_only.py chain_merge=1 2>/dev/null
python tools/tracker_only.py chain_merge=0 2>/dev/null
