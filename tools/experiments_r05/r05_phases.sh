#!/bin/bash
# tracker path alone: ab_tmp/hoist.so (before) vs in-tree, twice each; then the sequence tests
cd $GRAFT_REPO_ROOT
for i in 1 2; do
SS_LIB_PATH=$PWD/ab_tmp/hoist.so python tools/tracker_only.py 2>&1 | tail -1 | cut -c1-200 | sed 's/^/before /'
python tools/tracker_only.py 2>&1 | tail -1 | cut -c1-200 | sed 's/^/after  /'
done
timeout 1200 python -m pytest tests/test_gpu_sequence.py -x -q -m gpu 2>&1 | tail -2
