#!/bin/bash
# the fp32 detector on its own kernels: tests, eager time per 32 frames (own / library), kernel trace
cd $GRAFT_REPO_ROOT
timeout 1000 python -m pytest tests/test_gpu_detector32.py -x -q 2>&1 | tail -5
python tools/det32_eager.py 5 32 2>&1 | grep "^ok"
SS32_DET=0 python tools/det32_eager.py 5 32 2>&1 | grep "^ok"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_det32own_prof -o run -- python $GRAFT_REPO_ROOT/tools/det32_eager.py 3 32 > $GRAFT_REPO_ROOT/gpurun_out/r06_det32own_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/r06_det32own_prof -name "*kernel_stats.csv" | head -1) 4 25
