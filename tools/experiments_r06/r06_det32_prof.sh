#!/bin/bash
# where does the fp32 detector (PyTorch-ROCm library convolutions) spend its 10 ms per 32 frames?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_det32_prof -o run -- python $GRAFT_REPO_ROOT/tools/det32_eager.py 3 32 > $GRAFT_REPO_ROOT/gpurun_out/r06_det32_prof.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r06_det32_prof.log
python $GRAFT_REPO_ROOT/tools/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/r06_det32_prof -name "*kernel_stats.csv" | head -1) 3 40
