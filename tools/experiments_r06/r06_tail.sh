#!/bin/bash
# the persistent fp32 tail: numerics, time per 1024 crops, kernel trace
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets32.py -x -q 2>&1 | tail -3
python tools/osnet32_time.py 20 1024 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_tail_prof -o run -- python $GRAFT_REPO_ROOT/tools/osnet32_eager.py 3 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06_tail_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/r06_tail_prof -name "*kernel_stats.csv" | head -1) 3 40 | grep "k32"
