#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nets32.py -x -q -k "chains or block_parts" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for pr in 0 1; do
  SS32_CHAINS_SKEW=0 SS32_CHAINS_PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_fourth_prof_$pr -o run -- python $GRAFT_REPO_ROOT/tools/osnet32_eager.py 3 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06_fourth_prof_$pr.log 2>&1
  echo "== probe $pr"; python $GRAFT_REPO_ROOT/tools/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/r06_fourth_prof_$pr -name "*kernel_stats.csv" | head -1) 3 40 | grep chainsR
done
cd $GRAFT_REPO_ROOT
SS32_CHAINS_SKEW=0 timeout 300 python tools/osnet32_time.py 20 1024 2>&1 | grep -v amdgpu.ids
