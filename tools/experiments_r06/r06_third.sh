#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for pr in 0 1; do
  SS32_CHAINS_SKEW=0 SS32_CHAINS_PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_third_prof_$pr -o run -- python $GRAFT_REPO_ROOT/tools/osnet32_eager.py 3 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06_third_prof_$pr.log 2>&1
  echo "== probe $pr"; python $GRAFT_REPO_ROOT/tools/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/r06_third_prof_$pr -name "*kernel_stats.csv" | head -1) 3 40 | grep chainsR
done
