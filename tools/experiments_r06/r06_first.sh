#!/bin/bash
# round 6, first lease: the register-stream fp32 chain kernel (k32_chainsR): numerics, time per 1024 crops per form, kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_nets32.py -x -q -k "chains or block_parts or whole" > gpurun_out/r06_first_tests.txt 2>&1
tail -5 gpurun_out/r06_first_tests.txt
for f in 2 1; do SS32_CHAINS_FORM=$f timeout 300 python tools/osnet32_time.py 20 1024 >> gpurun_out/r06_first_time.txt 2>&1; done
cat gpurun_out/r06_first_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_first_prof -o run -- python $GRAFT_REPO_ROOT/tools/osnet32_eager.py 3 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06_first_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(find gpurun_out/r06_first_prof -name "*kernel_stats.csv" | head -1) 3 30
