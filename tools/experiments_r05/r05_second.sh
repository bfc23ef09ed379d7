#!/bin/bash
# round 5, second lease: all fp32 tests (incl. the 150-frame streams), kernel trace of the fp32 OSNet, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 > gpurun_out/t32b.log 2>&1
tail -30 gpurun_out/t32b.log
bash tools/prof.sh osnet32 python tools/osnet32_time.py 3 1024 1 > gpurun_out/prof_osnet32.log 2>&1
f=$(find gpurun_out/prof_osnet32 -name "*kernel_stats.csv" | head -1); grep -E "k32_|Name" "$f" | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path > gpurun_out/bench_r05_a.json 2> gpurun_out/bench_r05_a.err
tail -c 3000 gpurun_out/bench_r05_a.json; tail -5 gpurun_out/bench_r05_a.err
