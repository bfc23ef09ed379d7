#!/bin/bash
# round 5, first lease: fp32 kernel tests + timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 -k "not true_reid_path and not f16_mode" > gpurun_out/t32.log 2>&1
tail -40 gpurun_out/t32.log
python tools/osnet32_time.py 10 1024 > gpurun_out/osnet32_time.json 2> gpurun_out/osnet32_time.err
cat gpurun_out/osnet32_time.json; tail -3 gpurun_out/osnet32_time.err
