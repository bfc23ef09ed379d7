#!/bin/bash
# PMC passes over the fp32 OSNet (eager, 2 passes x 1024 crops): which pipe is busy in each kernel
cd $GRAFT_REPO_ROOT
SS32_CHAINS_SKEW=${SKEW:-0} bash tools/pmc_run.sh osnet32_r06 python tools/osnet32_eager.py 2 1024 2>&1 | grep -v amdgpu.ids | tail -20
cp gpurun_out/pmc_osnet32_r06/summary.json gpurun_out/r06_pmc_osnet32.json
