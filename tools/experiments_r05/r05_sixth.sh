#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SS32_CHAINS_PRE=0 PMC_GROUPS=0,1,3,4 bash tools/pmc_run.sh osnet32 python tools/osnet32_eager.py 3 1024 > gpurun_out/pmc_osnet32.log 2>&1
cat gpurun_out/pmc_osnet32.log | tail -20
