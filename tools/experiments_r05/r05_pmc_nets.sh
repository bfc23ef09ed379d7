#!/bin/bash
# PMC of the f16 network kernels with the round's final binary (32 frames / 1 024 crops, eager launches): SQ activity, FETCH_SIZE, WRITE_SIZE
cd $GRAFT_REPO_ROOT
PMC_GROUPS=0,3,4 bash tools/pmc_run.sh r05_nets python tools/nets_eager.py 4 32 > gpurun_out/pmc_r05_nets.txt 2>&1
tail -16 gpurun_out/pmc_r05_nets.txt | cut -c1-330
