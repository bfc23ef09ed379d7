#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 -k "not true_reid_path and not f16_mode" > gpurun_out/t32f.log 2>&1
tail -5 gpurun_out/t32f.log
SS32_CHAINS_PRE=0 python tools/osnet32_time.py 10 1024 2>/dev/null | tee gpurun_out/osnet32_time_pre0.json
SS32_CHAINS_PRE=1 python tools/osnet32_time.py 10 1024 2>/dev/null | tee gpurun_out/osnet32_time_pre1.json
for pre in 0 1; do
SS32_CHAINS_PRE=$pre bash tools/prof.sh osnet32e$pre python tools/osnet32_eager.py 3 1024 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_osnet32e$pre/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k32_chains' in r['Name'] or 'stem' in r['Name']: print(r['Name'][:58].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
