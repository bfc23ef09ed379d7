#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 -k "not true_reid_path and not f16_mode" > gpurun_out/t32e.log 2>&1
tail -8 gpurun_out/t32e.log
SS32_CHAINS_PRE=0 python tools/osnet32_time.py 10 1024 2>/dev/null | tee gpurun_out/osnet32_time_pre0.json
bash tools/prof.sh osnet32d python tools/osnet32_time.py 5 1024 1 > gpurun_out/prof_osnet32d.log 2>&1
grep crops gpurun_out/prof_osnet32d/cmd.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_osnet32d/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k32' in r['Name']: print(r['Name'][:58].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
