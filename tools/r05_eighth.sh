#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 -k "not true_reid_path and not f16_mode" > gpurun_out/t32g.log 2>&1
tail -5 gpurun_out/t32g.log
python tools/osnet32_time.py 10 1024 2>/dev/null | tee gpurun_out/osnet32_time_tail2.json
bash tools/prof.sh osnet32f python tools/osnet32_eager.py 3 1024 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_osnet32f/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k32' in r['Name']: print(r['Name'][:58].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
