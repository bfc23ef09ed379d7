#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode"
for i in 1 2; do
python bench.py --steps 30 --warmup 5 $X > gpurun_out/bench_quick_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_quick_$i.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['id_match_rate'], d['frames_bit_exact_timed'], r['mean_launch_us'], r['frac'], r['frac_of_hbm_peak'], r['launches_excluded_as_dispatch_stalls'], d['ms_per_step_distribution'])
PY
done
python bench.py --steps 30 --warmup 5 $X --opt chain_merge=0 > gpurun_out/bench_quick_m0.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_quick_m0.json').read().strip().splitlines()[-1])
print('merge0', d['value'], d['ms_per_step'], d['roofline']['mean_launch_us'])
PY
