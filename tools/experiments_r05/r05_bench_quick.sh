#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode"
for o in "" "--opt assoc_pack=0" "" "--opt assoc_pack=0"; do
python bench.py --steps 30 --warmup 5 $X $o > gpurun_out/bench_quick.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1])
r=d['roofline']
print("$o", d['value'], d['ms_per_step'], d['id_match_rate'], d['frames_bit_exact_timed'], 'assoc', r['mean_launch_us'], r['frac'], r['frac_of_hbm_peak'], r['inkernel_mean_us'], r['launches_excluded_as_dispatch_stalls'], r['launch_us_distribution'])
PY
done
