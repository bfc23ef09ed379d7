#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode"
for sp in 1 2 3 5 7; do
python bench.py --steps 20 --warmup 4 $X --reid-fp32 --reid-split $sp > gpurun_out/bench_sw.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_sw.json').read().strip().splitlines()[-1])
print("fp32 split $sp", d['value'], d['ms_per_step'], d['id_match_rate'], d['roofline']['mean_launch_us'])
PY
done
for sp in 4 6; do
python bench.py --steps 30 --warmup 5 $X --reid-split $sp > gpurun_out/bench_sw.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_sw.json').read().strip().splitlines()[-1])
print("f16 split $sp", d['value'], d['ms_per_step'], d['id_match_rate'], d['roofline']['mean_launch_us'])
PY
done
