#!/bin/bash
# the default line without the side legs, three times (+ the kernel sequences of the two networks)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode"
for i in 1 2 3; do
python bench.py --steps 30 --warmup 5 $X > gpurun_out/bench_quick.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['id_match_rate'], d['frames_bit_exact_timed'], 'assoc', r['mean_launch_us'], r['frac'])
PY
done
python tools/detector_sequence.py > gpurun_out/det_seq.txt 2>&1; tail -50 gpurun_out/det_seq.txt
python tools/osnet_sequence.py > gpurun_out/osnet_seq.txt 2>&1; tail -45 gpurun_out/osnet_seq.txt
