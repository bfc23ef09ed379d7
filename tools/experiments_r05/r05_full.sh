#!/bin/bash
# the whole GPU suite + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/gputests.log 2>&1
tail -6 gpurun_out/gputests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_c.json 2> gpurun_out/bench_r05_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05_c.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['id_match_rate'], d['roofline']['mean_launch_us'], d['roofline']['frac'])
print(json.dumps({k:v for k,v in d['accuracy_mode'].items() if k not in ('note','net_outputs_check')}))
print(json.dumps({k:v for k,v in (d.get('det_f16_vs_f32') or {}).items() if k!='note'}))
print(d.get('tracker_only')); print(d.get('api_path'))
PY
tail -3 gpurun_out/bench_r05_c.err
