#!/bin/bash
# round 5, third lease: register-streamed chain kernel + 2-workgroup stem: tests, A/B timing, kernel stats, bench line with det_f16_vs_f32
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nets32.py -q --maxfail=30 -k "not true_reid_path and not f16_mode" > gpurun_out/t32c.log 2>&1
tail -25 gpurun_out/t32c.log
SS32_CHAINS_FORM=0 python tools/osnet32_time.py 10 1024 > gpurun_out/osnet32_time_form0.json 2>/dev/null; cat gpurun_out/osnet32_time_form0.json
bash tools/prof.sh osnet32b python tools/osnet32_time.py 5 1024 1 > gpurun_out/prof_osnet32b.log 2>&1
cat gpurun_out/prof_osnet32b/cmd.log | grep crops
f=$(find gpurun_out/prof_osnet32b -name "*kernel_stats.csv" | head -1); grep -E "k32_" "$f" | cut -d, -f1-4 | sed 's/(float const.*)"/"/' | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path > gpurun_out/bench_r05_b.json 2> gpurun_out/bench_r05_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05_b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['id_match_rate'])
print(json.dumps({k:v for k,v in d['accuracy_mode'].items() if k not in ('note','net_outputs_check')}))
print(json.dumps({k:v for k,v in (d.get('det_f16_vs_f32') or {}).items() if k!='note'}))
PY
tail -3 gpurun_out/bench_r05_b.err
