#!/bin/bash
# measurement lines: per-frame call breakdown, streams x frame-batch with the networks on, 8 ranks on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/api_profile.py > gpurun_out/api_profile.txt 2>&1; cat gpurun_out/api_profile.txt | grep -v amdgpu.ids
python tools/api_profile2.py > gpurun_out/api_profile2.txt 2>&1; cat gpurun_out/api_profile2.txt | grep -v amdgpu.ids
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode"
python bench.py --streams 8 --frame-batch 4 --steps 40 --warmup 5 $X > gpurun_out/bench_r05_s8_fb4.json 2>/dev/null
python bench.py --streams 32 --frame-batch 1 --steps 40 --warmup 5 $X > gpurun_out/bench_r05_s32_fb1.json 2>/dev/null
python bench.py --streams 4 --frame-batch 8 --steps 40 --warmup 5 $X > gpurun_out/bench_r05_s4_fb8.json 2>/dev/null
SS_BENCH_SINGLE_DEVICE=1 SS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 10 --warmup 2 $X > gpurun_out/bench_r05_8ranks_one_gpu.json 2> gpurun_out/bench_r05_8ranks_one_gpu.err
python - <<'PY'
import json
for f in ('s8_fb4','s32_fb1','s4_fb8','8ranks_one_gpu'):
    try:
        d=json.loads(open(f'gpurun_out/bench_r05_{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['frames_per_step'], d['id_match_rate'], d['host_enqueue_cpu_ms_per_frame'], d.get('per_rank_value'), d.get('per_rank_host_enqueue_cpu_ms_per_frame'), d.get('cpu_affinity_rank0'))
    except Exception as e:
        print(f, 'failed', e)
PY
tail -5 gpurun_out/bench_r05_8ranks_one_gpu.err
