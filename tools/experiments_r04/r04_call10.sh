#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c10; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sequence.py -x -q -m gpu -k "work_areas or stream_parity or frame_groups_equal" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
for c in 0 32 64 0 32 64; do timeout 300 $B --opt frame_caps=$c >> $out/bench_caps$c.json 2>>$out/bench_caps$c.err; done
for sp in 5 7; do timeout 300 $B --opt frame_caps=32 --reid-split $sp > $out/bench_caps32_split$sp.json 2>/dev/null; done
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]).read().strip().splitlines():
    try:
        d=json.loads(line); r=d["roofline"]
        print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_distribution",{}).get("p50"), "assoc us", r["mean_launch_us"], "exact", d["frames_bit_exact"])
    except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
python tools/batched_assoc.py 1 32 30 1280 720 frame_caps=32 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tracker path frames/s caps32', d['tracker_path_frames_per_s'])"
python tools/batched_assoc.py 1 32 30 1280 720 frame_caps=0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tracker path frames/s caps0', d['tracker_path_frames_per_s'])"
