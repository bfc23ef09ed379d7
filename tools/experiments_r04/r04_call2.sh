#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c2; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sequence.py tests/test_gpu_pipeline.py -x -q -m gpu -k "staging or benchmarked" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
SS_OPTS=assoc_stage=5 SS_TL_DUMP=12 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage5.txt 2>&1
SS_OPTS=assoc_stage=4,assoc_xcd_map=1 SS_TL_DUMP=12 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage4_map1.txt 2>&1
SS_OPTS=assoc_stage=5,assoc_xcd_map=1 SS_TL_DUMP=4 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage5_map1.txt 2>&1
SS_OPTS=assoc_stage=4 SS_TL_DUMP=12 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage4_done.txt 2>&1
for cfg in "4 0" "5 0" "4 1" "5 1" "0 1"; do set -- $cfg
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --opt assoc_stage=$1 --opt assoc_xcd_map=$2 > $out/bench_stage$1_map$2.json 2>$out/bench_stage$1_map$2.err
done
( time timeout 900 python bench.py --steps 20 --warmup 5 --opt assoc_stage=4 > $out/bench_full_stage4.json 2>$out/bench_full_stage4.err ) 2> $out/bench_full_time.txt
tail -3 $out/pytest.txt
for f in $out/bench_stage*.json $out/bench_full_stage4.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "assoc us", r["mean_launch_us"], r.get("launch_us_distribution"), "inkernel", r.get("inkernel_mean_us"), "exact", d["frames_bit_exact"], d.get("ms_per_step_distribution"))
    for k in ("net_outputs_check","reid_f16_vs_f32"):
        if d.get(k): print(k, json.dumps(d[k]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
grep -h "in-kernel duration" $out/timeline_*.txt; cat $out/bench_full_time.txt
