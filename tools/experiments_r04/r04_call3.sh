#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c3; mkdir -p $out; cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.txt 2>&1 ) 2> $out/pytest_time.txt; echo "pytest rc $?" >> $out/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --opt track_graph=0 > $out/bench_nograph.json 2>$out/bench_nograph.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check > $out/bench_graph.json 2>$out/bench_graph.err
( time timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_full.json 2>$out/bench_full.err ) 2> $out/bench_full_time.txt
tail -4 $out/pytest_gpu.txt; cat $out/pytest_time.txt
for f in $out/bench_nograph.json $out/bench_graph.json $out/bench_full.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "assoc us", r["mean_launch_us"], r.get("launch_us_distribution"), "host enq ms/frame", d["host_enqueue_ms_per_frame"], "exact", d["frames_bit_exact"])
    for k in ("tracker_only","reid_f16_vs_f32","api_path"):
        if d.get(k): print(k, json.dumps(d[k]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $out/bench_full_time.txt; tail -3 $out/bench_full.err
