#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c9; mkdir -p $out; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
timeout 300 $B > $out/bench_default.json 2>$out/bench_default.err
for sp in 0 1 2 3 5; do timeout 300 $B --tracker-stream --reid-split $sp > $out/bench_ts_split$sp.json 2>$out/bench_ts_split$sp.err; done
timeout 300 $B > $out/bench_default_b.json 2>$out/bench_default_b.err
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_distribution",{}).get("p50"), "assoc us", r["mean_launch_us"], "exact", d["frames_bit_exact"])
except Exception as e: print(sys.argv[1], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
