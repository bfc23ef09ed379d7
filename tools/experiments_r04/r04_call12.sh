#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c12; mkdir -p $out; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
for i in 1 2 3 4 5 6 7 8; do timeout 300 $B > $out/run$i.json 2>/dev/null; done
for f in $out/run*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], "mean", r["mean_launch_us"], "inkernel", r["inkernel_mean_us"], r["launch_us_in_order"])
PY
done
