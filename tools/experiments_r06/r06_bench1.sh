#!/bin/bash
# first whole-pipeline lines of round 6: default (fp32 ReID) at stage cuts 1..4, then the full default line
cd $GRAFT_REPO_ROOT
for sp in 1 2 3 4 6; do
  python bench.py --steps 20 --warmup 5 --reid-split $sp --no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode > gpurun_out/r06_bench_split$sp.json 2> gpurun_out/r06_bench_split$sp.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06_bench_split$sp.json"))
print("split $sp", d["value"], d["ms_per_step"], d["id_match_rate"], d["reid_precision"])
PY
done
