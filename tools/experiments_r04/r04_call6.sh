#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c6; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu > $out/pytest_nets.txt 2>&1; echo "pytest rc $?" >> $out/pytest_nets.txt
tail -3 $out/pytest_nets.txt
for cfg in "SS_FUSED_HEAD=0" "SS_FUSED_HEAD=1"; do
  for rep in 1 2; do env $cfg timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1 | sed "s/^/$cfg : /" >> $out/det_time.txt; done
done
cat $out/det_time.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
for t in head0_a head1_a head0_b head1_b; do
  h=${t:4:1}; timeout 300 $B --fused HEAD=$h > $out/bench_$t.json 2>$out/bench_$t.err
done
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "assoc us", r["mean_launch_us"], "exact", d["frames_bit_exact"], d["net_outputs_check"]["head_tensor_equal_to_eager_rerun"] if d.get("net_outputs_check") else None)
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
