#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c5; mkdir -p $out; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > $out/bench_$tag.json 2>$out/bench_$tag.err; }
EXTRA="" run head1_a SS_FUSED_HEAD=1
EXTRA="" run head0_a SS_FUSED_HEAD=0
EXTRA="" run head1_b SS_FUSED_HEAD=1
EXTRA="" run head0_b SS_FUSED_HEAD=0
EXTRA="--reid-split 4" run split4 SS_FUSED_HEAD=1
EXTRA="--reid-split 6" run split6 SS_FUSED_HEAD=1
EXTRA="--frame-batch 64" run fb64 SS_FUSED_HEAD=1
EXTRA="--frame-batch 64 --reid-split 4" run fb64_s4 SS_FUSED_HEAD=1
EXTRA="--preset c6" run c6 SS_FUSED_HEAD=1
EXTRA="--preset c1" run c1 SS_FUSED_HEAD=1
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "benchmarked and yolo11" > $out/pytest_c6.txt 2>&1; tail -2 $out/pytest_c6.txt
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "assoc us", r["mean_launch_us"], "enq", d["host_enqueue_ms_per_frame"], d.get("host_enqueue_cpu_ms_per_frame"), "exact", d["frames_bit_exact"], d["net_outputs_check"]["head_tensor_equal_to_eager_rerun"] if d.get("net_outputs_check") else None)
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
