#!/bin/bash
# yolov8 neck: the down-path concats as placement — parity, listing, c2 line (short)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c29; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nets.py -q -m gpu -k "fused_ops or segmentation or detector_with or detect_head" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -4 $out/pytest.txt
for rep in 1 2; do timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1; done
for rep in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check 2>$out/bench_c2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 value',d['value'],'ms/step',d['ms_per_step'],'exact',d.get('frames_bit_exact'), d['net_outputs_check']['head_tensor_equal_to_eager_rerun'])"; done
