#!/bin/bash
# C3 blocks (yolov5u) with placed outputs: parity vs the torch modules, kernel listing, short c1 / c6 bench lines
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c28; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nets.py -q -m gpu -k "fused_ops or segmentation" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -4 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof; mkdir -p $out/prof
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 3 32 yolov5n > $out/prof/log.txt 2>&1)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
(cd $GRAFT_REPO_ROOT && python tools/detector_kernels.py $f 14) > $out/kernels_yolov5n.txt 2>&1; tail -16 $out/kernels_yolov5n.txt
rm -rf $out/prof
cd $GRAFT_REPO_ROOT
for p in c1 c6; do timeout 600 python bench.py --preset $p --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check 2>$out/bench_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p value',d['value'],'ms/step',d['ms_per_step'],'exact',d.get('frames_bit_exact'), d['net_outputs_check']['head_tensor_equal_to_eager_rerun'])"; done
