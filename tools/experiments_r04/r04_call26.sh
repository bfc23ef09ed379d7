#!/bin/bash
# C2PSA attention in one launch: parity tests, yolo11n-pose kernel listing, short c6 bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c26; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nets.py -q -m gpu -k "psa or fused_ops or segmentation or extra_rows" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -12 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof; mkdir -p $out/prof
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 3 32 yolo11n-pose > $out/prof/log.txt 2>&1)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
(cd $GRAFT_REPO_ROOT && python tools/detector_kernels.py $f 22) > $out/kernels_yolo11n-pose.txt 2>&1; tail -24 $out/kernels_yolo11n-pose.txt
rm -rf $out/prof
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --preset c6 --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check 2>$out/bench_c6.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c6 value',d['value'],'ms/step',d['ms_per_step'],'exact',d.get('frames_bit_exact'), d['net_outputs_check']['head_tensor_equal_to_eager_rerun'])"
