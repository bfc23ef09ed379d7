#!/bin/bash
# pose / seg / v11 detectors on the fused paths: parity tests vs the torch modules, kernel listing, graph-replay times
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c20; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py -q -m gpu -x -k "fused_ops or segmentation or detect_head or head_level" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -12 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
for m in yolov8n-pose yolo11n-pose yolo11n; do
  rm -rf $out/prof; mkdir -p $out/prof
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 3 32 $m > $out/prof/log.txt 2>&1)
  f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
  (cd $GRAFT_REPO_ROOT && echo "== $m" && python tools/detector_kernels.py $f 16) > $out/kernels_$m.txt 2>&1
  tail -20 $out/kernels_$m.txt
done
rm -rf $out/prof
