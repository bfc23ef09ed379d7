#!/bin/bash
# where yolov7 (preset c4) spends its time: eager launches, 32 frames of 1920x1080 letterboxed to 640x384
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c24; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
for m in yolov5n yolov8n-seg; do
  rm -rf $out/prof; mkdir -p $out/prof
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 3 32 $m > $out/prof/log.txt 2>&1)
  f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
  (cd $GRAFT_REPO_ROOT && echo "== $m" && python tools/detector_kernels.py $f 18) > $out/kernels_$m.txt 2>&1
  tail -20 $out/kernels_$m.txt
done
rm -rf $out/prof
