#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c4; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "head_level or detect_head or detector_with_head or fused_ops" > $out/pytest_head.txt 2>&1; echo "pytest rc $?" >> $out/pytest_head.txt
tail -5 $out/pytest_head.txt
for cfg in "SS_FUSED_HEAD=0" "SS_FUSED_HEAD=1" "SS_FUSED_HEAD=1 SS_HEAD_TILE16=1"; do
  for rep in 1 2; do env $cfg timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1 | sed "s/^/$cfg : /" >> $out/det_time.txt; done
done
cat $out/det_time.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof; mkdir -p $out/prof
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 4 32 > $out/prof/log.txt 2>&1)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT; python tools/detector_sequence.py $f > $out/detector_sequence.txt 2>&1; tail -12 $out/detector_sequence.txt
find $out/prof -name "*.csv" -size +8M -delete
