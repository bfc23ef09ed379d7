#!/bin/bash
# rocprofv3 kernel stats + per-network kernel sequences + GPU-busy summary of the default pipeline command (no side legs)
out=$GRAFT_REPO_ROOT/gpurun_out/r05_q; mkdir -p $out; cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_bench; mkdir -p $out/prof_bench
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode > $out/prof_bench/cmd.log 2>&1)
cd $GRAFT_REPO_ROOT
f=$(find $out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv
t=$(find $out/prof_bench -name "*kernel_trace.csv" | head -1)
python tools/detector_sequence.py "$t" > $out/detector_sequence.txt 2>&1
python tools/osnet_sequence.py "$t" > $out/osnet_sequence.txt 2>&1
python tools/trace_busy.py "$t" 16 > $out/gpu_busy.txt 2>&1
find $out/prof_bench -name "*.csv" -size +2M -delete
head -45 $out/gpu_busy.txt
