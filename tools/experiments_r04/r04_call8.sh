#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c8; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof; mkdir -p $out/prof
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check > $out/prof/cmd.log 2>&1)
cd $GRAFT_REPO_ROOT
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py "$t" 16 > $out/busy.txt 2>&1; cat $out/busy.txt
grep '"metric"' $out/prof/cmd.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
find $out/prof -name "*.csv" -size +2M -delete
