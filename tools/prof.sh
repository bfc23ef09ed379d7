#!/bin/bash
# usage: tools/prof.sh <tag> <command...>   -> gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + stats, csv)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $out/cmd.log 2>&1
grep '"metric"' $out/cmd.log | tail -1 > $out/bench.json
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
head -40 "$f"
# keep the merged-back payload small: drop the per-dispatch trace, keep the stats
find $out -name "*kernel_trace.csv" -size +20M -delete
