#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ (kernel stats)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o run -- python bench.py "$@" > $out/bench.log 2>&1
tail -2 $out/bench.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
head -45 "$f"
