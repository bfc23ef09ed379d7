#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c7; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in A B C; do for h in 0 1; do
  SS_LIB_PATH=$GRAFT_REPO_ROOT/exp/libss_$v.so SS_FUSED_HEAD=$h timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1 | sed "s/^/lib$v head$h : /" >> $out/ab.txt
done; done; done
cat $out/ab.txt
