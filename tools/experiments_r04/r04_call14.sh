#!/bin/bash
# same-box A/B of the chain kernel's step forms: layers skewed by one step or not, scheduling pattern or not
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c14; mkdir -p $out; cd $GRAFT_REPO_ROOT; rm -f $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for lib in default exp/lib_max-ilp_skew0.so exp/lib_max-ilp_skew1.so exp/lib_max-memory-clause_skew0.so; do
  if [ $lib = default ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  for rep in 1 2; do (cd $GRAFT_REPO_ROOT && timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1 | sed "s|^|$lib : |") >> $out/ab.txt; done
  rm -rf $out/prof; mkdir -p $out/prof
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 4 32 > $out/prof/log.txt 2>&1)
  f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
  (cd $GRAFT_REPO_ROOT && python tools/osnet_sequence.py $f | grep "chains\|streams\|stem\|tail<16, 64, 16" | sed "s|^|$lib : |") >> $out/ab.txt
done
rm -rf $out/prof
cat $out/ab.txt
