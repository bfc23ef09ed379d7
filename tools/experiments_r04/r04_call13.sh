#!/bin/bash
# depthwise 3x3 of the LightConv kernels on the matrix cores: parity tests, OSNet time, kernel sequence
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c13; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py -q -m gpu > $out/pytest_nets.txt 2>&1; echo "pytest rc $?" >> $out/pytest_nets.txt
tail -15 $out/pytest_nets.txt
for rep in 1 2 3; do timeout 200 python tools/osnet_time.py 30 32 2>/dev/null | tail -1 >> $out/osnet_time.txt; done
cat $out/osnet_time.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof; mkdir -p $out/prof
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python tools/nets_eager.py 4 32 > $out/prof/log.txt 2>&1)
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT; python tools/osnet_sequence.py $f > $out/osnet_sequence.txt 2>&1; tail -30 $out/osnet_sequence.txt
find $out/prof -name "*.csv" -size +8M -delete
