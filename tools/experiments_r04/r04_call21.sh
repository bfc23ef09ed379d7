#!/bin/bash
# after the pose / v11 head and block paths: full GPU suite, short bench lines of c5 / c6 / c2
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c21; mkdir -p $out; cd $GRAFT_REPO_ROOT; rm -f $out/summary.txt
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt
tail -6 $out/pytest_gpu.txt
for p in c5 c6 c2 c3; do
  timeout 400 python bench.py --preset $p --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check > $out/bench_$p.json 2> $out/bench_$p.err || echo "rc $? for $p" >> $out/summary.txt
  python - $out/bench_$p.json $p <<'PY' >> $out/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2],'value',d['value'],'ms/step',d['ms_per_step'],'assoc us',r.get('mean_launch_us'),'exact',d.get('frames_bit_exact'), 'net check', {k:v for k,v in (d.get('net_outputs_check') or {}).items() if 'equal' in k})
except Exception as e:
    print(sys.argv[2],'failed',e)
PY
done
cat $out/summary.txt
