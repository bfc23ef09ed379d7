#!/bin/bash
# after the matrix-core depthwise: full GPU suite, bench (driver form), OSNet time
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c15; mkdir -p $out; cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt
tail -6 $out/pytest_gpu.txt
( time python bench.py --steps 20 --warmup 5 > $out/bench_c2.json 2> $out/bench_c2.err ) 2> $out/bench_time.txt; tail -3 $out/bench_time.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_c15/bench_c2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',d['value'],'ms/step',d['ms_per_step'],'assoc us',r.get('mean_launch_us'),r.get('mean_launch_us_all'),'excluded',r.get('launches_excluded_as_dispatch_stalls'),'frac',r.get('frac'),'traffic',r.get('traffic'))
print('exact',d.get('parity') or d.get('frames_bit_exact'))
for k in ('reid_f16_vs_f32','api_path','tracker_only'):
    if k in d: print(k, json.dumps(d[k])[:700])
PY
