#!/bin/bash
# detached tracker chain on reserved compute units: same-box A/B of chain_cus = 0 / 8 / 16 / 32 (short bench lines, checker legs off)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c16; mkdir -p $out; cd $GRAFT_REPO_ROOT
for n in 0 -1 0 -1; do
  tag=cus${n}_$(date +%s)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --pipe chain_cus=$n > $out/$tag.json 2> $out/$tag.err || echo "rc $? for $n" >> $out/summary.txt
  python - $out/$tag.json $n <<'PY' >> $out/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('chain_cus',sys.argv[2],'value',d['value'],'ms/step',d['ms_per_step'],'assoc us',r.get('mean_launch_us'),'exact',d.get('frames_bit_exact'),'timed exact',d.get('frames_bit_exact_timed'),'step p50/p95',d.get('step_ms_distribution',{}).get('p50'),d.get('step_ms_distribution',{}).get('p95'))
except Exception as e:
    print('chain_cus',sys.argv[2],'failed',e)
PY
done
cat $out/summary.txt; tail -3 $out/*.err | tail -20
