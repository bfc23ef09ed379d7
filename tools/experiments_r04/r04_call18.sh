#!/bin/bash
# k_assoc: wave priority of the two workgroups a CU holds (assoc_prio 0 / 1 / 2), same box, the pipeline's own launches
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c18; mkdir -p $out; cd $GRAFT_REPO_ROOT; rm -f $out/summary.txt
for n in 0 1 2 0 1 2; do
  tag=prio${n}_$(date +%s)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --opt assoc_prio=$n > $out/$tag.json 2> $out/$tag.err || echo "rc $? for $n" >> $out/summary.txt
  python - $out/$tag.json $n <<'PY' >> $out/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('assoc_prio',sys.argv[2],'value',d['value'],'assoc us',r.get('mean_launch_us'),'all',r.get('mean_launch_us_all'),'inkernel',r.get('in_kernel_us'),'dist',r.get('launch_us_distribution'),'exact',d.get('frames_bit_exact'))
except Exception as e:
    print('assoc_prio',sys.argv[2],'failed',e)
PY
done
for n in 0 1 2; do SS_OPTS=assoc_prio=$n timeout 120 python tools/batched_assoc.py 1 32 assoc_prio=$n 2>/dev/null | tail -1 | sed "s/^/alone assoc_prio=$n : /" >> $out/summary.txt; done
cat $out/summary.txt
