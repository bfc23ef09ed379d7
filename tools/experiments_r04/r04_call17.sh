#!/bin/bash
# stage cut inside OSNet after the LightConv change: same-box sweep of --reid-split (short lines, checker legs off)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c17; mkdir -p $out; cd $GRAFT_REPO_ROOT; rm -f $out/summary.txt
for n in 5 4 6 3 5 4; do
  tag=split${n}_$(date +%s)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --reid-split $n > $out/$tag.json 2> $out/$tag.err || echo "rc $? for $n" >> $out/summary.txt
  python - $out/$tag.json $n <<'PY' >> $out/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('reid_split',sys.argv[2],'value',d['value'],'ms/step',d['ms_per_step'],'assoc us',r.get('mean_launch_us'),'exact',d.get('frames_bit_exact'))
except Exception as e:
    print('reid_split',sys.argv[2],'failed',e)
PY
done
cat $out/summary.txt
