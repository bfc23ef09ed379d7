#!/bin/bash
# stage cut inside OSNet for the pose presets after their detectors got cheaper (and c1): same-box sweep
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c22; mkdir -p $out; cd $GRAFT_REPO_ROOT; rm -f $out/summary.txt
for p in c5 c6 c1; do for n in 2 4 5; do
  timeout 300 python bench.py --preset $p --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --reid-split $n > $out/b_${p}_$n.json 2> $out/b_${p}_$n.err || echo "rc $? for $p $n" >> $out/summary.txt
  python - $out/b_${p}_$n.json $p $n <<'PY' >> $out/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2],'reid_split',sys.argv[3],'value',d['value'],'ms/step',d['ms_per_step'],'exact',d.get('frames_bit_exact'))
except Exception as e:
    print(sys.argv[2],sys.argv[3],'failed',e)
PY
done; done
cat $out/summary.txt
