#!/bin/bash
# where the two-stage pipeline is cut inside the ReID network (parts before the cut run on the detector's stream), final kernels
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode --steps 30 --warmup 5"
run() { python bench.py $X "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.ljust(20), d['value'], d['ms_per_step'], d['id_match_rate'], d['roofline']['mean_launch_us'])"; }
for i in 1 2; do
run
run --reid-split 3
run --reid-split 4
run --reid-split 6
run --reid-split 7
done
