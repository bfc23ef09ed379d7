#!/bin/bash
# pipeline switches around the tracker call, with the round's final kernels (chain 20 us per frame): default vs each switch flipped
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode --steps 30 --warmup 5"
run() { python bench.py $X "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.ljust(28), d['value'], d['ms_per_step'], d['id_match_rate'], d['roofline']['mean_launch_us'])"; }
for i in 1 2; do
run
run --pipe assoc_gate=0
run --pipe track_priority=0
run --defer-track 0
done
