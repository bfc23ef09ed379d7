#!/bin/bash
# tracker-path microbench: tools/mb.sh "<S list>" [extra bench args]
for S in $1; do python bench.py --no-batched --no-nets --streams $S --steps 200 --warmup 10 --no-cpu-baseline --check-frames 120 ${@:2} 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('S=%d fps=%.0f ms/step=%.3f id=%.3f  k_cosine %.2f us  %.0f GB/s frac=%.3f' % (d['config']['streams_per_gpu'], d['value'], d['ms_per_step'], d['id_match_rate'], r['mean_launch_us'], r['achieved'], r['frac']))"; done
