#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step_distribution']); t=d['throughput_mode']; print('tm', t['frames_per_s'], t['ms_per_step_distribution']); a=d['all_fp32']; print('all32', a['frames_per_s'], a['ms_per_step_distribution'])"
