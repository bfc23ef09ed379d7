#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c11; mkdir -p $out; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check"
timeout 300 $B > $out/a_default.json 2>/dev/null
timeout 300 $B --pipe skip_tracker=1 > $out/b_notracker.json 2>/dev/null
timeout 300 $B --pipe skip_tracker=1 --reid-split 2 > $out/c_notracker_split2.json 2>/dev/null
timeout 300 $B --pipe skip_tracker=1 --overlap 0 --frame-batch 32 > $out/d_notracker_seq.json 2>$out/d.err
timeout 300 $B --pipe track_priority=0 > $out/e_noprio.json 2>/dev/null
timeout 300 $B --pipe assoc_gate=0 > $out/f_nogate.json 2>/dev/null
timeout 300 $B --defer-track 0 > $out/g_nodefer.json 2>/dev/null
timeout 300 $B > $out/h_default.json 2>/dev/null
for f in $out/*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]).read().strip().splitlines():
    try:
        d=json.loads(line); r=d.get("roofline") or {}
        print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "assoc us", r.get("mean_launch_us"), "exact", d["frames_bit_exact"])
    except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
