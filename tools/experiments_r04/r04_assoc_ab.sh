#!/bin/bash
# Round-4 A/B of the association kernel's operand staging (assoc_stage 0 / 1 / 2 / 4) and of the kernel-argument placement:
# correctness of every variant, in-kernel timelines, tracker-only launch times, the bench line per variant.
# usage (GPU box): bash tools/experiments_r04/r04_assoc_ab.sh        -> gpurun_out/r04_ab/
out=$GRAFT_REPO_ROOT/gpurun_out/r04_ab; mkdir -p $out; cd $GRAFT_REPO_ROOT
nproc > $out/nproc.txt
timeout 600 python -m pytest tests/test_gpu_sequence.py -x -q -m gpu -k "staging or frame_groups_equal" > $out/pytest_staging.txt 2>&1; echo "pytest rc $?" >> $out/pytest_staging.txt
for v in 0 1 2 4; do
  SS_OPTS=assoc_stage=$v SS_TL_DUMP=1 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage$v.txt 2>&1
  timeout 120 python tools/batched_assoc.py 1 32 30 1280 720 assoc_stage=$v > $out/batched_s1_f32_stage$v.json 2>$out/batched_s1_f32_stage$v.err
done
HIP_FORCE_DEV_KERNARG=1 SS_OPTS=assoc_stage=0 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage0_devkernarg1.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 SS_OPTS=assoc_stage=0 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage0_devkernarg0.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 SS_OPTS=assoc_stage=4 timeout 120 python tools/assoc_timeline.py 1 32 > $out/timeline_stage4_devkernarg1.txt 2>&1
for v in 0 2 4; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --opt assoc_stage=$v > $out/bench_stage$v.json 2>$out/bench_stage$v.err
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --opt assoc_stage=0 > $out/bench_stage0_devkernarg1.json 2>$out/bench_stage0_devkernarg1.err
HIP_FORCE_DEV_KERNARG=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --opt assoc_stage=0 > $out/bench_stage0_devkernarg0.json 2>$out/bench_stage0_devkernarg0.err
timeout 120 python tools/batched_assoc.py 32 32 30 1280 720 assoc_stage=0 > $out/batched_s32_f32_stage0.json 2>/dev/null
timeout 120 python tools/batched_assoc.py 32 32 30 1280 720 assoc_stage=4 > $out/batched_s32_f32_stage4.json 2>/dev/null
timeout 120 python tools/batched_assoc.py 32 32 30 1280 720 assoc_stage=2 > $out/batched_s32_f32_stage2.json 2>/dev/null
tail -2 $out/pytest_staging.txt
for f in $out/bench_stage*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "assoc us", r["mean_launch_us"], "inkernel", r.get("inkernel_mean_us"), "exact", d["frames_bit_exact"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
grep -h "in-kernel duration" $out/timeline_*.txt
