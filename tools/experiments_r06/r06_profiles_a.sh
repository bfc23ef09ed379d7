#!/bin/bash
# Round-6 evidence, part A: the whole GPU suite, the driver-form bench line (fp32 ReID default), the f16 line, the other presets, and the
# rocprofv3 kernel stats + k_assoc launches of the DEFAULT pipeline command.  Summaries -> gpurun_out/r06_prof/.
out=$GRAFT_REPO_ROOT/gpurun_out/r06_prof; mkdir -p $out; cd $GRAFT_REPO_ROOT
nproc > $out/nproc.txt
( time timeout 1500 python -m pytest tests/ -q -m gpu > $out/pytest_gpu_full.txt 2>&1 ) 2> $out/pytest_time.txt; tail -2 $out/pytest_gpu_full.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $out/r06_bench_c2_s1_driverargs.json 2>$out/bench_c2.err ) 2> $out/bench_c2_time.txt
timeout 600 python bench.py --steps 20 --warmup 5 --reid-f16 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode > $out/r06_bench_c2_s1_reid_f16.json 2>$out/bench_c2_f16.err
timeout 600 python bench.py --steps 20 --warmup 5 --det-fp32 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode > $out/r06_bench_c2_s1_all_fp32.json 2>$out/bench_c2_all32.err
for p in c3 c5 c6 c1; do timeout 600 python bench.py --steps 20 --warmup 5 --preset $p --no-cpu-baseline --no-reid-check --no-accuracy-mode > $out/r06_bench_${p}_s1.json 2>$out/bench_$p.err; done
timeout 900 python bench.py --steps 10 --warmup 3 --preset c4 --no-cpu-baseline --no-reid-check --no-accuracy-mode > $out/r06_bench_c4_s1.json 2>$out/bench_c4.err
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_bench; mkdir -p $out/prof_bench
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check --no-accuracy-mode > $out/prof_bench/cmd.log 2>&1)
cd $GRAFT_REPO_ROOT
grep '"metric"' $out/prof_bench/cmd.log | tail -1 > $out/r06_rocprofv3_bench_line_c2_s1_driverargs.json
f=$(find $out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $out/r06_rocprofv3_kernel_stats_c2_s1_driverargs.csv
t=$(find $out/prof_bench -name "*kernel_trace.csv" | head -1)
python tools/assoc_trace_filter.py "$t" 20 > $out/r06_rocprofv3_kernel_trace_k_assoc_pipeline_c2_s1.csv
python tools/trace_busy.py "$t" 16 > $out/r06_gpu_busy_c2_s1.txt 2>&1
find $out/prof_bench -name "*.csv" -size +2M -delete
tail -3 $out/r06_rocprofv3_kernel_trace_k_assoc_pipeline_c2_s1.csv; cat $out/bench_c2_time.txt
for f in $out/r06_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "assoc us", r["mean_launch_us"], r["frac"], r["bound"], "exact", d["frames_bit_exact"], (d.get("throughput_mode") or {}).get("frames_per_s"), (d.get("all_fp32") or {}).get("frames_per_s"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
