#!/bin/bash
# the committed driver-form line and the kernel stats of the same command, with the round's final binary
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c32; mkdir -p $out; cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py --steps 20 --warmup 5 > $out/r04_bench_c2_s1_driverargs.json 2>$out/bench_c2.err ) 2> $out/bench_c2_time.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_bench; mkdir -p $out/prof_bench
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-api-path --no-reid-check > $out/prof_bench/cmd.log 2>&1)
cd $GRAFT_REPO_ROOT
grep '"metric"' $out/prof_bench/cmd.log | tail -1 > $out/r04_rocprofv3_bench_line_c2_s1_driverargs.json
f=$(find $out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $out/r04_rocprofv3_kernel_stats_c2_s1_driverargs.csv
t=$(find $out/prof_bench -name "*kernel_trace.csv" | head -1)
python tools/assoc_trace_filter.py "$t" 20 > $out/r04_rocprofv3_kernel_trace_k_assoc_pipeline_c2_s1.csv
python tools/detector_sequence.py "$t" > $out/r04_detector_sequence_32frames.txt 2>&1
python tools/osnet_sequence.py "$t" > $out/r04_osnet_sequence.txt 2>&1
find $out/prof_bench -name "*.csv" -size +2M -delete
tail -2 $out/r04_rocprofv3_kernel_trace_k_assoc_pipeline_c2_s1.csv; tail -3 $out/r04_osnet_sequence.txt; tail -2 $out/r04_detector_sequence_32frames.txt; cat $out/bench_c2_time.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_c32/r04_bench_c2_s1_driverargs.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'assoc', r['mean_launch_us'], r['frac'], 'exact', d['frames_bit_exact'], 'api', d['api_path']['track_stream_frames_per_s'], d['api_path']['per_frame_track_frames_per_s'], 'tracker_only', d['tracker_only']['frames_per_s'])
rf=d['reid_f16_vs_f32']; print({k:rf[k] for k in ('embedding_unit_max_abs_err','cost_matrix_cosine_max_abs_err','id_match_rate','id_match_rate_up_to_relabeling')}, rf['fp32_reid_mode']['cost_matrix_cosine_max_abs_err'], rf['fp32_reid_mode']['id_match_rate'])
print('cpu', d['cpu_baseline']['value'])
PY
