for f in 0 1; do SS_OSNET_CHAINS=$f python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_c$f.json; python -c "
import json; d=json.load(open('gpurun_out/bench_c$f.json')); print('chains=$f', d['value'], d['roofline']['mean_launch_us'], d['api_path']['track_stream_frames_per_s'], d['api_path']['per_frame_track_frames_per_s'])"; done
