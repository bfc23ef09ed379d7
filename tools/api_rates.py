"""The drop-in calls on their own (bench.api_path without the rest of the bench: per-frame YOLO.track and YOLO.track_stream with host
frames in, Results out, checked against the oracle).  usage: python tools/api_rates.py [preset=c2] [repeats=2]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig
from strongsort_yolo_amd.engine import scale_geometry
from strongsort_yolo_amd.pipeline import FramePipeline
preset = sys.argv[1] if len(sys.argv) > 1 else "c2"
detector, W, H, n_ids, rb = bench.PRESETS[preset]
cfg, dcfg = StrongSortConfig(), DetectConfig()
p = FramePipeline(detector, 1, (H, W), half=True, reid_batch=rb, cfg=cfg, dcfg=dcfg, det_source="synthetic", feat_source="by_anchor", graph="none", run_nets=False)
gs, nc, A = scale_geometry(p.geom, H, W), p.nc, p.n_anchors
p.close()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
  for fp32 in (True, False):
    r = bench.api_path(detector, W, H, n_ids, gs, nc, A, cfg, dcfg, reid_fp32=fp32)
    print(json.dumps({"preset": preset, "reid_fp32": fp32, **{k: r[k] for k in ("per_frame_track_frames_per_s", "track_stream_frames_per_s", "track_stream_batch", "frames_identical_to_oracle")}}), flush=True)
