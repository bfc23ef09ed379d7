import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig
from strongsort_yolo_amd.engine import scale_geometry
from strongsort_yolo_amd.pipeline import FramePipeline
detector, W, H, n_ids, rb = bench.PRESETS["c2"]
cfg, dcfg = StrongSortConfig(), DetectConfig()
p = FramePipeline(detector, 1, (H, W), half=True, reid_batch=rb, cfg=cfg, dcfg=dcfg, det_source="synthetic", feat_source="by_anchor", graph="none", run_nets=False)
gs, nc, A = scale_geometry(p.geom, H, W), p.nc, p.n_anchors
p.close()
for timed in (192, 960):
    r = bench.api_path(detector, W, H, n_ids, gs, nc, A, cfg, dcfg, reid_fp32=True, timed=timed)
    print(json.dumps({"timed": timed, **{k: r[k] for k in ("per_frame_track_frames_per_s", "track_stream_frames_per_s", "frames_identical_to_oracle")}}), flush=True)
