"""Does a pipeline that is not the first of its process run slower?  Builds the bench's overlapped pipeline (f16 ReID by default) N times in ONE
process — run, close, build again — and prints ms per 32-frame step and the tracker call's share for each.
usage: python tools/pipeline_reuse.py [n=4] [f16|fp32] [name=value ... pipeline switches, e.g. track_priority=0] [gc=1]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from strongsort_yolo_amd.config import StrongSortConfig, DetectConfig
from strongsort_yolo_amd.pipeline import OverlappedPipeline
import bench
from strongsort_yolo_amd.engine import scale_geometry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
half = (sys.argv[2] if len(sys.argv) > 2 else "f16") == "f16"
sw = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)
do_gc = sw.pop("gc", "0") != "0"
sw = {k: v != "0" for k, v in sw.items()}
W, H, FB, steps, warm = 1280, 720, 32, 12, 4
cfg, dcfg = StrongSortConfig(), DetectConfig()
for it in range(n):
    pipe = OverlappedPipeline("yolov8n", 1, (H, W), device=0, half=True, reid_batch=32, cfg=cfg, dcfg=dcfg, det_source="synthetic", feat_source="by_anchor",
                              graph="front", reid_half=half, n_stages=2, frame_batch=FB, reid_split=(bench.REID_SPLIT if half else bench.REID_SPLIT_FP32)["c2"],
                              defer_track=True, **sw)
    gs = scale_geometry(pipe.geom, H, W)
    total = bench.PREFILL + (warm + steps) * FB
    wl = bench.make_workload(1000, W, H, 30, total, gs, pipe.nc, pipe.n_anchors, pipe.nk)
    dev = pipe.dev
    p = {k: torch.from_numpy(wl[k]).to(dev) for k in ("preds", "agt", "feats", "pixels")}
    npix = p["pixels"].shape[0]
    cyc = p["pixels"].repeat((FB + npix - 1) // npix + 1, 1, 1, 1)

    def run(k0, k1):
        for g0 in range(k0, k1, FB):
            m = min(FB, k1 - g0)
            b = pipe.begin_frame()
            with torch.cuda.stream(pipe.s_in):
                b.frames[:m].copy_(cyc[g0 % npix:g0 % npix + m]); b.pred_in[:m].copy_(p["preds"][g0:g0 + m])
                b.anchor_gt[:m].copy_(p["agt"][g0:g0 + m]); b.gt_feats[:m].copy_(p["feats"][g0:g0 + m])
            pipe.submit(m)
    run(0, bench.PREFILL + warm * FB)
    pipe.flush(); torch.cuda.synchronize()
    pipe.trace = []
    t0 = time.perf_counter()
    run(bench.PREFILL + warm * FB, total)
    pipe.flush(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr, pipe.trace = pipe.trace, None
    ts = [a[2].elapsed_time(b[2]) for a, b in zip(tr, tr[1:]) if a[0] == "track_start" and b[0] == "track_end"]
    print(f"pipeline {it + 1}: {dt / steps * 1e3:6.3f} ms per step = {steps * FB / dt:8.0f} frames/s; tracker call {np.median(ts):.2f} ms (median of {len(ts)})", flush=True)
    pipe.close()
    del pipe, p, cyc, wl
    if do_gc:
        gc.collect(); torch.cuda.empty_cache()
