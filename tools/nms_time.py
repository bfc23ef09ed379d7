"""Batched NMS (a3) of 32 yolov8n head tensors [84][5040] with ~30 kept boxes each: microseconds per call by HIP events.
usage: python tools/nms_time.py [reps=50] [identities=30]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd.config import StrongSortConfig, DetectConfig
from strongsort_yolo_amd.engine import TrackerEngine, letterbox_geometry, scale_geometry
from strongsort_yolo_amd.synth import make_stream, synth_prediction
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = TrackerEngine(StrongSortConfig(), 1, 0)
dev, dcfg = eng.device, DetectConfig()
W, H, nc, B = 1280, 720, 80, 32
g = letterbox_geometry(H, W)
gain, px, py = scale_geometry(g, H, W)
N = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
NID = int(sys.argv[2]) if len(sys.argv) > 2 else 30
st, rng = make_stream(5, W, H, NID), np.random.default_rng(2)
preds = np.stack([synth_prediction(st.next_frame().dets, N, nc, gain, (px, py), rng)[0] for _ in range(B)])
pred = torch.from_numpy(preds).to(dev)
rows, keep, count = torch.zeros(B, 128, 6, device=dev), torch.zeros(B, 128, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
geom = torch.tensor([[gain, px, py, W, H]] * B, dtype=torch.float32, device=dev)
fn = lambda: eng.nms_batch(pred, nc, dcfg, geom, rows=rows, keep=keep, count=count, max_det=128)
for _ in range(3):
    fn()
s = torch.cuda.current_stream(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(reps):
    fn()
e1.record(s); s.synchronize()
print({"images": B, "anchors": N, "us_per_call": round(e0.elapsed_time(e1) / reps * 1e3, 1), "kept_mean": float(count.float().mean()), "rows_checksum": float(rows.double().sum())})
