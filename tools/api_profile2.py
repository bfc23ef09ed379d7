"""Per-statement host timing of the per-frame YOLO.track body + end-to-end rates of the two drop-in call forms."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
from strongsort_yolo_amd.yolo import YOLO
m = YOLO("yolov8n.pt", random_init_ok=True, reid_batch=32)
m.overrides.update(conf=0.9, iou=0.4, agnostic_nms=False, max_det=1000)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(8)]
for _ in range(5): m.track(imgs[0])
p = m._pipe
steps = [
 ("upload", lambda i: p.eng.upload(p.frames[0], imgs[i % 8])),
 ("step", lambda i: p.step(track=True)),
 ("results", lambda i: p.eng.pack_results(p.ndets, p.dets[0], p.nout, p.out[0], m._h_res)),
 ("sync", lambda i: torch.cuda.current_stream().synchronize()),
]
acc = {k: 0.0 for k, _ in steps}
for i in range(40):
    for k, fn in steps:
        t0 = time.perf_counter(); fn(i); acc[k] += time.perf_counter() - t0
for k, _ in steps: print(f"{k:10s} {acc[k] / 40 * 1e3:8.3f} ms")
t0 = time.perf_counter()
for i in range(100): m.track(imgs[i % 8])
print("model.track per frame: %.2f ms  (%.0f frames/s)" % ((time.perf_counter() - t0) * 10, 100 / (time.perf_counter() - t0)))
n = 0; t0 = None
for k, res in enumerate(m.track_stream((imgs[i % 8] for i in range(16 * 30)), batch=16)):
    if k == 16 * 5 - 1: t0 = time.perf_counter()
    n += 1
print("track_stream: %.0f frames/s" % ((n - 80) / (time.perf_counter() - t0)))
