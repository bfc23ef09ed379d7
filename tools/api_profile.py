"""Where the per-frame YOLO.track call spends its time (component timings on the model's own pipeline)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
from strongsort_yolo_amd.yolo import YOLO
m = YOLO("yolov8n.pt", random_init_ok=True, reid_batch=32)
m.overrides.update(conf=0.9, iou=0.4, agnostic_nms=False, max_det=1000)
img = np.random.default_rng(0).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
for _ in range(5):
    m.track(img)
def t(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
p = m._pipe
sync = lambda: torch.cuda.current_stream(p.dev).synchronize()
print("per-frame track: %.3f ms" % t(lambda: m.track(img)))
print("upload (host call only): %.3f ms" % t(lambda: p.eng.upload(p.frames[0], img)))
print("upload + sync: %.3f ms" % t(lambda: (p.eng.upload(p.frames[0], img), sync())))
print("3 graph replays, no sync (host enqueue): %.3f ms" % t(lambda: p.step(track=True)))
print("3 graph replays + sync: %.3f ms" % t(lambda: (p.step(track=True), sync())))
for i, nm in enumerate(("detect", "reid", "tracker")):
    print(f"{nm} graph + sync: %.3f ms" % t(lambda: (p.graph[i].replay(), sync())))
def body():
    p.eng.upload(p.frames[0], img); p.step(track=True)
    p.eng.pack_results(p.ndets, p.dets[0], p.nout, p.out[0], m._h_res); sync()
print("_run body without check_errors / Results: %.3f ms" % t(body))
print("check_errors: %.3f ms" % t(lambda: p.eng.check_errors()))
