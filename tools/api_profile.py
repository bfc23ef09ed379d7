"""Where the per-frame YOLO.track call spends its time (host profile + component timings)."""
import cProfile, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
from strongsort_yolo_amd.yolo import YOLO
m = YOLO("yolov8n.pt", random_init_ok=True, reid_batch=32)
m.overrides.update(conf=0.9, iou=0.4, agnostic_nms=False, max_det=1000)
img = np.random.default_rng(0).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
for _ in range(5):
    m.track(img)
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("per-frame track: %.2f ms" % t(lambda: m.track(img)))
p = m._pipe
src = torch.from_numpy(img)
print("numpy -> pinned copy: %.3f ms" % t(lambda: m._h_frame.copy_(src)))
print("pinned -> device: %.3f ms" % t(lambda: p.frames[0].copy_(m._h_frame, non_blocking=True)))
print("pageable -> device: %.3f ms" % t(lambda: p.frames[0].copy_(src)))
print("3 graph replays, no sync: %.3f ms" % t(lambda: p.step(track=True)))
def one():
    p.step(track=True); torch.cuda.current_stream().synchronize()
print("3 graph replays + sync each: %.3f ms" % t(one))
print("detect graph only + sync: %.3f ms" % t(lambda: (p.graph[0].replay(), torch.cuda.current_stream().synchronize())))
print("reid graph only + sync: %.3f ms" % t(lambda: (p.graph[1].replay(), torch.cuda.current_stream().synchronize())))
print("tracker graph only + sync: %.3f ms" % t(lambda: (p.graph[2].replay(), torch.cuda.current_stream().synchronize())))
print("check_errors: %.3f ms" % t(lambda: p.eng.check_errors()))
def one_idle():
    t0 = time.perf_counter(); p.step(track=True); torch.cuda.current_stream().synchronize(); dt = time.perf_counter() - t0
    time.sleep(0.005); return dt
print("3 graph replays + sync with 5 ms idle gaps: %.3f ms" % (sum(one_idle() for _ in range(50)) / 50 * 1e3))
def full_no_results():
    m._h_frame.copy_(src); p.frames[0].copy_(m._h_frame, non_blocking=True); p.step(track=True)
    m._d_cnt[0:1].copy_(p.ndets); m._h_dets.copy_(p.dets[0], non_blocking=True); m._d_cnt[1:2].copy_(p.nout)
    m._h_rows.copy_(p.out[0], non_blocking=True); m._h_cnt.copy_(m._d_cnt, non_blocking=True)
    torch.cuda.current_stream().synchronize()
print("_run body without Results: %.3f ms" % t(full_no_results))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): full_no_results()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(6)
os.system("rocm-smi --showperflevel --showclocks 2>&1 | head -20")
