"""In-kernel timeline of the association kernel (wave 0 of every workgroup, first work item).
usage: python tools/assoc_timeline.py [streams=1] [frame_batch=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.engine import TrackerEngine
from strongsort_yolo_amd.synth import make_stream
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
FB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = 128
eng = TrackerEngine(StrongSortConfig(), S, 0)
dev = eng.device
hd, hf, hn = np.zeros((frames, S, 128, 6), np.float32), np.zeros((frames, S, 128, 512), np.float32), np.zeros((frames, S), np.int32)
for s in range(S):
    st = make_stream(5000 + s, 1280, 720, 30)
    for k in range(frames):
        f = st.next_frame(); n = len(f.dets)
        hd[k, s, :n], hf[k, s, :n], hn[k, s] = f.dets, f.feats, n
dets, feats, nd = torch.from_numpy(hd).to(dev), torch.from_numpy(hf).to(dev), torch.from_numpy(hn).to(dev)
hw = torch.tensor([[720, 1280]] * S, dtype=torch.int32, device=dev)
out, nout = torch.zeros(FB, S, 256, 8, device=dev), torch.zeros(FB, S, dtype=torch.int32, device=dev)
for k0 in range(0, frames, FB):
    if k0 + FB >= frames:
        torch.cuda.synchronize(); eng.assoc_inkernel_timing(2)
    eng.update_group(FB, dets[k0:k0 + FB], nd[k0:k0 + FB], feats[k0:k0 + FB], hw, out, nout)
torch.cuda.synchronize()
tl = eng.assoc_timeline(512)
us, n = eng.assoc_inkernel_timing(0)
act = tl[:, 0] > 0
t0 = tl[act, 0].min()
rel = (tl[act] - t0) / 100.0
names = ["entry", "record", "loads", "staged"] + [f"seg{2 * i}" for i in range(8)] + ["done"]
print(f"S={S} FB={FB}: in-kernel duration {us:.2f} us over {n} launch(es); {int(act.sum())} workgroups had work")
print("stamp     min     median  max   (us from the first workgroup's entry)")
for i, nm in enumerate(names):
    c = rel[:, i]
    c = c[tl[act][:, i] > 0]                     # runs shorter than 15 segments leave the later stamps empty
    if len(c):
        print(f"{nm:7s} {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f}   ({len(c)} workgroups)")
if os.environ.get("SS_TL_DUMP"):
    i = int(os.environ["SS_TL_DUMP"])
    d = (tl[:, i] - tl[:, 0]) / 100.0
    print(f"stamp {i} - entry, by XCD (rows) x workgroup index within XCD (64 cols, us):")
    for x in range(8):
        print(x, " ".join(f"{v:4.1f}" for v in d[x::8][:64]))
