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
frames = 160                      # galleries full (nn_budget rows) from frame ~103 on
eng = TrackerEngine(StrongSortConfig(nn_budget=int(os.environ.get("SS_BUDGET", "100"))), S, 0)
for kv in os.environ.get("SS_OPTS", "").split(","):
    if kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
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
tl_all = eng.assoc_timeline(4096)
tl, cy = tl_all[:512], tl_all[2048:2048 + 512]
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
    j = int(os.environ.get("SS_TL_FROM", "0"))
    d = (tl[:, i] - tl[:, j]) / 100.0
    print(f"stamp {i} - entry, by XCD (rows) x workgroup index within XCD (64 cols, us):")
    for x in range(8):
        print(x, " ".join(f"{v:4.1f}" for v in d[x::8][:64]))

# shader-clock stamps (s_memtime) of the same points: cycles per interval and the clock they imply
print("interval            us(median)  cycles(median)  GHz")
a = act
for i0, i1 in [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 12)]:
    ok = a & (tl[:, i0] > 0) & (tl[:, i1] > 0)
    if ok.any():
        du = np.median((tl[ok, i1] - tl[ok, i0]) / 100.0); dc = np.median(cy[ok, i1] - cy[ok, i0])
        print(f"{names[i0]:7s}->{names[i1]:7s} {du:9.2f} {dc:12.0f} {dc / max(du, 1e-9) / 1e3:8.2f}")

nts = tl[act, 13] & 0xffff
print("tiles per record (first record of every workgroup):", dict(zip(*np.unique(nts, return_counts=True))), " composite records:", int((tl[act, 13] >> 32).sum()),
      " records per list:", sorted(set(tl[act, 14].tolist())))

# per XCD list (workgroup b serves list b % 8): is the start-up / finish spread systematic?
print("per XCD: median us of record / staged / done, tiles of its records (sum), latest done")
for x in range(8):
    m = act.copy(); m[np.arange(512) % 8 != x] = False
    r = (tl[m] - t0) / 100.0
    print(f"  xcd {x}: record {np.median(r[:, 1]):5.2f}  staged {np.median(r[:, 3]):5.2f}  done {np.median(r[:, 12]):5.2f}  max done {r[:, 12].max():5.2f}  tiles {int((tl[m, 13] & 0xffff).sum())}")
# ... and by position inside the list (dispatch order): quartiles of the workgroup index
q = np.arange(512) // 8
for lo in range(0, 64, 16):
    m = act & (q >= lo) & (q < lo + 16)
    r = (tl[m] - t0) / 100.0
    print(f"  list positions {lo:2d}-{lo + 15:2d}: entry {np.median(r[:, 0]):5.2f}  record {np.median(r[:, 1]):5.2f}  staged {np.median(r[:, 3]):5.2f}  done {np.median(r[:, 12]):5.2f}  max done {r[:, 12].max():5.2f}")
