"""Phase times inside k_frame / k_postnew (needs a library built with -DSS_FRAME_STAMPS: make FLAGS_ss_track="-mllvm -amdgpu-kernarg-preload-count=8
-DSS_FRAME_STAMPS"; SS_LIB_PATH=<that .so>).  Tracker path alone, one stream, 32 frames per call; microseconds per frame, mean over the timed frames.
usage: python tools/frame_phases.py [name=value ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.engine import TrackerEngine
from strongsort_yolo_amd.synth import make_stream
FB, frames, timed = 32, 352, 192
eng = TrackerEngine(StrongSortConfig(), 1, 0)
for kv in sys.argv[1:]:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = eng.device
hd, hf, hn = np.zeros((frames, 1, 128, 6), np.float32), np.zeros((frames, 1, 128, 512), np.float32), np.zeros((frames, 1), np.int32)
st = make_stream(6000, 1280, 720, 30)
for k in range(frames):
    f = st.next_frame(); n = len(f.dets)
    hd[k, 0, :n], hf[k, 0, :n], hn[k, 0] = f.dets, f.feats, n
dets, feats, nd = torch.from_numpy(hd).to(dev), torch.from_numpy(hf).to(dev), torch.from_numpy(hn).to(dev)
hw = torch.tensor([[720, 1280]], dtype=torch.int32, device=dev)
out, nout = torch.zeros(FB, 1, 256, 8, device=dev), torch.zeros(FB, 1, dtype=torch.int32, device=dev)
def run(a, b):
    for k0 in range(a, b, FB):
        eng.update_group(FB, dets[k0:k0 + FB].contiguous(), nd[k0:k0 + FB].contiguous(), feats[k0:k0 + FB].contiguous(), hw, out, nout)
    torch.cuda.synchronize()
run(0, frames - timed)
t0 = eng.assoc_timeline(4096)[3000:3004].copy()
run(frames - timed, frames)
t1 = eng.assoc_timeline(4096)[3000:3004]
d = (t1 - t0).astype(np.float64)
names = ["load state, gate factorisation", "scan confirmed + sync", "cost build (maha, blend) + sync", "assignment A + match", "scan B + lists + sync",
         "IoU cost + assignment B", "state machine (stage C)", "scans C", "births scan + sync", "births + order + counters"]
n = d[0, 15]
print("k_frame phases, us per frame (", int(n), "frames ):")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {d[0, i] / n / 100:.2f}")
print(f"  {'sum':36s} {d[0, :10].sum() / n / 100:.2f}")
print(f"k_postnew: post_track workgroup 0 {d[1, 0] / max(d[1, 15], 1) / 100:.2f} us, first new-row workgroup {d[2, 0] / max(d[2, 15], 1) / 100:.2f} us")
fine = ["counts loaded", "order loaded", "predicted mean / covariance loaded", "tsu loaded", "gate factorised, box written", "state loaded"]
print("inside the first phase, us from kernel start:", {nm: round(d[3, i] / n / 100, 2) for i, nm in enumerate(fine)})
eng.close()
