"""k_assoc's launch duration (HIP events on its dispatch + the kernel's own stamps) in the tracker-only loop: one stream, 30 identities,
32 frames per call, galleries full.  usage: python tools/assoc_time.py [name=value ...] (ss_set_option switches, e.g. assoc_pack=0; ids=N: identities of the stream, > 60: 1920x1080)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.engine import TrackerEngine
from strongsort_yolo_amd.synth import make_stream
FB, frames, W, H = 32, 32 * 14, 1280, 720
eng = TrackerEngine(StrongSortConfig(), 1, 0)
IDS = 30
for kv in [a for a in sys.argv[1:] if a.startswith("ids=")]:
    IDS = int(kv.split("=")[1]); sys.argv.remove(kv)
if IDS > 60:
    W, H = 1920, 1080
for kv in sys.argv[1:]:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = eng.device
hd, hf, hn = np.zeros((frames, 1, 128, 6), np.float32), np.zeros((frames, 1, 128, 512), np.float32), np.zeros((frames, 1), np.int32)
st = make_stream(6000, W, H, IDS)
for k in range(frames):
    f = st.next_frame(); n = len(f.dets)
    hd[k, 0, :n], hf[k, 0, :n], hn[k, 0] = f.dets, f.feats, n
dets, feats, nd = torch.from_numpy(hd).to(dev), torch.from_numpy(hf).to(dev), torch.from_numpy(hn).to(dev)
hw = torch.tensor([[H, W]], dtype=torch.int32, device=dev)
out, nout = torch.zeros(FB, 1, 256, 8, device=dev), torch.zeros(FB, 1, dtype=torch.int32, device=dev)
for k0 in range(0, 32 * 4, FB):
    eng.update_group(FB, dets[k0:k0 + FB], nd[k0:k0 + FB], feats[k0:k0 + FB], hw, out, nout)
torch.cuda.synchronize()
eng.assoc_timing(True); eng.assoc_inkernel_timing(True)
for k0 in range(32 * 4, frames, FB):
    eng.update_group(FB, dets[k0:k0 + FB], nd[k0:k0 + FB], feats[k0:k0 + FB], hw, out, nout)
torch.cuda.synchronize()
ms, n = eng.assoc_timing(False)
vals = np.sort(eng.assoc_timing_values().astype(np.float64) * 1e3)
ik, ikn = eng.assoc_inkernel_timing(False)
eng.check_errors()
print(json.dumps({"opts": sys.argv[1:], "launches": int(n), "event_us_mean": round(ms * 1e3, 2), "event_us_median": round(float(np.median(vals)), 2),
                  "event_us_min": round(float(vals[0]), 2), "inkernel_us_mean": round(float(ik), 2), "mean_dets": round(float(hn.mean()), 2), "ids": IDS}))
