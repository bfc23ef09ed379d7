import os, sys, torch, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd.engine import TrackerEngine, letterbox_geometry, scale_geometry
from strongsort_yolo_amd.config import DetectConfig
from strongsort_yolo_amd.synth import make_stream, synth_prediction
eng = TrackerEngine(None, 1, 0)
dc = DetectConfig(); g = letterbox_geometry(720, 1280); gs = scale_geometry(g, 720, 1280)
A = 5040; dev = eng.device
pred = torch.zeros(84, A, device=dev); rows = torch.zeros(128, 6, device=dev); keep = torch.zeros(128, dtype=torch.int32, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
st = make_stream(0); rng = np.random.default_rng(1)
preds = []
for k in range(6):
    fr = st.next_frame(); p, _ = synth_prediction(fr.dets, A, 80, gs[0], (gs[1], gs[2]), rng); preds.append(torch.from_numpy(p).to(dev))
def run(): eng.nms(pred, 80, dc, gs[0], gs[1], gs[2], 1280, 720, rows=rows, keep=keep, count=cnt)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.use_current_stream()
    pred.copy_(preds[0]); run(); s.synchronize(); print("eager count", int(cnt))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        run()
torch.cuda.current_stream().wait_stream(s)
eng.use_current_stream()
for k in range(6):
    pred.copy_(preds[k]); gr.replay(); torch.cuda.synchronize(); print("replay", k, "count", int(cnt), flush=True)
print("OK")
