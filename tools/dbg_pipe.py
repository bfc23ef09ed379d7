import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd.pipeline import FramePipeline
from strongsort_yolo_amd.engine import scale_geometry
from strongsort_yolo_amd.synth import make_stream, synth_prediction
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
nets_on = (sys.argv[2] != "nonets") if len(sys.argv) > 2 else True
pipe = FramePipeline("yolov8n", 1, (720, 1280), graph=mode, det_source="synthetic", feat_source=os.environ.get("DBG_FEAT","by_anchor"), run_nets=nets_on)
gs = scale_geometry(pipe.geom, 720, 1280)
st = make_stream(0); rng = np.random.default_rng(1)
for k in range(8):
    fr = st.next_frame()
    pred, agt = synth_prediction(fr.dets, pipe.n_anchors, pipe.nc, gs[0], (gs[1], gs[2]), rng)
    pipe.frames[0].copy_(torch.from_numpy(st.frame_pixels(k)).to(pipe.dev))
    pipe.pred_in[0].copy_(torch.from_numpy(pred).to(pipe.dev)); pipe.anchor_gt[0].copy_(torch.from_numpy(agt).to(pipe.dev))
    f = np.zeros((128, 512), np.float32); f[:len(fr.feats)] = fr.feats
    pipe.gt_feats[0].copy_(torch.from_numpy(f).to(pipe.dev))
    if k == 5:
        pipe.eng.assoc_timing(True)
    pipe.step(); torch.cuda.synchronize()
    print("frame", k, "ndets", int(pipe.ndets[0]), "nout", int(pipe.nout[0]), flush=True)
print("timing", pipe.eng.assoc_timing(False))
print("OK", mode)
