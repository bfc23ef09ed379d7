"""Regenerates tests/golden/*.npz from the exact-order C oracle (run from the repo root:
`python tests/golden/make_golden.py`).

There is no reference implementation or reference fixture to generate vectors from (parity
unpinned), so these files pin THIS repo's oracle: inputs and the oracle's outputs are stored together, and
tests/test_golden.py replays the stored inputs — any later change to the oracle's arithmetic or lifecycle
rules shows up as a diff against a committed vector instead of silently moving the parity target.
Inputs are stored (not re-generated) so the check is independent of the NumPy build of the machine.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cexact                                    # noqa: E402
from oracle.strongsort_np import OracleStrongSort            # noqa: E402
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig   # noqa: E402
from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry   # noqa: E402
from strongsort_yolo_amd.synth import make_stream, synth_prediction          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tracker_vector():
    W, H, n_ids, frames = 640, 480, 6, 24
    st = make_stream(1234, W, H, n_ids, p_vanish=0.08, vanish_max=4)
    orc = OracleStrongSort(StrongSortConfig(), "c")
    dets, feats, counts, rows, nrows, cost = [], [], [], [], [], []
    for k in range(frames):
        f = st.next_frame()
        if k == 9:                                           # one empty frame
            f.dets, f.feats = f.dets[:0], f.feats[:0]
        d = np.zeros((n_ids, 6), np.float32); d[:len(f.dets)] = f.dets
        x = np.zeros((n_ids, 512), np.float32); x[:len(f.feats)] = f.feats
        r = orc.update(f.dets, f.feats, (H, W))
        rr = np.zeros((n_ids * 2, 8), np.float32); rr[:len(r)] = r
        c = np.zeros((n_ids * 2, n_ids)); ca = orc.last["cost_a"]; c[:ca.shape[0], :ca.shape[1]] = ca
        dets.append(d); feats.append(x); counts.append(len(f.dets)); rows.append(rr); nrows.append(len(r)); cost.append(c)
    snap = orc.snapshot()
    np.savez_compressed(os.path.join(HERE, "tracker_6ids_24frames.npz"), hw=np.array([H, W]), dets=np.array(dets),
                        feats=np.array(feats).astype(np.float16 if False else np.float32), counts=np.array(counts),
                        rows=np.array(rows), nrows=np.array(nrows), cost_a=np.array(cost), final_mean=snap["mean"],
                        final_cov=snap["cov"], final_ids=snap["track_id"], next_id=np.array(snap["next_id"]))


def front_vector():
    rng = np.random.default_rng(77)
    H, W = 45, 80
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W, imgsz=64, stride=32)
    lb = cexact.letterbox(img, g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
    gain, px, py = scale_geometry(g, H, W)
    boxes = np.array([[5.2, 3.1, 40.7, 30.9, 0.9, 0], [38.0, 10.0, 79.9, 44.9, 0.8, 1], [-3.0, -2.0, 4.0, 6.0, 0.7, 0]], np.float32)
    crops = cexact.crop_norm(img, boxes, 32, 16)
    dc = DetectConfig()
    N, nc = 84, 3
    pred, _ = synth_prediction(boxes, N, nc, gain, (px, py), rng, dup=4, clutter=20)
    keep, rows = cexact.nms(pred, nc, dc.conf, dc.iou, dc.agnostic_nms, dc.max_wh, dc.max_nms, dc.max_det)
    rows = cexact.scale_boxes(rows, gain, px, py, W, H)
    np.savez_compressed(os.path.join(HERE, "front_small.npz"), img=img, geom=np.array([g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left]),
                        letterbox=lb, boxes=boxes, crops=crops, pred=pred, nms_keep=keep, nms_rows=rows,
                        scale=np.array([gain, px, py], np.float64))


def lsap_vector():
    rng = np.random.default_rng(5)
    mats, sols = [], []
    for shape in [(6, 6), (4, 9), (9, 4), (12, 12)]:
        m = np.zeros((12, 12)); c = rng.integers(0, 4, shape).astype(np.float64)
        m[:shape[0], :shape[1]] = c
        r, cc = cexact.lsap(c)
        s = np.full(12, -1, np.int64); s[r] = cc
        mats.append(m); sols.append(s)
    np.savez_compressed(os.path.join(HERE, "lsap_ties.npz"), shapes=np.array([(6, 6), (4, 9), (9, 4), (12, 12)]), mats=np.array(mats), sols=np.array(sols))


if __name__ == "__main__":
    tracker_vector(); front_vector(); lsap_vector()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
