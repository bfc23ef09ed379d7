"""The drop-in classes on the GPU: StrongSORT.update(dets, frame) and YOLO.track()/.predict()."""
import numpy as np
import pytest
import torch

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.synth import make_stream

pytestmark = pytest.mark.gpu


def test_strongsort_update_injected_features_equals_oracle():
    from strongsort_yolo_amd.tracker import StrongSORT
    trk, orc = StrongSORT(), OracleStrongSort(StrongSortConfig(), "c")
    sg, so = make_stream(21, 640, 480, 9), make_stream(21, 640, 480, 9)
    for k in range(15):
        fg, fo = sg.next_frame(), so.next_frame()
        got = trk.update(fg.dets, sg.frame_pixels(k), features=fg.feats)
        ref = orc.update(fo.dets, fo.feats, (480, 640))
        assert got.tobytes() == ref.tobytes() and got.shape == ref.shape
    trk.close()


def test_strongsort_update_with_reid_net_runs_and_is_deterministic():
    from strongsort_yolo_amd.tracker import StrongSORT
    outs = []
    for _ in range(2):
        trk, st = StrongSORT(), make_stream(22, 640, 480, 6)
        rows = []
        for k in range(6):
            f = st.next_frame()
            rows.append(trk.update(f.dets, st.frame_pixels(k)))
        outs.append(np.concatenate(rows))
        trk.close()
    assert outs[0].shape[1] == 8 and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("weights", ["yolov8n.pt", "yolo11n-pose.pt"])
def test_yolo_track_and_predict_contract(weights):
    from strongsort_yolo_amd.yolo import YOLO
    model = YOLO(weights)
    model.overrides.update(conf=0.9, iou=0.4, agnostic_nms=False, max_det=50)    # random-init head: keep it sparse
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    res = model.predict(img, verbose=False, device=0)
    assert len(res) == 1 and res[0].boxes is not None and res[0].boxes.xyxy.shape[1] == 4
    for _ in range(3):
        res = model.track(img, verbose=False, device=0, persist=True, tracker="botsort.yaml")
    r = res[0]
    if "pose" in weights and r.keypoints is not None:
        assert len(r.keypoints) == len(r.boxes)
        for bbox, kp in zip(r.boxes, r.keypoints):               # yolo_multi_model.py:58-62
            pts = kp.xy.tolist()
            assert len(pts) == 1 and len(pts[0]) == 17 and len(pts[0][0]) == 2
    if r.boxes.id is None:                 # nothing confirmed: the reference skips such frames (yolo_multi_model.py:54)
        assert len(r.boxes) == 0
        return
    assert len(r.boxes.id) == len(r.boxes.conf) == len(r.boxes.cls)
    for bbox in r.boxes:
        for scores, classes, xyxy, id_ in zip(bbox.conf, bbox.cls, bbox.xyxy, bbox.id):
            assert int(id_) >= 1 and 0 <= float(scores) <= 1 and int(classes) in r.names
