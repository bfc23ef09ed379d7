"""Host-side logic of the boundary (no GPU): geometry, result duck types, stream sharding."""
import numpy as np
import torch

from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
from strongsort_yolo_amd.streams import assign_streams
from strongsort_yolo_amd.yolo import Boxes, Keypoints, Results, YOLO


def test_letterbox_geometry_table():
    # 1280x720 and 1920x1080 -> 640x384; 640x480 -> 640x480
    for (h, w), (oh, ow, top) in {(720, 1280): (384, 640, 12), (1080, 1920): (384, 640, 12), (480, 640): (480, 640, 0)}.items():
        g = letterbox_geometry(h, w)
        assert (g.out_h, g.out_w, g.pad_top, g.pad_left) == (oh, ow, top, 0)
        gain, px, py = scale_geometry(g, h, w)
        assert abs(gain - min(oh / h, ow / w)) < 1e-12 and (px, py) == (0.0, float(top))


def test_results_duck_type_matches_reference_usage():
    # the access pattern of /root/reference/yolo_multi_model.py:45-169
    b = Boxes(torch.tensor([[1., 2., 3., 4.], [5., 6., 7., 8.]]), torch.tensor([0.9, 0.8]), torch.tensor([0., 2.]),
              torch.tensor([7., 9.]))
    r = Results(np.zeros((4, 4, 3), np.uint8), {0: "person", 2: "car"}, b, Keypoints(torch.zeros(2, 17, 3)))
    ids = [int(bbox.id) for predictions in [r] if predictions is not None for bbox in predictions.boxes if bbox.id is not None]
    assert ids == [7, 9]
    rows = []
    for bbox, kp in zip(r.boxes, r.keypoints):
        assert len(kp.xy.tolist()) == 1 and len(kp.xy.tolist()[0]) == 17
        for scores, classes, bbox_coords, id_ in zip(bbox.conf, bbox.cls, bbox.xyxy, bbox.id):
            rows.append((int(id_), r.names[int(classes)], round(float(scores) * 100, 1), [int(v) for v in bbox_coords]))
    assert rows == [(7, "person", 90.0, [1, 2, 3, 4]), (9, "car", 80.0, [5, 6, 7, 8])]
    assert r.masks is None and Results(None, {}, None).boxes is None


def test_yolo_overrides_and_names():
    m = YOLO("yolo11n-pose.pt")
    m.overrides["conf"], m.overrides["iou"], m.overrides["agnostic_nms"], m.overrides["max_det"] = 0.3, 0.4, False, 1000
    d = m._dcfg()
    assert (d.conf, d.iou, d.agnostic_nms, d.max_det) == (0.3, 0.4, False, 1000)
    assert m.names == {0: "person"} and len(YOLO("yolov8n.pt").names) == 80
    inv = {v: k for k, v in YOLO("yolov5n.pt").names.items()}          # yolo_multi_model.py:23-24
    assert inv["person"] == 0


def test_assign_streams_partitions_everything():
    for n, w in [(8, 8), (8, 2), (5, 4), (1, 8), (0, 2)]:
        parts = assign_streams(n, w)
        assert len(parts) == w and sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_weights_policy_is_loud(tmp_path, monkeypatch):
    """a missing weights file must not silently become a random-init network; checkpoints are loaded
    with weights_only=True and must be plain state_dicts."""
    import pytest
    from strongsort_yolo_amd import nets
    m = nets.build_reid(1)
    monkeypatch.delenv("SS_RANDOM_INIT", raising=False)
    with pytest.raises(FileNotFoundError):
        nets.load_weights(m, str(tmp_path / "osnet_x0_25.pt"), "reid")
    with pytest.warns(RuntimeWarning):
        assert nets.load_weights(m, None, "reid", random_init_ok=True) is False
    sd = {k: v + 1 for k, v in m.state_dict().items()}
    torch.save({"state_dict": sd}, tmp_path / "w.pt")
    assert nets.load_weights(m, str(tmp_path / "w.pt"), "reid") is True
    assert torch.equal(m.fc.bias, sd["fc.bias"])
    torch.save({"model": nets.build_reid(2)}, tmp_path / "pickled.pt")           # Ultralytics-style module checkpoint
    with pytest.raises(Exception):
        nets.load_weights(m, str(tmp_path / "pickled.pt"), "reid")


def test_pose_head_decodes_keypoints_like_ultralytics():
    """xy = (2 k + anchor - 0.5) * stride, visibility = sigmoid (Ultralytics Pose.kpts_decode); ADVICE r1."""
    from strongsort_yolo_amd import nets
    det = nets.build_detector("yolov8n-pose", 0).float()
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        feats = None

        def grab(mod, inp):
            nonlocal feats
            feats = inp[0]
        h = det.detect.register_forward_pre_hook(grab)
        out = det(x)
        h.remove()
        raw = torch.cat([det.detect.cv4[i](f).view(1, 51, -1) for i, f in enumerate(feats)], 2).view(1, 17, 3, -1)
    A = out.shape[2]
    assert out.shape == (1, 4 + 1 + 51, A) and A == sum((64 // s) * (96 // s) for s in (8, 16, 32))
    k = out[0, 5:].view(17, 3, A)
    a, col = 0, []
    for s in (8, 16, 32):
        hh, ww = 64 // s, 96 // s
        gy, gx = torch.meshgrid(torch.arange(hh), torch.arange(ww), indexing="ij")
        col.append(torch.stack((gx.reshape(-1), gy.reshape(-1), torch.full((hh * ww,), s))).float())
    g = torch.cat(col, 1)                                                       # [3, A]: grid x, grid y, stride
    assert torch.allclose(k[:, 0], (raw[0, :, 0] * 2 + g[0]) * g[2], atol=1e-5)
    assert torch.allclose(k[:, 1], (raw[0, :, 1] * 2 + g[1]) * g[2], atol=1e-5)
    assert torch.allclose(k[:, 2], raw[0, :, 2].sigmoid(), atol=1e-6)


def test_byte_crop_table_is_the_oracle_crop_normalisation():
    """fused32.crops_from_u8 (the host form of what k32_stemW's staging applies to byte crops): on a constant-colour frame every crop pixel is
    ((v / 255) - mean) / sd of that colour — the C oracle's crop_norm and the table give the same float32 bits for all 256 values of every channel
    (the table is built with IEEE operations on the host; PyTorch-ROCm's device division is not correctly rounded)."""
    import numpy as np
    import torch
    from oracle import cexact
    from strongsort_yolo_amd import fused32
    for v0 in range(0, 256, 5):
        bgr = np.array([v0, (v0 * 7 + 3) % 256, (v0 * 13 + 11) % 256], np.uint8)
        img = np.broadcast_to(bgr, (64, 48, 3)).copy()
        ref = cexact.crop_norm(img, np.array([[4.0, 6.0, 40.0, 60.0, 1.0, 0.0]], np.float32))      # [1, 3, 256, 128] RGB
        rgb = bgr[::-1].copy()
        x = torch.from_numpy(np.broadcast_to(rgb.reshape(1, 3, 1, 1), (1, 3, 256, 128)).copy()).contiguous(memory_format=torch.channels_last)
        got = fused32.crops_from_u8(x).numpy()
        assert got.tobytes() == np.ascontiguousarray(ref).tobytes()


def test_decode_into_hands_out_the_callers_buffer_only_when_it_fits():
    import torch
    from strongsort_yolo_amd import fused
    buf = torch.zeros(2, 84, 100)
    assert fused.decode_into.target((2, 84, 100), buf.device).data_ptr() != buf.data_ptr()          # no block active: a fresh tensor
    with fused.decode_into(buf):
        assert fused.decode_into.target((2, 84, 100), buf.device).data_ptr() == buf.data_ptr()
        assert fused.decode_into.target((2, 85, 100), buf.device).data_ptr() != buf.data_ptr()      # another shape: a fresh tensor
        with fused.decode_into(None):
            assert fused.decode_into.target((2, 84, 100), buf.device).data_ptr() != buf.data_ptr()
        assert fused.decode_into.target((2, 84, 100), buf.device).data_ptr() == buf.data_ptr()      # the outer block is restored
    assert fused.decode_into.target((2, 84, 100), buf.device).data_ptr() != buf.data_ptr()
