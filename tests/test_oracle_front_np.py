"""Independent NumPy restatements of the frame-side stages (written the way upstream-style code would write them:
vectorised, float64 interpolation) cross-checking the exact-order C oracle.  Like the NumPy tracker back end,
this guards the oracle against a systematic mistake that a self-comparison could not see."""
import numpy as np
import pytest

from oracle import cexact
from strongsort_yolo_amd.engine import letterbox_geometry


def _bilinear_f64(src, out_h, out_w):
    """cv2.INTER_LINEAR geometry (half-pixel centres, edge clamp) in float64, no intermediate rounding."""
    H, W = src.shape[:2]
    fy = (np.arange(out_h) + 0.5) * (H / out_h) - 0.5
    fx = (np.arange(out_w) + 0.5) * (W / out_w) - 0.5
    y0 = np.floor(fy).astype(int); x0 = np.floor(fx).astype(int)
    wy = fy - y0; wx = fx - x0
    wy[y0 < 0] = 0; y0 = np.maximum(y0, 0); wy[y0 >= H - 1] = 0; y0 = np.minimum(y0, H - 1); y1 = np.minimum(y0 + 1, H - 1)
    wx[x0 < 0] = 0; x0 = np.maximum(x0, 0); wx[x0 >= W - 1] = 0; x0 = np.minimum(x0, W - 1); x1 = np.minimum(x0 + 1, W - 1)
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - wx)[None, :, None] + s[y0][:, x1] * wx[None, :, None]
    bot = s[y1][:, x0] * (1 - wx)[None, :, None] + s[y1][:, x1] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


@pytest.mark.parametrize("hw", [(72, 128), (90, 77), (48, 64)])
def test_letterbox_vs_numpy(hw):
    H, W = hw
    img = np.random.default_rng(H).integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W, imgsz=64, stride=32)
    got = cexact.letterbox(img, g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
    ref = np.full((g.out_h, g.out_w, 3), 114.0)
    ref[g.pad_top:g.pad_top + g.new_h, g.pad_left:g.pad_left + g.new_w] = np.floor(_bilinear_f64(img, g.new_h, g.new_w) + 0.5)
    ref = (ref[:, :, ::-1] / 255.0).transpose(2, 0, 1)
    d = np.abs(got - ref)
    assert d.max() <= 1 / 255 + 1e-6            # at most one uint8 step where float32 vs float64 rounding straddles .5
    assert (d > 1e-6).mean() < 0.01


def test_crop_vs_numpy():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)
    dets = np.array([[10.3, 5.8, 50.2, 55.1, .9, 0], [-4, -4, 30, 20, .8, 0], [70, 40, 120, 90, .7, 0]], np.float32)
    got = cexact.crop_norm(img, dets, 32, 16)
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    for i, b in enumerate(dets):
        x1, y1 = max(int(b[0]), 0), max(int(b[1]), 0)
        x2, y2 = min(int(b[2]), 89), min(int(b[3]), 59)
        crop = img[y1:max(y2, y1 + 1), x1:max(x2, x1 + 1)]
        q = np.floor(_bilinear_f64(crop, 32, 16) + 0.5)[:, :, ::-1] / 255.0
        ref = ((q - mean) / std).transpose(2, 0, 1)
        d = np.abs(got[i] - ref)
        assert d.max() <= (1 / 255) / std.min() + 1e-5 and (d > 1e-5).mean() < 0.01


def _nms_numpy(pred, nc, conf, iou_thr, max_wh=7680.0):
    """textbook greedy NMS (Ultralytics recipe: best class, class offset) in vectorised float32."""
    box, cls = pred[:4].T, pred[4:4 + nc].T
    score, lab = cls.max(1), cls.argmax(1)
    idx = np.nonzero(score > conf)[0]
    order = idx[np.lexsort((idx, -score[idx]))]            # score desc, anchor asc
    xy, wh = box[order, :2], box[order, 2:] / np.float32(2)
    off = (lab[order].astype(np.float32) * np.float32(max_wh))[:, None]
    b = np.concatenate([xy - wh + off, xy + wh + off], 1).astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(len(order), bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(order[i])
        lt, rb = np.maximum(b[i, :2], b[i + 1:, :2]), np.minimum(b[i, 2:], b[i + 1:, 2:])
        whi = np.maximum(np.float32(0), rb - lt)
        inter = whi[:, 0] * whi[:, 1]
        alive[i + 1:] &= ~(inter / (area[i] + area[i + 1:] - inter) > np.float32(iou_thr))
    return np.array(keep, dtype=np.int32)


@pytest.mark.parametrize("seed", range(6))
def test_nms_vs_numpy(seed):
    rng = np.random.default_rng(seed)
    N, nc = 700, 4
    pred = np.zeros((4 + nc, N), np.float32)
    pred[0] = rng.uniform(0, 300, N); pred[1] = rng.uniform(0, 200, N)
    pred[2] = rng.uniform(8, 70, N); pred[3] = rng.uniform(8, 70, N)
    pred[4:] = rng.uniform(0, 1, (nc, N)) ** 2
    keep, rows = cexact.nms(pred, nc, 0.3, 0.4)
    assert np.array_equal(keep, _nms_numpy(pred, nc, np.float32(0.3), 0.4))
