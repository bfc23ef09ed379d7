"""N2/N3 on the CPU: the overlay command builder restates the reference's drawing sequence, the NumPy rasteriser
(oracle of the HIP kernel) obeys painter's order, and the frame sinks write what they are given."""
import json

import numpy as np
import torch

from oracle.overlay_np import rasterise
from strongsort_yolo_amd.cli import FrameSink
from strongsort_yolo_amd.overlay import CIRCLE, FILL, LINE, RECT, TEXT, CommandList, Overlay, bgr
from strongsort_yolo_amd.overlay_font import font_table
from strongsort_yolo_amd.yolo import Boxes, Keypoints, Results


def _results(ids=(7, 9)):
    n = len(ids)
    xyxy = torch.tensor([[40., 60., 120., 200.], [150., 30., 210., 180.]][:n])
    kp = torch.zeros(n, 17, 3)
    kp[0, 3] = torch.tensor([60., 90., 0.9])
    return [Results(np.zeros((240, 320, 3), np.uint8), {0: "person", 2: "car"}, Boxes(xyxy, torch.tensor([0.91, 0.5][:n]), torch.tensor([0., 2.][:n]),
                                                                                        torch.tensor([float(i) for i in ids])), Keypoints(kp))]


def test_commands_follow_the_reference_sequence():
    ov = Overlay({0: "person", 2: "car"})
    c1 = ov.commands(_results()).arrays()[0]
    # frame 1: keypoint (2 circles + index text), per box: outline, label plate, label text; no trails yet
    assert c1[:, 0].tolist() == [CIRCLE, CIRCLE, TEXT] + [RECT, FILL, TEXT] * 2
    assert c1[3, 1:5].tolist() == [40, 60, 120, 200] and c1[3, 5] == bgr(0, 0, 225) and c1[3, 6] == 2       # yolo_multi_model.py:80
    cl = ov.commands(_results(), counts={"person": 1}, fps_text="FPS: 30.00")
    prims, chars = cl.arrays()
    assert (prims[:, 0] == LINE).sum() == 2                              # one trajectory segment per id from the 2nd frame on (:156-162)
    assert prims[-1, 0] == TEXT and prims[-1, 7] == (2 << 1) and bytes(chars[prims[-1, 6]:prims[-1, 6] + prims[-1, 3]]) == b"FPS: 30.00"
    assert prims[-3, 0] == FILL and prims[-3, 7] & 1 and prims[-2, 7] & 1                                    # blended count plate (:311-318)
    label = bytes(chars[prims[5, 6]:prims[5, 6] + prims[5, 3]]).decode()
    assert label == " ID: 7 person 91.0%"                                                                    # :86
    ov.commands(_results(ids=(7,)))                                      # id 9 left the scene: its trajectory is dropped (:45-47)
    assert set(ov.trails) == {7}
    for _ in range(8):
        ov.commands(_results(ids=(7,)))
    assert len(ov.trails[7]) == 5                                        # deque(maxlen=5), :102


def test_numpy_rasteriser_painters_order_and_blend():
    font = font_table()
    img = np.full((40, 60, 3), 100, np.uint8)
    cl = CommandList()
    cl.fill(5, 5, 20, 15, bgr(10, 20, 30))
    cl.rect(10, 10, 30, 30, bgr(0, 0, 225), 2)          # drawn later: wins where both cover
    cl.circle(50, 8, 3, bgr(0, 255, 0))
    cl.line(0, 39, 59, 35, bgr(255, 255, 255), 2)
    cl.fill(40, 20, 55, 30, bgr(0, 0, 0), group=True)   # blended plate: 0.7 * black + 0.3 * 100 -> 30
    cl.text("Az", 2, 38, bgr(1, 2, 3))
    prims, chars = cl.arrays()
    out = rasterise(img, prims, chars, font)
    assert tuple(out[6, 6]) == (10, 20, 30) and tuple(out[10, 10]) == (0, 0, 225) and tuple(out[9, 9]) == (0, 0, 225)
    assert tuple(out[12, 12]) == (10, 20, 30) and tuple(out[20, 20]) == (100, 100, 100)       # inside the outline: untouched
    assert tuple(out[8, 50]) == (0, 255, 0) and tuple(out[8, 54]) == (100, 100, 100)
    assert tuple(out[25, 45]) == (30, 30, 30)
    assert (out[32:39, 2:7] == np.array([1, 2, 3])).all(axis=2).sum() == 18                  # the 18 pixels of the glyph 'A'
    assert np.array_equal(img, np.full((40, 60, 3), 100, np.uint8))                          # input not modified
    # clipping: primitives partly / fully outside the frame
    cl = CommandList(); cl.circle(-2, -2, 5, bgr(9, 9, 9)); cl.rect(50, 30, 80, 70, bgr(1, 1, 1), 3); cl.text("x", 100, 100, 0)
    out = rasterise(img, *cl.arrays(), font)
    assert tuple(out[0, 0]) == (9, 9, 9) and tuple(out[39, 49]) == (1, 1, 1)


def test_frame_sinks(tmp_path):
    frames = np.random.default_rng(0).integers(0, 255, (3, 6, 8, 3), dtype=np.uint8)
    for name in ("a.npy", "b.bgr", "dir"):
        s = FrameSink(str(tmp_path / name))
        for f in frames:
            s.write(f)
        s.close()
    assert np.array_equal(np.load(tmp_path / "a.npy"), frames)
    assert np.array_equal(np.fromfile(tmp_path / "b.bgr", np.uint8).reshape(frames.shape), frames)
    assert json.load(open(tmp_path / "b.bgr.json")) == {"width": 8, "height": 6, "fps": 15, "frames": 3, "pix_fmt": "bgr24"}
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "dir" / "frame_000002.png"))[:, :, ::-1], frames[2])


# ---- segmentation overlay (/root/reference/yolo_multi_model.py:71-121, :195-221) ------------------------------------------------
def _seg_results(tracked=True):
    from strongsort_yolo_amd.yolo import Masks
    xyxy = torch.tensor([[40., 60., 120., 200.], [150., 30., 210., 180.]])
    proto = torch.zeros(2, 60, 80)
    proto[0, 20:45, 12:28] = 1.0
    proto[1, 10:40, 40:50] = 1.0
    mk = Masks(proto, torch.eye(2), torch.tensor([[0., 0., 320., 240.]] * 2), (240, 320), (240, 320, 3), 1.0, (0.0, 0.0))
    ids = torch.tensor([7., 9.]) if tracked else None
    return [Results(np.zeros((240, 320, 3), np.uint8), {0: "person", 2: "car"}, Boxes(xyxy, torch.tensor([0.91, 0.5]), torch.tensor([0., 2.]), ids), None, mk)]


def test_polygon_fill_equals_the_exact_oracle():
    from oracle.overlay_np import blend_polygon_np, polygon_mask_np
    from strongsort_yolo_amd.overlay import blend_polygon_, polygon_mask
    rng = np.random.default_rng(4)
    polys = [np.array([[2, 2], [12, 2], [12, 9], [7, 5], [2, 9]]), np.array([[5, 5], [30, 25], [5, 25], [30, 5]]),        # concave, bow tie
             np.array([[-6, 3], [20, -4], [44, 18], [10, 40]]), np.array([[3, 3], [9, 3]]), np.array([[4, 4], [4, 4], [4, 4]])]
    polys += [rng.integers(-5, 45, (int(rng.integers(3, 12)), 2)) for _ in range(12)]
    for q in polys:
        ref = polygon_mask_np(q, 36, 40)
        got = polygon_mask(torch.from_numpy(np.asarray(q, np.int64)), 0, 0, 36, 40).numpy() if len(q) >= 3 else np.zeros((36, 40), bool)
        assert np.array_equal(got, ref), q.tolist()
        frame = rng.integers(0, 256, (36, 40, 3), dtype=np.uint8)
        color = tuple(int(c) for c in rng.integers(0, 256, 3))
        out = blend_polygon_(torch.from_numpy(frame.copy()), q, color).numpy()
        assert np.array_equal(out, blend_polygon_np(frame, q, color))
        assert np.array_equal(out[~ref], frame[~ref])                     # outside: addWeighted(x, .5, x, .5) = x


def test_mask_commands_interleave_boxes_and_fills():
    from strongsort_yolo_amd.overlay import CLASS_COLORS
    ov = Overlay({0: "person", 2: "car"})
    cl = ov.commands(_seg_results())
    prims = cl.arrays()[0]
    assert len(cl.blends) == 2
    (p0, q0, c0), (p1, q1, c1) = cl.blends
    # pair 0: box (outline, plate, text), its polygon's edges, THEN its fill (a POLY primitive at position p0: one ordered list, one
    # launch); pair 1's box comes after that fill (drawn over it)
    from strongsort_yolo_amd.overlay import POLY
    assert prims[:3, 0].tolist() == [RECT, FILL, TEXT] and (prims[3:p0, 0] == LINE).all() and p0 - 3 == len(q0)
    assert prims[p0, 0] == POLY and prims[p0 + 1:p0 + 4, 0].tolist() == [RECT, FILL, TEXT] and prims[p0 + 1, 1:5].tolist() == [150, 30, 210, 180]
    assert (prims[3:p0, 5] == bgr(255, 0, 0)).all() and c0 == tuple(int(v) for v in CLASS_COLORS[0]) and c1 == tuple(int(v) for v in CLASS_COLORS[2])
    assert q0[:, 0].min() >= 40 and q0[:, 0].max() <= 120 and p1 == len(prims) - 1 and prims[p1, 0] == POLY
    # the POLY row: bounding box, colour, vertex count and a 4-byte aligned offset to the vertices in the character buffer
    chars = cl.arrays()[1]
    assert prims[p0, 1:5].tolist() == [q0[:, 0].min(), q0[:, 1].min(), q0[:, 0].max(), q0[:, 1].max()] and prims[p0, 5] == bgr(*c0)
    assert prims[p0, 6] % 4 == 0 and prims[p0, 7] == len(q0) << 1 and len(chars) % 4 == 0
    assert np.array_equal(np.frombuffer(chars[prims[p0, 6]:prims[p0, 6] + 8 * len(q0)].tobytes(), np.int32).reshape(-1, 2), q0)
    n_lines_first = int((prims[:, 0] == LINE).sum())
    cl2 = ov.commands(_seg_results())                                    # second frame: one trail segment per id, drawn per pair (:101-110)
    assert int((cl2.arrays()[0][:, 0] == LINE).sum()) == n_lines_first + 1 + 2
    det = Overlay({0: "person", 2: "car"}).commands(_seg_results(tracked=False))
    lab = det.arrays()
    assert len(det.blends) == 2 and bytes(lab[1][lab[0][2, 6]:lab[0][2, 6] + lab[0][2, 3]]).decode() == " person 91.0%"


def test_rasterise_with_blends_is_painters_order():
    from oracle.overlay_np import rasterise_with_blends
    font = font_table()
    frame = np.full((240, 320, 3), 90, np.uint8)
    cl = Overlay({0: "person", 2: "car"}).commands(_seg_results())
    prims, chars = cl.arrays()
    out = rasterise_with_blends(frame, prims, chars, font, cl.blends)
    (p0, q0, c0), _ = cl.blends
    cx, cy = int(q0[:, 0].mean()), int(q0[:, 1].mean())
    assert tuple(out[cy, cx]) == tuple(int(np.rint((90 + c) / 2)) for c in c0)                   # inside mask 0: blended once
    assert tuple(out[130, 40]) == (0, 0, 225) and tuple(out[100, 150]) == (0, 0, 225)            # both box outlines (left edges) survive
    assert np.array_equal(rasterise_with_blends(frame, prims, chars, font, []), rasterise(frame, prims, chars, font, skip_poly=True))
    # the one-pass form (POLY primitives in the list: what the kernel executes) == the two-pass form (stretches + fills)
    assert np.array_equal(rasterise(frame, prims, chars, font), out)
    noisy = np.random.default_rng(3).integers(0, 256, frame.shape, dtype=np.uint8)
    cl2 = Overlay({0: "person", 2: "car"}).commands(_seg_results(), {"person": 1}, "FPS: 1.00")      # + blended count plate and FPS text
    assert np.array_equal(rasterise(noisy, *cl2.arrays(), font), rasterise_with_blends(noisy, *cl2.arrays(), font, cl2.blends))
