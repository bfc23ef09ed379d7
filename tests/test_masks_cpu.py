"""Host half of the segmentation results (yolo.Masks: /root/reference/yolo_multi_model.py:71-72, :112-121 iterate `masks` with
the boxes and draw `masks.xy`): mask assembly from prototypes + coefficients, outline tracing, the Results duck type."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from strongsort_yolo_amd.yolo import Masks, assemble_masks, mask_polygon, trace_outline


def test_outline_of_simple_shapes():
    m = np.zeros((8, 10), bool)
    m[2:6, 3:8] = True                                            # rectangle: its four corners, clockwise from the top-left one
    assert trace_outline(m).tolist() == [[3, 2], [7, 2], [7, 5], [3, 5]]
    m[:] = False
    m[4, 4] = True                                                # a single pixel
    assert trace_outline(m).tolist() == [[4, 4]]
    m[:] = False
    m[1, 1:6] = True                                              # a one-pixel-wide bar is walked there and back
    assert trace_outline(m).tolist() == [[1, 1], [5, 1]]
    m[:] = False
    m[1:7, 2] = True
    m[6, 2:8] = True                                              # an L: the chain turns at the knee and comes back along itself
    p = trace_outline(m)
    assert p[0].tolist() == [2, 1] and [2, 6] in p.tolist() and [7, 6] in p.tolist()
    m[:] = False
    m[1:7, 1:9] = True
    m[3:5, 3:6] = False                                           # a hole is not part of the OUTER boundary
    assert trace_outline(m).tolist() == [[1, 1], [8, 1], [8, 6], [1, 6]]
    assert trace_outline(np.zeros((3, 3), bool)).shape == (0, 2)


@pytest.mark.parametrize("seed", range(8))
def test_outline_properties_on_random_blobs(seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    m = ndimage.binary_dilation(rng.random((40, 60)) > 0.97, iterations=3)
    lab, n = ndimage.label(m, structure=np.ones((3, 3)))
    for i in range(1, min(n, 5) + 1):
        comp = lab == i
        p = trace_outline(comp)
        ys, xs = np.nonzero(comp)
        assert p[:, 0].min() == xs.min() and p[:, 0].max() == xs.max() and p[:, 1].min() == ys.min() and p[:, 1].max() == ys.max()
        q = np.vstack([p, p[:1]])
        for a, b in zip(q[:-1], q[1:]):                           # every edge is a straight 8-direction run over component pixels
            d = b - a
            steps = int(np.abs(d).max())
            assert steps == 0 or (np.abs(d)[np.abs(d) > 0] == steps).all()
            for t in range(steps + 1):
                x, y = a + (d // max(steps, 1)) * t
                assert comp[y, x]
                nb = comp[max(y - 1, 0):y + 2, max(x - 1, 0):x + 2]
                assert (not nb.all()) or y in (0, 39) or x in (0, 59)      # ... that touch the background (or the frame edge)


def test_mask_polygon_takes_the_longest_outline():
    m = np.zeros((30, 30), bool)
    m[2:5, 2:5] = True
    m[10:25, 8:28:1] = True
    m[12:23:2, 8] = False                                         # a ragged left edge: many vertices
    p = mask_polygon(m)
    assert p.dtype == np.float32 and p[:, 1].min() == 10 and p[:, 0].max() == 27
    assert mask_polygon(np.zeros((4, 4), bool)).shape == (0, 2)


def test_assemble_masks_is_the_published_recipe():
    g = torch.Generator().manual_seed(5)
    nm, mh, mw, ih, iw = 8, 12, 20, 48, 80
    proto = torch.randn(nm, mh, mw, generator=g).half()
    coef = torch.randn(3, nm, generator=g)
    boxes = torch.tensor([[8.0, 4.0, 60.0, 40.0], [0.0, 0.0, 80.0, 48.0], [30.5, 10.2, 33.0, 14.9]])
    got = assemble_masks(proto, coef, boxes, (ih, iw))
    assert got.shape == (3, ih, iw) and got.dtype == torch.bool
    for i in range(3):                                            # one mask at a time, written out
        lin = (coef[i].view(nm, 1, 1) * proto.float()).sum(0)
        x1, y1, x2, y2 = (boxes[i] * torch.tensor([mw / iw, mh / ih, mw / iw, mh / ih])).tolist()
        keep = torch.zeros(mh, mw)
        for r in range(mh):
            for c in range(mw):
                keep[r, c] = float(x1 <= c < x2 and y1 <= r < y2)
        up = F.interpolate((lin * keep)[None, None], (ih, iw), mode="bilinear", align_corners=False)[0, 0]
        assert torch.equal(got[i], up > 0)
    # nothing survives far outside its box (one prototype cell of bleed at most: 4 input pixels here)
    ys, xs = torch.nonzero(got[0], as_tuple=True)
    assert xs.min() >= 8 - 4 and xs.max() <= 60 + 4 and ys.min() >= 4 - 4 and ys.max() <= 40 + 4
    assert assemble_masks(proto, coef[:0], boxes[:0], (ih, iw)).shape == (0, ih, iw)


def test_masks_object_follows_the_boxes_and_scales_polygons_back():
    nm, mh, mw, ih, iw = 4, 24, 40, 96, 160
    proto = torch.zeros(nm, mh, mw)
    proto[0, 6:18, 10:30] = 1.0                                   # a rectangle of prototype cells
    proto[1] = -1.0
    coef = torch.tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0]])
    boxes = torch.tensor([[0.0, 0.0, 160.0, 96.0]] * 2)
    gain, pad = 0.5, (0.0, 8.0)                                   # a 320 x 160 source letterboxed into 160 x 96
    mk = Masks(proto, coef, boxes, (ih, iw), (160, 320, 3), gain, pad)
    assert len(mk) == 2 and mk.data.shape == (2, ih, iw) and mk.data[0].any() and not mk.data[1].any()
    p0, p1 = mk.xy
    assert p1.shape == (0, 2) and p0.shape[1] == 2 and p0.dtype == np.float32
    # prototype cells 10..29 x 6..17 cover input pixels ~38..121 x ~22..73 -> source pixels (x - 0) / 0.5, (y - 8) / 0.5
    assert 70 <= p0[:, 0].min() <= 84 and 236 <= p0[:, 0].max() <= 248 and 24 <= p0[:, 1].min() <= 36 and 124 <= p0[:, 1].max() <= 136
    one = list(mk)[0]
    assert len(one) == 1 and np.array_equal(one.xy[0], p0) and torch.equal(mk[torch.tensor([1, 0])].data[1], mk.data[0])
    assert all((q >= 0).all() and (q <= 1).all() for q in mk.xyn if len(q))


def test_results_carry_masks_in_track_order():
    """YOLO._results: masks are cut with the DETECTION boxes and follow `det_idx` of the tracker rows, like the keypoints."""
    from types import SimpleNamespace
    from strongsort_yolo_amd.yolo import YOLO
    nm, mh, mw = 4, 24, 40
    pipe = SimpleNamespace(nk=0, nm=nm, gain=0.5, pad_x=0.0, pad_y=8.0, geom=SimpleNamespace(out_h=96, out_w=160))
    proto = torch.zeros(nm, mh, mw)
    proto[0, 4:10, 4:12] = 1.0
    proto[1, 12:20, 20:36] = 1.0
    dets = torch.zeros(2, 6 + nm)
    dets[0, :6] = torch.tensor([20.0, 0.0, 110.0, 70.0, 0.9, 0.0])          # source-frame boxes around the two blobs
    dets[1, :6] = torch.tensor([150.0, 70.0, 300.0, 150.0, 0.8, 2.0])
    dets[0, 6], dets[1, 7] = 1.0, 1.0
    img = np.zeros((160, 320, 3), np.uint8)
    model = YOLO("yolov8n-seg.pt", random_init_ok=True)
    r = model._results(img, pipe, dets, None, proto)[0]                     # predict: detection order
    assert len(r.masks) == 2 and r.masks.data[0].any() and r.masks.data[1].any()
    assert r.masks.xy[0][:, 0].max() < 120 and r.masks.xy[1][:, 0].min() > 140
    rows = torch.tensor([[150.0, 70, 300, 150, 7, 2, 0.8, 1], [20.0, 0, 110, 70, 3, 0, 0.9, 0], [0, 0, 5, 5, 9, 0, 0.5, -1]])
    t = model._results(img, pipe, dets, rows, proto)[0]                     # track: tracker rows, det_idx 1 then 0; a coasting row dropped
    assert t.boxes.id.tolist() == [7.0, 3.0] and len(t.masks) == 2
    assert np.array_equal(t.masks.xy[0], r.masks.xy[1]) and np.array_equal(t.masks.xy[1], r.masks.xy[0])
    for box, m in zip(t.boxes, t.masks):                                     # the reference's loop shape (:71-72)
        assert len(m.xy) == 1 and box.xyxy.shape == (1, 4)
    assert model._results(img, pipe, dets, rows[2:], proto)[0].masks is None
