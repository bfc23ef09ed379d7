"""Property tests (hypothesis) of the segmentation results' host code: the traced outline is a closed 8-connected chain on the
component's boundary that spans its bounding box; the exact polygon fill agrees with the per-pixel rational oracle and is
invariant under rotation of the vertex list and under reversal of its orientation."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle.overlay_np import polygon_mask_np
from strongsort_yolo_amd.overlay import polygon_mask
from strongsort_yolo_amd.yolo import mask_polygon, trace_outline


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(4, 24), st.integers(4, 24), st.floats(0.25, 0.9))
def test_outline_is_a_closed_chain_on_the_boundary(seed, h, w, fill):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    lab, n = ndimage.label(rng.random((h, w)) < fill, structure=np.ones((3, 3)))
    if n == 0:
        return
    comp = lab == 1
    p = trace_outline(comp)
    ys, xs = np.nonzero(comp)
    assert p[:, 0].min() == xs.min() and p[:, 0].max() == xs.max() and p[:, 1].min() == ys.min() and p[:, 1].max() == ys.max()
    assert p[0].tolist() == [int(xs[0]), int(ys[0])]                      # starts at the first pixel in raster order
    q = np.vstack([p, p[:1]])
    for a, b in zip(q[:-1], q[1:]):
        d = b - a
        steps = int(np.abs(d).max())
        assert len(p) == 1 or steps > 0
        if steps:
            assert (np.abs(d)[np.abs(d) > 0] == steps).all()             # horizontal, vertical or diagonal run
            for t in range(steps + 1):
                x, y = a + (d // steps) * t
                assert comp[y, x]
    if len(p) > 2:                                                         # no vertex in the middle of a straight run
        din, dout = np.sign(p - np.roll(p, 1, 0)), np.sign(np.roll(p, -1, 0) - p)
        assert (din != dout).any(1).all()
    assert len(mask_polygon(lab > 0)) >= 1


@settings(max_examples=80, deadline=None)
@given(st.lists(st.tuples(st.integers(-6, 30), st.integers(-6, 26)), min_size=3, max_size=9), st.integers(0, 8))
def test_polygon_fill_matches_the_rational_oracle(pts, shift):
    H, W = 22, 26
    q = np.asarray(pts, np.int64)
    ref = polygon_mask_np(q, H, W)
    got = polygon_mask(torch.from_numpy(q), 0, 0, H, W).numpy()
    assert np.array_equal(got, ref)
    rolled = np.roll(q, shift % len(q), 0)
    assert np.array_equal(polygon_mask(torch.from_numpy(rolled), 0, 0, H, W).numpy(), ref)
    assert np.array_equal(polygon_mask(torch.from_numpy(q[::-1].copy()), 0, 0, H, W).numpy(), ref)
    # a window of the grid gives the same pixels as the full grid
    sub = polygon_mask(torch.from_numpy(q), 3, 2, 10, 12).numpy()
    assert np.array_equal(sub, ref[2:12, 3:15])
