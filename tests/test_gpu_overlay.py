"""N2 on the GPU: ss_overlay == the NumPy rasteriser, pixel for pixel (random primitive lists with overlaps, clipping,
blended groups, text), batches of frames, and the host-frame round trip used by the CLI's --save."""
import numpy as np
import pytest
import torch

from oracle.overlay_np import rasterise
from strongsort_yolo_amd.overlay import CommandList, Overlay, bgr
from strongsort_yolo_amd.overlay_font import font_table
from tests.gpu_util import engine

pytestmark = pytest.mark.gpu


def _random_list(rng, W, H, n):
    cl = CommandList()
    for _ in range(n):
        k = int(rng.integers(0, 6))
        x0, y0 = int(rng.integers(-20, W + 20)), int(rng.integers(-20, H + 20))
        x1, y1 = x0 + int(rng.integers(-60, 120)), y0 + int(rng.integers(-40, 90))
        col, grp = int(rng.integers(0, 1 << 24)), bool(rng.random() < 0.25)
        if k == 0: cl.rect(x0, y0, x1, y1, col, int(rng.integers(1, 5)), grp)
        elif k == 1: cl.fill(x0, y0, x1, y1, col, grp)
        elif k == 2: cl.circle(x0, y0, int(rng.integers(0, 14)), col, grp)
        elif k == 3: cl.line(x0, y0, x1, y1, col, int(rng.integers(1, 6)), grp)
        elif k == 5:                                             # mask fill: random (self-intersecting) polygon, even-odd interior, half-half blend
            nv = int(rng.integers(1, 14))
            cl.blend(np.stack([x0 + rng.integers(-70, 90, nv), y0 + rng.integers(-60, 70, nv)], 1), (col & 255, (col >> 8) & 255, (col >> 16) & 255))
        else: cl.text("".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(0, 24)))), x0, y0, col, int(rng.integers(1, 4)), grp)
    return cl


@pytest.mark.parametrize("wh,n", [((320, 240), 40), ((1280, 720), 700), ((333, 97), 300)])
def test_overlay_kernel_equals_numpy_rasteriser(wh, n):
    W, H = wh
    rng = np.random.default_rng(W + n)
    eng = engine(debug=False)
    ov = Overlay({}, eng)
    B = 3
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    lists = [_random_list(rng, W, H, n if b else 0) for b in range(B)]            # frame 0: nothing to draw -> untouched
    d = torch.from_numpy(frames).to(eng.device)
    ov.draw_device(d, lists)
    got = d.cpu().numpy()
    font = font_table()
    for b in range(B):
        ref = rasterise(frames[b], *lists[b].arrays(), font)
        assert np.array_equal(got[b], ref), f"frame {b}: {int((got[b] != ref).any(axis=2).sum())} pixels differ"
    assert np.array_equal(got[0], frames[0])
    eng.close()


def test_overlay_draw_results_round_trip():
    from tests.test_overlay_cpu import _results
    eng = engine(debug=False)
    ov, ov_ref = Overlay({0: "person", 2: "car"}, eng), Overlay({0: "person", 2: "car"})
    frame = np.random.default_rng(1).integers(0, 256, (240, 320, 3), dtype=np.uint8)
    font = font_table()
    for k in range(3):                                                   # trajectories grow from frame to frame
        got = ov.draw(frame, _results(), counts={"person": 1, "car": 1}, fps_text="FPS: 12.50")
        ref = rasterise(frame, *ov_ref.commands(_results(), {"person": 1, "car": 1}, "FPS: 12.50").arrays(), font)
        assert np.array_equal(got, ref) and not np.array_equal(got, frame)
    eng.close()


def test_overlay_mask_fills_equal_the_oracle():
    """Segmentation overlay (yolo_multi_model.py:112-121): polygon outlines as line primitives, fills as exact even-odd blends between
    the primitive stretches — device result == oracle.overlay_np.rasterise_with_blends, pixel for pixel; tracked and detect-only."""
    from oracle.overlay_np import rasterise_with_blends
    from tests.test_overlay_cpu import _seg_results
    eng = engine(debug=False)
    frame = np.random.default_rng(2).integers(0, 256, (240, 320, 3), dtype=np.uint8)
    font = font_table()
    for tracked in (True, False):
        ov, ov_ref = Overlay({0: "person", 2: "car"}, eng), Overlay({0: "person", 2: "car"})
        for k in range(2):
            got = ov.draw(frame, _seg_results(tracked), fps_text="FPS: 9.00")
            cl = ov_ref.commands(_seg_results(tracked), None, "FPS: 9.00")
            ref = rasterise_with_blends(frame, *cl.arrays(), font, cl.blends)
            assert len(cl.blends) == 2 and np.array_equal(got, ref), f"tracked={tracked} frame {k}: {int((got != ref).any(axis=2).sum())} pixels differ"
    eng.close()
