"""NumPy rasteriser of the overlay primitive list — the oracle of csrc/ss_overlay.hip.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against the reference: /root/reference/yolo_multi_model.py:58-162 draws with
OpenCV (cv2.rectangle / circle / line / putText / addWeighted), which is not installed here and whose rasteriser is not
restated; this file fixes the coverage rules of the primitives (integer arithmetic, painter's order) that the HIP kernel
must reproduce bit for bit."""
import numpy as np

RECT, FILL, CIRCLE, LINE, TEXT, POLY = 0, 1, 2, 3, 4, 5


def _coverage(p, chars, font, H, W):
    """(y0, x0, mask) of primitive p clipped to the frame, or None."""
    t, x0, y0, x1, y1, _, a, b = (int(v) for v in p)
    if t == RECT:
        o = a >> 1
        bx0, by0, bx1, by1 = min(x0, x1) - o, min(y0, y1) - o, max(x0, x1) + o, max(y0, y1) + o
    elif t == FILL:
        bx0, by0, bx1, by1 = min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)
    elif t == CIRCLE:
        bx0, by0, bx1, by1 = x0 - a, y0 - a, x0 + a, y0 + a
    elif t == LINE:
        o = (a + 1) >> 1
        bx0, by0, bx1, by1 = min(x0, x1) - o, min(y0, y1) - o, max(x0, x1) + o, max(y0, y1) + o
    elif t == POLY:
        bx0, by0, bx1, by1 = x0, y0, x1, y1
    else:
        sc = max(b >> 1, 1)
        bx0, bx1, by1, by0 = x0, x0 + x1 * 6 * sc - 1, y0, y0 - 7 * sc + 1
    cx0, cy0, cx1, cy1 = max(bx0, 0), max(by0, 0), min(bx1, W - 1), min(by1, H - 1)
    if cx0 > cx1 or cy0 > cy1:
        return None
    ys, xs = np.mgrid[cy0:cy1 + 1, cx0:cx1 + 1].astype(np.int64)
    if t == RECT:
        X0, X1, Y0, Y1, inn = min(x0, x1), max(x0, x1), min(y0, y1), max(y0, y1), (a + 1) >> 1
        m = ~((xs >= X0 + inn) & (xs <= X1 - inn) & (ys >= Y0 + inn) & (ys <= Y1 - inn))
    elif t == FILL:
        m = np.ones(xs.shape, bool)
    elif t == CIRCLE:
        m = (xs - x0) ** 2 + (ys - y0) ** 2 <= a * a
    elif t == LINE:
        dx, dy, px, py = x1 - x0, y1 - y0, xs - x0, ys - y0
        L, tt, a2 = dx * dx + dy * dy, px * dx + py * dy, a * a
        cr = px * dy - py * dx
        m = np.where(tt <= 0, 4 * (px * px + py * py) <= a2,
                     np.where(tt >= L, 4 * ((xs - x1) ** 2 + (ys - y1) ** 2) <= a2, 4 * cr * cr <= a2 * L))
    elif t == POLY:
        pts = np.frombuffer(chars[a:a + 8 * (b >> 1)].tobytes(), np.int32).reshape(-1, 2)
        if len(pts) < 3:                                           # a degenerate polygon covers nothing
            return
        full = polygon_mask_np(pts, H, W)
        m = full[cy0:cy1 + 1, cx0:cx1 + 1]
    else:
        sc = max(b >> 1, 1)
        cx, cy = xs - x0, ys - (y0 - 7 * sc + 1)
        ci = cx // (6 * sc)
        col, row = (cx - ci * 6 * sc) // sc, cy // sc
        ch = chars[a + np.clip(ci, 0, max(x1 - 1, 0))].astype(np.int64)
        ch = np.where((ch < 32) | (ch > 126), ord("?"), ch)
        m = (col < 5) & (((font[ch - 32, np.minimum(col, 4)] >> row) & 1) == 1)
    return cy0, cx0, m


def _mix(top, under):
    return ((179 * top.astype(np.int64) + 77 * under.astype(np.int64) + 128) >> 8).astype(np.uint8)


def _half_even(a, b):
    t = a.astype(np.int64) + b.astype(np.int64)
    return ((t >> 1) + ((t & 1) & ((t >> 1) & 1))).astype(np.uint8)


def rasterise(frame, prims, chars, font, skip_poly=False):
    """frame uint8 [H,W,3] BGR (copied), prims int32 [n,8], chars uint8, font uint8 [95,5] -> annotated frame.  A POLY
    primitive blends its even-odd interior half and half (ties to even) with what is there; like every non-group primitive
    it first closes a blended group pending under the pixels it covers."""
    out = frame.copy()
    H, W = frame.shape[:2]
    grp = np.zeros_like(out)
    in_grp = np.zeros((H, W), bool)
    for p in prims:
        if skip_poly and int(p[0]) == POLY:
            continue
        cov = _coverage(p, chars, font, H, W)
        if cov is None:
            continue
        y0, x0, m = cov
        color = np.array([p[5] & 255, (p[5] >> 8) & 255, (p[5] >> 16) & 255], np.uint8)
        sl = (slice(y0, y0 + m.shape[0]), slice(x0, x0 + m.shape[1]))
        if p[7] & 1:
            grp[sl][m] = color
            in_grp[sl] |= m
        else:
            close = in_grp[sl] & m                                   # a group under this pixel ends here
            if close.any():
                o, g = out[sl], grp[sl]
                o[close] = _mix(g[close], o[close])
                in_grp[sl] &= ~m
            if int(p[0]) == POLY:
                o = out[sl]
                o[m] = _half_even(o[m], color[None, :])
            else:
                out[sl][m] = color
    if in_grp.any():
        out[in_grp] = _mix(grp[in_grp], out[in_grp])
    return out


# ---- mask fills (the reference's fillPoly + addWeighted 0.5, /root/reference/yolo_multi_model.py:112-121) ------------------------
def polygon_mask_np(poly, H, W):  # noqa: E302
    """Even-odd interior of the closed integer polygon [k, 2] on the H x W pixel grid: a pixel is inside when an odd number of
    edges straddle its y (exactly one end point with y <= the pixel's) strictly to its right, in exact rational arithmetic.
    Written per pixel and edge, the slow way (the product side vectorises over both)."""
    from fractions import Fraction
    q = [(int(x), int(y)) for x, y in np.asarray(poly).reshape(-1, 2)]
    m = np.zeros((H, W), bool)
    if len(q) < 3:
        return m
    ys = [p[1] for p in q]
    for py in range(max(min(ys), 0), min(max(ys), H - 1) + 1):
        xs = []
        for (ax, ay), (bx, by) in zip(q, q[1:] + q[:1]):
            if (ay <= py) != (by <= py):
                xs.append(ax + Fraction((py - ay) * (bx - ax), by - ay))
        for px in range(W):
            m[py, px] = sum(px < x for x in xs) % 2 == 1
    return m


def blend_polygon_np(frame, poly, color):
    """frame uint8 [H, W, 3] copied; inside the polygon (frame + color) / 2 rounded half to even."""
    out = frame.copy()
    m = polygon_mask_np(poly, *frame.shape[:2])
    mixed = np.rint((out[m].astype(np.float64) + np.asarray(color, np.float64)) / 2.0)
    out[m] = mixed.astype(np.uint8)
    return out


def rasterise_with_blends(frame, prims, chars, font, blends):
    """Primitive stretches and mask fills in painter's order: blends = [(primitives drawn before it, polygon, colour)]."""
    out, at = frame, 0
    for pos, poly, color in list(blends) + [(len(prims), None, None)]:
        if pos > at:
            out = rasterise(out, prims[at:pos], chars, font, skip_poly=True)     # the list's own POLY rows are what `blends` restates
        at = pos
        if poly is not None:
            out = blend_polygon_np(out, poly, color)
    return out if out is not frame else frame.copy()
