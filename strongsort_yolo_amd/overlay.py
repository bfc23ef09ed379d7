"""Annotation overlay — the drawing half of the reference's `process()` and count plate
(/root/reference/yolo_multi_model.py:45-162, :284-331) as an ordered primitive list rasterised on the MI355X.

`Overlay.commands(results, ...)` restates the reference's per-frame drawing sequence as data (it also keeps the
5-point trajectories, :28, :101-110); `Overlay.draw(frame, results, ...)` uploads the frame (if it is a host array),
runs `ss_overlay` and returns the annotated frame.  The stroke font of cv2.putText is replaced by a 5x7 raster font
(overlay_font.py); rectangle / circle / line coverage is defined by integer tests (csrc/ss_overlay.hip), not by
OpenCV's rasteriser — pixel parity with cv2 is therefore not claimed, parity with oracle/overlay_np.py is exact.
"""
from __future__ import annotations

import ctypes as C
from collections import deque
from typing import Dict, Optional

import numpy as np
import torch

from .overlay_font import ADVANCE, font_table

RECT, FILL, CIRCLE, LINE, TEXT, POLY = 0, 1, 2, 3, 4, 5


def bgr(b, g, r):
    return int(b) | int(g) << 8 | int(r) << 16


# per-class fill colours of the mask blend; the reference draws 80 random ones at import (yolo_multi_model.py:25, unseeded) — here a fixed table
CLASS_COLORS = np.random.default_rng(25).integers(0, 255, size=(80, 3), dtype=np.uint8)


def polygon_mask(poly: torch.Tensor, x0: int, y0: int, h: int, w: int) -> torch.Tensor:
    """Even-odd interior of the closed integer polygon `poly` [k, 2] (x, y) on the pixel grid x0..x0+w-1 x y0..y0+h-1, in exact
    integer arithmetic (an edge is crossed when exactly one end point has y <= the pixel's y and the pixel is strictly left of
    the crossing).  Stands for cv2.fillPoly's interior (yolo_multi_model.py:118); cv2's rasteriser — which also paints the
    boundary pixels — is not restated, the boundary is the polyline drawn just before.  -> bool [h, w] on poly's device."""
    dev = poly.device
    p0 = poly.to(torch.int64)
    p1 = torch.roll(p0, -1, 0)
    px = (torch.arange(w, device=dev, dtype=torch.int64) + x0).view(1, 1, w)
    py = (torch.arange(h, device=dev, dtype=torch.int64) + y0).view(1, h, 1)
    ax, ay, bx, by = (t.view(-1, 1, 1) for t in (p0[:, 0], p0[:, 1], p1[:, 0], p1[:, 1]))
    straddle = (ay <= py) != (by <= py)
    dy = by - ay
    lhs, rhs = (px - ax) * dy, (py - ay) * (bx - ax)                     # px < ax + (py - ay) (bx - ax) / dy, cross-multiplied
    left = torch.where(dy > 0, lhs < rhs, lhs > rhs)
    return ((straddle & left).sum(0) & 1).bool()


def blend_polygon_(frame: torch.Tensor, poly: np.ndarray, color, alpha_num: int = 1, alpha_den: int = 2) -> torch.Tensor:
    """frame uint8 [H, W, 3] (any device), in place: inside the polygon round_half_even((frame + color) / 2) — what
    cv2.addWeighted(image, 0.5, filled_copy, 0.5, 0) leaves there (:119-121); outside nothing changes."""
    H, W = frame.shape[:2]
    q = np.asarray(poly, np.int64).reshape(-1, 2)
    if len(q) < 3:
        return frame
    x0, y0 = max(int(q[:, 0].min()), 0), max(int(q[:, 1].min()), 0)
    x1, y1 = min(int(q[:, 0].max()), W - 1), min(int(q[:, 1].max()), H - 1)
    if x0 > x1 or y0 > y1:
        return frame
    m = polygon_mask(torch.from_numpy(q).to(frame.device), x0, y0, y1 - y0 + 1, x1 - x0 + 1)
    reg = frame[y0:y1 + 1, x0:x1 + 1]
    c = torch.tensor([int(v) for v in color], dtype=torch.int32, device=frame.device).view(1, 1, 3)
    tot = reg.to(torch.int32) + c                                        # (a + b) / 2, ties to even
    half = (tot >> 1) + ((tot & 1) & ((tot >> 1) & 1))
    reg.copy_(torch.where(m.unsqueeze(-1), half.to(torch.uint8), reg))
    return frame


class CommandList:
    """Ordered primitives of one frame: int32 [n,8] rows + the characters of its text primitives."""

    def __init__(self):
        self.rows, self.chars = [], bytearray()
        self.blends = []                       # (primitives drawn before it, int32 polygon [k, 2], BGR colour): mask fills, in painter's order

    def blend(self, poly, color):
        """Mask fill: the polygon's even-odd interior mixed half and half with the frame — an ordered primitive like the others
        (POLY: bounding box + the vertices, stored 4-byte aligned in the character buffer), so a frame with masks is still ONE
        launch in painter's order.  `blends` restates the fills for the two-pass checker (oracle rasterise_with_blends)."""
        q = np.ascontiguousarray(np.asarray(poly, np.int32).reshape(-1, 2))
        col = tuple(int(c) for c in color)
        self.blends.append((len(self.rows), q, col))
        if len(q) < 3:
            self.rows.append((POLY, 1, 1, 0, 0, 0, 0, 0))       # degenerate polygon: 0 vertices, covers nothing (ov_covers: n < 3), keeps positions
            return
        self.chars += b"\0" * (-len(self.chars) % 4)
        self.rows.append((POLY, int(q[:, 0].min()), int(q[:, 1].min()), int(q[:, 0].max()), int(q[:, 1].max()), bgr(*col), len(self.chars), len(q) << 1))
        self.chars += q.tobytes()

    def polyline(self, poly, color, thickness=2):
        q = np.asarray(poly, np.int32).reshape(-1, 2)
        for a, b in zip(q, np.roll(q, -1, 0)) if len(q) > 1 else ():
            self.line(a[0], a[1], b[0], b[1], color, thickness)

    def rect(self, x0, y0, x1, y1, color, thickness=2, group=False):
        self.rows.append((RECT, int(x0), int(y0), int(x1), int(y1), color, int(thickness), int(group)))

    def fill(self, x0, y0, x1, y1, color, group=False):
        self.rows.append((FILL, int(x0), int(y0), int(x1), int(y1), color, 0, int(group)))

    def circle(self, x, y, r, color, group=False):
        self.rows.append((CIRCLE, int(x), int(y), 0, 0, color, int(r), int(group)))

    def line(self, x0, y0, x1, y1, color, thickness=2, group=False):
        self.rows.append((LINE, int(x0), int(y0), int(x1), int(y1), color, int(thickness), int(group)))

    def text(self, s, x, y, color, scale=1, group=False):
        s = s.encode("ascii", "replace")
        self.rows.append((TEXT, int(x), int(y), len(s), 0, color, len(self.chars), (int(scale) << 1) | int(group)))
        self.chars += s

    def arrays(self):
        ch = bytes(self.chars) + b"\0" * (-len(self.chars) % 4)       # a multiple of 4 bytes: frames are concatenated, vertex offsets stay aligned
        return (np.asarray(self.rows, dtype=np.int32).reshape(-1, 8), np.frombuffer(ch or b"\0\0\0\0", dtype=np.uint8).copy())


class Overlay:
    """Per-stream drawing state (trajectories) + the device launcher."""

    def __init__(self, names: Dict[int, str], engine=None, trail_len: int = 5):
        self.names, self.eng, self.trails = names, engine, {}
        self.trail_len = trail_len
        self._scratch = None
        if engine is not None:
            t = np.ascontiguousarray(font_table())
            engine._ck(engine.L.ss_overlay_set_font(engine.ctx, t.ctypes.data_as(C.c_void_p)))

    # ---- the reference's drawing sequence as data ---------------------------------------------------------
    def commands(self, results, counts: Optional[dict] = None, fps_text: str = "") -> CommandList:
        cl = CommandList()
        live = {int(i) for r in results if r is not None and r.boxes is not None and r.boxes.id is not None for i in r.boxes.id}
        for id_ in list(self.trails):                                   # yolo_multi_model.py:45-47
            if id_ not in live:
                del self.trails[id_]
        for r in results:
            if r is None or r.boxes is None:
                continue
            tracked = r.boxes.id is not None
            if r.keypoints is not None and (tracked or not hasattr(r.boxes, "id")):           # :58-67 / :182-191
                for kp in r.keypoints:
                    for pts in kp.xy.tolist():
                        for idx, (x, y) in enumerate(pts):
                            if (x, y) != (0.0, 0.0):
                                cl.circle(x, y, 5, bgr(0, 255, 0)); cl.circle(x, y, 2, bgr(0, 0, 0))
                                cl.text(str(idx), int(x) + 5, int(y) - 5, bgr(0, 0, 255))
            if getattr(r, "masks", None) is not None:                                         # :71-121 / :195-221: box, (trails,) mask, per pair
                ids = r.boxes.id if tracked else [None] * len(r.boxes)
                for conf, cls, xyxy, id_, polys in zip(r.boxes.conf, r.boxes.cls, r.boxes.xyxy, ids, r.masks.xy):
                    name = f"{self.names.get(int(cls), int(cls))} {round(float(conf) * 100, 1)}%"
                    self._box(cl, xyxy, f" ID: {int(id_)} {name}" if tracked else f" {name}")
                    if tracked:
                        t = self.trails.setdefault(int(id_), deque(maxlen=self.trail_len))
                        t.append(((float(xyxy[0]) + float(xyxy[2])) / 2, (float(xyxy[1]) + float(xyxy[3])) / 2))
                        for tr in self.trails.values():                                          # :106-110, inside the pair loop
                            for i in range(1, len(tr)):
                                cl.line(int(tr[i - 1][0]), int(tr[i - 1][1]), int(tr[i][0]), int(tr[i][1]), bgr(255, 255, 255), 2)
                    for poly in (polys if isinstance(polys, (list, tuple)) else [polys]):
                        q = np.int32(poly)                                                       # np.int32(polygon), :114
                        if len(q):
                            cl.polyline(q, bgr(255, 0, 0), 2)                                    # :114
                            cl.blend(q, CLASS_COLORS[int(cls) % len(CLASS_COLORS)])              # :116-121
                continue
            if not tracked:
                for conf, cls, xyxy in zip(r.boxes.conf, r.boxes.cls, r.boxes.xyxy):          # detection only, :193-237
                    self._box(cl, xyxy, f" {self.names.get(int(cls), int(cls))} {round(float(conf) * 100, 1)}%")
                continue
            for conf, cls, xyxy, id_ in zip(r.boxes.conf, r.boxes.cls, r.boxes.xyxy, r.boxes.id):   # :126-153
                self._box(cl, xyxy, f" ID: {int(id_)} {self.names.get(int(cls), int(cls))} {round(float(conf) * 100, 1)}%")
                t = self.trails.setdefault(int(id_), deque(maxlen=self.trail_len))
                t.append(((float(xyxy[0]) + float(xyxy[2])) / 2, (float(xyxy[1]) + float(xyxy[3])) / 2))
            for t in self.trails.values():                                                           # :156-162
                for i in range(1, len(t)):
                    cl.line(int(t[i - 1][0]), int(t[i - 1][1]), int(t[i][0]), int(t[i][1]), bgr(255, 255, 255), 2)
        if counts is not None:                                                                       # :311-318 blended plate
            s = str(counts)
            cl.fill(10, 11, max(10 + ADVANCE * 2 * len(s) + 20, 60), 70, bgr(0, 0, 0), group=True)
            cl.text(s, 20, 52, bgr(210, 210, 210), scale=2, group=True)
        if fps_text:
            cl.text(fps_text, 10, 30, bgr(0, 0, 255), scale=2)                                       # :331
        return cl

    @staticmethod
    def _box(cl, xyxy, label):
        x0, y0, x1, y1 = (int(v) for v in xyxy)
        cl.rect(x0, y0, x1, y1, bgr(0, 0, 225), 2)                                                   # :80 / :132
        cl.fill(x0, y0, x0 + ADVANCE * len(label) + 2, y0 - 12, bgr(30, 30, 30))                     # :92 label plate
        cl.text(label, x0, y0 - 3, bgr(255, 255, 255))                                               # :95

    # ---- device -------------------------------------------------------------------------------------------
    def draw_device(self, frames: torch.Tensor, command_lists, stream=None) -> torch.Tensor:
        """frames: uint8 [B,H,W,3] (or [H,W,3]) on the device, annotated in place with one CommandList per frame: ONE launch for the
        batch, mask fills included (POLY primitives).  The command upload and the launch are enqueued on `stream` (default: the
        current torch stream)."""
        e = self.eng
        fr = frames if frames.dim() == 4 else frames.unsqueeze(0)
        if len(command_lists) != fr.shape[0]:
            raise ValueError("one command list per frame")
        st = torch.cuda.current_stream(fr.device) if stream is None else stream
        self._keep = []                                                 # command buffers of this call's launch, alive until the next call
        arr = [c.arrays() for c in command_lists]
        off = np.zeros(len(arr) + 1, np.int32)
        coff = 0
        prims = []
        for i, (p, ch) in enumerate(arr):
            p = p.copy()
            p[(p[:, 0] == TEXT) | (p[:, 0] == POLY), 6] += coff        # character / vertex offsets into the concatenated buffer
            prims.append(p)
            off[i + 1] = off[i] + len(p)
            coff += len(ch)
        prims = np.concatenate(prims) if prims else np.zeros((0, 8), np.int32)
        chars = np.concatenate([ch for _, ch in arr])
        dev = fr.device
        with torch.cuda.stream(st):                                     # the copies are ordered on the stream the kernel runs on
            d_prims = torch.from_numpy(prims if len(prims) else np.zeros((1, 8), np.int32)).to(dev)
            d_off, d_chars = torch.from_numpy(off).to(dev), torch.from_numpy(chars).to(dev)
        e._ck(e.L.ss_overlay(e.ctx, C.c_void_p(st.cuda_stream), C.c_void_p(fr.data_ptr()), fr.shape[0], fr.stride(0), fr.shape[1],
                             fr.shape[2], fr.stride(1), C.c_void_p(d_prims.data_ptr()), C.c_void_p(d_off.data_ptr()),
                             C.c_void_p(d_chars.data_ptr())))
        for t in (d_prims, d_off, d_chars):
            t.record_stream(st)                                         # the allocator must not hand the block out under the pending launch
        self._keep.append((d_prims, d_off, d_chars))
        return frames

    def draw_resident(self, dev_frame: torch.Tensor, results, counts: Optional[dict] = None, fps_text: str = "") -> np.ndarray:
        """A frame that is ALREADY on the device (uint8 [H,W,3], e.g. `Results.orig_img_device` of track_stream) -> annotated
        host frame: ss_overlay in place, one download, no upload of the frame."""
        self.draw_device(dev_frame, [self.commands(results, counts, fps_text)])
        out = np.empty(tuple(dev_frame.shape), np.uint8)
        self.eng.download(out, dev_frame)
        return out

    def draw(self, frame: np.ndarray, results, counts: Optional[dict] = None, fps_text: str = "") -> np.ndarray:
        """Host frame in, annotated host frame out (upload -> ss_overlay -> download)."""
        e = self.eng
        if self._scratch is None or self._scratch.shape != frame.shape:
            self._scratch = torch.empty(frame.shape, dtype=torch.uint8, device=e.device)
        e.upload(self._scratch, frame)
        self.draw_device(self._scratch, [self.commands(results, counts, fps_text)])
        out = np.empty_like(frame)
        e.download(out, self._scratch)
        return out
