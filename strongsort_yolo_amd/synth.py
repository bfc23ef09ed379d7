"""Seeded synthetic video streams for the tracking hot path.

There is no video decoder, no weights and no dataset in this environment, so every measurement
and parity test is driven from here: frames are seeded uint8 noise, detections are persistent
ground-truth identities moving with constant velocity + jitter, and (when features are injected
rather than produced by the ReID net) each identity owns a unit-norm 512-d prototype.

Deviation from SURVEY §8(d), recorded in oracle/DECISIONS.md D-12: feature noise sigma defaults
to 0.02, not 0.1.  With sigma = 0.1 the noise vector has norm 0.1*sqrt(512) = 2.26 >> 1, the
cosine distance between two sightings of one identity is ~0.84 > max_dist (0.2) and the
appearance stage would never fire; sigma = 0.02 gives same-identity distances ~0.09-0.17,
i.e. both sides of the 0.2 gate are exercised.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .config import FEAT_DIM


@dataclass
class SynthConfig:
    width: int = 1280
    height: int = 720
    n_ids: int = 30
    n_classes: int = 1
    feat_sigma: float = 0.02
    p_vanish: float = 0.01          # per identity per frame
    vanish_min: int = 1
    vanish_max: int = 10
    jitter_px: float = 1.0
    frame_pool: int = 4             # distinct noise frames cycled per stream
    seed: int = 0


@dataclass
class SynthFrame:
    index: int
    dets: np.ndarray                # [n,6] f32 x1,y1,x2,y2,conf,cls  (descending conf)
    feats: np.ndarray               # [n,512] f32 unit rows
    gt_ids: np.ndarray              # [n] int32 ground-truth identity of each row


class SynthStream:
    """Deterministic stream `seed`; iterate with next_frame()."""

    def __init__(self, cfg: Optional[SynthConfig] = None, **kw):
        cfg = cfg or SynthConfig(**kw)
        self.cfg = cfg
        rng = np.random.default_rng(cfg.seed)
        self.rng = rng
        n = cfg.n_ids
        self.w = rng.uniform(40, 120, n)
        self.h = rng.uniform(80, 240, n)
        self.cx = rng.uniform(self.w / 2, cfg.width - self.w / 2)
        self.cy = rng.uniform(self.h / 2, cfg.height - self.h / 2)
        self.vx = rng.uniform(-3, 3, n)
        self.vy = rng.uniform(-2, 2, n)
        self.cls = rng.integers(0, cfg.n_classes, n).astype(np.float32)
        proto = rng.standard_normal((n, FEAT_DIM))
        self.proto = (proto / np.linalg.norm(proto, axis=1, keepdims=True)).astype(np.float32)
        self.hidden_until = np.zeros(n, dtype=np.int64)   # frame index until which id is absent
        self.t = 0
        self._frames = None

    # -- pixel data ---------------------------------------------------------------------------
    def frame_pixels(self, index: int) -> np.ndarray:
        """uint8 BGR [H,W,3]; a small seeded pool cycled (content is irrelevant to association)."""
        if self._frames is None:
            prng = np.random.default_rng(self.cfg.seed + 7919)
            self._frames = prng.integers(
                0, 256, (self.cfg.frame_pool, self.cfg.height, self.cfg.width, 3), dtype=np.uint8)
        return self._frames[index % self.cfg.frame_pool]

    def render(self, frame: "SynthFrame") -> np.ndarray:
        """uint8 BGR [H,W,3] with every visible identity DRAWN into its box: an identity-specific seeded 8 x 4 colour field,
        bilinearly interpolated over the box, on a flat background; identities are painted in ascending id order, so an
        overlap looks the same in every frame.  For the legs that put the real ReID network in the loop (SURVEY §8d:
        "rendered boxes when the real ReID net is in the loop"): crops of one identity look alike from frame to frame (box
        jitter and occlusion aside), crops of different identities do not."""
        cfg = self.cfg
        img = np.full((cfg.height, cfg.width, 3), 96, np.uint8)
        if not hasattr(self, "_tex"):
            self._tex = np.random.default_rng(cfg.seed + 15485863).integers(0, 256, (cfg.n_ids, 9, 5, 3)).astype(np.float32)
        for i in np.argsort(frame.gt_ids, kind="stable"):
            x1, y1, x2, y2 = (int(v) for v in frame.dets[i, :4])
            w, h = max(x2 - x1, 1), max(y2 - y1, 1)
            t = self._tex[int(frame.gt_ids[i])]
            fy, fx = (np.arange(h) + 0.5) / h * 8, (np.arange(w) + 0.5) / w * 4
            y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
            ay, ax = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
            p = (t[y0][:, x0] * (1 - ay) * (1 - ax) + t[y0 + 1][:, x0] * ay * (1 - ax) + t[y0][:, x0 + 1] * (1 - ay) * ax
                 + t[y0 + 1][:, x0 + 1] * ay * ax)
            img[y1:y1 + h, x1:x1 + w] = p.astype(np.uint8)[:img.shape[0] - y1, :img.shape[1] - x1]
        return img

    # -- detections -----------------------------------------------------------------------------
    def next_frame(self) -> SynthFrame:
        cfg, rng = self.cfg, self.rng
        n = cfg.n_ids
        # motion (reflect at borders so boxes stay inside the frame)
        self.cx += self.vx
        self.cy += self.vy
        for pos, vel, half, lim in ((self.cx, self.vx, self.w / 2, cfg.width),
                                    (self.cy, self.vy, self.h / 2, cfg.height)):
            lo = pos < half
            hi = pos > lim - half
            pos[lo] = 2 * half[lo] - pos[lo]
            pos[hi] = 2 * (lim - half[hi]) - pos[hi]
            vel[lo | hi] *= -1
        # occlusions
        start = (rng.random(n) < cfg.p_vanish) & (self.hidden_until <= self.t)
        dur = rng.integers(cfg.vanish_min, cfg.vanish_max + 1, n)
        self.hidden_until = np.where(start, self.t + dur, self.hidden_until)
        visible = self.hidden_until <= self.t
        ids = np.nonzero(visible)[0]
        jit = rng.normal(0, cfg.jitter_px, (n, 4))
        conf = rng.uniform(0.35, 0.95, n)
        noise = rng.standard_normal((n, FEAT_DIM)) * cfg.feat_sigma
        x1 = self.cx - self.w / 2 + jit[:, 0]
        y1 = self.cy - self.h / 2 + jit[:, 1]
        x2 = self.cx + self.w / 2 + jit[:, 2]
        y2 = self.cy + self.h / 2 + jit[:, 3]
        x1 = np.clip(x1, 0, cfg.width - 2)
        y1 = np.clip(y1, 0, cfg.height - 2)
        x2 = np.clip(x2, x1 + 1, cfg.width - 1)
        y2 = np.clip(y2, y1 + 1, cfg.height - 1)
        dets = np.stack([x1, y1, x2, y2, conf, self.cls], axis=1).astype(np.float32)
        f = self.proto.astype(np.float64) + noise
        f = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
        order = ids[np.argsort(-conf[ids], kind="stable")]
        out = SynthFrame(self.t, dets[order], f[order], order.astype(np.int32))
        self.t += 1
        return out


def make_stream(seed: int = 0, width: int = 1280, height: int = 720, n_ids: int = 30, **kw) -> SynthStream:
    return SynthStream(SynthConfig(width=width, height=height, n_ids=n_ids, seed=seed, **kw))


def synth_prediction(dets: np.ndarray, n_anchors: int, n_classes: int, lb_scale: float,
                     lb_pad: tuple, rng: np.random.Generator, dup: int = 6,
                     clutter: int = 200) -> np.ndarray:
    """A detector-head-shaped tensor [4+nc, N] (YOLOv8 layout: xywh in letterboxed pixels + class
    scores) whose NMS result is exactly `dets` (in descending-score order): every true box gets
    `dup` overlapping lower-scored duplicates, plus sub-threshold clutter elsewhere.
    lb_scale / lb_pad map original pixels to letterboxed pixels.
    Returns (pred, anchor_gt) where anchor_gt[a] = row of `dets` anchor a was drawn from, or -1."""
    n = dets.shape[0]
    anchor_gt = np.full(n_anchors, -1, dtype=np.int64)
    pred = np.zeros((4 + n_classes, n_anchors), dtype=np.float32)
    # background: tiny scores everywhere
    pred[4:, :] = rng.uniform(0.0, 0.05, (n_classes, n_anchors)).astype(np.float32)
    pred[0, :] = rng.uniform(0, 640, n_anchors)
    pred[1, :] = rng.uniform(0, 384, n_anchors)
    pred[2, :] = rng.uniform(8, 64, n_anchors)
    pred[3, :] = rng.uniform(8, 64, n_anchors)
    slots = rng.permutation(n_anchors)[: n * (dup + 1) + clutter]
    k = 0
    px, py = lb_pad
    for i in range(n):
        x1, y1, x2, y2, conf, cls = dets[i]
        cx = (x1 + x2) / 2 * lb_scale + px
        cy = (y1 + y2) / 2 * lb_scale + py
        w = (x2 - x1) * lb_scale
        h = (y2 - y1) * lb_scale
        for d in range(dup + 1):
            a = slots[k]; k += 1
            anchor_gt[a] = i
            if d == 0:
                pred[0:4, a] = (cx, cy, w, h)
                pred[4:, a] = 0.01
                pred[4 + int(cls), a] = conf
            else:
                j = rng.normal(0, 0.03, 4)
                pred[0:4, a] = (cx + j[0] * w, cy + j[1] * h, w * (1 + j[2]), h * (1 + j[3]))
                pred[4:, a] = 0.01
                pred[4 + int(cls), a] = max(0.31, conf - rng.uniform(0.02, 0.3))
    for _ in range(clutter):
        a = slots[k]; k += 1
        pred[4 + int(rng.integers(0, n_classes)), a] = rng.uniform(0.1, 0.29)
    return pred, anchor_gt
