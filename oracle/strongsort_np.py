"""CPU restatement of the StrongSORT tracker update (the oracle).

TEST INFRASTRUCTURE ONLY — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product package (strongsort_yolo_amd/) never does.

PARITY UNPINNED.  The reference snapshot holds no tracker source: its hot path is
the opaque third-party call /root/reference/yolo_multi_model.py:41
(`model.track(image, ..., persist=True, tracker="botsort.yaml")`) into the unpinned `ultralytics`
pip package, and the yolov5/ yolov7/ directories that once vendored StrongSORT are empty.  This
file restates the *published* StrongSORT algorithm that BASELINE.json's north_star names
(OSNet features, cosine gallery distance, Mahalanobis gate, lambda blend, Hungarian assignment,
IoU fallback, NSA Kalman, EMA features); each frozen choice is listed in oracle/DECISIONS.md.

Two numeric back ends share one lifecycle implementation:
  * "numpy": BLAS / SciPy calls, the way a NumPy implementation is normally written.  Used as the
    CPU baseline (bench.py cpu_baseline, kind "port") and as an independent cross-check.
  * "c":     oracle/csrc/ss_oracle.c — identical algorithm with a *defined* operation order, which
    the gfx950 kernels reproduce bit for bit.  This is the parity target of the -m gpu tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

TENTATIVE, CONFIRMED, DELETED = 1, 2, 3


# ==================================================================================================
# numeric back ends
# ==================================================================================================
class NumpyNumerics:
    """BLAS/SciPy arithmetic (summation order = whatever the libraries do)."""
    name = "numpy"

    def __init__(self, cfg):
        import scipy.linalg
        import scipy.optimize
        self._la = scipy.linalg
        self._opt = scipy.optimize
        self.cfg = cfg
        self.wp, self.wv = cfg.std_weight_position, cfg.std_weight_velocity
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)

    # appearance
    def normalize(self, v):
        v = np.asarray(v, dtype=np.float32)
        n = np.linalg.norm(v)
        return (v / n).astype(np.float32) if n > 0 else np.zeros_like(v)

    def cosine_min(self, gallery, feats):
        d = np.float32(1.0) - gallery @ feats.T          # [B, D] sgemm
        return d.min(axis=0)

    def ema(self, smooth, feat, alpha):
        s = np.float32(alpha) * smooth + np.float32(1.0 - alpha) * feat
        return (s / np.linalg.norm(s)).astype(np.float32)

    # kalman
    def kf_initiate(self, z):
        h = z[3]
        sd = np.array([2 * self.wp * h, 2 * self.wp * h, 1e-2, 2 * self.wp * h,
                       10 * self.wv * h, 10 * self.wv * h, 1e-5, 10 * self.wv * h])
        return np.r_[z, np.zeros(4)], np.diag(np.square(sd))

    def kf_predict(self, mean, cov):
        h = mean[3]
        sd = np.array([self.wp * h, self.wp * h, 1e-2, self.wp * h,
                       self.wv * h, self.wv * h, 1e-5, self.wv * h])
        return self.F @ mean, self.F @ (cov @ self.F.T) + np.diag(np.square(sd))

    def kf_project(self, mean, cov, conf):
        h = mean[3]
        sd = (1.0 - conf) * np.array([self.wp * h, self.wp * h, 1e-1, self.wp * h])
        return self.H @ mean, self.H @ cov @ self.H.T + np.diag(np.square(sd))

    def gating(self, mean, cov, Z):
        m4, S = self.kf_project(mean, cov, 0.0)
        L = np.linalg.cholesky(S)
        y = self._la.solve_triangular(L, (Z - m4).T, lower=True, check_finite=False)
        return np.sum(y * y, axis=0)

    def kf_update(self, mean, cov, z, conf):
        m4, S = self.kf_project(mean, cov, conf)
        cf = self._la.cho_factor(S, lower=True, check_finite=False)
        K = self._la.cho_solve(cf, (cov @ self.H.T).T, check_finite=False).T
        return mean + (z - m4) @ K.T, cov - K @ (S @ K.T)

    # cost / assignment
    def blend(self, cosd, maha):
        c = self.cfg
        gated = maha > c.gating_threshold
        v = np.where(gated, c.gated_cost, cosd.astype(np.float64))
        v = c.mc_lambda * v + (1.0 - c.mc_lambda) * maha
        v[v > c.max_dist] = c.max_dist + 1e-5
        return v, gated.astype(np.uint8)

    def iou_cost(self, tlwh, det_tlwh):
        c = self.cfg
        tl = np.maximum(tlwh[:2], det_tlwh[:, :2])
        br = np.minimum(tlwh[:2] + tlwh[2:], det_tlwh[:, :2] + det_tlwh[:, 2:])
        wh = np.maximum(0.0, br - tl)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (tlwh[2] * tlwh[3] + det_tlwh[:, 2] * det_tlwh[:, 3] - inter)
        v = 1.0 - iou
        v[v > c.max_iou_distance] = c.max_iou_distance + 1e-5
        return v

    def lsap(self, cost):
        return self._opt.linear_sum_assignment(cost)

    def camera_update(self, mean, warp):
        w, h = mean[2] * mean[3], mean[3]
        p1 = np.array([mean[0] - w / 2, mean[1] - h / 2, 1.0])
        p2 = p1 + np.array([w, h, 0.0])
        a, b = warp @ p1, warp @ p2
        nw, nh = b[0] - a[0], b[1] - a[1]
        out = mean.copy()
        out[:4] = [a[0] + nw / 2, a[1] + nh / 2, nw / nh, nh]
        return out


class CNumerics:
    """Defined-order arithmetic from oracle/csrc/ss_oracle.c."""
    name = "c"

    def __init__(self, cfg):
        from . import cexact
        self.x = cexact
        self.cfg = cfg
        self.wp, self.wv = cfg.std_weight_position, cfg.std_weight_velocity

    def normalize(self, v): return self.x.normalize(v)
    def cosine_min(self, gallery, feats): return self.x.cosine_min(gallery, feats)
    def ema(self, smooth, feat, alpha): return self.x.ema(smooth, feat, alpha)
    def kf_initiate(self, z): return self.x.kf_initiate(z, self.wp, self.wv)
    def kf_predict(self, mean, cov): return self.x.kf_predict(mean, cov, self.wp, self.wv)
    def kf_project(self, mean, cov, conf): return self.x.kf_project(mean, cov, conf, self.wp)
    def gating(self, mean, cov, Z): return self.x.gating(mean, cov, Z, self.wp)
    def kf_update(self, mean, cov, z, conf): return self.x.kf_update(mean, cov, z, conf, self.wp)

    def blend(self, cosd, maha):
        c = self.cfg
        return self.x.blend(cosd, maha, c.mc_lambda, c.gating_threshold, c.gated_cost, c.max_dist)

    def iou_cost(self, tlwh, det_tlwh):
        return self.x.iou_cost(tlwh, det_tlwh, self.cfg.max_iou_distance)

    def lsap(self, cost): return self.x.lsap(cost)
    def camera_update(self, mean, warp): return self.x.camera_update(mean, warp)


# ==================================================================================================
# tracker lifecycle
# ==================================================================================================
@dataclass
class Track:
    track_id: int
    mean: np.ndarray
    cov: np.ndarray
    smooth: np.ndarray                  # current EMA feature (unit norm)
    class_id: int
    conf: float
    hits: int = 1
    age: int = 1
    tsu: int = 0                        # time since update
    state: int = TENTATIVE
    gallery: List[np.ndarray] = field(default_factory=list)
    det_idx: int = -1                   # detection matched in the current frame


def xyxy_to_tlwh64(dets):
    """f32 xyxy -> float64 tlwh (differences of f32 values are exact in f64)."""
    d = np.asarray(dets[:, :4], dtype=np.float64)
    return np.stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]], axis=1)


def tlwh_to_xyah(t):
    return np.stack([t[:, 0] + t[:, 2] / 2, t[:, 1] + t[:, 3] / 2, t[:, 2] / t[:, 3], t[:, 3]], axis=1)


class OracleStrongSort:
    """One stream's tracker.  update() consumes detections + their (raw) ReID features."""

    def __init__(self, cfg, numerics: str = "c"):
        self.cfg = cfg
        self.nx = CNumerics(cfg) if numerics == "c" else NumpyNumerics(cfg)
        self.tracks: List[Track] = []
        self.next_id = 1
        self.frame = 0
        self.last = {}                  # stage intermediates of the latest update (for KATs)

    # -- helpers --------------------------------------------------------------------------------
    @staticmethod
    def _track_tlwh(mean):
        w = mean[2] * mean[3]
        return np.array([mean[0] - w / 2, mean[1] - mean[3] / 2, w, mean[3]])

    def _match(self, cost, thr):
        """LSAP + threshold: returns (pairs, unmatched_rows, unmatched_cols) over row/col indices."""
        T, D = cost.shape
        if T == 0 or D == 0:
            return [], list(range(T)), list(range(D))
        rows, cols = self.nx.lsap(cost)
        pairs, urow, ucol = [], set(range(T)), set(range(D))
        for r, c in zip(rows, cols):
            if cost[r, c] > thr:
                continue
            pairs.append((int(r), int(c)))
            urow.discard(int(r)); ucol.discard(int(c))
        return pairs, sorted(urow), sorted(ucol)

    # -- the per-frame update ---------------------------------------------------------------------
    def update(self, dets: np.ndarray, feats: np.ndarray, img_hw, warp: Optional[np.ndarray] = None) -> np.ndarray:
        """warp (N4, optional): 2x3 camera-motion matrix previous frame -> this frame in full-frame pixels; applied to
        every track's box before the prediction (upstream tracker.camera_update ahead of tracker.predict)."""
        cfg, nx = self.cfg, self.nx
        dets = np.asarray(dets, dtype=np.float32).reshape(-1, 6)
        D = dets.shape[0]
        H, W = int(img_hw[0]), int(img_hw[1])
        f = np.stack([nx.normalize(feats[d]) for d in range(D)]) if D else np.zeros((0, feats.shape[-1] if feats.ndim == 2 else 512), np.float32)
        tlwh = xyxy_to_tlwh64(dets) if D else np.zeros((0, 4))
        xyah = tlwh_to_xyah(tlwh) if D else np.zeros((0, 4))

        # 0. camera-motion compensation, 1. predict every live track
        for t in self.tracks:
            if warp is not None:
                t.mean = nx.camera_update(t.mean, np.asarray(warp, dtype=np.float64).reshape(2, 3))
            t.mean, t.cov = nx.kf_predict(t.mean, t.cov)
            t.age += 1
            t.tsu += 1
            t.det_idx = -1

        confirmed = [i for i, t in enumerate(self.tracks) if t.state == CONFIRMED]
        unconfirmed = [i for i, t in enumerate(self.tracks) if t.state != CONFIRMED]

        # 2. appearance + motion association of confirmed tracks (vanilla global matching, D-02)
        cost_a = np.zeros((len(confirmed), D))
        cos_a = np.zeros((len(confirmed), D), dtype=np.float32)
        maha_a = np.zeros((len(confirmed), D))
        gate_a = np.zeros((len(confirmed), D), dtype=np.uint8)
        if D:
            for r, ti in enumerate(confirmed):
                t = self.tracks[ti]
                cos_a[r] = nx.cosine_min(np.stack(t.gallery), f)
                maha_a[r] = nx.gating(t.mean, t.cov, xyah)
                cost_a[r], gate_a[r] = nx.blend(cos_a[r], maha_a[r])
        pairs_a, urow_a, ucol_a = self._match(cost_a, cfg.max_dist)
        matches = [(confirmed[r], c) for r, c in pairs_a]
        unmatched_conf = [confirmed[r] for r in urow_a]

        # 3. IoU association: unconfirmed tracks, then confirmed tracks missed for exactly 1 frame
        cand = unconfirmed + [k for k in unmatched_conf if self.tracks[k].tsu == 1]
        unmatched_tracks = [k for k in unmatched_conf if self.tracks[k].tsu != 1]
        cost_b = np.zeros((len(cand), len(ucol_a)))
        if len(ucol_a):
            for r, ti in enumerate(cand):
                t = self.tracks[ti]
                if t.tsu > 1:
                    cost_b[r] = cfg.max_iou_distance + 1e-5
                else:
                    cost_b[r] = nx.iou_cost(self._track_tlwh(t.mean), tlwh[ucol_a])
        pairs_b, urow_b, ucol_b = self._match(cost_b, cfg.max_iou_distance)
        matches += [(cand[r], ucol_a[c]) for r, c in pairs_b]
        unmatched_tracks += [cand[r] for r in urow_b]
        unmatched_dets = [ucol_a[c] for c in ucol_b]

        self.last = dict(feats=f, xyah=xyah, tlwh=tlwh, confirmed=confirmed, cos=cos_a, maha=maha_a,
                         gated=gate_a, cost_a=cost_a, pairs_a=pairs_a, cand=cand, cols_b=list(ucol_a),
                         cost_b=cost_b, pairs_b=pairs_b)

        # 4. matched tracks: NSA Kalman update + EMA feature
        for ti, d in matches:
            t = self.tracks[ti]
            t.mean, t.cov = nx.kf_update(t.mean, t.cov, xyah[d], float(dets[d, 4]))
            t.smooth = nx.ema(t.smooth, f[d], cfg.ema_alpha)
            t.conf = float(dets[d, 4])
            t.class_id = int(dets[d, 5])
            t.hits += 1
            t.tsu = 0
            t.det_idx = d
            if t.state == TENTATIVE and t.hits >= cfg.n_init:
                t.state = CONFIRMED
        # 5. missed tracks
        for ti in unmatched_tracks:
            t = self.tracks[ti]
            if t.state == TENTATIVE or t.tsu > cfg.max_age:
                t.state = DELETED
        # 6. births, ascending detection index (D-11)
        for d in sorted(unmatched_dets):
            mean, cov = nx.kf_initiate(xyah[d])
            self.tracks.append(Track(self.next_id, mean, cov, f[d].copy(), int(dets[d, 5]),
                                     float(dets[d, 4]), det_idx=d))
            self.next_id += 1
        # 7. drop deleted (stable), 8. gallery append for every confirmed track (D-05)
        self.tracks = [t for t in self.tracks if t.state != DELETED]
        for t in self.tracks:
            if t.state == CONFIRMED:
                t.gallery.append(t.smooth)
                if len(t.gallery) > cfg.nn_budget:
                    t.gallery = t.gallery[-cfg.nn_budget:]
        # 9. output rows (confirmed, seen within the last frame)
        rows = []
        for t in self.tracks:
            if t.state != CONFIRMED or t.tsu > 1:
                continue
            x, y, w, h = self._track_tlwh(t.mean)
            x1, y1 = max(int(x), 0), max(int(y), 0)
            x2, y2 = min(int(x + w), W - 1), min(int(y + h), H - 1)
            rows.append([x1, y1, x2, y2, t.track_id, t.class_id, t.conf, t.det_idx])
        self.frame += 1
        return np.asarray(rows, dtype=np.float32).reshape(-1, 8)

    # -- state snapshot in the device table's terms (for table-level parity checks) ----------------
    def snapshot(self):
        return dict(
            track_id=np.array([t.track_id for t in self.tracks], dtype=np.int32),
            state=np.array([t.state for t in self.tracks], dtype=np.int32),
            hits=np.array([t.hits for t in self.tracks], dtype=np.int32),
            age=np.array([t.age for t in self.tracks], dtype=np.int32),
            tsu=np.array([t.tsu for t in self.tracks], dtype=np.int32),
            mean=np.array([t.mean for t in self.tracks]).reshape(-1, 8),
            cov=np.array([t.cov for t in self.tracks]).reshape(-1, 8, 8),
            smooth=np.array([t.smooth for t in self.tracks], dtype=np.float32).reshape(-1, 512),
            gal_count=np.array([len(t.gallery) for t in self.tracks], dtype=np.int32),
            next_id=self.next_id,
        )
