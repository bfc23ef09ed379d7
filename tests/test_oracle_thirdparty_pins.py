"""Third-party pins of the oracle (VERDICT r2 'missing' 1: the reference holds no vector, so the only lever is checking the
oracle against INDEPENDENT implementations that ship in this image, the way LSAP is pinned to scipy.optimize):
bilinear resampling -> torch.nn.functional.interpolate, squared Mahalanobis gating -> scipy.spatial.distance.cdist,
cosine distance -> scipy cdist / scikit-learn, the chi-square gate -> scipy.stats, the Kalman update -> a textbook solve
with scipy.linalg.  None of these libraries is used by the oracle's C back end (oracle/csrc/ss_oracle.c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cexact
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.engine import letterbox_geometry


def _torch_bilinear_u8(img, oh, ow):
    """torch's bilinear (half-pixel centres, no antialias) on float64, rounded half up to the integer grey level."""
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    r = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    return np.floor(r + 0.5)


@pytest.mark.parametrize("hw", [(72, 128), (90, 77), (48, 64), (200, 120)])
def test_letterbox_resampling_is_torch_interpolate(hw):
    H, W = hw
    img = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W, imgsz=64, stride=32)
    got = cexact.letterbox(img, g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
    inner = got[:, g.pad_top:g.pad_top + g.new_h, g.pad_left:g.pad_left + g.new_w]
    ref = (_torch_bilinear_u8(img, g.new_h, g.new_w)[:, :, ::-1] / 255.0).transpose(2, 0, 1)      # BGR -> RGB, CHW, [0, 1]
    d = np.abs(inner - ref)
    assert d.max() <= 1 / 255 + 1e-6             # one grey level where float32 / float64 interpolation straddles .5
    assert (d > 1e-6).mean() < 0.01
    border = np.ones_like(got, dtype=bool)
    border[:, g.pad_top:g.pad_top + g.new_h, g.pad_left:g.pad_left + g.new_w] = False
    assert np.all(got[border] == np.float32(114 / 255))


def test_crop_resampling_is_torch_interpolate():
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    dets = np.array([[10.3, 5.8, 70.2, 110.1, .9, 0], [-4, -4, 30, 60, .8, 0], [100, 40, 220, 190, .7, 0]], np.float32)
    got = cexact.crop_norm(img, dets, 64, 32)
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    for i, b in enumerate(dets):
        x1, y1 = max(int(b[0]), 0), max(int(b[1]), 0)
        x2, y2 = min(int(b[2]), 159), min(int(b[3]), 119)
        q = _torch_bilinear_u8(img[y1:y2, x1:x2], 64, 32)[:, :, ::-1] / 255.0
        ref = ((q - mean) / std).transpose(2, 0, 1)
        d = np.abs(got[i] - ref)
        assert d.max() <= (1 / 255) / std.min() + 1e-5 and (d > 1e-5).mean() < 0.01


def test_gate_threshold_is_the_chi_square_quantile():
    from scipy.stats import chi2
    assert abs(StrongSortConfig().gating_threshold - chi2.ppf(0.95, df=4)) < 5e-5


def _state(rng, cfg, steps):
    z = np.array([rng.uniform(50, 1800), rng.uniform(50, 1000), rng.uniform(0.2, 0.8), rng.uniform(40, 300)])
    m, c = cexact.kf_initiate(z, cfg.std_weight_position, cfg.std_weight_velocity)
    for _ in range(steps):
        m, c = cexact.kf_predict(m, c, cfg.std_weight_position, cfg.std_weight_velocity)
        m, c = cexact.kf_update(m, c, z + rng.normal(0, 1, 4) * [2, 2, 0.01, 2], rng.uniform(0.3, 0.95), cfg.std_weight_position)
        z = z + [3, 1, 0, 0]
    return m, c


@pytest.mark.parametrize("seed", range(5))
def test_gating_distance_is_scipy_mahalanobis(seed):
    from scipy.spatial.distance import cdist
    cfg, rng = StrongSortConfig(), np.random.default_rng(seed)
    m, c = _state(rng, cfg, seed)
    Z = m[:4] + rng.normal(0, 5, (40, 4)) * [1, 1, 0.01, 1]
    got = cexact.gating(m, c, Z, cfg.std_weight_position)
    zm, S = cexact.kf_project(m, c, 0.0, cfg.std_weight_position)
    ref = cdist(Z, zm[None], "mahalanobis", VI=np.linalg.inv(S))[:, 0] ** 2
    assert np.allclose(got, ref, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("seed", range(4))
def test_kalman_update_is_the_textbook_solve(seed):
    from scipy.linalg import solve
    cfg, rng = StrongSortConfig(), np.random.default_rng(100 + seed)
    m, P = _state(rng, cfg, 2 + seed)
    z, conf = m[:4] + rng.normal(0, 3, 4) * [1, 1, 0.01, 1], rng.uniform(0.3, 0.95)
    m2, P2 = cexact.kf_update(m, P, z, conf, cfg.std_weight_position)
    zm, S = cexact.kf_project(m, P, conf, cfg.std_weight_position)          # NSA noise (1 - conf)^2 R inside
    H = np.eye(4, 8)
    K = solve(S, H @ P, assume_a="pos").T                                   # P H^T S^-1
    assert np.allclose(m2, m + K @ (z - zm), rtol=1e-9, atol=1e-9)
    assert np.allclose(P2, P - K @ S @ K.T, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("B,D", [(100, 30), (7, 64), (1, 1)])
def test_cosine_min_is_scipy_and_sklearn_cosine_distance(B, D):
    from scipy.spatial.distance import cdist
    from sklearn.metrics.pairwise import cosine_distances
    rng = np.random.default_rng(B + D)
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    gal, feats = unit(rng.standard_normal((B, 512))), unit(rng.standard_normal((D, 512)))
    got = cexact.cosine_min(gal, feats)
    for ref in (cdist(gal.astype(np.float64), feats.astype(np.float64), "cosine").min(0),
                cosine_distances(gal.astype(np.float64), feats.astype(np.float64)).min(0)):
        assert np.allclose(got, ref, atol=2e-6)                             # float32 dot products against float64 ones
