"""Exact-order C oracle vs the independent NumPy/SciPy back end (tolerances stated per stage)."""
import numpy as np
import pytest

from oracle import cexact
from oracle.strongsort_np import CNumerics, NumpyNumerics


def _unit(rng, n, F=512):
    x = rng.standard_normal((n, F)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_dot_definition():
    rng = np.random.default_rng(0)
    g, f = _unit(rng, 1)[0], _unit(rng, 1)[0]
    segs = []
    for s in range(8):
        p = np.float32(0)
        for k in range(64 * s, 64 * s + 64):   # fmaf == exact product in f64, one rounding
            p = np.float32(np.float64(g[k]) * np.float64(f[k]) + np.float64(p))
        segs.append(p)
    tot = segs[0]
    for p in segs[1:]:
        tot = np.float32(tot + p)
    # double rounding can differ from a true fma in rare cases; allow 1 ulp here, bit-exactness of
    # fmaf itself is covered by the GPU-vs-C tests.
    assert abs(float(cexact.dot(g, f)) - float(tot)) <= 2e-7
    assert abs(float(cexact.dot(g, f)) - float(np.dot(g.astype(np.float64), f.astype(np.float64)))) < 1e-6


def test_cosine_min_vs_numpy(cfg):
    rng = np.random.default_rng(1)
    gal, f = _unit(rng, 100), _unit(rng, 30)
    a = CNumerics(cfg).cosine_min(gal, f)
    b = NumpyNumerics(cfg).cosine_min(gal, f)
    assert a.dtype == np.float32
    assert np.abs(a - b).max() <= 1e-5          # north_star float tolerance is 1e-4


def test_normalize_and_ema(cfg):
    rng = np.random.default_rng(2)
    v = rng.standard_normal(512).astype(np.float32)
    a, b = CNumerics(cfg), NumpyNumerics(cfg)
    assert np.abs(a.normalize(v) - b.normalize(v)).max() <= 1e-6
    s, f = a.normalize(v), a.normalize(rng.standard_normal(512).astype(np.float32))
    assert np.abs(a.ema(s, f, 0.9) - b.ema(s, f, 0.9)).max() <= 1e-6
    assert abs(np.linalg.norm(a.ema(s, f, 0.9)) - 1) < 1e-6


def _track(rng, nx):
    z = np.array([rng.uniform(100, 1000), rng.uniform(100, 600), rng.uniform(0.3, 0.7), rng.uniform(80, 240)])
    mean, cov = nx.kf_initiate(z)
    for _ in range(5):
        mean, cov = nx.kf_predict(mean, cov)
        mean, cov = nx.kf_update(mean, cov, z + rng.normal(0, 1, 4) * [2, 2, 0.01, 2], rng.uniform(0.3, 0.9))
        z = z + [2, 1, 0, 0]
    return mean, cov


def test_kalman_vs_numpy(cfg):
    rng = np.random.default_rng(3)
    a, b = CNumerics(cfg), NumpyNumerics(cfg)
    for _ in range(20):
        st = rng.bit_generator.state
        ma, ca = _track(rng, a)
        rng.bit_generator.state = st
        mb, cb = _track(rng, b)
        assert np.allclose(ma, mb, rtol=1e-10, atol=1e-10)
        assert np.allclose(ca, cb, rtol=1e-9, atol=1e-12)
        assert np.allclose(ca, ca.T, rtol=1e-9, atol=1e-12)          # symmetric
        assert np.linalg.eigvalsh((ca + ca.T) / 2).min() > 0          # PSD
        Z = ma[:4] + rng.normal(0, 3, (30, 4)) * [3, 3, 0.02, 3]
        ga, gb = a.gating(ma, ca, Z), b.gating(mb, cb, Z)
        assert np.allclose(ga, gb, rtol=1e-9)


def test_blend_and_iou_vs_numpy(cfg):
    rng = np.random.default_rng(4)
    a, b = CNumerics(cfg), NumpyNumerics(cfg)
    cosd = rng.uniform(0, 1, 64).astype(np.float32)
    maha = rng.uniform(0, 20, 64)
    ca, ga = a.blend(cosd, maha)
    cb, gb = b.blend(cosd, maha)
    assert np.array_equal(ga, gb) and np.array_equal(ca, cb)     # same f64 ops -> identical
    t = np.array([100.0, 50.0, 60.0, 120.0])
    d = np.c_[rng.uniform(50, 150, 40), rng.uniform(0, 100, 40), rng.uniform(30, 90, 40), rng.uniform(60, 180, 40)]
    assert np.array_equal(a.iou_cost(t, d), b.iou_cost(t, d))


def test_gating_monotone(cfg):
    a = CNumerics(cfg)
    rng = np.random.default_rng(5)
    m, c = _track(rng, a)
    dirs = np.array([1.0, 0.5, 0.0, 0.2])
    vals = a.gating(m, c, m[:4] + np.outer(np.linspace(0, 50, 20), dirs))
    assert np.all(np.diff(vals) > 0) and vals[0] == 0
