"""Stage known-answer tests through the C ABI: gfx950 kernels vs the exact-order C oracle.
Bar: bit-exact (stronger than north_star's 1e-4 on floats; ids/indices must be exact anyway)."""
import numpy as np
import pytest

from oracle import cexact
from tests.gpu_util import engine, bits_equal, unit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine()
    yield e
    e.close()


def test_feat_normalize_bit_exact(eng):
    rng = np.random.default_rng(0)
    raw = (rng.standard_normal((37, 512)) * rng.uniform(0.1, 5)).astype(np.float32)
    got = eng.feat_normalize(raw).cpu().numpy()
    ref = np.stack([cexact.normalize(r) for r in raw])
    assert bits_equal(got, ref)


def test_ema_bit_exact(eng):
    rng = np.random.default_rng(1)
    s, f = unit(rng, 21), unit(rng, 21)
    got = eng.ema(s, f).cpu().numpy()
    ref = np.stack([cexact.ema(s[i], f[i], eng.cfg.ema_alpha) for i in range(21)])
    assert bits_equal(got, ref)


def _states(rng, n, cfg):
    wp, wv = cfg.std_weight_position, cfg.std_weight_velocity
    means, covs = [], []
    for _ in range(n):
        z = np.array([rng.uniform(50, 1800), rng.uniform(50, 1000), rng.uniform(0.2, 0.8), rng.uniform(40, 300)])
        m, c = cexact.kf_initiate(z, wp, wv)
        for _ in range(int(rng.integers(0, 6))):
            m, c = cexact.kf_predict(m, c, wp, wv)
            m, c = cexact.kf_update(m, c, z + rng.normal(0, 1, 4) * [2, 2, 0.01, 2], rng.uniform(0.3, 0.95), wp)
            z = z + [3, 1, 0, 0]
        means.append(m); covs.append(c)
    return np.array(means), np.array(covs)


def test_kalman_bit_exact(eng):
    cfg = eng.cfg
    wp, wv = cfg.std_weight_position, cfg.std_weight_velocity
    rng = np.random.default_rng(2)
    z0 = np.c_[rng.uniform(50, 1800, 50), rng.uniform(50, 1000, 50), rng.uniform(0.2, 0.8, 50), rng.uniform(40, 300, 50)]
    m, c = eng.kf_initiate(z0)
    ref = [cexact.kf_initiate(z, wp, wv) for z in z0]
    assert bits_equal(m.cpu().numpy(), np.array([r[0] for r in ref]))
    assert bits_equal(c.cpu().numpy(), np.array([r[1] for r in ref]))
    means, covs = _states(rng, 70, cfg)
    m, c = eng.kf_predict(means, covs)
    ref = [cexact.kf_predict(means[i], covs[i], wp, wv) for i in range(70)]
    assert bits_equal(m.cpu().numpy(), np.array([r[0] for r in ref]))
    assert bits_equal(c.cpu().numpy(), np.array([r[1] for r in ref]))
    z = means[:, :4] + rng.normal(0, 2, (70, 4)) * [1, 1, 0.005, 1]
    conf = rng.uniform(0.3, 0.95, 70)
    m, c = eng.kf_update(means, covs, z, conf)
    ref = [cexact.kf_update(means[i], covs[i], z[i], conf[i], wp) for i in range(70)]
    assert bits_equal(m.cpu().numpy(), np.array([r[0] for r in ref]))
    assert bits_equal(c.cpu().numpy(), np.array([r[1] for r in ref]))
    # a7 on its own (ss_kf_project): projected mean + innovation covariance with the NSA noise, and with conf = 0
    for cf in (conf, None):
        zm, S = eng.kf_project(means, covs, cf)
        ref = [cexact.kf_project(means[i], covs[i], 0.0 if cf is None else cf[i], wp) for i in range(70)]
        assert bits_equal(zm.cpu().numpy(), np.array([r[0] for r in ref]))
        assert bits_equal(S.cpu().numpy().reshape(70, 16), np.array([np.asarray(r[1]).reshape(16) for r in ref]))


@pytest.mark.parametrize("T,D,B", [(30, 30, 100), (100, 100, 100), (5, 1, 1), (3, 33, 37), (1, 128, 128), (17, 64, 32)])
def test_assoc_cost_bit_exact(eng, T, D, B):
    """fused a7+a8 at the BASELINE sizes (30,30,100,512), (100,100,100,512) and ragged edge cases."""
    cfg = eng.cfg
    rng = np.random.default_rng(T * 1000 + D)
    protos = unit(rng, T)
    gal = np.empty((T, B, 512), np.float32)
    for t in range(T):
        g = protos[t] + 0.02 * rng.standard_normal((B, 512)).astype(np.float32)
        gal[t] = g / np.linalg.norm(g, axis=1, keepdims=True)
    counts = rng.integers(1, B + 1, T).astype(np.int32)
    counts[0] = B
    f = protos[rng.integers(0, T, D)] + 0.02 * rng.standard_normal((D, 512)).astype(np.float32)
    f = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
    means, covs = _states(rng, T, cfg)
    xyah = means[rng.integers(0, T, D), :4] + rng.normal(0, 4, (D, 4)) * [1, 1, 0.01, 1]
    frag = eng.gallery_pack(gal)
    cost, cosd, maha, gated = (x.cpu().numpy() for x in eng.assoc_cost(frag, counts, f, means, covs, xyah))
    wp = cfg.std_weight_position
    for t in range(T):
        rc = cexact.cosine_min(gal[t, :counts[t]], f)
        rm = cexact.gating(means[t], covs[t], xyah, wp)
        rcost, rg = cexact.blend(rc, rm, cfg.mc_lambda, cfg.gating_threshold, cfg.gated_cost, cfg.max_dist)
        assert bits_equal(cosd[t], rc), f"cosine row {t}"
        assert bits_equal(maha[t], rm), f"maha row {t}"
        assert np.array_equal(gated[t], rg)
        assert bits_equal(cost[t], rcost)
    if T >= 17:
        assert gated.any() and (~gated.astype(bool)).any()               # both sides of the gate exercised


def test_iou_cost_bit_exact(eng):
    rng = np.random.default_rng(5)
    t = np.c_[rng.uniform(0, 1000, 40), rng.uniform(0, 600, 40), rng.uniform(30, 120, 40), rng.uniform(60, 240, 40)]
    d = t[rng.integers(0, 40, 55)] + rng.normal(0, 15, (55, 4))
    d[:, 2:] = np.abs(d[:, 2:]) + 1
    got = eng.iou_cost(t, d).cpu().numpy()
    ref = np.stack([cexact.iou_cost(t[i], d, eng.cfg.max_iou_distance) for i in range(40)])
    assert bits_equal(got, ref)


@pytest.mark.parametrize("seed", range(12))
def test_lsap_identical_to_oracle_and_scipy(eng, seed):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(seed)
    shapes = [(30, 30), (100, 100), (128, 128), (1, 1), (7, 90), (90, 7), (200, 33), (33, 200), (256, 256)]
    nr, nc = shapes[seed % len(shapes)]
    if seed % 2:
        cost = rng.integers(0, 4, (nr, nc)).astype(np.float64)          # tie-heavy
    else:
        cost = rng.random((nr, nc))
        cost[cost > 0.7] = 0.2 + 1e-5                                   # thresholded plateau
    r2c = eng.lsap(cost).cpu().numpy()
    rows, cols = cexact.lsap(cost)
    ref = np.full(nr, -1, np.int32)
    ref[rows] = cols
    assert np.array_equal(r2c, ref)
    sr, sc = linear_sum_assignment(cost)
    assert np.array_equal(rows, sr) and np.array_equal(cols, sc)
