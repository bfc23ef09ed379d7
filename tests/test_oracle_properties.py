"""Property tests of the oracle building blocks: hypothesis-driven, CPU only."""
import itertools

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import cexact


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.integers(1, 6), st.integers(0, 10_000))
def test_lsap_is_optimal_against_brute_force(nr, nc, seed):
    rng = np.random.default_rng(seed)
    cost = rng.integers(0, 9, (nr, nc)).astype(np.float64)
    r, c = cexact.lsap(cost)
    got = cost[r, c].sum()
    k = min(nr, nc)
    best = np.inf
    if nr <= nc:
        for cols in itertools.permutations(range(nc), k):
            best = min(best, cost[np.arange(k), list(cols)].sum())
    else:
        for rows in itertools.permutations(range(nr), k):
            best = min(best, cost[list(rows), np.arange(k)].sum())
    assert len(r) == k and len(set(c.tolist())) == k and got == best


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10_000), st.integers(5, 300))
def test_nms_sorted_idempotent_and_non_overlapping(seed, n):
    rng = np.random.default_rng(seed)
    nc = 3
    pred = np.zeros((4 + nc, n), np.float32)
    pred[0] = rng.uniform(0, 200, n); pred[1] = rng.uniform(0, 200, n)
    pred[2] = rng.uniform(5, 60, n); pred[3] = rng.uniform(5, 60, n)
    pred[4:] = rng.uniform(0, 1, (nc, n))
    keep, rows = cexact.nms(pred, nc, 0.3, 0.4)
    assert np.all(np.diff(rows[:, 4]) <= 0)                                   # score order
    assert len(set(keep.tolist())) == len(keep)
    keep2, rows2 = cexact.nms(np.ascontiguousarray(pred[:, keep]), nc, 0.3, 0.4)
    assert len(keep2) == len(keep)                                            # idempotent
    # survivors of one class never overlap above the threshold
    for i in range(len(rows)):
        for j in range(i + 1, len(rows)):
            if rows[i, 5] != rows[j, 5]:
                continue
            a, b = rows[i, :4], rows[j, :4]
            iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0])); ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
            inter = iw * ih
            iou = inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)
            assert iou <= 0.4 + 1e-5


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 10_000))
def test_kalman_covariance_stays_symmetric_psd(seed):
    rng = np.random.default_rng(seed)
    wp, wv = 1 / 20, 1 / 160
    z = np.array([rng.uniform(0, 1900), rng.uniform(0, 1000), rng.uniform(0.2, 1.0), rng.uniform(20, 400)])
    m, c = cexact.kf_initiate(z, wp, wv)
    for _ in range(25):
        m, c = cexact.kf_predict(m, c, wp, wv)
        if rng.random() < 0.7:
            m, c = cexact.kf_update(m, c, m[:4] + rng.normal(0, 1, 4) * [2, 2, 0.01, 2], rng.uniform(0.3, 0.99), wp)
        assert np.allclose(c, c.T, rtol=1e-8, atol=1e-10)
        assert np.linalg.eigvalsh((c + c.T) / 2).min() > 0
    g = cexact.gating(m, c, m[:4][None] + np.zeros((1, 4)), wp)
    assert g[0] == 0.0


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10_000), st.integers(1, 40), st.integers(1, 20))
def test_cosine_min_is_a_min_over_rows(seed, B, D):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((B, 512)).astype(np.float32); g /= np.linalg.norm(g, axis=1, keepdims=True)
    f = rng.standard_normal((D, 512)).astype(np.float32); f /= np.linalg.norm(f, axis=1, keepdims=True)
    full = cexact.cosine_min(g, f)
    parts = np.stack([cexact.cosine_min(g[b:b + 1], f) for b in range(B)])
    assert np.array_equal(full, parts.min(axis=0))                            # exact: min is order-free
    perm = rng.permutation(B)
    assert np.array_equal(full, cexact.cosine_min(g[perm], f))                # ring-buffer order is irrelevant
