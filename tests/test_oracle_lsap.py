"""The C oracle's LSAP must give SciPy's pairs exactly, ties included (DECISIONS exactness contract)."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import cexact


@pytest.mark.parametrize("seed", range(40))
def test_lsap_matches_scipy_on_ties(seed):
    rng = np.random.default_rng(seed)
    nr, nc = rng.integers(1, 40, 2)
    hi = int(rng.integers(2, 6))           # tiny integer range -> many ties
    cost = rng.integers(0, hi, (nr, nc)).astype(np.float64)
    r, c = linear_sum_assignment(cost)
    r2, c2 = cexact.lsap(cost)
    assert np.array_equal(r, r2) and np.array_equal(c, c2)


@pytest.mark.parametrize("shape", [(30, 30), (100, 100), (7, 33), (64, 5), (1, 1), (128, 128)])
def test_lsap_matches_scipy_float(shape):
    rng = np.random.default_rng(sum(shape))
    cost = rng.random(shape)
    cost[cost > 0.6] = 0.2 + 1e-5          # thresholded plateau as in min_cost_matching
    r, c = linear_sum_assignment(cost)
    r2, c2 = cexact.lsap(cost)
    assert np.array_equal(r, r2) and np.array_equal(c, c2)
    assert cost[r, c].sum() == cost[r2, c2].sum()


def test_lsap_constant_matrix_is_identity():
    r, c = cexact.lsap(np.ones((9, 9)))
    assert np.array_equal(c, np.arange(9))


def test_lsap_empty():
    r, c = cexact.lsap(np.zeros((0, 5)))
    assert len(r) == 0
