"""Committed vectors (tests/golden/, made by tests/golden/make_golden.py) replayed through the oracle.
Parity is unpinned w.r.t. the reference (nothing to generate vectors from); these pin the oracle itself."""
import os

import numpy as np
import pytest

from oracle import cexact
from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("numerics", ["c", "numpy"])
def test_tracker_vector(numerics):
    z = np.load(os.path.join(G, "tracker_6ids_24frames.npz"))
    orc = OracleStrongSort(StrongSortConfig(), numerics)
    H, W = z["hw"]
    for k in range(len(z["counts"])):
        n = int(z["counts"][k])
        r = orc.update(z["dets"][k, :n], z["feats"][k, :n], (H, W))
        ref = z["rows"][k, : int(z["nrows"][k])]
        ca = orc.last["cost_a"]
        gold = z["cost_a"][k][: ca.shape[0], : ca.shape[1]]
        if numerics == "c":
            assert r.tobytes() == ref.tobytes(), f"frame {k}"
            assert np.array_equal(ca, gold)
        else:                                        # library summation order: ids exact, floats within 1e-5
            assert r.shape == ref.shape and np.array_equal(r[:, [4, 5, 7]], ref[:, [4, 5, 7]])
            assert np.abs(ca - gold).max(initial=0) <= 1e-5
    snap = orc.snapshot()
    assert np.array_equal(snap["track_id"], z["final_ids"]) and snap["next_id"] == int(z["next_id"])
    if numerics == "c":
        assert np.array_equal(snap["mean"], z["final_mean"]) and np.array_equal(snap["cov"], z["final_cov"])


def test_front_vector():
    z = np.load(os.path.join(G, "front_small.npz"))
    oh, ow, nh, nw, pt, pl = (int(v) for v in z["geom"])
    assert np.array_equal(cexact.letterbox(z["img"], oh, ow, nh, nw, pt, pl), z["letterbox"])
    assert np.array_equal(cexact.crop_norm(z["img"], z["boxes"], 32, 16), z["crops"])
    dc = DetectConfig()
    keep, rows = cexact.nms(z["pred"], 3, dc.conf, dc.iou, dc.agnostic_nms, dc.max_wh, dc.max_nms, dc.max_det)
    gain, px, py = z["scale"]
    rows = cexact.scale_boxes(rows, gain, px, py, z["img"].shape[1], z["img"].shape[0])
    assert np.array_equal(keep, z["nms_keep"]) and np.array_equal(rows, z["nms_rows"])
    assert len(keep) == 3


def test_lsap_vector_and_scipy():
    from scipy.optimize import linear_sum_assignment
    z = np.load(os.path.join(G, "lsap_ties.npz"))
    for (nr, nc), m, s in zip(z["shapes"], z["mats"], z["sols"]):
        c = m[:nr, :nc]
        r, cc = cexact.lsap(c)
        got = np.full(12, -1, np.int64); got[r] = cc
        assert np.array_equal(got, s)
        sr, sc = linear_sum_assignment(c)
        assert np.array_equal(r, sr) and np.array_equal(cc, sc)
