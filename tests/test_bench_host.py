"""CPU-side pieces of bench.py (workload synthesis, oracle replay, CPU baseline plumbing) and tracker invariants."""
import importlib.util
import os

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig
from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
from strongsort_yolo_amd.synth import make_stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workload_and_oracle_replay():
    b = _bench()
    W, H = 640, 480
    g = letterbox_geometry(H, W)
    gs = scale_geometry(g, H, W)
    A = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
    wl = b.make_workload(3, W, H, 6, 8, gs, 80, A)
    assert wl["preds"].shape == (8, 84, A) and wl["agt"].shape == (8, A) and wl["pixels"].dtype == np.uint8
    rows = b.oracle_rows(wl, 8, W, H, gs, 80, StrongSortConfig(), DetectConfig())
    assert len(rows) == 8 and rows[-1].shape[1] == 8 and len(rows[-1]) > 0        # confirmed tracks by frame 3
    assert set(b.PRESETS) >= {"c2", "c3", "c4"} and b.PREFILL >= 103


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 10_000), st.integers(2, 14))
def test_tracker_invariants(seed, n_ids):
    s = make_stream(seed, 640, 480, n_ids, p_vanish=0.1, vanish_max=6)
    t = OracleStrongSort(StrongSortConfig(max_age=4), "c")
    last_next = 1
    for _ in range(25):
        f = s.next_frame()
        rows = t.update(f.dets, f.feats, (480, 640))
        ids = rows[:, 4].astype(int)
        assert len(set(ids.tolist())) == len(ids)                                  # one row per track
        di = rows[:, 7].astype(int)
        used = di[di >= 0]
        assert len(set(used.tolist())) == len(used) and (used < len(f.dets)).all()  # a detection feeds one track
        assert t.next_id >= last_next                                               # ids are never reused
        last_next = t.next_id
        assert all(tr.state in (1, 2) for tr in t.tracks)                           # deleted tracks are gone
        assert [tr.track_id for tr in t.tracks] == sorted(tr.track_id for tr in t.tracks)
        assert (rows[:, 0] >= 0).all() and (rows[:, 2] <= 639).all() and (rows[:, 3] <= 479).all()
