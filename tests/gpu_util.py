import numpy as np
import pytest


def engine(cfg=None, n_streams=1, debug=True):
    from strongsort_yolo_amd.engine import TrackerEngine
    return TrackerEngine(cfg, n_streams=n_streams, debug=debug)


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def unit(rng, n, F=512):
    x = rng.standard_normal((n, F)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
