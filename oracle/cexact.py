"""ctypes binding of the exact-order C oracle (oracle/csrc/ss_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY UNPINNED (see ss_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libss_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "csrc", "ss_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libss_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        f32p, f64p, i32p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint8))
        L = _lib
        L.so_dot.restype = C.c_float
        L.so_dot.argtypes = [f32p, f32p, C.c_int]
        L.so_cosine_min.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, f32p]
        L.so_sumsq.restype = C.c_float
        L.so_sumsq.argtypes = [f32p, C.c_int]
        L.so_normalize.argtypes = [f32p, f32p, C.c_int]
        L.so_ema.argtypes = [f32p, f32p, C.c_float, C.c_float, f32p, C.c_int]
        L.so_kf_initiate.argtypes = [f64p, C.c_double, C.c_double, f64p, f64p]
        L.so_kf_predict.argtypes = [f64p, f64p, C.c_double, C.c_double]
        L.so_kf_project.argtypes = [f64p, f64p, C.c_double, C.c_double, f64p, f64p]
        L.so_chol4.argtypes = [f64p, f64p]
        L.so_gating.argtypes = [f64p, f64p, f64p, C.c_int, C.c_double, f64p]
        L.so_kf_update.argtypes = [f64p, f64p, f64p, C.c_double, C.c_double]
        L.so_blend.argtypes = [f32p, f64p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, f64p, u8p]
        L.so_iou_cost.argtypes = [f64p, f64p, C.c_int, C.c_double, f64p]
        L.so_lsap.restype = C.c_int
        L.so_lsap.argtypes = [C.c_int, C.c_int, f64p, i32p]
        L.so_nms.restype = C.c_int
        L.so_nms.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                             C.c_int, C.c_int, i32p, f32p]
        L.so_nms_classes.restype = C.c_int
        L.so_nms_classes.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                     C.c_int, C.c_int, u8p, i32p, f32p]
        L.so_scale_boxes.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
        L.so_gray_small.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.so_ecc.restype = C.c_int
        L.so_ecc.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_double, f64p]
        L.so_camera_update.argtypes = [f64p, f64p]
        L.so_sincos.argtypes = [C.c_double, f64p, f64p]
        L.so_letterbox.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p] + [C.c_int] * 7
        L.so_crop_norm.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- appearance ---------------------------------------------------------------------------------
def dot(g, f):
    g, f = _f32(g), _f32(f)
    return np.float32(lib().so_dot(_p(g, C.c_float), _p(f, C.c_float), g.shape[0]))


def cosine_min(gallery, feats):
    gallery, feats = _f32(gallery), _f32(feats)
    D, F = feats.shape
    out = np.empty(D, dtype=np.float32)
    lib().so_cosine_min(_p(gallery, C.c_float), gallery.shape[0], _p(feats, C.c_float), D, F,
                        _p(out, C.c_float))
    return out


def sumsq(v):
    v = _f32(v)
    return np.float32(lib().so_sumsq(_p(v, C.c_float), v.shape[0]))


def normalize(v):
    v = _f32(v)
    out = np.empty_like(v)
    lib().so_normalize(_p(v, C.c_float), _p(out, C.c_float), v.shape[0])
    return out


def ema(smooth, feat, alpha):
    smooth, feat = _f32(smooth), _f32(feat)
    out = np.empty_like(smooth)
    a = np.float32(alpha)
    b = np.float32(1.0 - alpha)
    lib().so_ema(_p(smooth, C.c_float), _p(feat, C.c_float), a, b, _p(out, C.c_float), smooth.shape[0])
    return out


# ---- kalman ----------------------------------------------------------------------------------------
def kf_initiate(z, wp, wv):
    z = _f64(z)
    mean = np.empty(8)
    cov = np.empty((8, 8))
    lib().so_kf_initiate(_p(z, C.c_double), wp, wv, _p(mean, C.c_double), _p(cov, C.c_double))
    return mean, cov


def kf_predict(mean, cov, wp, wv):
    mean, cov = _f64(mean).copy(), _f64(cov).copy()
    lib().so_kf_predict(_p(mean, C.c_double), _p(cov, C.c_double), wp, wv)
    return mean, cov


def kf_project(mean, cov, conf, wp):
    mean, cov = _f64(mean), _f64(cov)
    m4 = np.empty(4)
    S = np.empty((4, 4))
    lib().so_kf_project(_p(mean, C.c_double), _p(cov, C.c_double), conf, wp, _p(m4, C.c_double), _p(S, C.c_double))
    return m4, S


def gating(mean, cov, Z, wp):
    mean, cov, Z = _f64(mean), _f64(cov), _f64(Z).reshape(-1, 4)
    out = np.empty(Z.shape[0])
    lib().so_gating(_p(mean, C.c_double), _p(cov, C.c_double), _p(Z, C.c_double), Z.shape[0], wp, _p(out, C.c_double))
    return out


def kf_update(mean, cov, z, conf, wp):
    mean, cov, z = _f64(mean).copy(), _f64(cov).copy(), _f64(z)
    lib().so_kf_update(_p(mean, C.c_double), _p(cov, C.c_double), _p(z, C.c_double), conf, wp)
    return mean, cov


# ---- cost / assignment ---------------------------------------------------------------------------
def blend(cosd, maha, lam, gate_thr, gated_cost, max_dist):
    cosd, maha = _f32(cosd), _f64(maha)
    D = cosd.shape[0]
    cost = np.empty(D)
    gated = np.empty(D, dtype=np.uint8)
    lib().so_blend(_p(cosd, C.c_float), _p(maha, C.c_double), D, lam, gate_thr, gated_cost, max_dist,
                   _p(cost, C.c_double), _p(gated, C.c_uint8))
    return cost, gated


def iou_cost(tlwh, det_tlwh, max_dist):
    tlwh, det_tlwh = _f64(tlwh), _f64(det_tlwh).reshape(-1, 4)
    out = np.empty(det_tlwh.shape[0])
    lib().so_iou_cost(_p(tlwh, C.c_double), _p(det_tlwh, C.c_double), det_tlwh.shape[0], max_dist, _p(out, C.c_double))
    return out


def lsap(cost):
    cost = _f64(cost)
    nr, nc = cost.shape
    r2c = np.empty(max(nr, 1), dtype=np.int32)
    rc = lib().so_lsap(nr, nc, _p(cost, C.c_double), _p(r2c, C.c_int))
    if rc != 0:
        raise ValueError("cost matrix is infeasible")
    r2c = r2c[:nr]
    rows = np.nonzero(r2c >= 0)[0]
    return rows.astype(np.int64), r2c[rows].astype(np.int64)


# ---- front end ---------------------------------------------------------------------------------------
def nms(pred, nc, conf_thres, iou_thres, agnostic=False, max_wh=7680.0, max_nms=8192, max_det=1000, classes=None):
    pred = _f32(pred)
    N = pred.shape[1]
    keep = np.empty(max(min(N, max_det), 1), dtype=np.int32)
    rows = np.empty((max(min(N, max_det), 1), 6), dtype=np.float32)
    allow = None
    if classes is not None:
        allow = np.zeros(nc, np.uint8)
        allow[list(classes)] = 1
    k = lib().so_nms_classes(_p(pred, C.c_float), N, nc, conf_thres, iou_thres, int(agnostic), max_wh,
                             max_nms, max_det, _p(allow, C.c_uint8) if allow is not None else None,
                             _p(keep, C.c_int), _p(rows, C.c_float))
    return keep[:k].copy(), rows[:k].copy()


def scale_boxes(rows, gain, pad_x, pad_y, w0, h0):
    rows = _f32(rows).copy()
    if rows.shape[0]:
        lib().so_scale_boxes(_p(rows, C.c_float), rows.shape[0], rows.shape[1], gain, pad_x, pad_y, w0, h0)
    return rows


def letterbox(img, out_h, out_w, new_h, new_w, pad_top, pad_left, pad_value=114):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    dst = np.empty((3, out_h, out_w), dtype=np.float32)
    lib().so_letterbox(_p(img, C.c_uint8), H, W, img.strides[0], _p(dst, C.c_float), out_h, out_w,
                       new_h, new_w, pad_top, pad_left, pad_value)
    return dst


def crop_norm(img, dets, out_h=256, out_w=128):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dets = _f32(dets)
    H, W = img.shape[:2]
    D = dets.shape[0]
    dst = np.empty((D, 3, out_h, out_w), dtype=np.float32)
    if D:
        lib().so_crop_norm(_p(img, C.c_uint8), H, W, img.strides[0], _p(dets, C.c_float),
                           dets.shape[1], D, _p(dst, C.c_float), out_h, out_w)
    return dst


# ---- N4 camera-motion compensation --------------------------------------------------------------------
def gray_small(img, hs, ws):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((hs, ws), dtype=np.uint8)
    lib().so_gray_small(_p(img, C.c_uint8), img.shape[0], img.shape[1], img.strides[0], _p(out, C.c_uint8), hs, ws)
    return out


def ecc(template, image, max_iter=100, eps=1e-5):
    """-> (warp [2,3] float64 mapping template to image coordinates (small-image pixels), iterations or -1)"""
    t, i = np.ascontiguousarray(template, dtype=np.uint8), np.ascontiguousarray(image, dtype=np.uint8)
    assert t.shape == i.shape
    warp = np.empty(6)
    it = lib().so_ecc(_p(t, C.c_uint8), _p(i, C.c_uint8), t.shape[0], t.shape[1], max_iter, eps, _p(warp, C.c_double))
    return warp.reshape(2, 3), it


def camera_update(mean, warp):
    mean, warp = _f64(mean).copy(), _f64(warp).reshape(6)
    lib().so_camera_update(_p(mean, C.c_double), _p(warp, C.c_double))
    return mean


def sincos(t):
    s, c = C.c_double(), C.c_double()
    lib().so_sincos(float(t), C.byref(s), C.byref(c))
    return s.value, c.value
