"""Host-side handle on the gfx950 StrongSORT library: one context = S device-resident streams.

torch is used for device memory and streams only; every arithmetic step of the hot path runs in
libstrongsort_hip.so (csrc/).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import sys as _sys
from dataclasses import dataclass

import numpy as np
import torch

from . import lib as _lib
from .config import StrongSortConfig, DetectConfig
from .lib import MAX_TRACKS, MAX_DETS, FEAT_DIM, OUT_COLS


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@dataclass
class LetterboxGeom:
    out_h: int
    out_w: int
    new_h: int
    new_w: int
    pad_top: int
    pad_left: int
    gain: float


def letterbox_geometry(h: int, w: int, imgsz: int = 640, stride: int = 32, auto: bool = True) -> LetterboxGeom:
    """Ultralytics LetterBox geometry (host integers; DECISIONS D-14).  Stands behind the
    preprocessing inside model.track / model.predict (yolo_multi_model.py:41, :173)."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return LetterboxGeom(new_h + top + bottom, new_w + left + right, new_h, new_w, top, left, r)


def scale_geometry(g: LetterboxGeom, h0: int, w0: int):
    """gain / pad used by scale_boxes to map letterboxed boxes back (Ultralytics scale_boxes)."""
    gain = min(g.out_h / h0, g.out_w / w0)
    pad_x = round((g.out_w - w0 * gain) / 2 - 0.1)
    pad_y = round((g.out_h - h0 * gain) / 2 - 0.1)
    return float(gain), float(pad_x), float(pad_y)


class TrackerEngine:
    def __init__(self, cfg: StrongSortConfig | None = None, n_streams: int = 1, device: int = 0, debug: bool = False):
        if not torch.cuda.is_available():
            raise _lib.SSError(_lib.SS_ERR_HIP, "no HIP device visible: the StrongSORT hot path has no CPU fallback")
        self.cfg = cfg or StrongSortConfig()
        self.S = n_streams
        self.device = torch.device("cuda", device)
        self.L = _lib.load()
        self.ctx = C.c_void_p()
        c = _lib.make_config(self.cfg, n_streams, debug)
        _lib.check(None, self.L.ss_create(C.byref(c), device, C.byref(self.ctx)))
        self.debug_enabled = debug
        self.use_current_stream()
        dev = self.device
        self.out = torch.zeros(n_streams, MAX_TRACKS, OUT_COLS, dtype=torch.float32, device=dev)
        self.nout = torch.zeros(n_streams, dtype=torch.int32, device=dev)

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            torch.cuda.synchronize(self.device)          # nothing of ours may still be in flight on any stream
            for h in getattr(self, "_streams_keep", []):
                self.L.ss_stream_destroy(self.ctx, C.c_void_p(h))
            self._streams_keep = []
            self.L.ss_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:                                             # interpreter teardown: the HIP runtime may already be gone
            if _sys is None or _sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.ctx, rc)

    def use_current_stream(self):
        self._ck(self.L.ss_set_hip_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def reset(self, stream: int = -1):
        self._ck(self.L.ss_reset(self.ctx, stream))

    def set_option(self, name: str, value: int):
        """cos_grid — see include/strongsort_hip.h."""
        self._ck(self.L.ss_set_option(self.ctx, name.encode(), int(value)))

    def check_errors(self):
        self._ck(self.L.ss_check_errors(self.ctx))

    def upload(self, dst: torch.Tensor, src: np.ndarray, stream=None):
        """Host array -> device tensor through the library's write-combined staging ring (asynchronous on `stream`,
        default the current torch stream).  `src` may be reused immediately."""
        src = np.ascontiguousarray(src)
        if src.nbytes != dst.numel() * dst.element_size() or not dst.is_contiguous():
            raise ValueError("upload: size / layout mismatch")
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        self._ck(self.L.ss_upload(self.ctx, C.c_void_p(st.cuda_stream), _ptr(dst), src.ctypes.data_as(C.c_void_p), src.nbytes))

    def upload_batch(self, dst: torch.Tensor, srcs, stream=None, threads: int = 4):
        """len(srcs) host arrays of one size -> dst[0 .. len(srcs)) (device, contiguous) in ONE asynchronous copy: the arrays are staged
        by `threads` host threads in a write-combined area (a single thread staging 32 frames of 720p costs more than the GPU needs
        for them).  The arrays may be reused immediately."""
        n = len(srcs)
        if n == 0:
            return
        srcs = [np.ascontiguousarray(a) for a in srcs]
        each = srcs[0].nbytes
        if (any(a.nbytes != each or a.shape != srcs[0].shape for a in srcs) or not dst.is_contiguous() or dst[0].numel() * dst.element_size() != each
                or dst.shape[0] < n or tuple(dst.shape[1:]) != tuple(srcs[0].shape)):
            raise ValueError("upload_batch: size / shape / layout mismatch")
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        arr = (C.c_void_p * n)(*[a.ctypes.data for a in srcs])
        self._ck(self.L.ss_upload_batch(self.ctx, C.c_void_p(st.cuda_stream), _ptr(dst), arr, n, each, int(threads)))

    def download(self, dst: np.ndarray, src: torch.Tensor, stream=None):
        """Device tensor -> host array (synchronous)."""
        if dst.nbytes != src.numel() * src.element_size() or not src.is_contiguous() or not dst.flags["C_CONTIGUOUS"]:
            raise ValueError("download: size / layout mismatch")
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        self._ck(self.L.ss_download(self.ctx, C.c_void_p(st.cuda_stream), dst.ctypes.data_as(C.c_void_p), _ptr(src), dst.nbytes))

    # ---- tracker --------------------------------------------------------------------------------
    def update_device(self, dets, ndets, feats, img_hw):
        """All streams, one frame; tensors live on the device ([S,128,6] f32, [S] i32, [S,128,512] f32,
        [S,2] i32).  Asynchronous; returns (rows [S,256,8], counts [S]) device tensors."""
        self._ck(self.L.ss_track_update(self.ctx, _ptr(dets), _ptr(ndets), _ptr(feats), _ptr(img_hw),
                                        _ptr(self.out), _ptr(self.nout)))
        return self.out, self.nout

    @property
    def max_group_frames(self) -> int:
        return int(self.L.ss_max_group_frames())

    def update_group(self, n_frames, dets, ndets, feats, img_hw, out, nout):
        """A group of n_frames (<= max_group_frames) consecutive frames of all streams: tensors [F,S,128,6] f32, [F,S] i32,
        [F,S,128,512] f32, [S,2] i32 -> rows out [F,S,256,8], counts nout [F,S] (device tensors, asynchronous).
        Frames are associated in order; the galleries are read once for the whole group."""
        self._ck(self.L.ss_track_update_group(self.ctx, int(n_frames), _ptr(dets), _ptr(ndets), _ptr(feats), _ptr(img_hw),
                                              _ptr(out), _ptr(nout)))
        return out, nout

    def cmc_estimate(self, frames: torch.Tensor, n_frames: int, warps: torch.Tensor = None, stream=None, n_valid: torch.Tensor = None):
        """N4: ECC camera-motion warps of a group: frames uint8 [F*S,H,W,3] ([F][S] order) -> warps float64 [F,S,8]
        (2x3 matrix previous -> current frame, [6] = iterations or -1).  Asynchronous.  n_valid: device int32 [1], the real
        frames of a partial group (the rest of the buffer is stale)."""
        if warps is None:
            warps = torch.zeros(n_frames, self.S, 8, dtype=torch.float64, device=self.device)
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        self._ck(self.L.ss_cmc_estimate(self.ctx, C.c_void_p(st.cuda_stream), _ptr(frames), int(n_frames), frames.stride(0),
                                        frames.shape[1], frames.shape[2], frames.stride(1), _ptr(n_valid), _ptr(warps)))
        return warps

    def set_cmc(self, warps):
        """The following tracker calls compensate camera motion with these warps ([F,S,8] float64); None: off."""
        self._cmc_keep = warps
        self._ck(self.L.ss_track_set_cmc(self.ctx, _ptr(warps)))

    def set_assoc_event(self, event):
        """`event` (torch.cuda.Event, recorded at least once, or None) is recorded on the tracker's stream right after the
        association launch of every following update_group call."""
        self._assoc_ev_keep = event
        self._ck(self.L.ss_track_set_assoc_event(self.ctx, C.c_void_p(event.cuda_event if event is not None else 0)))

    def track_join(self, stream):
        """`stream` (torch.cuda.Stream) waits for the detached per-frame chain of the last update_group (option "chain_cus")."""
        self._ck(self.L.ss_track_join(self.ctx, C.c_void_p(stream.cuda_stream)))

    def create_stream(self, skip_cus: int = 0):
        """A stream whose queue leaves the first `skip_cus` compute units of the CU mask alone (ss_stream_create) as a
        torch.cuda.ExternalStream; the handle lives as long as the engine."""
        import torch
        h = C.c_void_p()
        self._ck(self.L.ss_stream_create(self.ctx, int(skip_cus), C.byref(h)))
        if not hasattr(self, "_streams_keep"):
            self._streams_keep = []
        self._streams_keep.append(h.value)
        return torch.cuda.ExternalStream(h.value, device=self.device)

    def update_host(self, dets: np.ndarray, feats: np.ndarray, img_hw) -> np.ndarray:
        """Single-stream synchronous update with host arrays -> rows [M,8] float32."""
        dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, 6)
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, FEAT_DIM)
        n = dets.shape[0]
        out = np.empty((MAX_TRACKS, OUT_COLS), dtype=np.float32)
        n_out = C.c_int(0)
        f32p = C.POINTER(C.c_float)
        self._ck(self.L.ss_track_update_host(
            self.ctx, 0, dets.ctypes.data_as(f32p), n, feats.ctypes.data_as(f32p), int(img_hw[0]), int(img_hw[1]),
            out.ctypes.data_as(f32p), MAX_TRACKS, C.byref(n_out)))
        return out[: n_out.value].copy()

    # ---- inspection -------------------------------------------------------------------------------
    def tracks(self, stream: int = 0) -> dict:
        T = MAX_TRACKS
        n, nid = C.c_int(), C.c_int()
        ints = {k: np.zeros(T, np.int32) for k in ("track_id", "state", "hits", "age", "tsu", "class_id", "gal_count")}
        conf = np.zeros(T, np.float32)
        mean, cov = np.zeros((T, 8)), np.zeros((T, 8, 8))
        smooth = np.zeros((T, FEAT_DIM), np.float32)
        ip, fp, dp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
        self._ck(self.L.ss_get_tracks(
            self.ctx, stream, T, C.byref(n), C.byref(nid), ints["track_id"].ctypes.data_as(ip),
            ints["state"].ctypes.data_as(ip), ints["hits"].ctypes.data_as(ip), ints["age"].ctypes.data_as(ip),
            ints["tsu"].ctypes.data_as(ip), ints["class_id"].ctypes.data_as(ip), conf.ctypes.data_as(fp),
            mean.ctypes.data_as(dp), cov.ctypes.data_as(dp), smooth.ctypes.data_as(fp),
            ints["gal_count"].ctypes.data_as(ip)))
        k = n.value
        d = {key: v[:k].copy() for key, v in ints.items()}
        d.update(conf=conf[:k].copy(), mean=mean[:k].copy(), cov=cov[:k].copy(), smooth=smooth[:k].copy(),
                 next_id=nid.value)
        return d

    def debug(self, stream: int = 0, frame: int = 0) -> dict:
        T, D = MAX_TRACKS, MAX_DETS
        counts = np.zeros(6, np.int32)
        cosd = np.zeros((T, D), np.float32)
        maha, cost_a, cost_b = np.zeros((T, D)), np.zeros((T, D)), np.zeros((T, D))
        gated = np.zeros((T, D), np.uint8)
        lists = np.zeros((4, T), np.int32)
        ip, fp, dp, up = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        self._ck(self.L.ss_get_debug(self.ctx, stream, frame, counts.ctypes.data_as(ip), cosd.ctypes.data_as(fp),
                                     maha.ctypes.data_as(dp), gated.ctypes.data_as(up), cost_a.ctypes.data_as(dp),
                                     cost_b.ctypes.data_as(dp), lists.ctypes.data_as(ip)))
        nC, nCand, nCols, nD, path_a, path_b = (int(v) for v in counts)
        return dict(n_conf=nC, n_cand=nCand, n_cols=nCols, n_dets=nD, path_a=path_a, path_b=path_b, cos=cosd[:nC, :nD], maha=maha[:nC, :nD],
                    gated=gated[:nC, :nD], cost_a=cost_a[:nC, :nD], cost_b=cost_b[:nCand, :nCols],
                    pairs_a=lists[0, :nC], cand=lists[1, :nCand], cols_b=lists[2, :nCols], pairs_b=lists[3, :nCand])

    def gallery(self, stream: int, track_index: int) -> np.ndarray:
        rows = np.zeros((128, FEAT_DIM), np.float32)
        cnt = C.c_int()
        self._ck(self.L.ss_get_gallery(self.ctx, stream, track_index, rows.ctypes.data_as(C.POINTER(C.c_float)), 128, C.byref(cnt)))
        return rows[: cnt.value].copy()

    def assoc_inkernel_timing(self, enable):
        """(mean microseconds, launches) of the association kernel measured by the kernel itself since the last call."""
        us, n = C.c_double(), C.c_int()
        self._ck(self.L.ss_assoc_inkernel_timing(self.ctx, int(enable), C.byref(us), C.byref(n)))
        return us.value, n.value

    def assoc_timeline(self, n_workgroups: int = 512):
        """[n_workgroups, 16] wall-clock stamps (100 MHz) of the last association launch (assoc_inkernel_timing(2))."""
        buf = np.zeros((n_workgroups, 16), np.int64)
        self._ck(self.L.ss_assoc_timeline(self.ctx, buf.ctypes.data_as(C.POINTER(C.c_longlong)), n_workgroups))
        return buf

    def assoc_timing(self, enable: bool):
        ms, n = C.c_float(), C.c_int()
        self._ck(self.L.ss_assoc_timing(self.ctx, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def assoc_timing_values(self) -> np.ndarray:
        """Per-launch durations (ms) behind the mean the last assoc_timing() call returned."""
        n = C.c_int()
        self._ck(self.L.ss_assoc_timing_values(self.ctx, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float32)
        self._ck(self.L.ss_assoc_timing_values(self.ctx, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
        return out[: n.value]

    # ---- stage entry points (device tensors in, device tensors out) ---------------------------------
    def _dev(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device).contiguous()

    def feat_normalize(self, raw):
        raw = self._dev(raw, torch.float32)
        out = torch.empty_like(raw)
        self._ck(self.L.ss_feat_normalize(self.ctx, _ptr(raw), raw.shape[0], _ptr(out)))
        return out

    def ema(self, smooth, feat):
        s, f = self._dev(smooth, torch.float32), self._dev(feat, torch.float32)
        out = torch.empty_like(s)
        self._ck(self.L.ss_ema(self.ctx, _ptr(s), _ptr(f), s.shape[0], _ptr(out)))
        return out

    def kf_predict(self, mean, cov):
        m, c = self._dev(mean, torch.float64), self._dev(cov, torch.float64)
        self._ck(self.L.ss_kf_predict(self.ctx, _ptr(m), _ptr(c), m.shape[0]))
        return m, c

    def kf_update(self, mean, cov, z, conf):
        m, c = self._dev(mean, torch.float64), self._dev(cov, torch.float64)
        z, cf = self._dev(z, torch.float64), self._dev(conf, torch.float64)
        self._ck(self.L.ss_kf_update(self.ctx, _ptr(m), _ptr(c), _ptr(z), _ptr(cf), m.shape[0]))
        return m, c

    def kf_project(self, mean, cov, conf=None):
        """a7: (projected mean [n,4], innovation covariance [n,4,4]) of the states, NSA noise for `conf` (None: 0)."""
        m, c = self._dev(mean, torch.float64), self._dev(cov, torch.float64)
        cf = self._dev(conf, torch.float64) if conf is not None else None
        n = m.shape[0]
        z = torch.empty(n, 4, dtype=torch.float64, device=self.device)
        S = torch.empty(n, 4, 4, dtype=torch.float64, device=self.device)
        self._ck(self.L.ss_kf_project(self.ctx, _ptr(m), _ptr(c), _ptr(cf), n, _ptr(z), _ptr(S)))
        return z, S

    def kf_initiate(self, z):
        z = self._dev(z, torch.float64)
        n = z.shape[0]
        m = torch.empty(n, 8, dtype=torch.float64, device=self.device)
        c = torch.empty(n, 8, 8, dtype=torch.float64, device=self.device)
        self._ck(self.L.ss_kf_initiate(self.ctx, _ptr(z), n, _ptr(m), _ptr(c)))
        return m, c

    def gallery_pack(self, gallery):
        g = self._dev(gallery, torch.float32)            # [T,B,512]
        T, B = g.shape[0], g.shape[1]
        frag = torch.zeros(T, 4, 16384, dtype=torch.float32, device=self.device)
        self._ck(self.L.ss_gallery_pack(self.ctx, _ptr(g), T, B, _ptr(frag)))
        return frag

    def assoc_cost(self, gallery_frag, counts, feats, mean, cov, xyah):
        counts = self._dev(counts, torch.int32)
        feats = self._dev(feats, torch.float32)
        mean, cov, xyah = self._dev(mean, torch.float64), self._dev(cov, torch.float64), self._dev(xyah, torch.float64)
        T, D = counts.shape[0], feats.shape[0]
        cost = torch.empty(T, D, dtype=torch.float64, device=self.device)
        cosd = torch.empty(T, D, dtype=torch.float32, device=self.device)
        maha = torch.empty(T, D, dtype=torch.float64, device=self.device)
        gated = torch.empty(T, D, dtype=torch.uint8, device=self.device)
        self._ck(self.L.ss_assoc_cost(self.ctx, _ptr(gallery_frag), _ptr(counts), T, _ptr(feats), D, _ptr(mean),
                                      _ptr(cov), _ptr(xyah), _ptr(cost), _ptr(cosd), _ptr(maha), _ptr(gated)))
        return cost, cosd, maha, gated

    def iou_cost(self, track_tlwh, det_tlwh):
        t, d = self._dev(track_tlwh, torch.float64), self._dev(det_tlwh, torch.float64)
        cost = torch.empty(t.shape[0], d.shape[0], dtype=torch.float64, device=self.device)
        self._ck(self.L.ss_iou_cost(self.ctx, _ptr(t), t.shape[0], _ptr(d), d.shape[0], _ptr(cost)))
        return cost

    def lsap(self, cost):
        c = self._dev(cost, torch.float64)
        nr, nc = c.shape
        r2c = torch.full((max(nr, 1),), -1, dtype=torch.int32, device=self.device)
        self._ck(self.L.ss_lsap(self.ctx, _ptr(c), nr, nc, _ptr(r2c)))
        return r2c[:nr]

    # ---- front end ---------------------------------------------------------------------------------
    def letterbox(self, frame: torch.Tensor, g: LetterboxGeom, half: bool = False, pad_value: int = 114, out=None):
        """frame: uint8 [H,W,3] BGR on the device -> [3,out_h,out_w] float/half."""
        H, W = frame.shape[0], frame.shape[1]
        if out is None:
            out = torch.empty(3, g.out_h, g.out_w, dtype=torch.float16 if half else torch.float32, device=self.device)
        self._ck(self.L.ss_letterbox(self.ctx, _ptr(frame), H, W, frame.stride(0), _ptr(out), int(half), g.out_h,
                                     g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left, pad_value))
        return out

    def letterbox_batch(self, frames: torch.Tensor, g: LetterboxGeom, half: bool = False, pad_value: int = 114,
                        out=None, channels_last: bool = False):
        """frames: uint8 [B,H,W,3] BGR on the device -> [B,3,out_h,out_w] float/half in one launch;
        channels_last=True writes the NHWC memory format the convolutions read (no permute copy afterwards)."""
        B, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if out is None:
            out = torch.empty(B, 3, g.out_h, g.out_w, dtype=torch.float16 if half else torch.float32, device=self.device,
                              memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        flags = (_lib.DST_F16 if half else 0) | (_lib.DST_HWC if channels_last else 0)
        self._ck(self.L.ss_letterbox_batch(self.ctx, _ptr(frames), B, frames.stride(0), H, W, frames.stride(1), _ptr(out),
                                           flags, g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left, pad_value))
        return out

    def nms_batch(self, pred: torch.Tensor, nc: int, dcfg: DetectConfig, geom: torch.Tensor, n_extra: int = 0,
                  rows=None, keep=None, count=None, max_det: int | None = None):
        """pred: [B,(4+nc+n_extra),N] float32; geom: [B,5] float32 rows (gain, pad_x, pad_y, w0, h0), both on the
        device.  One set of launches for the whole batch."""
        B, N = pred.shape[0], pred.shape[2]
        md = min(dcfg.max_det, 1024) if max_det is None else max_det
        stride = 6 + n_extra
        if rows is None:
            rows = torch.zeros(B, md, stride, dtype=torch.float32, device=self.device)
            keep = torch.zeros(B, md, dtype=torch.int32, device=self.device)
            count = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._ck(self.L.ss_nms_batch(self.ctx, _ptr(pred), B, pred.stride(0), N, nc, n_extra, dcfg.conf, dcfg.iou,
                                     int(dcfg.agnostic_nms), dcfg.max_wh, md, _ptr(geom), _ptr(rows), rows.stride(1),
                                     rows.stride(0), _ptr(keep), keep.stride(0), _ptr(count)))
        return rows, keep, count

    def nms_set_classes(self, classes=None):
        """Keep only these class ids in the following NMS calls (None / empty: all) — overrides['classes']."""
        cl = [] if classes is None else ([int(classes)] if np.isscalar(classes) else [int(c) for c in classes])
        arr = (C.c_int * max(len(cl), 1))(*cl)
        self._ck(self.L.ss_nms_set_classes(self.ctx, arr, len(cl)))

    def crop_norm_batch(self, frames: torch.Tensor, dets: torch.Tensor, n: int, counts=None, half: bool = False,
                        out=None, channels_last: bool = False):
        """frames uint8 [B,H,W,3], dets [B,cap,>=4] float32, counts [B] int32 -> crops [B*n,3,256,128]."""
        B, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if out is None:
            out = torch.empty(B * n, 3, 256, 128, dtype=torch.float16 if half else torch.float32, device=self.device,
                              memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        flags = (_lib.DST_U8 if out.dtype == torch.uint8 else _lib.DST_F16 if half else 0) | (_lib.DST_HWC if channels_last else 0)
        self._ck(self.L.ss_crop_norm_batch(self.ctx, _ptr(frames), B, frames.stride(0), H, W, frames.stride(1), _ptr(dets),
                                           dets.stride(1), dets.stride(0), n, _ptr(counts), _ptr(out), flags))
        return out

    def crop_norm_packed(self, frames: torch.Tensor, dets: torch.Tensor, n: int, counts: torch.Tensor, offsets: torch.Tensor,
                         out: torch.Tensor, half: bool = True):
        """Packed crops (channels-last `out` [B*n,3,256,128]): offsets int32 [B+1] <- exclusive prefix of min(counts, n); crop d
        of image i at slot offsets[i] + d; offsets[B] = crops in total.  A uint8 `out` receives the rounded bilinear values before
        the normalisation (SS_DST_U8: the fp32 ReID stem applies /255, mean and std itself)."""
        B, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        flags = ((_lib.DST_U8 if out.dtype == torch.uint8 else _lib.DST_F16 if half else 0)) | _lib.DST_HWC
        self._ck(self.L.ss_crop_norm_packed(self.ctx, _ptr(frames), B, frames.stride(0), H, W, frames.stride(1), _ptr(dets),
                                            dets.stride(1), dets.stride(0), n, _ptr(counts), _ptr(offsets), _ptr(out), flags))
        return out

    def unpack_feats(self, emb: torch.Tensor, offsets: torch.Tensor, counts: torch.Tensor, n: int, feats: torch.Tensor):
        """feats[i, d, :] = emb[offsets[i] + d, :] for d < min(counts[i], n); emb [*, 512] half or float, feats [B, cap, 512] f32."""
        assert emb.is_contiguous() and feats.stride(2) == 1 and feats.stride(1) == 512
        self._ck(self.L.ss_unpack_feats(self.ctx, _ptr(emb), int(emb.dtype == torch.float16), _ptr(offsets), _ptr(counts),
                                        feats.shape[0], n, _ptr(feats), feats.stride(0)))

    def pack_results(self, n_dets: torch.Tensor, dets: torch.Tensor, n_out, out, dst: torch.Tensor):
        """One frame's counts, detection rows and track rows into `dst` (float32, device or PINNED host memory) on the engine's stream:
        dst[0:2] = the two counts as int32 bits, then dets rows, then (from 2 + dets.numel()) track rows (csrc ss_pack_results)."""
        assert dets.is_contiguous() and dets.dim() == 2 and dst.dtype == torch.float32 and dst.is_contiguous()
        need = 2 + dets.numel() + (out.numel() if out is not None else 0)
        assert dst.numel() >= need and (dst.is_cuda or dst.is_pinned())
        if out is not None:
            assert out.is_contiguous() and out.dim() == 2
        self._ck(self.L.ss_pack_results(self.ctx, _ptr(n_dets), _ptr(dets), dets.shape[1], dets.shape[0], _ptr(n_out) if out is not None else None,
                                        _ptr(out) if out is not None else None, out.shape[1] if out is not None else 0,
                                        out.shape[0] if out is not None else 0, _ptr(dst)))

    def nms(self, pred: torch.Tensor, nc: int, dcfg: DetectConfig, gain: float, pad_x: float, pad_y: float,
            w0: int, h0: int, n_extra: int = 0, rows=None, keep=None, count=None):
        """pred: [(4+nc+n_extra), N] float32 on the device."""
        N = pred.shape[1]
        md = min(dcfg.max_det, 1024)
        stride = 6 + n_extra
        if rows is None:
            rows = torch.zeros(md, stride, dtype=torch.float32, device=self.device)
            keep = torch.zeros(md, dtype=torch.int32, device=self.device)
            count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ck(self.L.ss_nms(self.ctx, _ptr(pred), N, nc, n_extra, dcfg.conf, dcfg.iou, int(dcfg.agnostic_nms),
                               dcfg.max_wh, md, gain, pad_x, pad_y, float(w0), float(h0), _ptr(rows), stride,
                               _ptr(keep), _ptr(count)))
        return rows, keep, count

    def crop_norm(self, frame: torch.Tensor, dets: torch.Tensor, n: int, count=None, half: bool = False, out=None):
        H, W = frame.shape[0], frame.shape[1]
        if out is None:
            out = torch.empty(n, 3, 256, 128, dtype=torch.float16 if half else torch.float32, device=self.device)
        self._ck(self.L.ss_crop_norm(self.ctx, _ptr(frame), H, W, frame.stride(0), _ptr(dets), dets.stride(0), n,
                                     _ptr(count), _ptr(out), int(half)))
        return out
