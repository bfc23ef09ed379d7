"""`YOLO(weights)` — the model object the reference script drives.

Mirrors exactly what /root/reference/yolo_multi_model.py touches:
  :17     model = YOLO("yolo11n-pose.pt")
  :18-21  model.overrides['conf'|'iou'|'agnostic_nms'|'max_det'] (+ optional 'classes' :22)
  :23     model.names
  :41     model.track(image, verbose=False, device=0, persist=True, tracker="botsort.yaml") -> [Results]
  :173    model.predict(image, verbose=False, device=0)                                       -> [Results]
and the Results duck type consumed at :45-169 / :175-237 (boxes.id/.conf/.cls/.xyxy as 1-row tensors
per box, keypoints[i].xy, masks[i].xy, names).  The frame -> rows path runs on the MI355X
(pipeline.FramePipeline); this file is host glue only.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from .config import DetectConfig, StrongSortConfig

COCO_NAMES = ("person bicycle car motorcycle airplane bus train truck boat traffic_light fire_hydrant stop_sign "
              "parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella handbag tie "
              "suitcase frisbee skis snowboard sports_ball kite baseball_bat baseball_glove skateboard surfboard "
              "tennis_racket bottle wine_glass cup fork knife spoon bowl banana apple sandwich orange broccoli carrot "
              "hot_dog pizza donut cake chair couch potted_plant bed dining_table toilet tv laptop mouse remote keyboard "
              "cell_phone microwave oven toaster sink refrigerator book clock vase scissors teddy_bear hair_drier "
              "toothbrush").split()


def _row(i, n):
    """An int index as a one-row slice (negative indices count from the end, as on a tensor); anything else unchanged."""
    if isinstance(i, (int, np.integer)):
        i = int(i)
        if not -n <= i < n:
            raise IndexError(f"index {i} out of range for {n} rows")
        i %= n
        return slice(i, i + 1)
    return i


class Boxes:
    """Rows of [x1,y1,x2,y2,(id),conf,cls]; iterating yields 1-row Boxes (yolo_multi_model.py:73,126)."""

    def __init__(self, xyxy, conf, cls, ids=None):
        self.xyxy, self.conf, self.cls, self.id = xyxy, conf, cls, ids

    def __len__(self):
        return self.xyxy.shape[0]

    def __getitem__(self, i):
        i = _row(i, len(self))
        return Boxes(self.xyxy[i], self.conf[i], self.cls[i], None if self.id is None else self.id[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def is_track(self):
        return self.id is not None


class Keypoints:
    def __init__(self, data):                    # [n,17,3]
        self.data = data
        self.xy = data[..., :2]
        self.conf = data[..., 2]

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return Keypoints(self.data[_row(i, len(self))])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def assemble_masks(proto, coef, boxes_in, in_hw):
    """Instance masks at the network-input resolution from a frame's prototypes [nm, mh, mw], the kept rows' coefficients [n, nm]
    and their DETECTION boxes in input pixels [n, 4] (the published Ultralytics recipe `process_mask(..., upsample=True)`: linear
    combination, cropped to the box on the prototype grid, bilinear up to the input size, > 0).  Host arithmetic in float32:
    the device hands over prototypes and coefficients, nothing of this is on the timed path.  -> bool [n, ih, iw]"""
    import torch.nn.functional as F
    c, mh, mw = proto.shape
    ih, iw = in_hw
    n = coef.shape[0]
    if n == 0:
        return torch.zeros(0, ih, iw, dtype=torch.bool)
    m = (coef.float() @ proto.float().reshape(c, -1)).view(n, mh, mw)
    b = boxes_in.float().clone()
    b[:, [0, 2]] *= mw / iw
    b[:, [1, 3]] *= mh / ih
    x1, y1, x2, y2 = (b[:, i].view(n, 1, 1) for i in range(4))
    col = torch.arange(mw, dtype=torch.float32).view(1, 1, mw)
    row = torch.arange(mh, dtype=torch.float32).view(1, mh, 1)
    m = m * ((col >= x1) & (col < x2) & (row >= y1) & (row < y2))
    return F.interpolate(m[None], (ih, iw), mode="bilinear", align_corners=False)[0] > 0.0


_RING = ((-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1))      # (dx, dy) clockwise from west, y down


def trace_outline(comp: np.ndarray) -> np.ndarray:
    """Outer boundary of ONE 8-connected component (bool [h, w]) by Moore-neighbour tracing, clockwise from its first pixel in
    raster order, runs of equal chain direction reduced to their end points (what cv2.CHAIN_APPROX_SIMPLE keeps).  cv2 is not
    installed and its border-following order is not restated: the polygon is the same closed pixel chain, the start point
    and orientation are this function's.  -> int32 [k, 2] (x, y)"""
    h, w = comp.shape
    ys, xs = np.nonzero(comp)
    if len(ys) == 0:
        return np.zeros((0, 2), np.int32)
    sx, sy = int(xs[0]), int(ys[0])                       # np.nonzero is row-major: top-most row, left-most pixel
    inside = lambda x, y: 0 <= x < w and 0 <= y < h and comp[y, x]
    pts, x, y, back = [(sx, sy)], sx, sy, 0               # back: ring index of a background neighbour (west of the start pixel)
    first = None
    for _ in range(4 * (h * w + 4)):
        nxt = None
        for k in range(1, 9):
            j = (back + k) % 8
            qx, qy = x + _RING[j][0], y + _RING[j][1]
            if inside(qx, qy):
                bx, by = x + _RING[(j - 1) % 8][0], y + _RING[(j - 1) % 8][1]      # the background pixel scanned just before
                nxt = (qx, qy, _RING.index((bx - qx, by - qy)))
                break
        if nxt is None:                                   # an isolated pixel
            break
        if (x, y) == (sx, sy) and first is not None and nxt[:2] == first:          # back at the start, leaving as the first time
            pts.pop()
            break
        if first is None:
            first = nxt[:2]
        x, y, back = nxt
        pts.append((x, y))
    p = np.asarray(pts, np.int32)
    if len(p) < 3:
        return p
    d_in, d_out = p - np.roll(p, 1, 0), np.roll(p, -1, 0) - p
    return p[(d_in != d_out).any(1)]


def mask_polygon(mask: np.ndarray) -> np.ndarray:
    """The polygon Ultralytics' `masks2segments(strategy="largest")` stands for: the outline with the most points among the
    mask's 8-connected components (the 16 largest are traced).  -> float32 [k, 2] in the mask's pixel coordinates."""
    from scipy import ndimage
    lab, n = ndimage.label(mask, structure=np.ones((3, 3), np.int8))
    if n == 0:
        return np.zeros((0, 2), np.float32)
    order = np.argsort(-np.bincount(lab.ravel(), minlength=n + 1)[1:], kind="stable")[:16] + 1
    best = max((trace_outline(lab == int(i)) for i in order), key=len)
    return best.astype(np.float32)


class Masks:
    """Instance masks of one frame: the part of Ultralytics' `Masks` the reference touches (/root/reference/yolo_multi_model.py
    :71-72 iterates them in step with the boxes, :112-121 draws `masks.xy`).  `data`: bool [n, ih, iw] at the network-input size,
    `xy`: one float32 [k, 2] polygon per mask in ORIGINAL-image pixels, `xyn`: the same normalised.  Assembled lazily on the host
    from the frame's prototypes and the kept rows' coefficients (`assemble_masks`)."""

    def __init__(self, proto, coef, boxes_in, in_hw, orig_shape, gain, pad_xy, _data=None):
        self._proto, self._coef, self._boxes, self._in_hw = proto, coef, boxes_in, tuple(int(v) for v in in_hw)
        self.orig_shape, self._gain, self._pad = tuple(int(v) for v in orig_shape[:2]), float(gain), (float(pad_xy[0]), float(pad_xy[1]))
        self._data, self._xy = _data, None

    @property
    def data(self):
        if self._data is None:
            self._data = assemble_masks(self._proto, self._coef, self._boxes, self._in_hw)
        return self._data

    @property
    def xy(self):
        if self._xy is None:
            H, W = self.orig_shape
            out = []
            for m in self.data.numpy():
                p = mask_polygon(m)
                if len(p):
                    p = (p - np.float32(self._pad)) / np.float32(self._gain)       # scale_coords: un-pad, un-scale, clip
                    p[:, 0], p[:, 1] = p[:, 0].clip(0, W), p[:, 1].clip(0, H)
                out.append(p.astype(np.float32))
            self._xy = out
        return self._xy

    @property
    def xyn(self):
        H, W = self.orig_shape
        return [p / np.float32((W, H)) for p in self.xy]

    def __len__(self):
        return self._coef.shape[0]

    def __getitem__(self, i):
        i = _row(i, len(self))
        return Masks(self._proto, self._coef[i], self._boxes[i], self._in_hw, self.orig_shape, self._gain, self._pad,
                     None if self._data is None else self._data[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Results:
    def __init__(self, orig_img, names, boxes: Optional[Boxes], keypoints: Optional[Keypoints] = None, masks: Optional[Masks] = None):
        self.orig_img, self.names, self.boxes, self.keypoints, self.masks = orig_img, names, boxes, keypoints, masks
        self.orig_img_device = None      # track_stream(keep_device_frames=True): the frame as a device tensor uint8 [H,W,3] (for Overlay.draw_resident)

    def __len__(self):
        return 0 if self.boxes is None else len(self.boxes)


class YOLO:
    """`model = YOLO("yolo11n-pose.pt")` (yolo_multi_model.py:17).  `.track(frame)` / `.predict(frame)` replay HIP graphs
    captured once per (frame shape, overrides): the frame goes through a pinned host buffer, every stage runs on the
    device, the results come back through pinned buffers and there is ONE stream synchronisation per call.
    `.track_stream(frames, batch=16)` is the throughput form for sources that can supply frames ahead (files): groups
    of `batch` frames through the two-stream pipeline, same rows.

    Changing `.overrides` (or the frame size) re-captures the graphs and restarts the tracker, as a new
    `model.track(..., persist=False)` would."""

    def __init__(self, weights: str = "yolov8n.pt", seed: int = 0, random_init_ok: bool = False, reid_batch: int = 128,
                 camera_motion: bool = False, reid_weights: Optional[str] = None, reid_fp32: bool = True, half: bool = True):
        """reid_fp32 (default since round 6): ReID crops + OSNet-x0.25 in fp32 on the fp32 kernels — appearance distances within 1e-4 of a CPU fp32
        network, which f16 activations miss by 330x (reid_fp32=False: the f16 throughput mode, ~1.2x the per-frame rate, 1.7x the stream rate).
        half=False: the DETECTOR in fp32 as well (the reference's own precision: it passes no half=, yolo_multi_model.py:41) on the
        fp32 convolution kernels (csrc/ss_ops32.hip k32_conv) — NMS keep lists then equal the CPU fp32 network's; implies reid_fp32."""
        self.weights = weights
        self.reid_weights = reid_weights          # OSNet-x0.25 state_dict; same policy as the detector's (raise unless random init is asked for)
        self.random_init_ok = random_init_ok
        self.arch = os.path.basename(weights).replace(".pt", "")
        self.overrides = {"conf": 0.25, "iou": 0.7, "agnostic_nms": False, "max_det": 300}
        self.seed = seed
        self.reid_batch = reid_batch
        pose = "pose" in self.arch
        self.names = {0: "person"} if pose else dict(enumerate(COCO_NAMES))
        self._pipe = None
        self._key = None
        self._stream_pipe = None
        self._stream_key = None
        self._pred_pipe = None          # detection-only pipeline for predict() when max_det exceeds the tracker's 128 rows
        self._pred_key = None
        # test / bench hooks (synthetic head tensors, no weights exist offline): extra pipeline keywords and a callable
        # fill(buffers, virtual_stream, frame_index) that writes pred_in / anchor_gt / gt_feats before a frame runs
        self._pipe_kw = {"cmc": True} if camera_motion else {}    # N4: ECC camera-motion compensation (off by default)
        if reid_fp32 or not half:                                  # OSNet in fp32 (pipeline.FramePipeline reid_half): float distances within 1e-4
            self._pipe_kw["reid_half"] = False
        if not half:
            self._pipe_kw["half"] = False
        self._fill = None
        self._frame_index = 0

    def _dcfg(self):
        o = self.overrides
        return DetectConfig(conf=float(o["conf"]), iou=float(o["iou"]), agnostic_nms=bool(o["agnostic_nms"]),
                            max_det=int(o["max_det"]))

    def _state_key(self, shape, device):
        cl = self.overrides.get("classes")
        cl = None if cl is None else tuple(int(c) for c in (cl if isinstance(cl, (list, tuple)) else [cl]))
        return (tuple(shape), int(device or 0), self._dcfg(), cl)

    def _build(self, cls, shape, device, need_reid=True, **kw):
        from . import nets
        args = dict(reid_batch=self.reid_batch, cfg=StrongSortConfig(), dcfg=self._dcfg(), det_source="detector",
                    feat_source="reid", seed=self.seed)
        args.update(self._pipe_kw)
        args.update(kw)
        pipe = cls(self.arch, 1, shape, device=int(device or 0), **args)
        # the networks' weights, before the first forward (the fused weight-prep caches are built from the loaded tensors).
        # The ReID weights are needed only once the TRACKER consumes OSNet embeddings: model.predict (yolo_multi_model.py:173)
        # works with detector weights alone; the first model.track on such a pipeline rebuilds it with the ReID weights.
        nets.load_weights(pipe.detector, self.weights, f"detector {self.arch}", self.random_init_ok)
        pipe.reid_loaded = False
        if need_reid and pipe.feat_source == "reid" and pipe.det_rows == 128:
            nets.load_weights(pipe.reid, self.reid_weights, "OSNet-x0.25 ReID", self.random_init_ok)
            pipe.reid_loaded = True
        pipe.eng.nms_set_classes(self.overrides.get("classes"))
        return pipe

    # ---- per-frame path -------------------------------------------------------------------------------------
    def _pipeline(self, image, device, need_reid=True):
        from .pipeline import FramePipeline
        key = self._state_key(image.shape[:2], device)
        if self._pipe is None or key != self._key or (need_reid and not self._pipe.reid_loaded and self._pipe.feat_source == "reid"):
            if self._pipe is not None:
                self._pipe.close()
            p = self._pipe = self._build(FramePipeline, image.shape[:2], device, need_reid=need_reid, graph="split")
            self._key = key
            self._frame_index = 0
            H, W = image.shape[:2]
            # one pinned buffer the device writes the frame's counts, detection rows and track rows into (csrc ss_pack_results): the host
            # reads it after the call's one synchronisation, no copies in between
            nd, no = p.dets.shape[1] * p.dets.shape[2], p.out.shape[1] * p.out.shape[2]
            self._h_res = torch.zeros(2 + nd + no).pin_memory()
            self._h_cnt = self._h_res[:2].view(torch.int32)
            self._h_dets = self._h_res[2:2 + nd].view(p.dets.shape[1], p.dets.shape[2])
            self._h_rows = self._h_res[2 + nd:].view(p.out.shape[1], p.out.shape[2])
            self._h_proto = torch.empty(p.proto.shape[1:], dtype=p.proto.dtype).pin_memory() if p.nm else None
        return self._pipe

    def _run(self, image, device, track):
        if not track and int(self.overrides["max_det"]) > 128:
            return self._run_predict_wide(image, device)
        pipe = self._pipeline(image, device, need_reid=track)
        pipe.eng.upload(pipe.frames[0], image)
        if self._fill is not None:
            self._fill(pipe, 0, self._frame_index)
        pipe.step(track=track)
        pipe.eng.pack_results(pipe.ndets, pipe.dets[0], pipe.nout if track else None, pipe.out[0] if track else None, self._h_res)
        if pipe.nm:
            self._h_proto.copy_(pipe.proto[0], non_blocking=True)
        torch.cuda.current_stream(pipe.dev).synchronize()            # the one synchronisation of the call
        pipe.eng.check_errors()
        self._frame_index += 1
        n, m = int(self._h_cnt[0]), int(self._h_cnt[1])
        self._warn_if_capped(pipe, n, track)
        return self._results(image, pipe, self._h_dets[:n].clone(), self._h_rows[:m].clone() if track else None,
                             self._h_proto.clone() if pipe.nm else None)

    def _warn_if_capped(self, pipe, n_kept, track):
        """The tracking pipeline carries at most pipe.max_det (<= 128, <= reid_batch) detections per frame, highest scores first;
        the reference's max_det = 1000 (yolo_multi_model.py:21) applies to its tracker too.  A frame that fills the cap may have
        lost lower-scored detections: say so once (capacity errors elsewhere are loud; this truncation used to be silent)."""
        if track and n_kept >= pipe.max_det and pipe.max_det < int(self.overrides["max_det"]) and not getattr(self, "_cap_warned", False):
            import warnings
            self._cap_warned = True
            warnings.warn(f"a frame filled the tracker's per-frame limit of {pipe.max_det} detections (overrides['max_det'] = "
                          f"{self.overrides['max_det']}): lower-scored detections beyond it are not tracked "
                          f"(limit = min(max_det, 128, reid_batch = {self.reid_batch}))", RuntimeWarning, stacklevel=3)

    def _run_predict_wide(self, image, device):
        """model.predict with max_det > 128 (the reference sets 1000, yolo_multi_model.py:21): a detection-only pipeline
        whose NMS keeps up to 1024 rows; the tracking pipeline (128 detections per frame) is not involved."""
        from .pipeline import FramePipeline
        key = self._state_key(image.shape[:2], device)
        if self._pred_pipe is None or key != self._pred_key:
            if self._pred_pipe is not None:
                self._pred_pipe.close()
            p = self._pred_pipe = self._build(FramePipeline, image.shape[:2], device, graph="split", detect_only_rows=1024)
            self._pred_key = key
            self._hp_dets = torch.empty(p.dets.shape[1], p.dets.shape[2]).pin_memory()
            self._hp_cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._hp_proto = torch.empty(p.proto.shape[1:], dtype=p.proto.dtype).pin_memory() if p.nm else None
        pipe = self._pred_pipe
        pipe.eng.upload(pipe.frames[0], image)
        if self._fill is not None:
            self._fill(pipe, 0, self._frame_index)
        pipe.step(track=False)
        self._hp_dets.copy_(pipe.dets[0], non_blocking=True)
        self._hp_cnt.copy_(pipe.ndets, non_blocking=True)
        if pipe.nm:
            self._hp_proto.copy_(pipe.proto[0], non_blocking=True)
        torch.cuda.current_stream(pipe.dev).synchronize()
        pipe.eng.check_errors()
        return self._results(image, pipe, self._hp_dets[:int(self._hp_cnt[0])].clone(), None,
                             self._hp_proto.clone() if pipe.nm else None)

    def _results(self, image, pipe, dets, rows, proto=None):
        kpts = masks = None
        if pipe.nk:
            k = dets[:, 6:6 + pipe.nk].reshape(dets.shape[0], pipe.nk // 3, 3).clone()
            k[..., 0] = (k[..., 0] - pipe.pad_x) / pipe.gain
            k[..., 1] = (k[..., 1] - pipe.pad_y) / pipe.gain
            kpts = k
        if pipe.nm and proto is not None:
            # masks are cut with the DETECTION boxes (as upstream: assembled at predict time, before the tracker replaces the boxes),
            # brought back to the network-input frame: x * gain + pad
            b = dets[:, :4].clone()
            b[:, [0, 2]] = b[:, [0, 2]] * pipe.gain + pipe.pad_x
            b[:, [1, 3]] = b[:, [1, 3]] * pipe.gain + pipe.pad_y
            masks = Masks(proto, dets[:, 6 + pipe.nk:6 + pipe.nk + pipe.nm].clone(), b, (pipe.geom.out_h, pipe.geom.out_w),
                          image.shape, pipe.gain, (pipe.pad_x, pipe.pad_y))
        if rows is None:
            return [Results(image, self.names, Boxes(dets[:, :4], dets[:, 4], dets[:, 5]),
                            None if kpts is None else Keypoints(kpts), masks)]
        rows = rows[rows[:, 7] >= 0]                   # ultralytics semantics: results[i] = results[i][det_idx]
        if rows.shape[0] == 0:
            return [Results(image, self.names, Boxes(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0), None))]
        di = rows[:, 7].long()
        return [Results(image, self.names, Boxes(rows[:, :4], rows[:, 6], rows[:, 5], rows[:, 4]),
                        None if kpts is None else Keypoints(kpts[di]), None if masks is None else masks[di])]

    TRACKERS = ("strongsort.yaml", "strongsort", "botsort.yaml", "bytetrack.yaml")

    def _check_tracker(self, tracker):
        """`tracker=`: this library has ONE tracker, StrongSORT (BASELINE north_star).  The reference passes "botsort.yaml"
        (yolo_multi_model.py:41) — the Ultralytics tracker configurations are accepted and answered by StrongSORT, once with a
        warning that says so; any other value is an error instead of being ignored."""
        name = os.path.basename(str(tracker))
        if name not in self.TRACKERS:
            raise ValueError(f"tracker={tracker!r}: this library tracks with StrongSORT only (accepted: {', '.join(self.TRACKERS)})")
        if not name.startswith("strongsort") and not getattr(self, "_tracker_warned", False):
            import warnings
            self._tracker_warned = True
            warnings.warn(f"tracker={tracker!r} is an Ultralytics configuration; tracking runs StrongSORT (OSNet-x0.25 appearance + NSA Kalman), "
                          f"its parameters are strongsort_yolo_amd.config.StrongSortConfig", RuntimeWarning, stacklevel=3)

    @torch.no_grad()
    def track(self, image, verbose=False, device=0, persist=True, tracker="strongsort.yaml", **kw) -> List[Results]:
        self._check_tracker(tracker)
        if not persist and self._pipe is not None:
            self._pipe.eng.reset(-1)
        return self._run(image, device, True)

    @torch.no_grad()
    def predict(self, image, verbose=False, device=0, **kw) -> List[Results]:
        return self._run(image, device, False)

    __call__ = predict

    # ---- throughput path ------------------------------------------------------------------------------------
    @torch.no_grad()
    def track_stream(self, frames, batch: int = 16, device=0, keep_device_frames: bool = False):
        """Generator over `frames` (BGR uint8 arrays of one size): yields the same [Results] `track(frame)` would, in
        order, `batch` frames at a time through the overlapped two-stream pipeline (stateless stages of group k+1 run
        while the tracker consumes group k; the tracker reads its galleries once per group).
        keep_device_frames: every Results also carries `orig_img_device`, the frame as a device tensor (a device-to-device copy
        of the group's input buffer taken on the tracker's stream), so that an annotated output needs no second upload
        (`Overlay.draw_resident`); valid until the generator has yielded `ring` more groups."""
        from .pipeline import OverlappedPipeline
        it = iter(frames)
        first = next(it, None)
        if first is None:
            return
        key = self._state_key(first.shape[:2], device) + (batch,)
        if self._stream_pipe is None or key != self._stream_key:
            if self._stream_pipe is not None:
                self._stream_pipe.close()
            self._stream_pipe = self._build(OverlappedPipeline, first.shape[:2], device, graph="front", frame_batch=batch,
                                            reid_split=(5 if self.arch == "yolov8n" else 2) if batch > 1 else None, defer_track=batch > 1)     # stage cut: bench.REID_SPLIT's sweep
            self._stream_key = key
        pipe = self._stream_pipe
        pipe.on_result = None
        pipe.flush()                                                      # groups an abandoned generator left in flight: tracked, results dropped
        F, H, W = batch, first.shape[0], first.shape[1]
        ring = 5                                                          # result slots: groups in flight (<= 3) + margin
        h_rows = torch.empty(ring, F, pipe.outs.shape[2], 8).pin_memory()
        h_dets = torch.empty(ring, F, pipe.bufs[0].dets.shape[1], pipe.bufs[0].dets.shape[2]).pin_memory()
        h_cnt = torch.zeros(ring, 2, F, dtype=torch.int32).pin_memory()
        h_proto = torch.empty((ring, F) + tuple(pipe.bufs[0].proto.shape[1:]), dtype=pipe.bufs[0].proto.dtype).pin_memory() if pipe.nm else None
        d_frames = torch.empty((ring, F, H, W, 3), dtype=torch.uint8, device=pipe.dev) if keep_device_frames else None
        done = [torch.cuda.Event() for _ in range(ring)]
        pending = []                                                      # (group index, frames of the group)
        state = {"group": 0, "first": {}, "enqueued": -1}                 # first frame index of a group -> group index

        def on_result(frame_idx, f):                                      # runs while a tracker call is being enqueued
            g = state["first"].get(frame_idx - f)
            if g is None:                                                 # a group left over from an abandoned generator
                return
            b, nv = pipe.cur_bufs, pipe.cur_valid                         # the set this tracker call read (the pipeline's own index, not g)
            if f != nv - 1:
                return
            slot = g % ring
            h_rows[slot, :nv].copy_(pipe.outs[:nv, 0], non_blocking=True)
            h_cnt[slot, 1, :nv].copy_(pipe.nouts[:nv, 0], non_blocking=True)
            h_cnt[slot, 0, :nv].copy_(b.ndets[:nv], non_blocking=True)
            h_dets[slot, :nv].copy_(b.dets[:nv], non_blocking=True)
            if h_proto is not None:
                h_proto[slot, :nv].copy_(b.proto[:nv], non_blocking=True)
            if d_frames is not None:
                d_frames[slot, :nv].copy_(b.frames[:nv], non_blocking=True)       # the set's frames, before the set is refilled
            done[slot].record(torch.cuda.current_stream(pipe.dev))
            state["enqueued"] = g
            del state["first"][frame_idx - f]

        pipe.on_result = on_result

        def finish(g, imgs):
            slot = g % ring
            done[slot].synchronize()
            for f, img in enumerate(imgs):
                n, m = int(h_cnt[slot, 0, f]), int(h_cnt[slot, 1, f])
                self._warn_if_capped(pipe, n, True)
                res = self._results(img, pipe, h_dets[slot, f, :n].clone(), h_rows[slot, f, :m].clone(),
                                    None if h_proto is None else h_proto[slot, f].clone())
                if d_frames is not None:
                    res[0].orig_img_device = d_frames[slot, f]
                yield res

        try:
            chunk = [first]
            while chunk:
                while len(chunk) < F:
                    nxt = next(it, None)
                    if nxt is None:
                        break
                    chunk.append(nxt)
                g = state["group"]
                b = pipe.begin_frame()                                    # waits until this buffer set's last group left the tracker
                with torch.cuda.stream(pipe.s_in):
                    pipe.eng.upload_batch(b.frames, chunk, pipe.s_in)      # the group's frames: staged by several host threads, one copy
                    if self._fill is not None:
                        for f in range(len(chunk)):
                            self._fill(b, f, self._frame_index + f)
                self._frame_index += len(chunk)
                state["first"][pipe.frames_in] = g                       # submit() numbers the group's frames from here
                pipe.submit(len(chunk))
                pending.append((g, chunk))
                state["group"] = g + 1
                while pending and pending[0][0] <= state["enqueued"] - 1:  # its results are enqueued, and so is a group after it
                    yield from finish(*pending.pop(0))
                nxt = next(it, None)
                chunk = [nxt] if nxt is not None else []
            pipe.flush()
            while pending:
                yield from finish(*pending.pop(0))
            pipe.eng.check_errors()
        finally:
            pipe.on_result = None

    def overlay(self):
        """Annotation overlay bound to this model's device context (drawing of yolo_multi_model.py:58-162 as a kernel)."""
        from .overlay import Overlay
        pipe = self._stream_pipe or self._pipe
        if pipe is None:
            raise RuntimeError("overlay(): run track()/predict()/track_stream() once first (the device context is created there)")
        return Overlay(self.names, pipe.eng)

    def close(self):
        for p in (self._pipe, self._stream_pipe, self._pred_pipe):
            if p is not None:
                p.close()
        self._pipe = self._stream_pipe = self._pred_pipe = None
