"""StrongSORT.update(dets, frame) — the tracker seam BASELINE.json's north_star names.

Stands behind the tracker callback inside `model.track(...)` (/root/reference/yolo_multi_model.py:41):
detections of one frame in, rows of confirmed tracks out.  All arithmetic runs in
libstrongsort_hip.so (crop-extract, feature normalise, Kalman, cost matrix, LSAP, bookkeeping) and in
the PyTorch-ROCm OSNet; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import nets
from .config import StrongSortConfig
from .engine import TrackerEngine
from .lib import MAX_DETS, FEAT_DIM


class StrongSORT:
    """One video stream.  `update(dets, frame)`:
         dets  [N,6] float  x1,y1,x2,y2,conf,cls in frame pixels (numpy or torch)
         frame uint8 [H,W,3] BGR (numpy, or a torch tensor already on the device)
       returns float32 [M,8]: x1,y1,x2,y2,track_id,class_id,conf,det_idx for every confirmed track seen
       within the last frame (det_idx = row of `dets` matched this frame, -1 while coasting)."""

    def __init__(self, model_weights: Optional[str] = None, device: int = 0, fp16: bool = True,
                 max_dist: float = 0.2, max_iou_distance: float = 0.7, max_age: int = 30, n_init: int = 3,
                 nn_budget: int = 100, mc_lambda: float = 0.995, ema_alpha: float = 0.9, reid_seed: int = 1,
                 random_init_ok: bool = False):
        self.cfg = StrongSortConfig(max_dist=max_dist, max_iou_distance=max_iou_distance, max_age=max_age,
                                    n_init=n_init, nn_budget=nn_budget, mc_lambda=mc_lambda, ema_alpha=ema_alpha)
        self.eng = TrackerEngine(self.cfg, 1, device)
        self.dev = self.eng.device
        self.dtype = torch.float16 if fp16 else torch.float32
        self.reid = nets.build_reid(reid_seed)
        nets.load_weights(self.reid, model_weights, "OSNet-x0.25 ReID", random_init_ok)
        self.reid = self.reid.to(self.dev, self.dtype).to(memory_format=torch.channels_last)
        self._dets = torch.zeros(1, MAX_DETS, 6, dtype=torch.float32, device=self.dev)
        self._feats = torch.zeros(1, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=self.dev)
        self._n = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._hw = torch.zeros(1, 2, dtype=torch.int32, device=self.dev)

    @torch.no_grad()
    def update(self, dets, frame, features=None) -> np.ndarray:
        self.eng.use_current_stream()                    # launch on the caller's current torch stream
        dets = torch.as_tensor(dets, dtype=torch.float32).reshape(-1, 6)
        n = dets.shape[0]
        if n > MAX_DETS:
            raise ValueError(f"at most {MAX_DETS} detections per frame (got {n})")
        frame_t = frame if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame))
        frame_t = frame_t.to(self.dev, non_blocking=True)
        H, W = int(frame_t.shape[0]), int(frame_t.shape[1])
        self._dets[0, :n].copy_(dets, non_blocking=True)
        self._n.fill_(n)
        self._hw.copy_(torch.tensor([[H, W]], dtype=torch.int32))
        if features is not None:
            self._feats[0, :n].copy_(torch.as_tensor(features, dtype=torch.float32).reshape(n, FEAT_DIM))
        elif n:
            crops = self.eng.crop_norm(frame_t, self._dets[0], n, half=self.dtype == torch.float16)
            self._feats[0, :n].copy_(self.reid(crops.contiguous(memory_format=torch.channels_last)))
        out, nout = self.eng.update_device(self._dets, self._n, self._feats, self._hw)
        torch.cuda.synchronize(self.dev)
        self.eng.check_errors()
        return out[0, : int(nout[0])].cpu().numpy()

    def reset(self):
        self.eng.reset(-1)

    def close(self):
        self.eng.close()
