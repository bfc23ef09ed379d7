"""`YOLO(weights)` — the model object the reference script drives.

Mirrors exactly what /root/reference/yolo_multi_model.py touches:
  :17     model = YOLO("yolo11n-pose.pt")
  :18-21  model.overrides['conf'|'iou'|'agnostic_nms'|'max_det'] (+ optional 'classes' :22)
  :23     model.names
  :41     model.track(image, verbose=False, device=0, persist=True, tracker="botsort.yaml") -> [Results]
  :173    model.predict(image, verbose=False, device=0)                                       -> [Results]
and the Results duck type consumed at :45-169 / :175-237 (boxes.id/.conf/.cls/.xyxy as 1-row tensors
per box, keypoints[i].xy, masks, names).  The frame -> rows path runs on the MI355X
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


class Boxes:
    """Rows of [x1,y1,x2,y2,(id),conf,cls]; iterating yields 1-row Boxes (yolo_multi_model.py:73,126)."""

    def __init__(self, xyxy, conf, cls, ids=None):
        self.xyxy, self.conf, self.cls, self.id = xyxy, conf, cls, ids

    def __len__(self):
        return self.xyxy.shape[0]

    def __getitem__(self, i):
        i = slice(i, i + 1) if isinstance(i, int) else i
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
        i = slice(i, i + 1) if isinstance(i, int) else i
        return Keypoints(self.data[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Results:
    def __init__(self, orig_img, names, boxes: Optional[Boxes], keypoints: Optional[Keypoints] = None):
        self.orig_img, self.names, self.boxes, self.keypoints, self.masks = orig_img, names, boxes, keypoints, None

    def __len__(self):
        return 0 if self.boxes is None else len(self.boxes)


class YOLO:
    def __init__(self, weights: str = "yolov8n.pt", seed: int = 0, random_init_ok: bool = False):
        self.weights = weights
        self.random_init_ok = random_init_ok
        self.arch = os.path.basename(weights).replace(".pt", "")
        self.overrides = {"conf": 0.25, "iou": 0.7, "agnostic_nms": False, "max_det": 300}
        self.seed = seed
        pose = "pose" in self.arch
        self.names = {0: "person"} if pose else dict(enumerate(COCO_NAMES))
        self._pipe = None
        self._shape = None

    def _dcfg(self):
        o = self.overrides
        return DetectConfig(conf=float(o["conf"]), iou=float(o["iou"]), agnostic_nms=bool(o["agnostic_nms"]),
                            max_det=int(o["max_det"]))

    def _pipeline(self, image, device):
        from .pipeline import FramePipeline
        shape = image.shape[:2]
        if self._pipe is None or self._shape != shape:
            if self._pipe is not None:
                self._pipe.close()
            self._pipe = FramePipeline(self.arch, 1, shape, device=int(device or 0), reid_batch=128, cfg=StrongSortConfig(),
                                       dcfg=self._dcfg(), det_source="detector", feat_source="reid", graph="none",
                                       seed=self.seed)
            from . import nets
            nets.load_weights(self._pipe.detector, self.weights, f"detector {self.arch}", self.random_init_ok)
            self._shape = shape
        self._pipe.dcfg = self._dcfg()
        return self._pipe

    def _run(self, image, device, track):
        pipe = self._pipeline(image, device)
        pipe.frames[0].copy_(torch.from_numpy(np.ascontiguousarray(image)))
        pipe.step(track=track)
        torch.cuda.synchronize(pipe.dev)
        n = int(pipe.ndets[0])
        dets = pipe.dets[0, :n].cpu()
        classes = self.overrides.get("classes")
        kpts = None
        if pipe.nk:
            k = dets[:, 6:].reshape(n, pipe.nk // 3, 3).clone()
            k[..., 0] = (k[..., 0] - pipe.pad_x) / pipe.gain
            k[..., 1] = (k[..., 1] - pipe.pad_y) / pipe.gain
            kpts = k
        if not track:
            keep = torch.ones(n, dtype=torch.bool)
            if classes is not None:
                cl = classes if isinstance(classes, (list, tuple)) else [classes]
                keep = torch.isin(dets[:, 5].long(), torch.tensor(cl))
            return [Results(image, self.names, Boxes(dets[keep, :4], dets[keep, 4], dets[keep, 5]),
                            None if kpts is None else Keypoints(kpts[keep]))]
        pipe.eng.check_errors()
        rows = pipe.out[0, : int(pipe.nout[0])].cpu()
        rows = rows[rows[:, 7] >= 0]                   # ultralytics semantics: results[i] = results[i][det_idx]
        if rows.shape[0] == 0:
            return [Results(image, self.names, Boxes(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0), None))]
        di = rows[:, 7].long()
        return [Results(image, self.names, Boxes(rows[:, :4], rows[:, 6], rows[:, 5], rows[:, 4]),
                        None if kpts is None else Keypoints(kpts[di]))]

    @torch.no_grad()
    def track(self, image, verbose=False, device=0, persist=True, tracker="strongsort.yaml", **kw) -> List[Results]:
        if not persist and self._pipe is not None:
            self._pipe.eng.reset(-1)
        return self._run(image, device, True)

    @torch.no_grad()
    def predict(self, image, verbose=False, device=0, **kw) -> List[Results]:
        return self._run(image, device, False)

    __call__ = predict
