"""Frozen constants of the StrongSORT hot path (shared by the product and, by value, the oracle).

Every number here is a *decision* recorded in oracle/DECISIONS.md: the reference snapshot pins only
conf/iou/agnostic_nms/max_det (/root/reference/yolo_multi_model.py:18-21); everything else follows
BASELINE.json's north_star + SURVEY.md Appendix A.1 (upstream recall, unverified).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

# Layout constants compiled into the HIP library (csrc/ss_common.h mirrors these).
FEAT_DIM = 512          # OSNet embedding width (F = 512)
SEG_LEN = 64            # dot-product segment length: 8 segments of 64, each an fmaf chain
ROW_TILE = 32           # gallery rows / detection columns per MFMA 32x32 tile
CROP_H, CROP_W = 256, 128

TENTATIVE, CONFIRMED, DELETED = 1, 2, 3


@dataclass(frozen=True)
class StrongSortConfig:
    # association
    max_dist: float = 0.2            # cosine matching threshold (appearance stage)
    max_iou_distance: float = 0.7    # IoU-stage threshold on 1-IoU
    max_age: int = 30
    n_init: int = 3
    nn_budget: int = 100             # gallery rows kept per track
    mc_lambda: float = 0.995         # appearance/motion blend
    ema_alpha: float = 0.9
    gating_threshold: float = 9.4877  # chi2inv95[4]
    gated_cost: float = 1e5           # INFTY_COST
    # Kalman (DeepSORT xyah, h-scaled)
    std_weight_position: float = 1.0 / 20
    std_weight_velocity: float = 1.0 / 160
    # capacities of the device-resident track table (per stream)
    max_tracks: int = 256
    max_dets: int = 128

    def as_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class DetectConfig:
    """The four overrides the reference pins (yolo_multi_model.py:18-21) + letterbox geometry."""
    conf: float = 0.3
    iou: float = 0.4
    agnostic_nms: bool = False
    max_det: int = 1000
    imgsz: int = 640
    stride: int = 32
    pad_value: int = 114
    max_wh: float = 7680.0            # per-class box offset (non-agnostic NMS)
    max_nms: int = 8192               # candidate cap after the confidence filter (LDS sort capacity)
