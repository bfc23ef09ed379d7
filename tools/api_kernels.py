"""Kernel time of ONE per-frame YOLO.track call (run under rocprofv3 --kernel-trace --stats; tools/kstats.py <stats.csv> <calls> prints per-call sums).
usage: python tools/api_kernels.py [calls=50] [fp32|f16|all32]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
import bench
from strongsort_yolo_amd.yolo import YOLO
from strongsort_yolo_amd.synth import make_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
m = YOLO("yolov8n.pt", random_init_ok=True, reid_batch=32, reid_fp32=mode != "f16", half=mode != "all32")
m.overrides.update(conf=0.25, iou=0.45, agnostic_nms=False, max_det=1000)
st = make_stream(2025, 1280, 720, 30)
imgs = [st.render(st.next_frame()) for _ in range(8)]
for i in range(n):
    m.track(imgs[i % 8])
torch.cuda.synchronize()
