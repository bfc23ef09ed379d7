"""Detector (batch 16 frames, 640x384) + OSNet (512 crops) forward, eager launches (rocprofv3 PMC passes cannot follow
HIP-graph replays here).  usage: python tools/nets_eager.py [iters=4] [frames=16] [detector=yolov8n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
det = nets.build_detector(sys.argv[3] if len(sys.argv) > 3 else "yolov8n").to(dev, torch.float16).to(memory_format=torch.channels_last)
reid = nets.build_reid().to(dev, torch.float16).to(memory_format=torch.channels_last)
x = torch.randn(F, 3, 384, 640, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
c = torch.randn(32 * F, 3, 256, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(n):
        det(x); reid(c)
torch.cuda.synchronize()
print("done")
