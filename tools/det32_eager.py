"""The detector in fp32, eagerly, a few passes over F letterboxed frames: the command kernel traces / PMC passes of the fp32 detector run.
usage: python tools/det32_eager.py [passes=3] [frames=32] [detector=yolov8n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused32
if os.environ.get("SS32_DET") == "0":
    fused32.DET = False
for kv in os.environ.get("SS32_OPTS", "").split(","):
    if "=" in kv:
        fused32.set_option(kv.split("=")[0], int(kv.split("=")[1]))
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name = sys.argv[3] if len(sys.argv) > 3 else "yolov8n"
dev = torch.device("cuda", 0)
net = nets.build_detector(name).to(dev).to(memory_format=torch.channels_last).eval()
x = torch.rand(F, 3, 384, 640, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    y = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(P):
        y = net(x)
    torch.cuda.synchronize()
print("ok", tuple((y[0] if isinstance(y, tuple) else y).shape), "own fp32 kernels:", bool(getattr(net, "_own32", False)), f"{(time.perf_counter() - t0) / P * 1e3:.3f} ms per pass (eager)")
