"""The fp32 OSNet (own kernels) eagerly, a few passes over N crops: the command the PMC passes / kernel traces run.
usage: python tools/osnet32_eager.py [passes=3] [crops=1024]; SS32_CHAINS_FORM / SS32_CHAINS_PRE as tools/osnet32_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused32
for k, o in (("SS32_CHAINS_FORM", "chains_form"), ("SS32_CHAINS_PRE", "chains_pre"), ("SS32_CHAINS_PROBE", "chains_probe")):
    if k in os.environ:
        fused32.set_option(o, int(os.environ[k]))
for kv in os.environ.get("SS32_OPTS", "").split(","):
    if "=" in kv:
        fused32.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
net = nets.build_reid().to(dev).to(memory_format=torch.channels_last)
x = torch.randn(N, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(P):
        y = net(x)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
