"""OSNet-x0.25 in fp32 on N crops (default 1024 = 32 frames x 32) as one replayed HIP graph: the hand-written fp32 kernels
(csrc/ss_ops32.hip), the library convolutions (fused32.ENABLED = False), and the f16 kernels beside them.
usage: python tools/osnet32_time.py [reps=20] [crops=1024] [eager_pass=0|1: one eager pass at the end for a kernel trace]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused32
if "SS32_CHAINS_PRE" in os.environ:
    fused32.set_option("chains_pre", int(os.environ["SS32_CHAINS_PRE"]))
if "SS32_CHAINS_FORM" in os.environ:
    fused32.set_option("chains_form", int(os.environ["SS32_CHAINS_FORM"]))
for k, o in (("SS32_CHAINS_PROBE", "chains_probe"),):
    if k in os.environ:
        fused32.set_option(o, int(os.environ[k]))
for kv in os.environ.get("SS32_OPTS", "").split(","):
    if "=" in kv:
        fused32.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eager = len(sys.argv) > 3 and sys.argv[3] == "1"
r32 = nets.build_reid().to(dev).to(memory_format=torch.channels_last)
r16 = nets.build_reid().to(dev, torch.float16).to(memory_format=torch.channels_last)
c32 = torch.randn(N, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)
c16 = c32.half().contiguous(memory_format=torch.channels_last)


def timed(fn):
    s = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(s):
        for _ in range(2):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"crops": N, "chains_form": os.environ.get("SS32_CHAINS_FORM", "2"), "chains_pre": os.environ.get("SS32_CHAINS_PRE", "1"), "osnet_fp32_own_kernels_ms": round(timed(lambda: r32(c32)), 4)}
fused32.ENABLED = False
out["osnet_fp32_library_ms"] = round(timed(lambda: r32(c32)), 4)
fused32.ENABLED = True
out["osnet_f16_own_kernels_ms"] = round(timed(lambda: r16(c16)), 4)
print(json.dumps(out))
if eager:
    with torch.no_grad():
        r32(c32)
    torch.cuda.synchronize()
