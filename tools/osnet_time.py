"""OSNet-x0.25 forward on 512 crops (16 frames x 32) and the detector on 16 frames, each as one replayed HIP graph:
milliseconds per replay.  A/B switches: SS_FUSED_<NAME>=0 (fused.py), SS_OP_OPTS=pw_splitk=0,... (ss_op_set_option).  usage: python tools/osnet_time.py [reps=20] [frames=16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused
fused.flags_from_env()                                        # SS_FUSED_<NAME>=0|1, SS_BNECK_C=...
for kv in os.environ.get("SS_OP_OPTS", "").split(","):          # e.g. SS_OP_OPTS=pw_splitk=0,osnet_chains=0
    if kv:
        fused.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
det = nets.build_detector("yolov8n").to(dev, torch.float16).to(memory_format=torch.channels_last)
reid = nets.build_reid().to(dev, torch.float16).to(memory_format=torch.channels_last)
x = torch.randn(F, 3, 384, 640, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
c = torch.randn(32 * F, 3, 256, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)


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


print({"frames": F, "osnet_ms": round(timed(lambda: reid(c)), 4), "detector_ms": round(timed(lambda: det(x)), 4),
       "flags": {k: v for k, v in os.environ.items() if k.startswith("SS_FUSED_")}})
