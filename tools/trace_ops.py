"""Which Python lines of nets.py / fused.py launch which device kernels in one detector forward (torch profiler)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "yolov8n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = (nets.build_reid() if name == "osnet" else nets.build_detector(name)).to(dev, torch.float16).to(memory_format=torch.channels_last)
shape = (B, 3, 256, 128) if name == "osnet" else (B, 3, 384, 640)
x = torch.randn(*shape, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        m(x); torch.cuda.synchronize()
by = collections.Counter(); tm = collections.Counter()
for ev in prof.events():
    if not ev.stack or not ev.name.startswith("aten::"): continue
    if any(c.name.startswith("aten::") for c in ev.cpu_children): continue     # leaf aten ops only
    line = next((s for s in ev.stack if "strongsort_yolo_amd" in s), "?")
    line = line.split("strongsort_yolo_amd/")[-1]
    by[(ev.name, line)] += 1; tm[(ev.name, line)] += max(getattr(ev, 'device_time_total', 0), getattr(ev, 'cuda_time_total', 0), 0)
tot = sum(tm.values())
print(f"{name} b{B}: device time {tot/1e3:.3f} ms in leaf ops")
for k, n in by.most_common(45):
    print(f"{tm[k]:9.1f} us {n:4d}x  {k[0]:34s} {k[1]}")
