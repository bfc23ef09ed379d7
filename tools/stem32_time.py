"""k32_stem (one band per workgroup) against k32_stemW (a workgroup walks 1 .. 16 bands) on N crops: replayed-graph time per launch.
usage: python tools/stem32_time.py [crops=1024,430,860,28] [reps=30]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused32
dev = torch.device("cuda", 0)
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024,430,860,28").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
m = nets.build_reid().to(dev).to(memory_format=torch.channels_last)
stem = m.conv1
for N in Ns:
    x = torch.randn(N, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)
    row = {"crops": N}
    for walk in (-1, 1, 2, 4, 8, 16, 0):
        fused32.set_option("stem_walk", walk)
        s = torch.cuda.Stream()
        with torch.no_grad(), torch.cuda.stream(s):
            for _ in range(2):
                fused32.stem(x, stem)
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(4):
                    fused32.stem(x, stem)
            g.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                g.replay()
            e1.record(s); s.synchronize()
        row[{-1: "band_form_us", 0: "auto_us"}.get(walk, f"walk{walk}_us")] = round(e0.elapsed_time(e1) * 1e3 / reps / 4, 1)
    fused32.set_option("stem_walk", 0)
    xb = torch.randint(0, 256, (N, 3, 256, 128), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
    s = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(s):
        for _ in range(2):
            fused32.stem(xb, stem)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                fused32.stem(xb, stem)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); s.synchronize()
    row["byte_crops_auto_us"] = round(e0.elapsed_time(e1) * 1e3 / reps / 4, 1)
    print(json.dumps(row))
