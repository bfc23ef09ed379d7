"""OSNet forward time at the pipeline's batch sizes, fused LightConv on/off (HIP-graph replay)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused
dev = torch.device("cuda", 0)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
reid = nets.build_reid().to(dev, torch.float16).to(memory_format=torch.channels_last)
for b in (32, 128, 256):
    c = torch.randn(b, 3, 256, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    for lc in (False, True):
        fused.LIGHTCONV = lc
        with torch.no_grad():
            t = timeit(lambda: reid(c))
        print(f"osnet b{b} lightconv={lc}: {t:.3f} ms", flush=True)
