import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets
dev = torch.device("cuda", 0)
def timeit(fn, n=50):
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
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for cl in (True, False):
        mf = torch.channels_last if cl else torch.contiguous_format
        det = nets.build_detector("yolov8n").to(dev, torch.float16).to(memory_format=mf)
        reid = nets.build_reid().to(dev, torch.float16).to(memory_format=mf)
        x = torch.randn(1, 3, 384, 640, device=dev, dtype=torch.float16).contiguous(memory_format=mf)
        c = torch.randn(32, 3, 256, 128, device=dev, dtype=torch.float16).contiguous(memory_format=mf)
        with torch.no_grad():
            td = timeit(lambda: det(x)); tr = timeit(lambda: reid(c))
        print(f"benchmark={bench} channels_last={cl}: yolov8n b1 {td:.3f} ms, osnet b32 {tr:.3f} ms", flush=True)
