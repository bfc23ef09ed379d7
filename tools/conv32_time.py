"""k32_conv on the convolution shapes of yolov8n at 32 frames of 384x640: microseconds and TFLOP/s per layer shape by HIP events (graph replay of 20 launches).
usage: python tools/conv32_time.py [name=value ...]   (ss_op32_set_option switches: conv_waves, conv_mt, conv_min)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd import nets, fused32
for kv in sys.argv[1:]:
    fused32.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
B = 32
# (name, k, s, cin, cout, h, w) input sizes
SHAPES = [("b1 3x3/2 16->32", 3, 2, 16, 32, 192, 320), ("b2.m 3x3 16->16", 3, 1, 16, 16, 96, 160), ("b2.cv2 1x1 48->32", 1, 1, 48, 32, 96, 160),
          ("b3 3x3/2 32->64", 3, 2, 32, 64, 96, 160), ("b4.m 3x3 32->32", 3, 1, 32, 32, 48, 80), ("b4.cv2 1x1 128->64", 1, 1, 128, 64, 48, 80),
          ("b5 3x3/2 64->128", 3, 2, 64, 128, 48, 80), ("b6.m 3x3 64->64", 3, 1, 64, 64, 24, 40), ("b7 3x3/2 128->256", 3, 2, 128, 256, 24, 40),
          ("b8.m 3x3 128->128", 3, 1, 128, 128, 12, 20), ("b9.cv2 1x1 512->256", 1, 1, 512, 256, 12, 20), ("h12.cv1 1x1 384->128", 1, 1, 384, 128, 24, 40),
          ("h15.cv1 1x1 192->64", 1, 1, 192, 64, 48, 80), ("det.cv2[0] 3x3 64->64 P3", 3, 1, 64, 64, 48, 80), ("det.cv3[0] 3x3 64->80 P3", 3, 1, 64, 80, 48, 80),
          ("det.cv3[0] 3x3 80->80 P3", 3, 1, 80, 80, 48, 80), ("det.cv3[1] 3x3 128->80 P4", 3, 1, 128, 80, 24, 40), ("det.cv3[2] 3x3 256->80 P5", 3, 1, 256, 80, 12, 20)]
tot_us = tot_fl = 0.0
for name, k, s, ci, co, h, w in SHAPES:
    m = nets.Conv(ci, co, k, s).to(dev)
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    st = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(3):
            y = m(x)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                y = m(x)
        g.replay(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); g.replay(); e1.record(st); st.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * B * y.shape[2] * y.shape[3] * co * ci * k * k
    tot_us += us; tot_fl += fl
    print(f"{name:30s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  ({fl / 1e9:.2f} GFLOP)")
print(f"sum {tot_us:.0f} us, {tot_fl / tot_us / 1e6:.1f} TFLOP/s")
