"""Packed ReID crops (a4) of 32 frames x ~27 boxes: microseconds per launch pair (offsets + crops) by HIP events, for small boxes (rows staged
through LDS) and large ones (the global-load fallback).  usage: python tools/crop_time.py [reps=50]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.engine import TrackerEngine
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = TrackerEngine(StrongSortConfig(), 1, 0)
B, n, H, W = 32, 32, 720, 1280
g = torch.Generator().manual_seed(1)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g).to(dev)
counts = torch.randint(24, 31, (B,), dtype=torch.int32, generator=g).to(dev)
offs = torch.zeros(B + 1, dtype=torch.int32, device=dev)
res = {}
for name, (bw, bh) in {"boxes 60x140": (60, 140), "boxes 110x260": (110, 260), "boxes 220x480 (fallback)": (220, 480)}.items():
    x1 = torch.rand(B, n, generator=g) * (W - bw - 2); y1 = torch.rand(B, n, generator=g) * (H - bh - 2)
    w = bw * (0.7 + 0.6 * torch.rand(B, n, generator=g)); h = bh * (0.7 + 0.6 * torch.rand(B, n, generator=g))
    dets = torch.stack([x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1), torch.ones(B, n), torch.zeros(B, n)], 2).to(dev).contiguous()
    for half in (True, False, "u8"):
        out = torch.empty(B * n, 3, 256, 128, dtype=torch.uint8 if half == "u8" else torch.float16 if half else torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
        s = torch.cuda.current_stream(dev)
        for _ in range(3):
            eng.crop_norm_packed(frames, dets, n, counts, offs, out, half=half is True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            eng.crop_norm_packed(frames, dets, n, counts, offs, out, half=half is True)
        e1.record(s); s.synchronize()
        res[f"{name} {'u8' if half == 'u8' else 'f16' if half else 'f32'}"] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
        res[f"{name} {'u8' if half == 'u8' else 'f16' if half else 'f32'} checksum"] = float(out[: int(offs[B])].float().sum())
print({"crops": int(offs[B]), "us": res})
