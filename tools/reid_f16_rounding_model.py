"""How much of the f16 ReID error is the FORM of the rounding?  CPU model of OSNet-x0.25 on calibrated weights (bench.calibrate_reid_)
and rendered crops, fp32 arithmetic with explicit roundings to half:
  fp32        no rounding (the oracle's network)
  once        every STORED tensor rounded once: conv + bias + ReLU (+ shortcut) in fp32 then one rounding; a LightConv layer (1x1 ->
              depthwise + bias + ReLU) as one fused unit; the gate-weighted sum, pools and the head likewise
  unfused     torch-f16's rounding points: the convolution rounded before the bias / activation pass, the 1x1 of a LightConv rounded
              before its depthwise, every elementwise step rounded (what the fused HIP kernels reproduce bit for bit)
  hybrid      f16 storage ('once') for the stem and the first stage only, fp32 from the second stage on
Prints the max abs error of the unit embeddings and of the pairwise cosine distances against fp32.  usage: python tools/reid_f16_rounding_model.py [crops=96]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import bench
from oracle import cexact
from strongsort_yolo_amd import nets
from strongsort_yolo_amd.synth import make_stream

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
torch.set_num_threads(min(os.cpu_count() or 1, 16))
cs = make_stream(2023, 1280, 720, 30)
cfr = [cs.next_frame() for _ in range(2)]
net = bench.calibrate_reid_(nets.build_reid(1).float(), torch.from_numpy(np.concatenate([cexact.crop_norm(cs.render(f), f.dets) for f in cfr])))
st = make_stream(2024, 1280, 720, 30)
crops = []
while sum(len(c) for c in crops) < N:
    f = st.next_frame()
    crops.append(cexact.crop_norm(st.render(f), f.dets))
x = torch.from_numpy(np.concatenate(crops)[:N])
h = lambda t: t.half().float()


def convbr(m, x, mode, res=None):
    w, b = (h(m.conv.weight), h(m.conv.bias)) if mode != "fp32" else (m.conv.weight, m.conv.bias)
    y = F.conv2d(x, w, None, m.conv.stride, m.conv.padding)
    if mode == "unfused":
        y = h(y)
    y = y + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if m.relu:
        y = F.relu(y)
    return h(y) if mode != "fp32" else y


def light(m, x, mode):
    wp, wd, bd = (h(m.pw.weight), h(m.dw.weight), h(m.dw.bias)) if mode != "fp32" else (m.pw.weight, m.dw.weight, m.dw.bias)
    y = F.conv2d(x, wp)
    if mode == "unfused":
        y = h(y)
    y = F.relu(F.conv2d(y, wd, bd, 1, 1, groups=wd.shape[0]))
    return h(y) if mode != "fp32" else y


def block(b, x, mode):
    r = (lambda t: t) if mode == "fp32" else h
    idn = x if b.down is None else convbr(b.down, x, mode)
    x1 = convbr(b.conv1, x, mode)
    g = b.gate
    w1, b1, w2, b2 = [(t if mode == "fp32" else h(t)) for t in (g.fc1.weight, g.fc1.bias, g.fc2.weight, g.fc2.bias)]
    x2 = 0
    for s in b.streams:
        y = x1
        for m in s:
            y = light(m, y, mode)
        gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(y.mean((2, 3), keepdim=True), w1, b1)), w2, b2))
        if mode == "unfused":
            x2 = r(x2 + r(y * r(gate)))
        else:
            x2 = x2 + y * gate
    x2 = r(x2)
    if mode == "unfused":                                   # conv rounded, + bias rounded, + shortcut and ReLU rounded
        return r(F.relu(convbr(b.conv3, x2, mode) + idn))
    w3, b3 = (b.conv3.conv.weight, b.conv3.conv.bias) if mode == "fp32" else (h(b.conv3.conv.weight), h(b.conv3.conv.bias))
    return r(F.relu(F.conv2d(x2, w3) + b3.view(1, -1, 1, 1) + idn))


def forward(net, x, mode):
    if mode == "hybrid":                                    # f16 storage (one rounding per tensor) for the stem and the 64 x 32 stage — 60 % of the
        x = h(x)                                            # network's time — fp32 from the 32 x 16 stage on
        x = h(F.max_pool2d(convbr(net.conv1, x, "once"), 3, 2, 1))
        x = block(net.conv2[0], x, "once"); x = block(net.conv2[1], x, "once")
        x = h(F.avg_pool2d(convbr(net.conv2[2], x, "once"), 2, 2))
        x = block(net.conv3[0], x, "fp32"); x = block(net.conv3[1], x, "fp32")
        x = F.avg_pool2d(convbr(net.conv3[2], x, "fp32"), 2, 2)
        x = block(net.conv4[0], x, "fp32"); x = block(net.conv4[1], x, "fp32")
        x = convbr(net.conv5, x, "fp32")
        return F.relu(F.linear(x.mean((2, 3)), net.fc.weight, net.fc.bias))
    r = (lambda t: t) if mode == "fp32" else h
    x = r(x)
    x = r(F.max_pool2d(convbr(net.conv1, x, mode), 3, 2, 1))
    for stage in (net.conv2, net.conv3):
        x = block(stage[0], x, mode); x = block(stage[1], x, mode)
        x = r(F.avg_pool2d(convbr(stage[2], x, mode), 2, 2))
    x = block(net.conv4[0], x, mode); x = block(net.conv4[1], x, mode)
    x = convbr(net.conv5, x, mode)
    w, b = (net.fc.weight, net.fc.bias) if mode == "fp32" else (h(net.fc.weight), h(net.fc.bias))
    return r(F.relu(F.linear(r(x.mean((2, 3))), w, b)))


with torch.no_grad():
    ref = forward(net, x, "fp32")
    assert torch.allclose(ref, net(x), atol=1e-4, rtol=1e-4)          # the model below IS the module
    u = lambda e: (e / e.norm(dim=1, keepdim=True)).double()
    out = {"crops": N}
    for mode in ("once", "unfused", "hybrid"):
        e = forward(net, x, mode)
        out[mode] = {"embedding_unit_max_abs_err": round(float((u(e) - u(ref)).abs().max()), 5),
                     "cosine_distance_max_abs_err": round(float(((1 - u(e) @ u(e).T) - (1 - u(ref) @ u(ref).T)).abs().max()), 5)}
print(out)
