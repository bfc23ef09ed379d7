"""Row a2 (detector forward) held to the fp32 oracle: the reference calls model.track / model.predict with no half=
(/root/reference/yolo_multi_model.py:18-21, :41, :173), so its detector arithmetic is fp32 and north_star wants the box indices
bit-exact for the same input frames.

* `FramePipeline(det_source="detector", half=False)` — HIP letterbox -> the detector in fp32 on the GPU -> HIP NMS — gives the keep
  list of the CPU fp32 network + the oracle's NMS on every one of 48 frames at configs[1] (1280x720, yolov8n), and rows that differ
  only by the convolutions' summation order (bounds written below).
* The f16 detector kernels (the throughput mode) do NOT reproduce those keep lists; how far they are off is pinned here the way
  tests/test_gpu_nets32.py pins the f16 ReID error, so a change that makes it worse fails and nobody can read the f16 mode as parity.

The network is bench.calibrated_detector's: seeded weights calibrated to unit-variance layer outputs, class biases shifted so that
~40 anchors per frame pass conf (no checkpoint exists offline) — the density of near-threshold scores is synthetic and the harshest
place for a rounding difference."""
import numpy as np
import pytest
import torch

from oracle import cexact
from strongsort_yolo_amd.config import DetectConfig
from strongsort_yolo_amd.synth import make_stream

pytestmark = pytest.mark.gpu

W, H, N_IDS, FRAMES = 1280, 720, 30, 48


@pytest.fixture(scope="module")
def runs():
    """48 rendered frames through the CPU fp32 network + oracle NMS, the GPU fp32 pipeline and the GPU f16 pipeline (same weights)."""
    import bench
    from strongsort_yolo_amd import fused
    from strongsort_yolo_amd.pipeline import FramePipeline
    dcfg = DetectConfig()
    kw = dict(device=0, reid_batch=32, dcfg=dcfg, det_source="detector", feat_source="injected", graph="none", seed=0)
    p16 = FramePipeline("yolov8n", 1, (H, W), half=True, **kw)
    p32 = FramePipeline("yolov8n", 1, (H, W), half=False, **kw)
    torch.set_num_threads(8)
    det32, shift = bench.calibrated_detector("yolov8n", W, H, N_IDS, dcfg, p32, target=40)
    for p in (p16, p32):
        p.detector.load_state_dict(det32.state_dict())
        fused.clear_prepared(p.detector)
    st = make_stream(2025, W, H, N_IDS)
    out = {"cpu": [], "g32": [], "g16": []}
    gs = (p32.gain, p32.pad_x, p32.pad_y)
    with torch.no_grad():
        for k in range(FRAMES):
            img = torch.from_numpy(st.render(st.next_frame())).to(p32.dev)
            for name, p in (("g16", p16), ("g32", p32)):
                p.frames[0].copy_(img)
                p.step(track=False)
                torch.cuda.synchronize(p.dev)
                n = int(p.ndets[0].item())
                out[name].append((p.keep[0, :n].cpu().numpy().copy(), p.dets[0, :n, :6].cpu().numpy().copy()))
            pc = det32(p32.lb.float().cpu().contiguous())          # the same letterboxed pixels (the letterbox is bit-exact: test_gpu_front.py)
            pc = (pc[0] if isinstance(pc, tuple) else pc)[0, :4 + p32.nc].numpy()
            kc, rc = cexact.nms(pc, p32.nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, p32.max_det)
            out["cpu"].append((np.asarray(kc), cexact.scale_boxes(rc, gs[0], gs[1], gs[2], W, H)))
    out["own32"] = bool(getattr(p32.detector, "_own32", False))
    p16.close(); p32.close()
    return out


def test_fp32_detector_gives_the_cpu_fp32_keep_lists_on_every_frame(runs):
    """Keep indices identical (integers: exact) on all 48 frames; rows within 0.05 px (boxes up to 1 280 px: 4e-5 relative; measured
    7e-3 px) / 1e-4 in score of the CPU network's — the GPU convolutions sum in another order, ~1e-6 relative per layer, and the DFL
    decode of this random network amplifies it.  (The f16 kernels: 2.4 px median / 12.7 px p95, test below.)"""
    kept = 0
    for k, ((kc, rc), (kg, rg)) in enumerate(zip(runs["cpu"], runs["g32"])):
        assert len(kc) == len(kg) and (kc == kg).all(), f"frame {k}: keep list differs ({len(kc)} vs {len(kg)})"
        assert rc.shape == rg.shape
        if len(kc):
            assert np.abs(rc[:, :4] - rg[:, :4]).max() <= 5e-2, f"frame {k}: box"
            assert np.abs(rc[:, 4] - rg[:, 4]).max() <= 1e-4, f"frame {k}: score"
            assert (rc[:, 5] == rg[:, 5]).all(), f"frame {k}: class"
        kept += len(kc)
    assert kept >= 20 * FRAMES, "the calibrated network should keep tens of anchors per frame"


def test_f16_detector_disagreement_is_bounded_and_recorded(runs):
    """The f16 kernels against the fp32 GPU side (= the CPU side by the test above).  Measured in round 5 on this network: identical
    ordered list on 3 of 48 frames, 79 of 1 578 kept anchors in one list only, common anchors 2.4 px (median) / 12.7 px (p95) apart,
    scores 0.009 / 0.035 apart.  Pinned with margin; the assertion that the lists are NOT generally identical keeps the f16 mode from
    being reported as index parity."""
    same = sym = total = 0
    dbox, dconf = [], []
    for (k32, r32), (k16, r16) in zip(runs["g32"], runs["g16"]):
        same += int(len(k16) == len(k32) and bool((k16 == k32).all()))
        s16, s32 = set(k16.tolist()), set(k32.tolist())
        sym += len(s16 ^ s32)
        total += len(s16 | s32)
        pos = {int(a): i for i, a in enumerate(k32)}
        for i, a in enumerate(k16):
            j = pos.get(int(a))
            if j is not None:
                dbox.append(float(np.abs(r16[i, :4] - r32[j, :4]).max()))
                dconf.append(float(abs(r16[i, 4] - r32[j, 4])))
    assert total > 0 and dbox
    assert sym / total <= 0.10, (sym, total)                       # measured 0.05
    assert np.percentile(dbox, 50) <= 5.0 and np.percentile(dbox, 95) <= 25.0, (np.percentile(dbox, 50), np.percentile(dbox, 95))
    assert np.percentile(dconf, 95) <= 0.07, np.percentile(dconf, 95)
    assert same < FRAMES, "f16 reproduced every fp32 keep list: re-measure and re-state the f16 mode's contract"


# ---- the fp32 convolution kernels themselves (csrc k32_conv / k32_conv0) against fp64 ------------------------------------------
def _close(got, ref64, rel=5e-6):
    ref, got = ref64.to(torch.float64).cpu(), got.to(torch.float64).cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, (err, scale, err / scale)


@pytest.mark.parametrize("k,s,ci,co,n,h,w", [
    (1, 1, 16, 16, 2, 7, 9), (1, 1, 32, 32, 1, 24, 40), (1, 1, 48, 32, 2, 12, 20), (1, 1, 128, 64, 3, 12, 20), (1, 1, 384, 128, 1, 24, 40),
    (1, 1, 512, 256, 2, 12, 20), (1, 1, 64, 80, 2, 6, 10), (1, 1, 80, 80, 1, 48, 80),
    (3, 1, 16, 16, 2, 9, 11), (3, 1, 64, 64, 1, 12, 20), (3, 1, 64, 80, 2, 24, 40), (3, 1, 80, 80, 1, 12, 20), (3, 1, 256, 64, 1, 12, 20),
    (3, 2, 16, 32, 2, 16, 24), (3, 2, 64, 128, 1, 24, 40), (3, 2, 128, 256, 2, 12, 20), (3, 2, 32, 64, 1, 17, 23),
])
def test_conv32_matches_fp64(k, s, ci, co, n, h, w):
    """Plain tensors, SiLU on / off, every (MT, PT) instantiation the launcher picks for these shapes (odd sizes: partial tiles, borders)."""
    import torch.nn.functional as F
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(k * 1000 + ci + co)
    for act in ("silu", "none"):
        m = nets.Conv(ci, co, k, s, act=(act == "silu"))
        with torch.no_grad():
            m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * (2.0 / (ci * k * k)) ** 0.5)
            m.conv.bias.copy_(torch.randn(co, generator=g))
        x = torch.randn(n, ci, h, w, generator=g)
        ref = F.conv2d(x.double(), m.conv.weight.double(), m.conv.bias.double(), s, k // 2)
        if act == "silu":
            ref = ref * torch.sigmoid(ref)
        m = m.to("cuda")
        xd = x.to("cuda").contiguous(memory_format=torch.channels_last)
        assert fused32.conv_ok(xd, m.conv)
        got = m(xd)
        assert got.is_contiguous(memory_format=torch.channels_last)
        _close(got, ref)


def test_conv32_on_channel_slices_with_shortcut_matches_fp64():
    """Input, output and shortcut as channel slices of wider channels-last tensors (what a C2f block does), shortcut after the SiLU."""
    import torch.nn.functional as F
    from strongsort_yolo_amd import nets
    g = torch.Generator().manual_seed(5)
    m = nets.Conv(32, 32, 3, 1)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.1)
        m.conv.bias.copy_(torch.randn(32, generator=g))
    wide = torch.randn(2, 96, 12, 20, generator=g)
    ref = F.conv2d(wide[:, 32:64].double(), m.conv.weight.double(), m.conv.bias.double(), 1, 1)
    ref = ref * torch.sigmoid(ref) + wide[:, 32:64].double()
    m = m.to("cuda")
    wd = wide.to("cuda").contiguous(memory_format=torch.channels_last)
    before = wd.clone()
    m(wd[:, 32:64], out=wd[:, 64:96], res=wd[:, 32:64])
    _close(wd[:, 64:96], ref)
    assert torch.equal(wd[:, :64], before[:, :64]), "the kernel wrote outside its output slice"


def test_conv0_32_matches_fp64():
    import torch.nn.functional as F
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(11)
    m = nets.Conv(3, 16, 3, 2)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.3)
        m.conv.bias.copy_(torch.randn(16, generator=g))
    w64, b64 = m.conv.weight.detach().double().clone(), m.conv.bias.detach().double().clone()
    md = m.to("cuda")
    for (n, h, w) in ((2, 384, 640), (1, 33, 47)):
        x = torch.rand(n, 3, h, w, generator=g)
        ref = F.conv2d(x.double(), w64, b64, 2, 1)
        ref = ref * torch.sigmoid(ref)
        xd = x.to("cuda").contiguous(memory_format=torch.channels_last)
        assert fused32.conv0_ok(xd, md.conv)
        _close(md(xd), ref)


@pytest.mark.parametrize("name", ["yolov8n", "yolov8s", "yolov8n-pose", "yolov8n-seg", "yolov5n", "yolo11n"])
def test_fp32_detector_on_own_kernels_matches_the_cpu_fp32_network(name):
    """Whole networks: fp32 CUDA (k32_conv / k32_conv0 behind every Conv block they cover) against the same modules on the CPU."""
    from strongsort_yolo_amd import nets
    net = nets.build_detector(name, 3).float().eval()
    x = torch.rand(2, 3, 192, 320, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = net(x)
        ref = (ref[0] if isinstance(ref, tuple) else ref).clone()
        got = net.to("cuda")(x.to("cuda").contiguous(memory_format=torch.channels_last))
    got = got[0] if isinstance(got, tuple) else got
    assert getattr(net, "_own32", None) in (True, None)
    _close(got, ref, rel=2e-5)


def test_the_pipeline_fp32_detector_runs_on_the_own_kernels(runs):
    assert runs["own32"], "FramePipeline(half=False) did not take the fp32 convolution kernels"
