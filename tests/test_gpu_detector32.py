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
    out = {"cpu": [], "g32": [], "g16": [], "own32": bool(getattr(p32.detector, "_own32", False))}
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
