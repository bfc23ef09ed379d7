"""Frame-side kernels through the C ABI vs the C oracle: letterbox (a1), NMS (a3), ReID crop (a4)."""
import numpy as np
import pytest
import torch

from oracle import cexact
from strongsort_yolo_amd.config import DetectConfig
from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
from strongsort_yolo_amd.synth import make_stream, synth_prediction
from tests.gpu_util import engine, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine(debug=False)
    yield e
    e.close()


@pytest.mark.parametrize("wh", [(1280, 720), (1920, 1080), (640, 480), (333, 517)])
def test_letterbox_bit_exact(eng, wh):
    W, H = wh
    rng = np.random.default_rng(W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W)
    got = eng.letterbox(torch.from_numpy(img).to(eng.device), g).cpu().numpy()
    ref = cexact.letterbox(img, g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
    assert got.shape == ref.shape == (3, g.out_h, g.out_w)
    assert bits_equal(got, ref)
    half = eng.letterbox(torch.from_numpy(img).to(eng.device), g, half=True).cpu().numpy()
    assert bits_equal(half, ref.astype(np.float16))


def test_letterbox_geometry_matches_survey():
    g = letterbox_geometry(720, 1280)
    assert (g.out_h, g.out_w) == (384, 640)
    g = letterbox_geometry(1080, 1920)
    assert (g.out_h, g.out_w) == (384, 640)
    g = letterbox_geometry(480, 640)
    assert (g.out_h, g.out_w) == (480, 640)


@pytest.mark.parametrize("n_ids,wh,nc", [(30, (1280, 720), 80), (100, (1920, 1080), 80), (5, (640, 480), 1)])
def test_nms_bit_exact_and_recovers_truth(eng, n_ids, wh, nc):
    W, H = wh
    dcfg = DetectConfig()
    g = letterbox_geometry(H, W)
    gain, px, py = scale_geometry(g, H, W)
    N = (g.out_h // 8) * (g.out_w // 8) + (g.out_h // 16) * (g.out_w // 16) + (g.out_h // 32) * (g.out_w // 32)
    s = make_stream(3, W, H, n_ids, n_classes=min(nc, 3))
    fr = s.next_frame()
    rng = np.random.default_rng(9)
    pred, _ = synth_prediction(fr.dets, N, nc, gain, (px, py), rng)
    rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, gain, px, py, W, H)
    k = int(count.item())
    rows, keep = rows.cpu().numpy()[:k], keep.cpu().numpy()[:k]
    rkeep, rrows = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
    rrows = cexact.scale_boxes(rrows, gain, px, py, W, H)
    assert np.array_equal(keep, rkeep)                       # integer outputs: exact
    assert bits_equal(rows, rrows)
    # every survivor is one of the true boxes (overlapping same-class truths may suppress each other)
    assert 0 < k <= len(fr.dets)
    for r in rows:
        err = np.abs(fr.dets[:, :4] - r[:4]).max(axis=1)
        j = int(err.argmin())
        assert err[j] < 0.01 and fr.dets[j, 5] == r[5] and abs(fr.dets[j, 4] - r[4]) < 1e-6


def test_nms_edge_cases(eng):
    dcfg = DetectConfig()
    N, nc = 5040, 80
    rng = np.random.default_rng(0)
    # nothing above threshold
    pred = rng.uniform(0, 0.2, (4 + nc, N)).astype(np.float32)
    _, _, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 0.5, 0.0, 12.0, 1280, 720)
    assert int(count.item()) == 0
    # many identical boxes + identical scores (ties): lowest anchor index must survive
    pred = np.zeros((4 + nc, N), np.float32)
    pred[0:4, :] = np.array([[100.0], [100.0], [50.0], [80.0]], np.float32)
    pred[4, :3000] = 0.9
    rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 0.5, 0.0, 12.0, 1280, 720)
    rkeep, _ = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, False, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
    assert int(count.item()) == 1 and keep.cpu().numpy()[0] == 0 == rkeep[0]
    # dense random field: hundreds of survivors, idempotence (NMS of the survivors keeps all)
    pred = np.zeros((4 + nc, N), np.float32)
    pred[0] = rng.uniform(0, 640, N); pred[1] = rng.uniform(0, 384, N)
    pred[2] = rng.uniform(10, 80, N); pred[3] = rng.uniform(10, 80, N)
    pred[4:6] = rng.uniform(0, 1, (2, N))
    rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 1.0, 0.0, 0.0, 640, 384)
    k = int(count.item())
    rkeep, rrows = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, False, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
    assert k == len(rkeep) and np.array_equal(keep.cpu().numpy()[:k], rkeep)
    assert bits_equal(rows.cpu().numpy()[:k], cexact.scale_boxes(rrows, 1.0, 0.0, 0.0, 640, 384))
    scores = rows.cpu().numpy()[:k, 4]
    assert np.all(np.diff(scores) <= 0)                      # sortedness
    sub = pred[:, keep.cpu().numpy()[:k]].copy()
    _, _, c2 = eng.nms(torch.from_numpy(np.ascontiguousarray(sub)).to(eng.device), nc, dcfg, 1.0, 0.0, 0.0, 640, 384)
    assert int(c2.item()) == k                               # idempotence


@pytest.mark.parametrize("wh,n", [((1280, 720), 30), ((1920, 1080), 100), ((640, 480), 3)])
def test_crop_norm_bit_exact(eng, wh, n):
    W, H = wh
    s = make_stream(4, W, H, n)
    fr = s.next_frame()
    img = s.frame_pixels(0)
    dets = fr.dets.copy()
    dets[0, :4] = [-5.0, -3.0, 20.5, 30.2]                   # clipped at the border
    if len(dets) > 1:
        dets[1, :4] = [W - 10.0, H - 12.0, W + 50.0, H + 50.0]
    if len(dets) > 2:
        dets[2, :4] = [100.0, 100.0, 100.4, 100.3]           # degenerate (sub-pixel) box
    dt = torch.from_numpy(dets).to(eng.device)
    got = eng.crop_norm(torch.from_numpy(img).to(eng.device), dt, len(dets)).cpu().numpy()
    ref = cexact.crop_norm(img, dets)
    assert bits_equal(got, ref)


def test_nms_carries_extra_channels(eng):
    """pose heads: 51 keypoint channels ride along with the kept anchors (n_extra)."""
    dcfg = DetectConfig()
    N, nc, ne = 1344, 1, 51
    rng = np.random.default_rng(3)
    pred = np.zeros((4 + nc + ne, N), np.float32)
    pred[0] = rng.uniform(0, 320, N); pred[1] = rng.uniform(0, 256, N)
    pred[2] = rng.uniform(10, 60, N); pred[3] = rng.uniform(10, 60, N)
    pred[4] = rng.uniform(0, 1, N)
    pred[5:] = rng.standard_normal((ne, N))
    rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 1.0, 0.0, 0.0, 320, 256, n_extra=ne)
    k = int(count.item())
    rows, keep = rows.cpu().numpy()[:k], keep.cpu().numpy()[:k]
    rkeep, rrows = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, False, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
    assert k > 10 and np.array_equal(keep, rkeep)
    assert bits_equal(rows[:, :6], cexact.scale_boxes(rrows, 1.0, 0.0, 0.0, 320, 256))
    assert bits_equal(rows[:, 6:], np.ascontiguousarray(pred[5:, keep].T))


# ---- batched launches (one per stage over all streams / frames of a group) -----------------------------
@pytest.mark.parametrize("half,hwc", [(False, False), (True, False), (False, True), (True, True)])
def test_letterbox_batch_bit_exact(eng, half, hwc):
    W, H, B = 1280, 720, 5
    rng = np.random.default_rng(11)
    imgs = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W)
    out = eng.letterbox_batch(torch.from_numpy(imgs).to(eng.device), g, half=half, channels_last=hwc)
    assert out.shape == (B, 3, g.out_h, g.out_w)
    assert out.is_contiguous(memory_format=torch.channels_last if hwc else torch.contiguous_format)
    got = out.cpu().numpy()                                  # logical NCHW view either way
    for b in range(B):
        ref = cexact.letterbox(imgs[b], g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
        assert bits_equal(got[b], ref.astype(np.float16) if half else ref)


@pytest.mark.parametrize("wh", [(1920, 1080), (640, 480), (3840, 2160), (720, 1280), (333, 500), (320, 240)])
@pytest.mark.parametrize("half", [True, False])
def test_letterbox_channels_last_other_geometries(eng, wh, half):
    """The channels-last form the pipeline uses at a down-scale by 3 and by 6, no resize, side padding, an unaligned row pitch and an up-scale.
    (A form that stages the source rows of 4 output rows in LDS, 4 pixels per thread, passed these too and was slower: 58 vs 40 us per 32
    720p frames, 88 vs 39 at 1080p - the byte loads of neighbouring threads hit the same cache lines; not kept.)"""
    W, H = wh
    rng = np.random.default_rng(W + H)
    imgs = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    g = letterbox_geometry(H, W)
    got = eng.letterbox_batch(torch.from_numpy(imgs).to(eng.device), g, half=half, channels_last=True).cpu().numpy()
    for b in range(2):
        ref = cexact.letterbox(imgs[b], g.out_h, g.out_w, g.new_h, g.new_w, g.pad_top, g.pad_left)
        assert bits_equal(got[b], ref.astype(np.float16) if half else ref)


def test_nms_batch_bit_exact_mixed_geometry(eng):
    """Images of one batch carry their own scale_boxes geometry; one of them is empty, one is dense."""
    dcfg = DetectConfig()
    nc, N = 80, 5040
    rng = np.random.default_rng(21)
    sizes = [(1280, 720), (1920, 1080), (1280, 720), (1280, 720), (1920, 1080), (1280, 720)]
    preds, geoms = [], []
    for b, (W, H) in enumerate(sizes):
        g = letterbox_geometry(H, W)
        gain, px, py = scale_geometry(g, H, W)
        if b == 2:                                           # nothing above threshold
            pred = rng.uniform(0, 0.2, (4 + nc, N)).astype(np.float32)
        elif b == 3:                                         # dense random field: hundreds of survivors
            pred = np.zeros((4 + nc, N), np.float32)
            pred[0] = rng.uniform(0, 640, N); pred[1] = rng.uniform(0, 384, N)
            pred[2] = rng.uniform(10, 80, N); pred[3] = rng.uniform(10, 80, N)
            pred[4:6] = rng.uniform(0, 1, (2, N))
        else:
            fr = make_stream(30 + b, W, H, 20 + 15 * b, n_classes=3).next_frame()
            pred, _ = synth_prediction(fr.dets, N, nc, gain, (px, py), rng)
        preds.append(pred); geoms.append([gain, px, py, float(W), float(H)])
    P = torch.from_numpy(np.stack(preds)).to(eng.device)
    G = torch.tensor(geoms, dtype=torch.float32, device=eng.device)
    rows, keep, count = eng.nms_batch(P, nc, dcfg, G)
    rows, keep, count = rows.cpu().numpy(), keep.cpu().numpy(), count.cpu().numpy()
    for b, (W, H) in enumerate(sizes):
        rkeep, rrows = cexact.nms(preds[b], nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
        rrows = cexact.scale_boxes(rrows, geoms[b][0], geoms[b][1], geoms[b][2], W, H)
        k = int(count[b])
        assert k == len(rkeep), b
        assert np.array_equal(keep[b, :k], rkeep)
        assert bits_equal(rows[b, :k], rrows)
    assert count[2] == 0 and count[3] > 100
    # the same batch again (counters re-armed in-kernel) and a smaller batch on the grown workspace
    rows2, keep2, count2 = eng.nms_batch(P, nc, dcfg, G)
    assert np.array_equal(count2.cpu().numpy(), count) and bits_equal(rows2.cpu().numpy(), rows)
    r1, k1, c1 = eng.nms_batch(P[4:5], nc, dcfg, G[4:5])
    assert int(c1[0]) == count[4] and bits_equal(r1.cpu().numpy()[0], rows[4])
    # and the single-image entry point agrees with its row of the batch
    r0, k0, c0 = eng.nms(P[1], nc, dcfg, geoms[1][0], geoms[1][1], geoms[1][2], 1920, 1080)
    assert int(c0.item()) == count[1] and bits_equal(r0.cpu().numpy()[:count[1]], rows[1, :count[1]])
    eng.check_errors()


def test_nms_candidate_overflow_is_reported(eng):
    """More than 8192 candidates above conf: the image yields no rows and ss_check_errors raises CAPACITY
    (the reference's max_nms=30000 truncation is out of the supported range and must not pass silently)."""
    from strongsort_yolo_amd.lib import SSError, SS_ERR_CAPACITY
    dcfg = DetectConfig()
    nc, N = 2, 21504
    rng = np.random.default_rng(5)
    pred = np.zeros((4 + nc, N), np.float32)
    pred[0] = rng.uniform(0, 640, N); pred[1] = rng.uniform(0, 640, N); pred[2:4] = 20.0
    pred[4] = rng.uniform(0.5, 1.0, N)
    _, _, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 1.0, 0.0, 0.0, 640, 640)
    assert int(count.item()) == 0
    with pytest.raises(SSError) as ei:
        eng.check_errors()
    assert ei.value.code == SS_ERR_CAPACITY
    eng.reset(-1)
    eng.check_errors()


@pytest.mark.parametrize("half,hwc", [(False, False), (True, True), (False, True)])
def test_crop_norm_batch_bit_exact(eng, half, hwc):
    W, H, B, n = 1280, 720, 3, 16
    imgs, dets, counts = [], np.zeros((B, 32, 6), np.float32), []
    for b in range(B):
        s = make_stream(40 + b, W, H, 6 + 5 * b)
        fr = s.next_frame()
        imgs.append(s.frame_pixels(0))
        k = min(len(fr.dets), n)
        dets[b, :k] = fr.dets[:k]
        counts.append(k)
    dets[0, 0, :4] = [-5.0, -3.0, 20.5, 30.2]
    dets[1, 1, :4] = [W - 10.0, H - 12.0, W + 50.0, H + 50.0]
    # the channels-last kernel stages a band's source rows in LDS: a box too large for it (global-load path), the last bytes
    # of the last frame, one-pixel-wide / one-pixel-high boxes, every byte alignment of the first column
    dets[2, 0, :4] = [0.0, 0.0, W, H]
    dets[2, 1, :4] = [W - 3.0, H - 2.0, W, H]
    dets[0, 1, :4] = [101.0, 50.0, 102.5, 400.0]
    dets[1, 0, :4] = [30.0, 300.0, 500.0, 301.2]
    for j in range(2, 6):
        dets[2, j, :4] = [200.0 + j, 100.0, 260.0 + 2 * j, 290.0]
    # tall boxes whose 64-row band does not fit the LDS but whose 16-row sub-bands do (staged four times per workgroup), at two
    # byte alignments, the second one touching the last row of the frame
    dets[0, 2, :4] = [300.0, 60.0, 520.0, 560.0]
    dets[1, 2, :4] = [701.0, 150.0, 990.0, H + 4.0]
    counts[0], counts[1], counts[2] = max(counts[0], 3), max(counts[1], 3), max(counts[2], 6)
    out = torch.full((B * n, 3, 256, 128), 7.0, dtype=torch.float16 if half else torch.float32, device=eng.device)
    if hwc:
        out = out.contiguous(memory_format=torch.channels_last)
    eng.crop_norm_batch(torch.from_numpy(np.stack(imgs)).to(eng.device), torch.from_numpy(dets).to(eng.device), n,
                        counts=torch.tensor(counts, dtype=torch.int32, device=eng.device), half=half, out=out,
                        channels_last=hwc)
    got = out.cpu().numpy().reshape(B, n, 3, 256, 128)
    for b in range(B):
        ref = cexact.crop_norm(imgs[b], dets[b, :counts[b]])
        assert bits_equal(got[b, :counts[b]], ref.astype(np.float16) if half else ref)
        assert np.all(got[b, counts[b]:] == 7.0)             # rows past the count stay untouched


def test_nms_classes_override(eng):
    """overrides['classes'] (yolo_multi_model.py:22): anchors whose best class is not listed never become candidates."""
    W, H, nc = 1280, 720, 80
    dcfg = DetectConfig()
    g = letterbox_geometry(H, W)
    gain, px, py = scale_geometry(g, H, W)
    N = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
    fr = make_stream(5, W, H, 40, n_classes=3).next_frame()
    pred, _ = synth_prediction(fr.dets, N, nc, gain, (px, py), np.random.default_rng(2))
    try:
        for classes in ([0, 2], [1], [79]):
            eng.nms_set_classes(classes)
            rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, gain, px, py, W, H)
            k = int(count.item())
            rkeep, rrows = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, dcfg.max_det,
                                      classes=classes)
            assert np.array_equal(keep.cpu().numpy()[:k], rkeep)
            assert bits_equal(rows.cpu().numpy()[:k], cexact.scale_boxes(rrows, gain, px, py, W, H))
            assert set(rrows[:, 5].astype(int).tolist()) <= set(classes) and (k > 0) == (79 not in classes)
    finally:
        eng.nms_set_classes(None)


@pytest.mark.parametrize("packed", [False, True])
def test_crop_norm_byte_output_is_the_float_crop_before_normalisation(eng, packed):
    """SS_DST_U8: the crop kernel hands over the rounded bilinear bytes; ((q / 255) - mean) / sd applied to them (fused32.crops_from_u8, the
    expression k32_stemW's staging uses) gives the float crops bit for bit — large boxes (global-load path), clipped and sub-pixel ones included."""
    from strongsort_yolo_amd import fused32
    W, H, B, n = 1280, 720, 3, 32
    rng = np.random.default_rng(31)
    frames = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(eng.device)
    x1 = rng.uniform(-10, W - 40, (B, n)); y1 = rng.uniform(-10, H - 60, (B, n))
    bw = rng.uniform(0.3, 500, (B, n)); bh = rng.uniform(0.3, 650, (B, n))
    dets = np.stack([x1, y1, x1 + bw, y1 + bh, np.ones((B, n)), np.zeros((B, n))], 2).astype(np.float32)
    dets = torch.from_numpy(dets).to(eng.device)
    counts = torch.tensor([32, 0, 17], dtype=torch.int32, device=eng.device)
    mk = lambda dt: torch.zeros(B * n, 3, 256, 128, dtype=dt, device=eng.device).contiguous(memory_format=torch.channels_last)
    f32, u8 = mk(torch.float32), mk(torch.uint8)
    if packed:
        off = torch.zeros(B + 1, dtype=torch.int32, device=eng.device)
        eng.crop_norm_packed(frames, dets, n, counts, off, f32, half=False)
        eng.crop_norm_packed(frames, dets, n, counts, off, u8)
        k = int(off[B].item())
        assert k == 49
        a, b = f32[:k], fused32.crops_from_u8(u8[:k])
    else:
        eng.crop_norm_batch(frames, dets, n, counts=counts, half=False, out=f32, channels_last=True)
        eng.crop_norm_batch(frames, dets, n, counts=counts, out=u8, channels_last=True)
        sel = torch.cat([torch.arange(0, 32), torch.arange(64, 81)]).to(eng.device)
        a, b = f32[sel], fused32.crops_from_u8(u8[sel])
    assert bits_equal(a.cpu().numpy(), b.cpu().numpy())


@pytest.mark.parametrize("n_cand", [0, 1, 2, 17, 40, 63, 64, 65, 130])
@pytest.mark.parametrize("agnostic", [False, True])
def test_nms_small_candidate_counts_one_wave_path(eng, n_cand, agnostic):
    """Up to 64 candidates an image is handled by ONE wave in registers (shuffle sort, IoU word per lane, greedy scan); 65 and more take the
    workgroup path.  Both against the oracle: heavily overlapping boxes, equal scores (ties broken by anchor index), two classes."""
    from dataclasses import replace
    dcfg = replace(DetectConfig(), agnostic_nms=agnostic)
    N, nc = 1344, 2
    rng = np.random.default_rng(1000 + n_cand)
    pred = np.zeros((4 + nc, N), np.float32)
    pred[0] = rng.uniform(0, 320, N); pred[1] = rng.uniform(0, 256, N); pred[2] = rng.uniform(10, 60, N); pred[3] = rng.uniform(10, 60, N)
    idx = rng.choice(N, n_cand, replace=False)
    centres = rng.uniform(60, 200, (max(1, n_cand // 5 + 1), 2))
    for j, a in enumerate(idx):                                  # clusters of ~5 boxes around a few centres: most pairs inside a cluster overlap
        c = centres[j % len(centres)]
        pred[0, a], pred[1, a] = c[0] + rng.uniform(-6, 6), c[1] + rng.uniform(-6, 6)
        pred[2, a], pred[3, a] = 40 + rng.uniform(-4, 4), 50 + rng.uniform(-4, 4)
        pred[4 + (j % nc), a] = [0.9, 0.9, 0.8, 0.8, 0.7][j % 5] if j % 7 else float(rng.uniform(0.4, 0.95))   # equal scores inside clusters
    rows, keep, count = eng.nms(torch.from_numpy(pred).to(eng.device), nc, dcfg, 1.0, 0.0, 0.0, 320, 256)
    k = int(count.item())
    rkeep, rrows = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, agnostic, dcfg.max_wh, dcfg.max_nms, dcfg.max_det)
    assert k == len(rkeep) and np.array_equal(keep.cpu().numpy()[:k], rkeep)
    assert bits_equal(rows.cpu().numpy()[:k, :6], cexact.scale_boxes(rrows, 1.0, 0.0, 0.0, 320, 256))
    if n_cand:
        assert 0 < k <= n_cand
