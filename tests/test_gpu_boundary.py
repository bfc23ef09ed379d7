"""The drop-in classes on the GPU: StrongSORT.update(dets, frame) and YOLO.track()/.predict()."""
import numpy as np
import pytest
import torch

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.synth import make_stream

pytestmark = pytest.mark.gpu


def test_strongsort_update_injected_features_equals_oracle():
    from strongsort_yolo_amd.tracker import StrongSORT
    trk, orc = StrongSORT(), OracleStrongSort(StrongSortConfig(), "c")
    sg, so = make_stream(21, 640, 480, 9), make_stream(21, 640, 480, 9)
    for k in range(15):
        fg, fo = sg.next_frame(), so.next_frame()
        got = trk.update(fg.dets, sg.frame_pixels(k), features=fg.feats)
        ref = orc.update(fo.dets, fo.feats, (480, 640))
        assert got.tobytes() == ref.tobytes() and got.shape == ref.shape
    trk.close()


def test_strongsort_update_with_reid_net_runs_and_is_deterministic():
    from strongsort_yolo_amd.tracker import StrongSORT
    outs = []
    for _ in range(2):
        trk, st = StrongSORT(), make_stream(22, 640, 480, 6)
        rows = []
        for k in range(6):
            f = st.next_frame()
            rows.append(trk.update(f.dets, st.frame_pixels(k)))
        outs.append(np.concatenate(rows))
        trk.close()
    assert outs[0].shape[1] == 8 and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("weights", ["yolov8n.pt", "yolo11n-pose.pt"])
def test_yolo_track_and_predict_contract(weights):
    from strongsort_yolo_amd.yolo import YOLO
    model = YOLO(weights)
    model.overrides.update(conf=0.9, iou=0.4, agnostic_nms=False, max_det=50)    # random-init head: keep it sparse
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    res = model.predict(img, verbose=False, device=0)
    assert len(res) == 1 and res[0].boxes is not None and res[0].boxes.xyxy.shape[1] == 4
    for _ in range(3):
        res = model.track(img, verbose=False, device=0, persist=True, tracker="botsort.yaml")
    r = res[0]
    if "pose" in weights and r.keypoints is not None:
        assert len(r.keypoints) == len(r.boxes)
        for bbox, kp in zip(r.boxes, r.keypoints):               # yolo_multi_model.py:58-62
            pts = kp.xy.tolist()
            assert len(pts) == 1 and len(pts[0]) == 17 and len(pts[0][0]) == 2
    if r.boxes.id is None:                 # nothing confirmed: the reference skips such frames (yolo_multi_model.py:54)
        assert len(r.boxes) == 0
        return
    assert len(r.boxes.id) == len(r.boxes.conf) == len(r.boxes.cls)
    for bbox in r.boxes:
        for scores, classes, xyxy, id_ in zip(bbox.conf, bbox.cls, bbox.xyxy, bbox.id):
            assert int(id_) >= 1 and 0 <= float(scores) <= 1 and int(classes) in r.names


# ---- the fast paths behind the drop-in calls ---------------------------------------------------------
H_, W_, NF_ = 480, 640, 37


def _synthetic_model(weights="yolov8n.pt", n_ids=9, nk=0, cmc=False):
    """YOLO object whose NMS consumes a synthetic head tensor (the random-init detector still runs for load) and whose
    tracker consumes the synthetic identity features: the oracle chain can be replayed on the same inputs."""
    from oracle import cexact
    from strongsort_yolo_amd.config import DetectConfig
    from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
    from strongsort_yolo_amd.synth import synth_prediction
    from strongsort_yolo_amd.yolo import YOLO
    model = YOLO(weights, random_init_ok=True, camera_motion=cmc)
    model.overrides.update(conf=0.3, iou=0.4, agnostic_nms=False, max_det=1000)            # yolo_multi_model.py:18-21
    model._pipe_kw.update(det_source="synthetic", feat_source="by_anchor", reid_batch=32)
    g = letterbox_geometry(H_, W_)
    gs = scale_geometry(g, H_, W_)
    A = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
    nc = 1 if nk else 80
    st, rng = make_stream(41, W_, H_, n_ids), np.random.default_rng(41)
    frames, preds, agts, feats = [], [], [], []
    for k in range(NF_):
        fr = st.next_frame()
        pred, agt = synth_prediction(fr.dets, A, nc, gs[0], (gs[1], gs[2]), rng)
        if nk:                                               # keypoint rows: any values, they must follow their anchor
            pred = np.concatenate([pred, rng.uniform(0, 400, (nk, A)).astype(np.float32)])
        f = np.zeros((128, 512), np.float32)
        f[:len(fr.feats)] = fr.feats
        frames.append(st.frame_pixels(k).copy()); preds.append(pred); agts.append(agt); feats.append(f)
    dev = torch.device("cuda", 0)
    dp, da, df = (torch.from_numpy(np.stack(x)).to(dev) for x in (preds, agts, feats))

    def fill(b, v, k):
        b.pred_in[v].copy_(dp[k]); b.anchor_gt[v].copy_(da[k]); b.gt_feats[v].copy_(df[k])

    model._fill = fill
    dcfg, orc, ref = DetectConfig(), OracleStrongSort(StrongSortConfig(), "c"), []
    for k in range(NF_):
        keep, r = cexact.nms(preds[k][:4 + nc], nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, 32)
        r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], W_, H_)
        warp = None
        if cmc and k > 0:                                    # N4: the oracle's own ECC on the same two frames
            hs, ws = int(H_ * 0.1), int(W_ * 0.1)
            wm, it = cexact.ecc(cexact.gray_small(frames[k - 1], hs, ws), cexact.gray_small(frames[k], hs, ws))
            if it >= 1:
                warp = wm.copy(); warp[0, 2] *= W_ / ws; warp[1, 2] *= H_ / hs
        rows = orc.update(r, feats[k][np.maximum(agts[k][keep], 0)], (H_, W_), warp)
        kp = None
        if nk:
            kp = preds[k][4 + nc:, keep].T.reshape(len(keep), nk // 3, 3).copy()
            kp[..., 0] = (kp[..., 0] - np.float32(gs[1])) / np.float32(gs[0])
            kp[..., 1] = (kp[..., 1] - np.float32(gs[2])) / np.float32(gs[0])
        ref.append((r, rows[rows[:, 7] >= 0], kp))
    return model, frames, ref


def _check_tracked(res, ref_rows, ref_kp, k):
    r = res[0]
    if len(ref_rows) == 0:
        assert len(r.boxes) == 0 and r.boxes.id is None, f"frame {k}"
        return
    assert np.array_equal(r.boxes.id.numpy(), ref_rows[:, 4]), f"frame {k}: ids"
    assert np.array_equal(r.boxes.xyxy.numpy(), ref_rows[:, :4]) and np.array_equal(r.boxes.cls.numpy(), ref_rows[:, 5])
    assert np.array_equal(r.boxes.conf.numpy(), ref_rows[:, 6])
    if ref_kp is not None:                                   # keypoints follow det_idx (BASELINE configs[4])
        assert np.allclose(r.keypoints.data.numpy(), ref_kp[ref_rows[:, 7].astype(int)], rtol=0, atol=1e-4), f"frame {k}: keypoints"


@pytest.mark.parametrize("weights,nk", [("yolov8n.pt", 0), ("yolov8n-pose.pt", 51)])
def test_yolo_track_graph_path_equals_oracle(weights, nk):
    """model.track(frame) per frame (yolo_multi_model.py:41): replayed HIP graphs, pinned buffers, one sync per call."""
    model, frames, ref = _synthetic_model(weights, nk=nk)
    for k in range(20):
        _check_tracked(model.track(frames[k], verbose=False, device=0, persist=True, tracker="botsort.yaml"), ref[k][1], ref[k][2], k)
    # model.predict on the same object: detection only, the tracker does not advance
    det = model.predict(frames[20], verbose=False, device=0)[0]
    assert np.array_equal(det.boxes.xyxy.numpy(), ref[20][0][:, :4]) and np.array_equal(det.boxes.conf.numpy(), ref[20][0][:, 4])
    model.close()


@pytest.mark.parametrize("weights,nk,stream", [("yolov8n.pt", 0, False), ("yolov8n-pose.pt", 51, False), ("yolov8n.pt", 0, True)])
def test_yolo_all_fp32_equals_oracle(weights, nk, stream):
    """YOLO(..., half=False): the reference's own precision (it passes no half=, yolo_multi_model.py:41) — detector on the fp32 convolution
    kernels, ReID crops + OSNet on the fp32 ReID kernels — behind the same drop-in calls, per frame and as a stream; the tracker side equals
    the oracle chain as in every other mode."""
    model, frames, ref = _synthetic_model(weights, nk=nk)
    model._pipe_kw.update(half=False, reid_half=False)
    if stream:
        for k, res in enumerate(model.track_stream(frames[:24], batch=8)):
            _check_tracked(res, ref[k][1], ref[k][2], k)
        assert k == 23
        pipe = model._stream_pipe
    else:
        for k in range(12):
            _check_tracked(model.track(frames[k], verbose=False, device=0, persist=True, tracker="botsort.yaml"), ref[k][1], ref[k][2], k)
        pipe = model._pipe
    assert pipe.dtype == torch.float32 and pipe.reid_dtype == torch.float32 and getattr(pipe.detector, "_own32", False)
    model.close()


@pytest.mark.parametrize("batch", [1, 4, 16, 32])
def test_yolo_track_stream_equals_oracle(batch):
    """the throughput form: groups of `batch` frames through the overlapped pipeline, partial last group included"""
    model, frames, ref = _synthetic_model()
    n = 0
    for k, res in enumerate(model.track_stream(iter(frames), batch=batch, device=0)):
        _check_tracked(res, ref[k][1], ref[k][2], k)
        assert res[0].orig_img is frames[k]
        n += 1
    assert n == NF_
    model.close()


@pytest.mark.parametrize("batch,cmc", [(4, False), (32, False), (4, True), (3, True)])
def test_yolo_track_stream_called_twice_and_after_an_abandoned_generator(batch, cmc):
    """track_stream on the same model object again: the cached pipeline's buffer-set rotation does not restart
    with the generator's group numbering, and a partial last group must leave the right 'previous frame' for the camera-motion
    estimate of the next call.  17 frames = 4 full groups + 1 frame at batch 4 (5 groups: not a multiple of the 3 buffer sets);
    then the rest of the clip; then a generator abandoned half way, and a third call picking up where it stopped."""
    model, frames, ref = _synthetic_model(cmc=cmc)
    n = 0
    for part in (frames[:17], frames[17:26]):
        for res in model.track_stream(iter(part), batch=batch, device=0):
            _check_tracked(res, ref[n][1], ref[n][2], n)
            assert res[0].orig_img is frames[n]
            n += 1
    assert n == 26
    model.close()
    # abandoned generator on one pipeline: consume a few results, drop it, go on from the frames it had taken in
    model, frames, ref = _synthetic_model(cmc=cmc)
    gen = model.track_stream(iter(frames[:20]), batch=batch, device=0)
    got = 0
    for res in gen:
        _check_tracked(res, ref[got][1], ref[got][2], got)
        got += 1
        if got == 2:
            break
    gen.close()
    k0 = model._frame_index                                   # frames the abandoned call had submitted: all tracked, in order
    assert got <= k0 <= 20
    n = k0
    for res in model.track_stream(iter(frames[k0:]), batch=batch, device=0):
        _check_tracked(res, ref[n][1], ref[n][2], n)
        n += 1
    assert n == NF_
    model.close()


@pytest.mark.parametrize("batch", [0, 4])
def test_yolo_with_camera_motion_compensation_equals_oracle(batch):
    """YOLO(..., camera_motion=True): ECC warps estimated beside the detector (per frame / per frame group) and applied
    by the tracker == the oracle tracker fed its own ECC warps (N4)"""
    model, frames, ref = _synthetic_model(cmc=True)
    if batch:
        for k, res in enumerate(model.track_stream(iter(frames), batch=batch, device=0)):
            _check_tracked(res, ref[k][1], ref[k][2], k)
    else:
        for k in range(20):
            _check_tracked(model.track(frames[k], verbose=False, device=0, persist=True), ref[k][1], ref[k][2], k)
    model.close()


def test_cli_process_video_on_the_gpu(tmp_path):
    """labels file + class counts of the reference's --track --count loop (yolo_multi_model.py:244-339) through the real
    model object, both the grouped and the per-frame call form giving the same file."""
    from strongsort_yolo_amd.cli import process_video
    outs = []
    for batch in (8, 1):
        model, frames, ref = _synthetic_model()
        np.save(tmp_path / f"clip{batch}.npy", np.stack(frames))
        out = process_video({"source": str(tmp_path / f"clip{batch}.npy"), "track": True, "count": True,
                             "outdir": str(tmp_path), "batch": batch}, model)
        assert out["frames"] == NF_ and out["counts"] == {"person": len({int(i) for _, rows, _ in ref for i in rows[:, 4]})}
        lines = open(tmp_path / f"clip{batch}_labels.txt").read().strip().split("\n")
        assert len(lines) == sum(len(rows) for _, rows, _ in ref)
        f0 = lines[-1].split()
        assert len(f0) == 12 and f0[0] == str(NF_ - 1) and f0[-4:] == ["-1"] * 4 and all("." not in v for v in f0[4:8])
        outs.append(lines)
        model.close()
    assert outs[0] == outs[1]


def test_cli_save_annotated_frames(tmp_path):
    """--save: overlay kernel + sink behind the reference's loop (yolo_multi_model.py:58-162, :331)"""
    from strongsort_yolo_amd.cli import process_video
    model, frames, ref = _synthetic_model()
    np.save(tmp_path / "clip.npy", np.stack(frames[:20]))
    out = process_video({"source": str(tmp_path / "clip.npy"), "track": True, "count": False, "outdir": str(tmp_path), "batch": 8,
                         "save": str(tmp_path / "annotated.bgr")}, model)
    ann = np.fromfile(tmp_path / "annotated.bgr", np.uint8).reshape(20, H_, W_, 3)
    assert out["frames"] == 20
    assert np.array_equal(ann[0], frames[0])                              # no confirmed track yet, no count plate: nothing drawn
    changed = (ann[10] != frames[10]).any(axis=2)
    assert changed.sum() > 500
    x1, y1 = int(ref[10][1][0, 0]), int(ref[10][1][0, 1])
    assert tuple(int(v) for v in ann[10][y1 + 20, x1]) == (0, 0, 225)     # left edge of the first track's box outline (BGR)
    model.close()


def test_cli_save_draws_on_the_resident_frame_one_upload_per_frame(tmp_path):
    """N2's reason to exist ("keeps frames on device until encode", VERDICT r3 'next' 7): with --save on the throughput
    path every frame is uploaded ONCE (by track_stream) and downloaded once — the overlay runs on the device copy the pipeline kept —
    and the annotated frames equal the ones the upload-draw-download form (Overlay.draw) produces."""
    from strongsort_yolo_amd.cli import process_video
    from strongsort_yolo_amd.engine import TrackerEngine
    calls = {"up": 0, "down": 0}
    up0, down0, upb0 = TrackerEngine.upload, TrackerEngine.download, TrackerEngine.upload_batch

    def up(self, *a, **k):
        calls["up"] += 1
        return up0(self, *a, **k)

    def upb(self, dst, srcs, *a, **k):                      # a frame group in one call: counts its frames
        calls["up"] += len(srcs)
        return upb0(self, dst, srcs, *a, **k)

    def down(self, *a, **k):
        calls["down"] += 1
        return down0(self, *a, **k)

    TrackerEngine.upload, TrackerEngine.download, TrackerEngine.upload_batch = up, down, upb
    try:
        model, frames, ref = _synthetic_model()
        np.save(tmp_path / "clip.npy", np.stack(frames[:24]))
        out = process_video({"source": str(tmp_path / "clip.npy"), "track": True, "count": True, "outdir": str(tmp_path), "batch": 8,
                             "save": str(tmp_path / "resident.npy")}, model)
        assert out["frames"] == 24 and calls == {"up": 24, "down": 24}, calls
        res_frames = np.load(tmp_path / "resident.npy")
        # the same run through the host-frame form (one more upload per frame)
        model2, _, _ = _synthetic_model()
        ov, host = None, []
        from strongsort_yolo_amd.cli import ClassCounter
        cnt = ClassCounter(model2.names)
        for k, res in enumerate(model2.track_stream(iter(frames[:24]), batch=8, device=0)):
            assert res[0].orig_img_device is None
            cnt.update(res)
            ov = ov or model2.overlay()
            fps = ""                                             # the FPS text depends on wall time: compare frames drawn without it
            host.append(ov.draw(frames[k], res, cnt.counts(), fps))
        model2.close()
    finally:
        TrackerEngine.upload, TrackerEngine.download, TrackerEngine.upload_batch = up0, down0, upb0
    # frames 0..8 carry no FPS text in either run (it appears from the 10th processed frame on): identical pixels there
    for k in range(9):
        assert np.array_equal(res_frames[k], host[k]), f"frame {k}"
    assert (res_frames[8] != frames[8]).any()
    model.close()


def test_predict_needs_no_reid_weights_and_track_asks_for_them(tmp_path, monkeypatch):
    """model.predict (yolo_multi_model.py:173) runs on detector weights alone; the first model.track on that object
    needs the OSNet weights and says so (no silent random-init ReID network)."""
    from strongsort_yolo_amd import nets
    from strongsort_yolo_amd.yolo import YOLO
    monkeypatch.delenv("SS_RANDOM_INIT", raising=False)
    wd, wr = str(tmp_path / "yolov8n.pt"), str(tmp_path / "osnet_x0_25.pt")
    torch.save(nets.build_detector("yolov8n", 3).state_dict(), wd)
    torch.save(nets.build_reid(4).state_dict(), wr)
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    model = YOLO(wd)
    model.overrides.update(conf=0.9, max_det=50)
    assert len(model.predict(img, verbose=False, device=0)) == 1 and len(model(img)) == 1
    with pytest.raises(FileNotFoundError, match="OSNet"):
        model.track(img, verbose=False, device=0, persist=True)
    model.close()
    model = YOLO(wd, reid_weights=wr)
    model.overrides.update(conf=0.9, max_det=50)
    assert len(model.predict(img)) == 1
    for _ in range(2):
        assert len(model.track(img, verbose=False, device=0, persist=True)) == 1
    model.close()


def test_true_wiring_detector_nms_reid_tracker_edges():
    """det_source='detector', feat_source='reid' (what a deployment runs): the head tensor the detector produced and the
    embeddings OSNet produced are read back and pushed through the oracle chain (C NMS + scale_boxes + tracker); the
    pipeline's rows must equal that bit for bit, i.e. the detector -> NMS and OSNet -> tracker data edges carry exactly
    those tensors.  Random-init networks: the detections are arbitrary, the plumbing is not."""
    from oracle import cexact
    from strongsort_yolo_amd.config import DetectConfig
    from strongsort_yolo_amd.engine import scale_geometry
    from strongsort_yolo_amd.pipeline import FramePipeline
    dcfg = DetectConfig(conf=0.52, iou=0.4)                      # random-init class scores sit around sigmoid(0) = 0.5
    for graph in ("none", "split"):
        pipe = FramePipeline("yolov8n", 1, (H_, W_), graph=graph, reid_batch=32, dcfg=dcfg, det_source="detector", feat_source="reid")
        gs = scale_geometry(pipe.geom, H_, W_)
        orc = OracleStrongSort(StrongSortConfig(), "c")
        st = make_stream(77, W_, H_, 6)
        seen = 0
        for k in range(8):
            pipe.frames[0].copy_(torch.from_numpy(st.frame_pixels(k)).to(pipe.dev))
            pipe.step()
            got = pipe.results()[0]
            pred = pipe.pred_in[0].cpu().numpy()
            keep, r = cexact.nms(pred, pipe.nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, 32)
            r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], W_, H_)
            n = int(pipe.ndets[0])
            assert n == len(r) and np.array_equal(pipe.dets[0, :n].cpu().numpy(), r), f"{graph} frame {k}: NMS edge"
            ref = orc.update(r, pipe.feats_in[0, :n].cpu().numpy(), (H_, W_))
            assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), f"{graph} frame {k}: tracker edge"
            seen += n
        assert seen > 0
        pipe.close()


def test_predict_keeps_more_than_128_detections():
    """model.predict with the reference's max_det = 1000 (yolo_multi_model.py:21): not capped at the tracker's 128 rows"""
    from oracle import cexact
    from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
    from strongsort_yolo_amd.synth import synth_prediction
    from strongsort_yolo_amd.yolo import YOLO
    W, H = 1920, 1080
    model = YOLO("yolov8n.pt", random_init_ok=True)
    model.overrides.update(conf=0.3, iou=0.4, agnostic_nms=False, max_det=1000)
    model._pipe_kw = dict(det_source="synthetic", feat_source="by_anchor", reid_batch=32)
    g = letterbox_geometry(H, W)
    gs = scale_geometry(g, H, W)
    A = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
    fr = make_stream(5, W, H, 220).next_frame()
    pred, _ = synth_prediction(fr.dets, A, 80, gs[0], (gs[1], gs[2]), np.random.default_rng(3), dup=2, clutter=50)
    dp = torch.from_numpy(pred).cuda()
    model._fill = lambda b, v, k: b.pred_in[v].copy_(dp)
    res = model.predict(np.zeros((H, W, 3), np.uint8), verbose=False, device=0)[0]
    keep, r = cexact.nms(pred, 80, 0.3, 0.4, False, 7680.0, 8192, 1000)
    r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], W, H)
    assert len(r) > 128 and np.array_equal(res.boxes.xyxy.numpy(), r[:, :4]) and np.array_equal(res.boxes.conf.numpy(), r[:, 4])
    model.close()


def test_two_ranks_on_one_gpu_each_identical_to_its_oracle():
    """The multi-rank path of bench.py with the PRODUCT pipeline in every rank (VERDICT r2 'next' item 6): two processes
    started exactly as the driver starts them (torch.distributed.run, 127.0.0.1 rendezvous), gloo for the barrier / max /
    gather because both ranks share this box's single GPU; every rank checks its own stream against its own oracle over all
    frames and rank 0 reports the minimum.  Stands for /root/reference/yolo_multi_model.py:351-354 (one process per source)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SS_BENCH_BACKEND="gloo", SS_BENCH_SINGLE_DEVICE="1", SS_RANDOM_INIT="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-api-path", "--no-batched"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_process_group"] == 2 and len(d["per_rank_value"]) == 2 and d["scaling"] == "weak"
    assert d["id_match_rate"] == 1.0
    n, m = d["frames_bit_exact_timed"].split("/")
    assert n == m and int(m) == 2 * 32
    assert abs(d["value"] - 2 * min(d["per_rank_value"])) <= 0.03                       # whole job = ranks x frames / max-rank time (each rounded to 2 decimals)


def test_yolo_segmentation_masks_follow_the_boxes():
    """YOLO("yolov8n-seg.pt") (yolo_multi_model.py:14): `Results.masks` built from the prototypes and mask coefficients the device
    holds for the frame — predict (detection order), track (tracker rows, `det_idx`), and the throughput form's host ring."""
    from strongsort_yolo_amd.yolo import YOLO, assemble_masks
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(6)]
    model = YOLO("yolov8n-seg.pt")
    model.overrides.update(conf=0.5, iou=0.4, agnostic_nms=False, max_det=20)       # random-init head: about half the anchors pass
    det = model.predict(frames[0], verbose=False, device=0)[0]
    pipe = model._pipe
    ih, iw = pipe.geom.out_h, pipe.geom.out_w
    assert pipe.nm == 32 and pipe.nk == 0 and tuple(pipe.proto.shape[1:]) == (32, ih // 4, iw // 4)
    n = len(det.boxes)
    assert det.masks is not None and len(det.masks) == n and tuple(det.masks.data.shape) == (n, ih, iw)

    def from_device(idx):
        d = pipe.dets[0].cpu()[idx]
        b = d[:, :4].clone()
        b[:, [0, 2]] = b[:, [0, 2]] * pipe.gain + pipe.pad_x
        b[:, [1, 3]] = b[:, [1, 3]] * pipe.gain + pipe.pad_y
        return assemble_masks(pipe.proto[0].cpu(), d[:, 6:38], b, (ih, iw))

    assert torch.equal(det.masks.data, from_device(torch.arange(n)))
    for box, mk in zip(det.boxes, det.masks):                                        # the reference's loop (:196)
        assert len(mk.xy) == 1 and mk.xy[0].dtype == np.float32
        if len(mk.xy[0]):
            assert mk.xy[0][:, 0].min() >= 0 and mk.xy[0][:, 0].max() <= 640 and mk.xy[0][:, 1].min() >= 0 and mk.xy[0][:, 1].max() <= 480
    for _ in range(4):                                                               # the same frame: tracks confirm on the third call
        r = model.track(frames[0], verbose=False, device=0, persist=True)[0]
    if r.boxes.id is not None:
        rows = pipe.out[0].cpu()[:int(pipe.nout[0])]
        di = rows[rows[:, 7] >= 0][:, 7].long()
        assert len(r.masks) == len(r.boxes) == len(di) and torch.equal(r.masks.data, from_device(di))
    model.close()
    outs = []
    for _ in range(2):                                                               # throughput form, twice from scratch: same results
        m2 = YOLO("yolov8n-seg.pt")
        m2.overrides.update(conf=0.5, iou=0.4, agnostic_nms=False, max_det=20)
        res = [r[0] for r in m2.track_stream(iter(frames), batch=2, device=0)]
        assert len(res) == len(frames)
        for r in res:
            assert (r.masks is None and len(r.boxes) == 0) or len(r.masks) == len(r.boxes)
        outs.append([(r.boxes.xyxy.numpy().copy(), [] if r.masks is None else [p.copy() for p in r.masks.xy]) for r in res])
        m2.close()
    for (b0, p0), (b1, p1) in zip(*outs):
        assert np.array_equal(b0, b1) and len(p0) == len(p1) and all(np.array_equal(u, v) for u, v in zip(p0, p1))


@pytest.mark.parametrize("pinned", [False, True])
def test_pack_results_writes_counts_and_rows_into_one_buffer(pinned):
    """ss_pack_results (the per-frame call's result hand-over): counts as int32 bits, n detection rows, m track rows at their fixed
    offsets; rows past the counts are left alone; a pinned host buffer is written by the kernel itself."""
    from tests.gpu_util import engine
    eng = engine(StrongSortConfig())
    dev = eng.device
    g = torch.Generator().manual_seed(5)
    dets = torch.randn(128, 6, generator=g).to(dev)
    out = torch.randn(64, 8, generator=g).to(dev)
    for n, m, with_out in ((0, 0, True), (17, 9, True), (128, 64, True), (500, 70, True), (23, 0, False)):
        nd = torch.tensor([n], dtype=torch.int32, device=dev)
        no = torch.tensor([m], dtype=torch.int32, device=dev)
        dst = torch.full((2 + dets.numel() + out.numel(),), -7.0)
        dst = dst.pin_memory() if pinned else dst.to(dev)
        eng.pack_results(nd, dets, no if with_out else None, out if with_out else None, dst)
        torch.cuda.synchronize()
        h = dst.cpu()
        cn, cm = min(n, 128), (min(m, 64) if with_out else 0)
        assert h[:2].view(torch.int32).tolist() == [cn, cm]
        assert torch.equal(h[2:2 + cn * 6].view(cn, 6), dets.cpu()[:cn])
        assert (h[2 + cn * 6:2 + 128 * 6] == -7.0).all()
        assert torch.equal(h[2 + 128 * 6:2 + 128 * 6 + cm * 8].view(cm, 8), out.cpu()[:cm])
        assert (h[2 + 128 * 6 + cm * 8:] == -7.0).all()
    eng.close()
