"""Full per-frame pipelines on the GPU (synthetic head tensor -> HIP NMS -> crops -> nets -> tracker):
sequential FramePipeline (eager and HIP-graph) and the two-stream OverlappedPipeline must all reproduce the
oracle chain (C NMS + scale_boxes + tracker) bit for bit."""
import numpy as np
import pytest
import torch

from oracle import cexact
from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig
from strongsort_yolo_amd.engine import scale_geometry
from strongsort_yolo_amd.synth import make_stream, synth_prediction

pytestmark = pytest.mark.gpu
H, W, N_IDS, FRAMES = 480, 640, 8, 14


def _workload(pipe):
    gs = scale_geometry(pipe.geom, H, W)
    st, rng = make_stream(31, W, H, N_IDS), np.random.default_rng(31)
    items = []
    for k in range(FRAMES):
        fr = st.next_frame()
        pred, agt = synth_prediction(fr.dets, pipe.n_anchors, pipe.nc, gs[0], (gs[1], gs[2]), rng)
        feats = np.zeros((128, 512), np.float32)
        feats[:len(fr.feats)] = fr.feats
        items.append((st.frame_pixels(k).copy(), pred, agt, feats))
    return gs, items


def _oracle(gs, items, nc):
    dcfg, orc, rows = DetectConfig(), OracleStrongSort(StrongSortConfig(), "c"), []
    for _, pred, agt, feats in items:
        keep, r = cexact.nms(pred, nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, 128)
        r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], W, H)
        rows.append(orc.update(r, feats[np.maximum(agt[keep], 0)], (H, W)))
    return rows


def _fill(b, item, dev):
    img, pred, agt, feats = item
    b.frames[0].copy_(torch.from_numpy(img).to(dev))
    b.pred_in[0].copy_(torch.from_numpy(pred).to(dev))
    b.anchor_gt[0].copy_(torch.from_numpy(agt).to(dev))
    b.gt_feats[0].copy_(torch.from_numpy(feats).to(dev))


@pytest.mark.parametrize("graph", ["none", "front", "all"])
def test_sequential_pipeline_equals_oracle(graph):
    from strongsort_yolo_amd.pipeline import FramePipeline
    pipe = FramePipeline("yolov8n", 1, (H, W), graph=graph, det_source="synthetic", feat_source="by_anchor")
    gs, items = _workload(pipe)
    ref = _oracle(gs, items, pipe.nc)
    for k, it in enumerate(items):
        _fill(pipe, it, pipe.dev)
        pipe.step()
        got = pipe.results()[0]
        assert got.shape == ref[k].shape and got.tobytes() == ref[k].tobytes(), f"{graph}: frame {k}"
    pipe.close()


@pytest.mark.parametrize("graph,n_stages,fb,kw", [("front", 4, 1, {}), ("all", 2, 1, {}), ("front", 2, 1, {}), ("front", 2, 3, {}),
                                                  ("front", 2, 3, {"reid_split": 2}), ("front", 2, 2, {"reid_split": 0}),
                                                  ("front", 2, 2, {"reid_split": 1, "tracker_stream": True}),
                                                  ("front", 2, 3, {"reid_split": 2, "defer_track": True}),
                                                  ("front", 2, 1, {"defer_track": True})])
def test_overlapped_pipeline_equals_oracle(graph, n_stages, fb, kw):
    """fb = 3 with 14 frames: groups of 3,3,3,3 and a partial group of 2.  kw: stage cut inside the ReID backbone,
    tracker on the last stage's stream (default) or on its own stream with three buffer sets; defer_track: the tracker call of
    a group enqueued after the wait for the next group's first stage (three buffer sets, results one group later)."""
    from strongsort_yolo_amd.pipeline import OverlappedPipeline
    pipe = OverlappedPipeline("yolov8n", 1, (H, W), graph=graph, det_source="synthetic", feat_source="by_anchor",
                              n_stages=n_stages, frame_batch=fb, **kw)
    assert pipe.n == n_stages
    assert (pipe.sT is not None) == (graph == "front" and kw.get("tracker_stream", False))
    assert pipe.nb == n_stages + (1 if (pipe.sT is not None or kw.get("defer_track", False)) else 0)
    gs, items = _workload(pipe)
    ref = _oracle(gs, items, pipe.nc)
    out_host = torch.empty(FRAMES, 256, 8).pin_memory()
    n_host = torch.empty(FRAMES, dtype=torch.int32).pin_memory()

    def fetch(idx, f):
        out_host[idx].copy_(pipe.outs[f][0], non_blocking=True)
        n_host[idx].copy_(pipe.nouts[f][0], non_blocking=True)

    pipe.on_result = fetch
    for g0 in range(0, FRAMES, fb):
        n = min(fb, FRAMES - g0)
        b = pipe.begin_frame()
        with torch.cuda.stream(pipe.sA):
            for f in range(n):
                img, pred, agt, feats = items[g0 + f]
                b.frames[f].copy_(torch.from_numpy(img).to(pipe.dev))
                b.pred_in[f].copy_(torch.from_numpy(pred).to(pipe.dev))
                b.anchor_gt[f].copy_(torch.from_numpy(agt).to(pipe.dev))
                b.gt_feats[f].copy_(torch.from_numpy(feats).to(pipe.dev))
        pipe.submit(n)
    pipe.flush()
    torch.cuda.synchronize()
    pipe.eng.check_errors()
    for k in range(FRAMES):
        g = out_host[k, : int(n_host[k])].numpy()
        assert g.shape == ref[k].shape and g.tobytes() == ref[k].tobytes(), f"{graph} fb={fb}: frame {k}"
    pipe.close()


def test_two_streams_in_one_pipeline_equal_their_oracles():
    """S = 2 streams batched in one context / one set of launches (streams are independent)."""
    from strongsort_yolo_amd.pipeline import FramePipeline
    pipe = FramePipeline("yolov8n", 2, (H, W), graph="front", det_source="synthetic", feat_source="by_anchor")
    gs = scale_geometry(pipe.geom, H, W)
    dcfg = DetectConfig()
    streams = [make_stream(50 + s, W, H, N_IDS + 2 * s) for s in range(2)]
    rngs = [np.random.default_rng(50 + s) for s in range(2)]
    orcs = [OracleStrongSort(StrongSortConfig(), "c") for _ in range(2)]
    for k in range(10):
        refs = []
        for s in range(2):
            fr = streams[s].next_frame()
            pred, agt = synth_prediction(fr.dets, pipe.n_anchors, pipe.nc, gs[0], (gs[1], gs[2]), rngs[s])
            feats = np.zeros((128, 512), np.float32); feats[:len(fr.feats)] = fr.feats
            pipe.frames[s].copy_(torch.from_numpy(streams[s].frame_pixels(k)).to(pipe.dev))
            pipe.pred_in[s].copy_(torch.from_numpy(pred).to(pipe.dev))
            pipe.anchor_gt[s].copy_(torch.from_numpy(agt).to(pipe.dev))
            pipe.gt_feats[s].copy_(torch.from_numpy(feats).to(pipe.dev))
            keep, r = cexact.nms(pred, pipe.nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, 128)
            r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], W, H)
            refs.append(orcs[s].update(r, feats[np.maximum(agt[keep], 0)], (H, W)))
        pipe.step()
        got = pipe.results()
        for s in range(2):
            assert got[s].shape == refs[s].shape and got[s].tobytes() == refs[s].tobytes(), f"stream {s} frame {k}"
    pipe.close()


def test_packed_reid_batches_give_the_same_embeddings_and_tracks():
    """feat_source="reid": the group's valid crops packed to the front of the fixed ReID batch (OSNet kernels skip the rest,
    embeddings scattered back by offset) vs one slot range per frame: identical feature rows for the detections and identical
    tracker output."""
    from strongsort_yolo_amd.pipeline import OverlappedPipeline
    res = {}
    for pack in ("1", "0"):
        pipe = OverlappedPipeline("yolov8n", 1, (H, W), graph="front", det_source="synthetic", feat_source="reid",
                                  n_stages=2, frame_batch=4, reid_split=2, reid_batch=32, defer_track=True, pack_crops=pack == "1")
        assert pipe.pack == (pack == "1")
        gs, items = _workload(pipe)
        rows, feats = [], []

        def fetch(idx, f, pipe=pipe, rows=rows):
            n = int(pipe.nouts[f][0].item())
            rows.append((idx, pipe.outs[f][0, :n].cpu().numpy().copy()))

        pipe.on_result = fetch
        for g0 in range(0, FRAMES, 4):
            n = min(4, FRAMES - g0)
            b = pipe.begin_frame()
            with torch.cuda.stream(pipe.sA):
                for f in range(n):
                    img, pred, agt, _ = items[g0 + f]
                    b.frames[f].copy_(torch.from_numpy(img).to(pipe.dev))
                    b.pred_in[f].copy_(torch.from_numpy(pred).to(pipe.dev))
            pipe.submit(n)
            if g0 == 4:                                       # one group's features, once its last stage has run
                pipe.flush(); torch.cuda.synchronize()
                bb = pipe.bufs[1 % pipe.nb]
                nd = bb.ndets[:4].cpu().numpy()
                feats = [bb.feats_v[f, :nd[f]].cpu().numpy().copy() for f in range(4)]
                assert nd.sum() > 8 and nd.max() <= 32
                if pack == "1":
                    off = bb.crop_off.cpu().numpy()
                    assert off[0] == 0 and (np.diff(off[:5]) == nd).all() and off[pipe.Sv] == nd.sum() + off[4] - off[4]
        pipe.flush()
        torch.cuda.synchronize()
        pipe.eng.check_errors()
        res[pack] = (sorted(rows, key=lambda t: t[0]), feats)
        pipe.close()
    assert [i for i, _ in res["1"][0]] == list(range(FRAMES))
    for (i, a), (_, b) in zip(res["1"][0], res["0"][0]):
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), i
    for a, b in zip(res["1"][1], res["0"][1]):
        assert a.shape == b.shape and a.tobytes() == b.tobytes() and np.abs(a).max() > 0


def _run_groups(pipe, items, fb):
    """Push `items` (frame, head tensor, anchor map, features) through an OverlappedPipeline in groups of fb; rows per frame."""
    n_frames = len(items)
    out_host = torch.empty(n_frames, 256, 8).pin_memory()
    n_host = torch.empty(n_frames, dtype=torch.int32).pin_memory()

    def fetch(idx, f):
        out_host[idx].copy_(pipe.outs[f][0], non_blocking=True)
        n_host[idx].copy_(pipe.nouts[f][0], non_blocking=True)

    pipe.on_result = fetch
    for g0 in range(0, n_frames, fb):
        n = min(fb, n_frames - g0)
        b = pipe.begin_frame()
        with torch.cuda.stream(pipe.sA):
            for f in range(n):
                img, pred, agt, feats = items[g0 + f]
                b.frames[f].copy_(torch.from_numpy(img).to(pipe.dev, non_blocking=True))
                b.pred_in[f].copy_(torch.from_numpy(pred).to(pipe.dev, non_blocking=True))
                b.anchor_gt[f].copy_(torch.from_numpy(agt).to(pipe.dev, non_blocking=True))
                b.gt_feats[f].copy_(torch.from_numpy(feats).to(pipe.dev, non_blocking=True))
        pipe.submit(n)
    pipe.flush()
    torch.cuda.synchronize()
    pipe.eng.check_errors()
    return [out_host[k, : int(n_host[k])].numpy().copy() for k in range(n_frames)]


@pytest.mark.parametrize("detector,w,h,n_ids,reid_batch,n_frames,split", [
    ("yolov8n", 1280, 720, 30, 32, 176, 6),    # bench.py default = BASELINE configs[1]: 5 groups of 32 + a partial group of 16 (cut after OSNet part 6)
    ("yolov7", 1920, 1080, 100, 128, 80, 2),   # bench.py --preset c4 = configs[3]: 2 groups of 32 + 16
    ("yolov8s", 1280, 720, 30, 32, 144, 2),    # --preset c3 = configs[2] per GPU: the stage cut the larger detectors keep (OSNet part 2)
    ("yolov8n-pose", 1280, 720, 30, 32, 144, 5),   # --preset c5 = configs[4] per GPU: 51 keypoint columns ride through NMS with the kept rows
    ("yolov5n", 640, 480, 8, 32, 80, 4),       # --preset c1 = configs[0]'s shape (the reference's CPU-runnable case) on the GPU path
    ("yolo11n-pose", 1280, 720, 30, 32, 112, 2),   # --preset c6: the reference's default weights file (yolo_multi_model.py:17), C3k2 / C2PSA graph, pose head
])
def test_benchmarked_configuration_equals_oracle(detector, w, h, n_ids, reid_batch, n_frames, split):
    """The f16-ReID form of every preset (bench.py --reid-f16, the `throughput_mode` of the default line)."""
    _benchmarked(detector, w, h, n_ids, reid_batch, n_frames, split)


@pytest.mark.parametrize("detector,w,h,n_ids,reid_batch,n_frames", [
    ("yolov8n", 1280, 720, 30, 32, 176),       # bench.py's default line since round 6: configs[1] with the fp32 ReID network on its own kernels
    ("yolov7", 1920, 1080, 100, 128, 80),      # --preset c4
    ("yolov8n-pose", 1280, 720, 30, 32, 112),  # --preset c5
])
def test_benchmarked_default_configuration_fp32_reid_equals_oracle(detector, w, h, n_ids, reid_batch, n_frames):
    """bench.py's headline configuration: the same pipeline with reid_half=False (crops + OSNet-x0.25 in fp32 on csrc/ss_ops32.hip) and
    the stage cut bench.REID_SPLIT_FP32 names."""
    import bench
    preset = {v[0]: k for k, v in bench.PRESETS.items()}[detector]
    _benchmarked(detector, w, h, n_ids, reid_batch, n_frames, bench.REID_SPLIT_FP32[preset], reid_half=False)


def test_detached_tracker_chain_on_reserved_compute_units_equals_oracle():
    """Pipeline option chain_cus=16 (library option "chain_cus", ss_stream_create, ss_track_join): the per-frame chain on a stream that
    owns two compute units of every XCD, the pipeline's streams without them, rows fetched on a results stream that joined the chain —
    the same bytes as the oracle (the form is off by default: profiles/r04_chain_cus_ab.txt)."""
    _benchmarked("yolov5n", 640, 480, 8, 32, 80, 4, chain_cus=16)


def _benchmarked(detector, w, h, n_ids, reid_batch, n_frames, split, **pipe_kw):
    """Exactly what bench.py times: frame batch 32, stage cut inside OSNet where bench.REID_SPLIT puts it, deferred tracker call + association gate,
    packed ReID crops, galleries filling up to nn_budget rows — every frame tobytes()-equal to the oracle chain
    (VERDICT r2 'next' item 1, r3 'next' item 1a; arithmetic behind /root/reference/yolo_multi_model.py:41).  Pose head: the keypoint
    columns of the group still in the buffers equal the head tensor's columns at the oracle's keep indices (carried by det_idx)."""
    import bench
    from strongsort_yolo_amd.pipeline import OverlappedPipeline
    preset = {v[0]: k for k, v in bench.PRESETS.items()}[detector]
    reid_half = pipe_kw.pop("reid_half", True)
    assert (bench.REID_SPLIT if reid_half else bench.REID_SPLIT_FP32)[preset] == split and bench.PRESETS[preset][1:] == (w, h, n_ids, reid_batch)
    pipe = OverlappedPipeline(detector, 1, (h, w), half=True, reid_batch=reid_batch, det_source="synthetic", reid_half=reid_half,
                              feat_source="by_anchor", graph="front", n_stages=2, frame_batch=32, reid_split=split, defer_track=True, **pipe_kw)
    assert pipe.reid_half == reid_half
    assert pipe.pack and pipe.defer and pipe.assoc_ev is not None and pipe.nb == 3 and pipe.eng.max_group_frames == 32
    assert (pipe.sR is not None) == bool(pipe_kw.get("chain_cus"))
    gs = scale_geometry(pipe.geom, h, w)
    st, rng = make_stream(77, w, h, n_ids), np.random.default_rng(77)
    items, dcfg, orc = [], DetectConfig(), OracleStrongSort(StrongSortConfig(), "c")
    ref, keeps = [], []
    nc, nk = pipe.nc, pipe.nk
    assert nk == (51 if "pose" in detector else 0)
    for k in range(n_frames):
        fr = st.next_frame()
        pred, agt = synth_prediction(fr.dets, pipe.n_anchors, nc, gs[0], (gs[1], gs[2]), rng)
        if nk:
            pred = np.concatenate([pred, rng.uniform(0, 640, (nk, pipe.n_anchors)).astype(np.float32)])
        feats = np.zeros((128, 512), np.float32)
        feats[:len(fr.feats)] = fr.feats
        items.append((st.frame_pixels(k), pred, agt, feats))
        keep, r = cexact.nms(pred[:4 + nc], nc, dcfg.conf, dcfg.iou, dcfg.agnostic_nms, dcfg.max_wh, dcfg.max_nms, 128)
        r = cexact.scale_boxes(r, gs[0], gs[1], gs[2], w, h)
        ref.append(orc.update(r, feats[np.maximum(agt[keep], 0)], (h, w)))
        keeps.append(keep)
    got = _run_groups(pipe, items, 32)
    if nk:                                            # the last group is still in its buffer set
        b, nv = pipe.bufs[(pipe.k - 1) % pipe.nb], n_frames % 32 or 32
        dets, nd = b.dets[:nv].cpu().numpy(), b.ndets[:nv].cpu().numpy()
        for f in range(nv):
            k = n_frames - nv + f
            assert nd[f] == len(keeps[k]) > 0
            assert dets[f, :nd[f], 6:6 + nk].tobytes() == np.ascontiguousarray(items[k][1][4 + nc:, keeps[k]].T).tobytes(), f"keypoints of frame {k}"
    pipe.close()
    assert sum(len(r) for r in ref[-16:]) >= 16 * n_ids * 0.7          # confirmed tracks are reported in the last (partial) group
    for k in range(n_frames):
        assert got[k].shape == ref[k].shape and got[k].tobytes() == ref[k].tobytes(), f"{detector}: frame {k}"
