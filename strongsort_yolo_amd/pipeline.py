"""Per-frame tracking pipeline for the S streams owned by one GPU (one process per GPU).

    frame (u8, HBM) -> letterbox [HIP] -> detector [own HIP conv kernels behind nets.py modules] -> NMS [HIP]
      -> ReID crops [HIP] -> OSNet [own HIP kernels] -> StrongSORT update [HIP: k_group_prep / k_assoc / k_frame / k_post / k_newrow]

This is what runs inside `model.track(frame, persist=True)` (/root/reference/yolo_multi_model.py:41).
Everything between the frame and the output rows stays on the device: detection counts, track
counts and assignment results are device-side values, so the whole step is a fixed launch sequence
that is captured once into a HIP graph and replayed per frame (no host round trip, no per-frame
allocation).

Synthetic-workload switches (no weights / no decoder exist offline):
  det_source = "detector"  NMS consumes the detector head output (true end to end)
             = "synthetic" NMS consumes `pred_in` filled by the caller (the detector still runs)
  feat_source = "reid"     the tracker consumes the OSNet embeddings
              = "injected" the tracker consumes `feats_in` filled by the caller (OSNet still runs)
              = "by_anchor" the tracker consumes gt_feats[anchor_gt[keep]]: the feature of the
                            synthetic identity whose anchor survived NMS (OSNet still runs)
  graph = "all"   whole step in one HIP graph;  "front" everything up to the ReID output in the
          graph and the tracker kernels launched eagerly (lets bench.py bracket the association
          kernel with HIP events);  "split" three graphs (detection | ReID | tracker) so that a
          detection-only call replays only the first;  "none" eager.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .config import StrongSortConfig, DetectConfig
from .engine import TrackerEngine, letterbox_geometry, scale_geometry
from .lib import MAX_DETS, MAX_TRACKS, FEAT_DIM
from . import nets


class FramePipeline:
    def __init__(self, detector: str = "yolov8n", n_streams: int = 1, frame_hw=(720, 1280), device: int = 0,
                 half: bool = True, reid_batch: int = 32, cfg: Optional[StrongSortConfig] = None,
                 dcfg: Optional[DetectConfig] = None, det_source: str = "detector", feat_source: str = "reid",
                 graph: str = "all", debug: bool = False, run_nets: bool = True, seed: int = 0, detect_only_rows: int = 0, cmc: bool = False,
                 reid_half: Optional[bool] = None, crops_u8: bool = True):
        self.cfg, self.dcfg = cfg or StrongSortConfig(), dcfg or DetectConfig()
        self.S, (self.H, self.W) = n_streams, frame_hw
        self.eng = TrackerEngine(self.cfg, n_streams, device, debug=debug)
        dev = self.dev = self.eng.device
        self.half, self.dtype = half, torch.float16 if half else torch.float32
        # reid_half=False: the ReID crops and OSNet in fp32 on the hand-written fp32 kernels (fused32.py, csrc/ss_ops32.hip:
        # v_mfma_f32_16x16x4_f32) beside a half detector — the accuracy mode: appearance distances then agree with a CPU fp32
        # OSNet to ~1e-5, which f16 activations do not (bench.py reid_f16_vs_f32; north_star's 1e-4 bound on float distances;
        # the reference passes no half=, yolo_multi_model.py:41).  Default: as the detector.
        self.reid_half = half if reid_half is None else bool(reid_half)
        self.reid_dtype = torch.float16 if self.reid_half else torch.float32
        # fp32 ReID: the crops cross HBM as BYTES (the rounded bilinear values; SS_DST_U8) and the fp32 stem applies /255, mean and std while
        # it stages them - the same floats, a quarter of the bytes (crops_u8=False: A/B switch, float crops)
        from . import fused32 as _f32
        self.crops_dtype = torch.uint8 if (not self.reid_half and crops_u8 and run_nets and _f32.ENABLED) else self.reid_dtype
        self.det_source, self.feat_source, self.run_nets = det_source, feat_source, run_nets
        self.RB = reid_batch
        if reid_batch > MAX_DETS:
            raise ValueError("reid_batch <= 128")
        # Only the first reid_batch detections of a frame are cropped and embedded: with OSNet features feeding the
        # tracker, NMS keeps at most that many (highest scores first), so no detection reaches it without a feature.
        self.max_det = min(self.dcfg.max_det, MAX_DETS, reid_batch if (feat_source == "reid" and run_nets) else MAX_DETS)
        # detect_only_rows > 0: a detection-only pipeline (model.predict, yolo_multi_model.py:173) whose NMS keeps up to
        # that many rows (<= 1024, the reference's max_det is 1000) — it never feeds the tracker (128 detections per frame)
        self.det_rows = MAX_DETS
        if detect_only_rows:
            self.det_rows = min(int(detect_only_rows), 1024)
            self.max_det = min(self.dcfg.max_det, self.det_rows)
        self.geom = letterbox_geometry(self.H, self.W, self.dcfg.imgsz, self.dcfg.stride)
        self.gain, self.pad_x, self.pad_y = scale_geometry(self.geom, self.H, self.W)
        self.detector = self.reid = None
        if run_nets:
            self.detector = nets.build_detector(detector, seed).to(dev, self.dtype).to(memory_format=torch.channels_last)
            self.reid = nets.build_reid(seed + 1).to(dev, self.reid_dtype).to(memory_format=torch.channels_last)
            self.nc, self.nk = self.detector.nc, self.detector.nk
            self.nm = getattr(self.detector, "nm", 0)           # mask coefficients of a segmentation head (after the keypoints' place)
        else:
            self.nc, self.nk, self.nm = 80, 0, 0
        self.nx = self.nk + self.nm                             # extra columns NMS carries along with every kept row
        g, S = self.geom, n_streams
        self.n_anchors = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
        # ---- static buffers (addresses are baked into the graph) ----
        self.frames = torch.zeros(S, self.H, self.W, 3, dtype=torch.uint8, device=dev)
        self.lb = torch.zeros(S, 3, g.out_h, g.out_w, dtype=self.dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.geom_dev = torch.tensor([[self.gain, self.pad_x, self.pad_y, float(self.W), float(self.H)]] * S,
                                     dtype=torch.float32, device=dev)      # per-image scale_boxes geometry for ss_nms_batch
        self.pred_in = torch.zeros(S, 4 + self.nc + self.nx, self.n_anchors, dtype=torch.float32, device=dev)
        self.dets = torch.zeros(S, self.det_rows, 6 + self.nx, dtype=torch.float32, device=dev)
        self.dets6 = self.dets if self.nx == 0 else torch.zeros(S, self.det_rows, 6, dtype=torch.float32, device=dev)
        # segmentation: the frame's mask prototypes [nm, out_h/4, out_w/4] stay here until the caller has built its Results
        self.proto = torch.zeros(S, self.nm, g.out_h // 4, g.out_w // 4, dtype=self.dtype, device=dev) if self.nm else None
        self.keep = torch.zeros(S, self.det_rows, dtype=torch.int32, device=dev)
        self.ndets = torch.zeros(S, dtype=torch.int32, device=dev)
        self.crops = torch.zeros(S * self.RB, 3, 256, 128, dtype=self.crops_dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.feats_in = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)
        self.img_hw = torch.tensor([[self.H, self.W]] * S, dtype=torch.int32, device=dev)
        self.out, self.nout = self.eng.out, self.eng.nout
        self.anchor_gt = torch.zeros(S, self.n_anchors, dtype=torch.int64, device=dev)
        self.gt_feats = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)
        self.graph = None
        self.graph_all = None
        self.graph_mode = graph
        # N4 (optional): ECC camera-motion warps estimated beside the detector, applied by the tracker before predicting
        self.cmc = bool(cmc)
        self.warps = torch.zeros(1, S, 8, dtype=torch.float64, device=dev) if self.cmc else None
        if self.cmc:
            self.eng.cmc_estimate(self.frames, 1, self.warps)            # sizes the small-frame buffer outside any capture
            self.eng.reset(-1)
            self.eng.set_cmc(self.warps)

    # ---- one frame for every stream, from the static buffers ----------------------------------------
    def _detect_impl(self):
        # every frame-side stage is ONE launch (set) over all S streams, written straight in the NHWC layout the
        # convolutions read
        e, g = self.eng, self.geom
        if self.cmc:
            e.cmc_estimate(self.frames, 1, self.warps)
        if self.run_nets:
            e.letterbox_batch(self.frames, g, half=self.half, pad_value=self.dcfg.pad_value, out=self.lb, channels_last=True)
            with self._decode_into(self.pred_in):
                pred = self._pred(self.detector(self.lb), self.proto)    # [S, 4+nc+nk+nm, A]
            if self.det_source == "detector" and pred.data_ptr() != self.pred_in.data_ptr():
                self.pred_in.copy_(pred)
        e.nms_batch(self.pred_in, self.nc, self.dcfg, self.geom_dev, n_extra=self.nx, rows=self.dets, keep=self.keep,
                    count=self.ndets, max_det=self.max_det)
        if self.nx:
            self.dets6.copy_(self.dets[:, :, :6])

    def _decode_into(self, buf):
        """The head's decode launch writes the NMS input in place (fused.decode_into) when the detector feeds the NMS."""
        import contextlib
        from . import fused
        return fused.decode_into(buf) if self.det_source == "detector" else contextlib.nullcontext()

    @staticmethod
    def _pred(out, proto_dst):
        """Detector output -> the row tensor; a segmentation head's prototypes go to their static buffer."""
        if isinstance(out, tuple):
            out, proto = out
            proto_dst.copy_(proto)
        return out

    def _reid_impl(self):
        e, S = self.eng, self.S
        if self.run_nets:
            e.crop_norm_batch(self.frames, self.dets6, self.RB, counts=self.ndets, half=self.reid_half, out=self.crops,
                              channels_last=True)
            emb = self.reid(self.crops)                         # [S*RB, 512]
            if self.feat_source == "reid":
                self.feats_in[:, :self.RB].copy_(emb.view(S, self.RB, FEAT_DIM))
        if self.feat_source == "by_anchor":
            idx = self.anchor_gt.gather(1, self.keep.long().clamp_(0, self.n_anchors - 1))      # [S,128]
            torch.gather(self.gt_feats, 1, idx.clamp_(min=0).unsqueeze(-1).expand(-1, -1, FEAT_DIM), out=self.feats_in)

    def _step_impl(self):
        self._detect_impl()
        self._reid_impl()

    def _track(self):
        self.eng.update_device(self.dets6, self.ndets, self.feats_in, self.img_hw)

    @torch.no_grad()
    def step(self, track: bool = True):
        """Run one frame (all streams).  Asynchronous; results in self.out / self.nout (device).
        track=False (graph "none" / "split" only skip the work): detection only — letterbox, detector, NMS."""
        if self.det_rows != MAX_DETS:
            if track:
                raise RuntimeError("a detect_only_rows pipeline cannot track")
            if self.graph_mode != "none":
                if self.graph is None:                       # one graph: letterbox, detector, NMS
                    st = torch.cuda.Stream(self.dev)
                    st.wait_stream(torch.cuda.current_stream(self.dev))
                    with torch.cuda.stream(st):
                        self.eng.use_current_stream()
                        for _ in range(3):
                            self._detect_impl()
                        st.synchronize()
                        self.graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.graph, stream=st):
                            self._detect_impl()
                    torch.cuda.current_stream(self.dev).wait_stream(st)
                    self.eng.use_current_stream()
                self.graph.replay()
                return
        if self.graph_mode == "none":
            self._detect_impl()
            if track:
                self._reid_impl()
                self._track()
            return
        if self.graph is None:
            # warm up on a side stream (MIOpen find, allocator), then capture
            st = torch.cuda.Stream(self.dev)
            st.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(st):
                self.eng.use_current_stream()
                for _ in range(3):
                    self._step_impl()
                    self._track()
                st.synchronize()
                self._restore_tracker()
                if self.graph_mode == "split":
                    # three graphs on the same static buffers: detection | ReID | tracker, so a detection-only call
                    # (model.predict, yolo_multi_model.py:173) replays just the first
                    self.graph = []
                    for fn in (self._detect_impl, self._reid_impl, self._track):
                        gph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gph, stream=st):
                            fn()
                        self.graph.append(gph)
                    # ... and the three stages once more as ONE graph for the tracking call (model.track, yolo_multi_model.py:41 / :278: the
                    # only call the reference makes per frame): one replay instead of three, no launch boundary between the stages
                    self.graph_all = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_all, stream=st):
                        self._detect_impl()
                        self._reid_impl()
                        self._track()
                else:
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph, stream=st):
                        self._step_impl()
                        if self.graph_mode == "all":
                            self._track()
            torch.cuda.current_stream(self.dev).wait_stream(st)
            self.eng.use_current_stream()
            self._restore_tracker()          # warm-up frames must not count: streams start fresh
        if self.graph_mode == "split":
            (self.graph_all if track else self.graph[0]).replay()
            return
        self.graph.replay()
        if self.graph_mode == "front":
            self._track()

    # the warm-up iterations must not advance the tracker: reset it (callers start streams fresh)
    def _restore_tracker(self):
        torch.cuda.synchronize(self.dev)
        self.eng.reset(-1)

    # ---- convenience ----------------------------------------------------------------------------------
    def results(self):
        """Synchronise and return per-stream rows (numpy [M,8])."""
        torch.cuda.synchronize(self.dev)
        self.eng.check_errors()
        out, n = self.out.cpu().numpy(), self.nout.cpu().numpy()
        return [out[s, :n[s]].copy() for s in range(self.S)]

    def detections(self):
        torch.cuda.synchronize(self.dev)
        d, n = self.dets.cpu().numpy(), self.ndets.cpu().numpy()
        return [d[s, :n[s]].copy() for s in range(self.S)]

    def close(self):
        torch.cuda.synchronize(self.dev)
        self.graph = None
        self.graph_all = None
        if hasattr(self, "graphs"):
            self.graphs = None                           # captured graphs reference the context's buffers: drop them first
        self.eng.close()


def _p(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


class _Bufs:
    """One set of per-frame buffers (static addresses, baked into that set's HIP graphs)."""

    def __init__(self, p: "FramePipeline"):
        dev, S, g = p.dev, getattr(p, "Sv", p.S), p.geom
        self.frames = torch.zeros(S, p.H, p.W, 3, dtype=torch.uint8, device=dev)
        self.lb = torch.zeros(S, 3, g.out_h, g.out_w, dtype=p.dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.pred_in = torch.zeros(S, 4 + p.nc + p.nx, p.n_anchors, dtype=torch.float32, device=dev)
        self.dets = torch.zeros(S, MAX_DETS, 6 + p.nx, dtype=torch.float32, device=dev)
        self.dets6 = self.dets if p.nx == 0 else torch.zeros(S, MAX_DETS, 6, dtype=torch.float32, device=dev)
        self.proto = torch.zeros(S, p.nm, p.geom.out_h // 4, p.geom.out_w // 4, dtype=p.dtype, device=dev) if p.nm else None
        self.keep = torch.zeros(S, MAX_DETS, dtype=torch.int32, device=dev)
        self.ndets = torch.zeros(S, dtype=torch.int32, device=dev)
        self.crops = torch.zeros(S * p.RB, 3, 256, 128, dtype=p.crops_dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.anchor_gt = torch.zeros(S, p.n_anchors, dtype=torch.int64, device=dev)
        self.gt_feats = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)
        self.feats_v = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)    # what the tracker reads
        self.crop_off = torch.zeros(S + 1, dtype=torch.int32, device=dev)                     # packed ReID batch: first crop of every image, total
        self.warps = torch.zeros(getattr(p, "F", 1), p.S, 8, dtype=torch.float64, device=dev) if getattr(p, "cmc", False) else None
        self.nvalid = torch.full((1,), getattr(p, "F", 1), dtype=torch.int32, device=dev)    # real frames of the group in this set (device side: read by captured kernels)


class OverlappedPipeline(FramePipeline):
    """Software pipeline over the frames of the same streams: stage j of frame k-j runs concurrently with the
    other stages on its own HIP stream,

        stage 0  letterbox -> detector backbone                                frame k
        stage 1  detector neck + head (-> head tensor)                         frame k-1
        stage 2  NMS -> ReID crops -> OSNet (first half)                       frame k-2
        stage 3  OSNet (second half) -> feature select -> StrongSORT update    frame k-3

    (stages 0/1 and 2/3 merge when the network has no split entry points, or with `n_stages=2`).  The
    detector and OSNet are stateless, so only the last stage carries the tracker recurrence, and it sees the
    frames strictly in order.  Results are identical to FramePipeline (same kernels, same order per frame);
    a frame still takes the sum of the stage times to come out, but throughput approaches the longest stage
    because the GPU is otherwise idle between the ~4 us launches of these batch-1 networks.  Every stage of
    every buffer set is one captured HIP graph; buffer sets = stages.  Measured at configs[1] on MI355X: 1 stage
    427, 2 stages 761, 4 stages 659 frames/s (four concurrent launch chains contend in the dispatcher), so the
    default is 2: [letterbox, detector] | [NMS, crops, OSNet, select, tracker]; `reid_split=k` moves the cut to
    after part k of the ReID backbone ([letterbox, detector, NMS, crops, OSNet parts <k] | [rest, tracker]) so
    the two streams carry equal work when frames are batched.
    """

    def __init__(self, *a, n_stages: int = 2, frame_batch: int = 1, reid_split: Optional[int] = None,
                 tracker_stream: bool = False, defer_track: bool = False, keep_net_outputs: bool = False, pack_crops: bool = True,
                 assoc_gate: bool = True, track_priority: bool = False, skip_tracker: bool = False, chain_cus: int = 0, **kw):
        kw = dict(kw)
        # keep_net_outputs: every buffer set keeps a reference to the head tensor and the embeddings its graphs produce
        # (b.head_out / b.emb_out: tensors of the graph's private pool, same address at every replay) even when the synthetic
        # workload does not consume them — bench.py compares them with an eager re-run after the timed region
        self.keep_net_outputs = bool(keep_net_outputs)
        self.skip_tracker = bool(skip_tracker)      # MEASUREMENT ONLY: no tracker calls (rows stay empty) — what the stateless stages alone cost
        kw["graph"] = kw.get("graph", "front")
        if kw["graph"] == "none":
            raise ValueError("OverlappedPipeline needs graph='front' or 'all'")
        super().__init__(*a, **kw)
        # frame batching: F consecutive frames of every stream travel through the stateless stages together as
        # S*F "virtual streams" (index f*S + s); the tracker then consumes them one frame at a time, in order.
        self.F = int(frame_batch)
        if self.F < 1:
            raise ValueError("frame_batch >= 1")
        if self.F > 1:
            self.graph_mode = "front"       # the per-frame tracker calls stay eager (partial last groups)
        self.Sv = self.S * self.F
        # packed ReID batches: the group's valid crops contiguous, the OSNet kernels skip the unused slots of the fixed batch
        # (~28 of 32 slots per frame are used at configs[1]); pack_crops=False: A/B switch
        self.pack = bool(self.run_nets and pack_crops)
        self.geom_dev = self.geom_dev[:1].repeat(self.Sv, 1).contiguous()
        self.outs = torch.zeros(self.F, self.S, MAX_TRACKS, 8, dtype=torch.float32, device=self.dev)
        self.nouts = torch.zeros(self.F, self.S, dtype=torch.int32, device=self.dev)
        split_det = self.run_nets and n_stages >= 4 and hasattr(self.detector, "forward_backbone")
        split_reid = self.run_nets and n_stages >= 4 and hasattr(self.reid, "forward_a")
        # two stages can also be cut INSIDE the ReID network to balance them: stage 0 = letterbox, detector, NMS,
        # crops and OSNet parts [0, reid_split); stage 1 = the remaining parts, feature select and the tracker
        self.reid_split = None
        if reid_split is not None and n_stages == 2:
            self.reid_split = int(reid_split) if (self.run_nets and hasattr(self.reid, "N_PARTS")) else 0
            if not 0 <= self.reid_split <= getattr(self.reid, "N_PARTS", 0):
                raise ValueError("reid_split out of range")
        st = []
        if self.reid_split is not None:
            st += [self._s_front_split, self._s_back_split]
        elif split_det:
            st += [self._s_backbone, self._s_head]
        else:
            st += [self._s_detector]
        if self.reid_split is not None:
            pass
        elif split_reid:
            st += [self._s_nms_crop_reid_a, self._s_reid_b_select]
        else:
            st += [self._s_nms_crop_reid_select]
        self.stages = st
        self.n = len(st)
        # track_priority=True gives the last stage's stream (the tracker's short dependent launches, the association kernel) high
        # priority.  OFF by default since round 6: with the association launch alone on the chip (assoc_gate) the first two pipelines
        # of a process run the same with and without it (13 700 / 8 900 frames/s, f16 / fp32 ReID), and from the THIRD pipeline a process
        # creates on, the high-priority stream makes things worse — the tracker call takes 0.9-3.1 ms instead of 1.4-1.6 and the step
        # 3.3-3.8 ms instead of 2.3 (f16), 4.4-5.0 instead of 3.6 (fp32): tools/pipeline_reuse.py.  (An application re-creates its
        # pipeline whenever `model.overrides` change; bench.py's side legs did.)
        hi = bool(track_priority)
        # chain_cus = n > 0: the tracker's per-frame chain (one- to 64-workgroup kernels, ~27 us a frame, strictly dependent) is
        # detached onto a library stream that owns n compute units (n / 8 per XCD), every stream of this pipeline is created without
        # them, and the rows are fetched on a results stream that waits for the chain: the chain then runs beside the networks
        # instead of between their kernels in the dispatcher.  (CU-masked streams have no priority: track_priority does not apply.)
        # chain_cus = -1: detached onto a plain high-priority stream, nothing reserved.
        self.chain_cus = int(chain_cus)
        if self.chain_cus and tracker_stream:
            # the tracker-stream branch releases a buffer set on that stream, not behind the detached chain's results stream: the
            # chain could still be reading the set's detections / features when stage 0 refills it
            self.eng.close()                                     # (the tracker context exists already: do not leak it)
            raise ValueError("tracker_stream and chain_cus cannot be combined")
        if self.chain_cus:
            self.eng.set_option("chain_cus", self.chain_cus)
        if self.chain_cus > 0:
            self.streams = [self.eng.create_stream(self.chain_cus) for _ in range(self.n)]
        else:
            self.streams = [torch.cuda.Stream(self.dev, priority=(-1 if (hi and j == self.n - 1) else 0)) for j in range(self.n)]
        self.sR = torch.cuda.Stream(self.dev) if self.chain_cus else None      # result copies / buffer-set release of a detached chain
        self._res_ev = None
        # tracker_stream=True gives the tracker (one-workgroup kernels, ~70 us a frame) a stream of its own next to the
        # last stage of the following group, with one more buffer set so that stage 0 does not wait for it.  Measured
        # at configs[1], same box, frame batch 8: 3110 frames/s with it vs 3280 without — a third concurrent launch
        # chain costs more in the dispatcher than the queue stall it removes (same finding as the 4-stage split) — so
        # it is off by default.
        self.sT = torch.cuda.Stream(self.dev, priority=-1 if hi else 0) if (tracker_stream and self.graph_mode == "front") else None
        # defer_track: the tracker call of group k is enqueued on the last stage's stream AFTER that stream has waited for
        # stage 0 of group k+1 — it then runs beside the START of the other stream's next group (letterbox, the detector's
        # short-lived workgroups) instead of beside whatever that stream happens to be in.  The association kernel needs
        # whole CUs' worth of registers at once; beside the OSNet row-stream kernel (waves that live ~100-200 us and own the
        # register file) its launch took 50-83 us instead of 30.  Results arrive one group later; one more buffer set.
        self.defer = bool(defer_track) and self.sT is None and self.graph_mode == "front"
        self._pending_track = None
        # ... and stage 0 of the next group waits for that call's association launch (not for the per-frame chain after it):
        # for ~40 us the association kernel has the chip to itself (alone: 39.8 us for 32 frames, beside the detector's first
        # layers: 61-65 us)
        self.assoc_ev = None
        if self.defer and assoc_gate:
            self.assoc_ev = torch.cuda.Event()
            self.assoc_ev.record(torch.cuda.current_stream(self.dev))
            self.eng.set_assoc_event(self.assoc_ev)
        self._gate = False
        self.nb = self.n + (1 if (self.sT is not None or self.defer) else 0)              # buffer sets
        self.bufs = [_Bufs(self) for _ in range(self.nb)]
        self.ev = [[torch.cuda.Event() for _ in range(self.nb)] for _ in range(self.n)]   # ev[stage][set]
        self.ev_graph = [torch.cuda.Event() for _ in range(self.nb)]                      # last stage's graph done (tracker may start)
        self.graphs = [[None] * self.nb for _ in range(self.n)]                           # graphs[stage][set]
        self.sA = self.streams[0]                                  # first stage's stream
        # input stream: the caller fills a buffer set's inputs here (`with torch.cuda.stream(pipe.s_in)`).  Its own stream, so
        # that the copies of group k+1 run as soon as their buffer set is free — beside the networks' kernels of the groups in
        # flight — instead of at the head of stage 0's stream, which is exactly when the deferred tracker call of an older group
        # launches its association kernel (the copies' blit kernels and k_assoc then share the chip: on some boxes / runs the
        # association launch took 86 us instead of 38).  Filling on `sA` stays correct (it is ordered before stage 0).
        self.s_in = self.eng.create_stream(self.chain_cus) if self.chain_cus > 0 else torch.cuda.Stream(self.dev)
        self.ev_in = [torch.cuda.Event() for _ in range(self.nb)]
        self.sB = self.sT if self.sT is not None else self.streams[-1]                    # tracker + result stream
        self.k = 0                          # groups (of frame_batch frames) submitted
        self.stage_done = [0] * self.n      # groups enqueued per stage
        self.valid = [self.F] * self.nb     # real frames in the group occupying each buffer set
        self.base = [0] * self.nb           # frame index of the first frame of that group
        self.frames_in = 0
        self._captured = False

    # ---- stage bodies (b = the frame's buffer set) -------------------------------------------------------
    def _letterbox(self, b):
        if self.cmc:                                           # the group's F warps, beside the detector (stateless stage)
            self.eng.cmc_estimate(b.frames, self.F, b.warps, n_valid=b.nvalid)    # a partial group: the last REAL frame becomes "previous"
        self.eng.letterbox_batch(b.frames, self.geom, half=self.half, pad_value=self.dcfg.pad_value, out=b.lb,
                                 channels_last=True)

    @staticmethod
    def _keep(b, name, tensors):
        """Stage outputs that cross a graph boundary live in per-set static buffers."""
        cur = getattr(b, name, None)
        if cur is None:
            cur = [torch.empty_like(t) for t in tensors]
            setattr(b, name, cur)
        for d, t in zip(cur, tensors):
            d.copy_(t)

    def _s_backbone(self, b):
        if self.run_nets:
            self._letterbox(b)
            self._keep(b, "pyr", self.detector.forward_backbone(b.lb))

    def _s_head(self, b):
        with self._decode_into(b.pred_in):
            pred = self._pred(self.detector.forward_head(*b.pyr), b.proto)
        if self.keep_net_outputs:
            b.head_out = pred
        if self.det_source == "detector" and pred.data_ptr() != b.pred_in.data_ptr():
            b.pred_in.copy_(pred)

    def _s_detector(self, b):
        if self.run_nets:
            self._letterbox(b)
            with self._decode_into(b.pred_in):
                pred = self._pred(self.detector(b.lb), b.proto)
            if self.keep_net_outputs:
                b.head_out = pred
            if self.det_source == "detector" and pred.data_ptr() != b.pred_in.data_ptr():
                b.pred_in.copy_(pred)

    def _nms_crop(self, b):
        e = self.eng                       # one launch set over all S*F virtual streams
        e.nms_batch(b.pred_in, self.nc, self.dcfg, self.geom_dev, n_extra=self.nx, rows=b.dets, keep=b.keep,
                    count=b.ndets, max_det=self.max_det)
        if self.nx:
            b.dets6.copy_(b.dets[:, :, :6])
        if self.run_nets:
            if self.pack:        # the group's valid crops contiguous; the ReID kernels skip the rest of the fixed-size batch
                e.crop_norm_packed(b.frames, b.dets6, self.RB, b.ndets, b.crop_off, b.crops, half=self.reid_half)
            else:
                e.crop_norm_batch(b.frames, b.dets6, self.RB, counts=b.ndets, half=self.reid_half, out=b.crops, channels_last=True)

    def _valid(self, b):
        from . import fused, fused32
        if not self.reid_half:                               # fp32 kernels take the count as a launch argument
            return fused32.valid_images(b.crop_off[self.Sv:] if self.pack else None)
        return fused.valid_images(b.crop_off[self.Sv:] if self.pack else None, self.Sv * self.RB)

    def _select(self, b, emb):
        if emb is not None and self.keep_net_outputs:
            b.emb_out = emb
        if emb is not None and self.feat_source == "reid":
            if self.pack:
                self.eng.unpack_feats(emb.contiguous(), b.crop_off, b.ndets, self.RB, b.feats_v)
            else:
                b.feats_v[:, :self.RB].copy_(emb.view(self.Sv, self.RB, FEAT_DIM))
        if self.feat_source == "by_anchor":
            idx = b.anchor_gt.gather(1, b.keep.long().clamp_(0, self.n_anchors - 1))
            torch.gather(b.gt_feats, 1, idx.clamp_(min=0).unsqueeze(-1).expand(-1, -1, FEAT_DIM), out=b.feats_v)

    def _s_nms_crop_reid_a(self, b):
        self._nms_crop(b)
        with self._valid(b):
            st = self.reid.forward_a(b.crops)                # a tensor or a tuple of tensors (nets.OSNet._block_part)
        b.mid_tuple = isinstance(st, tuple)
        self._keep(b, "mid", list(st) if b.mid_tuple else [st])

    def _s_reid_b_select(self, b):
        with self._valid(b):
            emb = self.reid.forward_b(tuple(b.mid) if b.mid_tuple else b.mid[0])
        self._select(b, emb)

    def _s_nms_crop_reid_select(self, b):
        self._nms_crop(b)
        emb = None
        if self.run_nets:
            with self._valid(b):
                emb = self.reid(b.crops)
        self._select(b, emb)

    def _s_front_split(self, b):
        self._s_detector(b)
        self._nms_crop(b)
        if self.run_nets and self.reid_split > 0:
            # no copy across the stage boundary: the tensor lives in this graph's private pool, keeps its address over
            # replays, and the next stage's graph (captured after this one, for the same buffer set) reads it there
            with self._valid(b):
                b.mid = [self.reid.forward_a(b.crops, self.reid_split)]

    def _s_back_split(self, b):
        emb = None
        if self.run_nets:
            with self._valid(b):
                emb = self.reid.forward_b(b.mid[0] if self.reid_split > 0 else b.crops, self.reid_split)
        self._select(b, emb)

    def _track_b(self, b: _Bufs, n_valid: int = None, group: int = None):
        """Tracker update of the group's frames in ONE call: the library associates them strictly in order (frame f =
        virtual streams f*S..) and reads the galleries once for all of them; `group` = index of the group's first
        frame (None while warming up / capturing: no callbacks)."""
        e = self.eng
        nv = self.F if n_valid is None else n_valid
        G, S = self.eng.max_group_frames, self.S             # frames per library call (SS_FMAX)
        if self.sR is not None and self._res_ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._res_ev)     # the previous group's rows have left self.outs
        for f0 in range(0, nv if not self.skip_tracker else 0, G):
            n, v0, v1 = min(G, nv - f0), f0 * S, min(nv, f0 + G) * S
            if self.cmc:
                e.set_cmc(b.warps[f0:f0 + n])
            e.update_group(n, b.dets6[v0:v1], b.ndets[v0:v1], b.feats_v[v0:v1], self.img_hw, self.outs[f0:f0 + n], self.nouts[f0:f0 + n])
        if self.sR is not None:                                 # detached chain: the rows exist once the chain's stream is done
            e.track_join(self.sR)
        if group is not None and self.on_result is not None:
            self.cur_bufs, self.cur_valid = b, nv               # the buffer set / real-frame count the callbacks below refer to
            with torch.cuda.stream(self.sR if self.sR is not None else torch.cuda.current_stream(self.dev)):
                for f in range(nv):
                    self.on_result(group + f, f)       # e.g. enqueue the D2H copy of self.outs[f] on this stream

    def _ss_stream(self, st):
        import ctypes as C
        self.eng._ck(self.eng.L.ss_set_hip_stream(self.eng.ctx, C.c_void_p(st.cuda_stream)))

    @torch.no_grad()
    def _capture(self):
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            st.wait_stream(cur)
        last = self.n - 1
        for i, b in enumerate(self.bufs):                       # eager warm-up in stage order (MIOpen find, allocator)
            for j, fn in enumerate(self.stages):
                with torch.cuda.stream(self.streams[j]):
                    self._ss_stream(self.streams[j])
                    for _ in range(2):
                        fn(b)
                        if j == last:
                            self._track_b(b)
                self.streams[j].synchronize()
        for i, b in enumerate(self.bufs):
            for j, fn in enumerate(self.stages):
                with torch.cuda.stream(self.streams[j]):
                    self._ss_stream(self.streams[j])
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.streams[j]):
                        fn(b)
                        if j == last and self.graph_mode == "all":
                            self._track_b(b)
                    self.graphs[j][i] = g
        torch.cuda.synchronize(self.dev)
        self._ss_stream(self.sB)                                 # the tracker lives on the last stage's stream
        self.eng.reset(-1)
        self._captured = True

    # ---- driver API --------------------------------------------------------------------------------------
    on_result = None
    cur_bufs, cur_valid = None, 0

    def begin_frame(self) -> _Bufs:
        """Buffer set for the next frame.  Fill its inputs inside `with torch.cuda.stream(pipe.sA):`."""
        if not self._captured:
            self._capture()
        i = self.k % self.nb
        self.sA.wait_event(self.ev[self.n - 1][i])               # the group that used this set has left the tracker
        self.s_in.wait_event(self.ev[self.n - 1][i])
        return self.bufs[i]

    @torch.no_grad()
    def _run_stage(self, j: int, frame_idx: int):
        i = frame_idx % self.nb
        st = self.streams[j]
        with torch.cuda.stream(st):
            if j > 0:
                st.wait_event(self.ev[j - 1][i])
            if j == self.n - 1 and self.defer:
                self._flush_track(st)                            # the previous group's tracker call, now that stage 0 moved on
            self._mark(f"stage{j}_start", frame_idx, st)
            self.graphs[j][i].replay()
            self._mark(f"stage{j}_end", frame_idx, st)
            if j == self.n - 1 and self.defer:
                self._pending_track = i
            elif j == self.n - 1 and self.sT is not None:        # tracker of this group on its own stream
                self.ev_graph[i].record(st)
                with torch.cuda.stream(self.sT):
                    self.sT.wait_event(self.ev_graph[i])
                    self._track_b(self.bufs[i], self.valid[i], self.base[i])
                    self.ev[j][i].record(self.sT)
            else:
                if j == self.n - 1:
                    if self.graph_mode == "front":
                        self._track_b(self.bufs[i], self.valid[i], self.base[i])
                    elif self.on_result is not None:
                        self.on_result(self.base[i], 0)          # graph == "all" implies frame_batch == 1
                    self._mark_tracked(i, st)
                else:
                    self.ev[j][i].record(st)
        self.stage_done[j] = frame_idx + 1

    def _mark_tracked(self, i, st):
        """Buffer set i has left the tracker (its rows are fetched): on the results stream when the chain is detached."""
        ev = self.ev[self.n - 1][i]
        ev.record(self.sR if self.sR is not None else st)
        self._res_ev = ev

    trace = None                                                 # a list: (name, group, timing event) of every stage / tracker boundary (tools/overlap_timeline.py)

    def _mark(self, name, k, st):
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st)
            self.trace.append((name, k, ev))

    def _flush_track(self, st):
        i = self._pending_track
        if i is not None:
            self._pending_track = None
            self._mark("track_start", self.base[i], st)
            self._track_b(self.bufs[i], self.valid[i], self.base[i])
            self._mark("track_end", self.base[i], st)
            self._mark_tracked(i, st)
            self._gate = self.assoc_ev is not None

    def submit(self, n_valid: int = None):
        """Stage 0 of the group of frames just filled (n_valid <= frame_batch of them are real), and stage j of
        group k-j for every j that has one."""
        k = self.k
        nv = self.F if n_valid is None else int(n_valid)
        self.valid[k % self.nb] = nv
        self.base[k % self.nb] = self.frames_in                 # index of the group's first frame (partial groups allowed)
        self.frames_in += nv
        self.ev_in[k % self.nb].record(self.s_in)               # inputs filled on the input stream: stage 0 waits for them
        self.sA.wait_event(self.ev_in[k % self.nb])
        if self.cmc:
            with torch.cuda.stream(self.sA):
                self.bufs[k % self.nb].nvalid.fill_(nv)
        for j in (range(self.n - 1, -1, -1) if self.defer else range(self.n)):     # deferred: the tracker call is enqueued first
            f = k - j
            if f >= 0 and self.stage_done[j] == f:
                if j == 0 and self._gate:
                    self.sA.wait_event(self.assoc_ev)           # the association launch just enqueued on the other stream
                    self._gate = False
                self._run_stage(j, f)
        self.k += 1

    def flush(self):
        """Push every submitted frame through the remaining stages (results are then enqueued on stream sB)."""
        for j in range(1, self.n):
            while self.stage_done[j] < self.stage_done[j - 1]:
                self._run_stage(j, self.stage_done[j])
        if self.defer and self._pending_track is not None:
            with torch.cuda.stream(self.streams[self.n - 1]):
                self._flush_track(self.streams[self.n - 1])

    def step(self, track: bool = True):
        raise RuntimeError("use begin_frame()/submit()/flush() on an OverlappedPipeline")
