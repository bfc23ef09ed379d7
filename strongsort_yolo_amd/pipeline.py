"""Per-frame tracking pipeline for the S streams owned by one GPU (one process per GPU).

    frame (u8, HBM) -> letterbox [HIP] -> detector [PyTorch-ROCm] -> NMS [HIP] -> ReID crops [HIP]
      -> OSNet [PyTorch-ROCm] -> StrongSORT update [HIP: k_pre / k_cosine / k_step]

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
          graph and the three tracker kernels launched eagerly (lets bench.py bracket the
          association kernel with HIP events);  "none" eager.
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
                 graph: str = "all", debug: bool = False, run_nets: bool = True, seed: int = 0,
                 track_grid: int = MAX_TRACKS):
        self.cfg, self.dcfg = cfg or StrongSortConfig(), dcfg or DetectConfig()
        self.S, (self.H, self.W) = n_streams, frame_hw
        self.eng = TrackerEngine(self.cfg, n_streams, device, debug=debug)
        dev = self.dev = self.eng.device
        if graph == "all":
            self.eng.set_track_grid(track_grid)
        self.half, self.dtype = half, torch.float16 if half else torch.float32
        self.det_source, self.feat_source, self.run_nets = det_source, feat_source, run_nets
        self.RB = reid_batch
        if reid_batch > MAX_DETS:
            raise ValueError("reid_batch <= 128")
        self.geom = letterbox_geometry(self.H, self.W, self.dcfg.imgsz, self.dcfg.stride)
        self.gain, self.pad_x, self.pad_y = scale_geometry(self.geom, self.H, self.W)
        self.detector = self.reid = None
        if run_nets:
            self.detector = nets.build_detector(detector, seed).to(dev, self.dtype).to(memory_format=torch.channels_last)
            self.reid = nets.build_reid(seed + 1).to(dev, self.dtype).to(memory_format=torch.channels_last)
            self.nc, self.nk = self.detector.nc, self.detector.nk
        else:
            self.nc, self.nk = 80, 0
        g, S = self.geom, n_streams
        self.n_anchors = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
        # ---- static buffers (addresses are baked into the graph) ----
        self.frames = torch.zeros(S, self.H, self.W, 3, dtype=torch.uint8, device=dev)
        self.lb = torch.zeros(S, 3, g.out_h, g.out_w, dtype=self.dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.lb_planar = torch.zeros(S, 3, g.out_h, g.out_w, dtype=self.dtype, device=dev)
        self.pred_in = torch.zeros(S, 4 + self.nc + self.nk, self.n_anchors, dtype=torch.float32, device=dev)
        self.dets = torch.zeros(S, MAX_DETS, 6 + self.nk, dtype=torch.float32, device=dev)
        self.dets6 = self.dets if self.nk == 0 else torch.zeros(S, MAX_DETS, 6, dtype=torch.float32, device=dev)
        self.keep = torch.zeros(S, MAX_DETS, dtype=torch.int32, device=dev)
        self.ndets = torch.zeros(S, dtype=torch.int32, device=dev)
        self.crops = torch.zeros(S * self.RB, 3, 256, 128, dtype=self.dtype, device=dev)
        self.feats_in = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)
        self.img_hw = torch.tensor([[self.H, self.W]] * S, dtype=torch.int32, device=dev)
        self.out, self.nout = self.eng.out, self.eng.nout
        self.anchor_gt = torch.zeros(S, self.n_anchors, dtype=torch.int64, device=dev)
        self.gt_feats = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)
        self.graph = None
        self.graph_mode = graph

    # ---- one frame for every stream, from the static buffers ----------------------------------------
    def _step_impl(self):
        e, S, g = self.eng, self.S, self.geom
        if self.run_nets:
            for s in range(S):
                e.letterbox(self.frames[s], g, half=self.half, pad_value=self.dcfg.pad_value, out=self.lb_planar[s])
            self.lb.copy_(self.lb_planar)                       # planar -> channels_last for MIOpen
            pred = self.detector(self.lb)                       # [S, 4+nc+nk, A]
            if self.det_source == "detector":
                self.pred_in.copy_(pred)
        md = min(self.dcfg.max_det, MAX_DETS)
        for s in range(S):
            self.eng._ck(e.L.ss_nms(e.ctx, _p(self.pred_in[s]), self.n_anchors, self.nc, self.nk, self.dcfg.conf,
                                    self.dcfg.iou, int(self.dcfg.agnostic_nms), self.dcfg.max_wh, md, self.gain,
                                    self.pad_x, self.pad_y, float(self.W), float(self.H), _p(self.dets[s]),
                                    6 + self.nk, _p(self.keep[s]), _p(self.ndets[s:s + 1])))
        if self.nk:
            self.dets6.copy_(self.dets[:, :, :6])
        if self.run_nets:
            for s in range(S):
                e.crop_norm(self.frames[s], self.dets6[s], self.RB, count=self.ndets[s:s + 1], half=self.half,
                            out=self.crops[s * self.RB:(s + 1) * self.RB])
            emb = self.reid(self.crops.contiguous(memory_format=torch.channels_last))     # [S*RB, 512]
            if self.feat_source == "reid":
                self.feats_in[:, :self.RB].copy_(emb.view(S, self.RB, FEAT_DIM))
        if self.feat_source == "by_anchor":
            idx = self.anchor_gt.gather(1, self.keep.long().clamp_(0, self.n_anchors - 1))      # [S,128]
            torch.gather(self.gt_feats, 1, idx.clamp_(min=0).unsqueeze(-1).expand(-1, -1, FEAT_DIM), out=self.feats_in)

    def _track(self):
        self.eng.update_device(self.dets6, self.ndets, self.feats_in, self.img_hw)

    def step(self, track: bool = True):
        """Run one frame (all streams).  Asynchronous; results in self.out / self.nout (device)."""
        if self.graph_mode == "none":
            self._step_impl()
            if track:
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
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=st):
                    self._step_impl()
                    if self.graph_mode == "all":
                        self._track()
            torch.cuda.current_stream(self.dev).wait_stream(st)
            self.eng.use_current_stream()
            self._restore_tracker()          # warm-up frames must not count: streams start fresh
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
        self.graph = None
        self.eng.close()


def _p(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


class _Bufs:
    """One set of per-frame buffers (static addresses, baked into that set's HIP graphs)."""

    def __init__(self, p: "FramePipeline"):
        dev, S, g = p.dev, p.S, p.geom
        self.frames = torch.zeros(S, p.H, p.W, 3, dtype=torch.uint8, device=dev)
        self.lb = torch.zeros(S, 3, g.out_h, g.out_w, dtype=p.dtype, device=dev).contiguous(memory_format=torch.channels_last)
        self.lb_planar = torch.zeros(S, 3, g.out_h, g.out_w, dtype=p.dtype, device=dev)
        self.pred_in = torch.zeros(S, 4 + p.nc + p.nk, p.n_anchors, dtype=torch.float32, device=dev)
        self.dets = torch.zeros(S, MAX_DETS, 6 + p.nk, dtype=torch.float32, device=dev)
        self.dets6 = self.dets if p.nk == 0 else torch.zeros(S, MAX_DETS, 6, dtype=torch.float32, device=dev)
        self.keep = torch.zeros(S, MAX_DETS, dtype=torch.int32, device=dev)
        self.ndets = torch.zeros(S, dtype=torch.int32, device=dev)
        self.crops = torch.zeros(S * p.RB, 3, 256, 128, dtype=p.dtype, device=dev)
        self.anchor_gt = torch.zeros(S, p.n_anchors, dtype=torch.int64, device=dev)
        self.gt_feats = torch.zeros(S, MAX_DETS, FEAT_DIM, dtype=torch.float32, device=dev)


class OverlappedPipeline(FramePipeline):
    """Two-stage software pipeline over the frames of the same streams, on two HIP streams:

        stream A:  [letterbox -> detector]                                        of frame k+1
        stream B:  [NMS -> ReID crops -> OSNet -> feature select -> StrongSORT update] of frame k

    The detector is stateless, so frame k+1's stage A does not depend on the tracker state of frame k; the
    tracker recurrence stays strictly in frame order on stream B.  Results are identical to FramePipeline
    (same kernels, same order per stream); throughput approaches max(stage A, stage B) instead of their
    sum at the price of one frame of latency.  Each stage of each buffer set is one captured HIP graph.
    """

    def __init__(self, *a, **kw):
        kw = dict(kw)
        kw["graph"] = kw.get("graph", "front")
        if kw["graph"] == "none":
            raise ValueError("OverlappedPipeline needs graph='front' or 'all'")
        super().__init__(*a, **kw)
        self.bufs = [_Bufs(self), _Bufs(self)]
        self.sA, self.sB = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self.evA = [torch.cuda.Event(), torch.cuda.Event()]
        self.evB = [torch.cuda.Event(), torch.cuda.Event()]
        self.gA, self.gB = [None, None], [None, None]
        self.k = 0                      # frames submitted
        self.done = 0                   # frames whose stage B has been enqueued
        self._captured = False

    # the two stages, parameterised by buffer set -------------------------------------------------------
    def _stage_a(self, b: _Bufs):
        """letterbox -> detector (the longest stage: ~240 launches for yolov8n)"""
        e, S, g = self.eng, self.S, self.geom
        if self.run_nets:
            for s in range(S):
                e.letterbox(b.frames[s], g, half=self.half, pad_value=self.dcfg.pad_value, out=b.lb_planar[s])
            b.lb.copy_(b.lb_planar)
            pred = self.detector(b.lb)
            if self.det_source == "detector":
                b.pred_in.copy_(pred)

    def _stage_b(self, b: _Bufs):
        """NMS -> ReID crops -> OSNet -> feature select (the tracker kernels follow on the same stream)"""
        e, S = self.eng, self.S
        md = min(self.dcfg.max_det, MAX_DETS)
        for s in range(S):
            e._ck(e.L.ss_nms(e.ctx, _p(b.pred_in[s]), self.n_anchors, self.nc, self.nk, self.dcfg.conf, self.dcfg.iou,
                             int(self.dcfg.agnostic_nms), self.dcfg.max_wh, md, self.gain, self.pad_x, self.pad_y,
                             float(self.W), float(self.H), _p(b.dets[s]), 6 + self.nk, _p(b.keep[s]), _p(b.ndets[s:s + 1])))
        if self.nk:
            b.dets6.copy_(b.dets[:, :, :6])
        if self.run_nets:
            for s in range(S):
                e.crop_norm(b.frames[s], b.dets6[s], self.RB, count=b.ndets[s:s + 1], half=self.half,
                            out=b.crops[s * self.RB:(s + 1) * self.RB])
            emb = self.reid(b.crops.contiguous(memory_format=torch.channels_last))
            if self.feat_source == "reid":
                self.feats_in[:, :self.RB].copy_(emb.view(S, self.RB, FEAT_DIM))
        if self.feat_source == "by_anchor":
            idx = b.anchor_gt.gather(1, b.keep.long().clamp_(0, self.n_anchors - 1))
            torch.gather(b.gt_feats, 1, idx.clamp_(min=0).unsqueeze(-1).expand(-1, -1, FEAT_DIM), out=self.feats_in)

    def _track_b(self, b: _Bufs):
        self.eng.update_device(b.dets6, b.ndets, self.feats_in, self.img_hw)

    def _ss_stream(self, st):
        import ctypes as C
        self.eng._ck(self.eng.L.ss_set_hip_stream(self.eng.ctx, C.c_void_p(st.cuda_stream)))

    def _capture(self):
        cur = torch.cuda.current_stream(self.dev)
        self.sA.wait_stream(cur); self.sB.wait_stream(cur)
        for i, b in enumerate(self.bufs):                       # warm-up (MIOpen find, allocator) eagerly, in order
            with torch.cuda.stream(self.sA):
                self._ss_stream(self.sA)
                for _ in range(2):
                    self._stage_a(b)
            self.sA.synchronize()
            with torch.cuda.stream(self.sB):
                self._ss_stream(self.sB)
                for _ in range(2):
                    self._stage_b(b)
                    self._track_b(b)
            self.sB.synchronize()
        for i, b in enumerate(self.bufs):
            with torch.cuda.stream(self.sA):
                self._ss_stream(self.sA)
                self.gA[i] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.gA[i], stream=self.sA):
                    self._stage_a(b)
            with torch.cuda.stream(self.sB):
                self._ss_stream(self.sB)
                self.gB[i] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.gB[i], stream=self.sB):
                    self._stage_b(b)
                    if self.graph_mode == "all":
                        self._track_b(b)
        torch.cuda.synchronize(self.dev)
        self._ss_stream(self.sB)                                 # the tracker lives on stream B from here on
        self.eng.reset(-1)
        self._captured = True

    # ---- driver API --------------------------------------------------------------------------------------
    def begin_frame(self) -> _Bufs:
        """Buffer set for the next frame.  Fill its inputs inside `with torch.cuda.stream(pipe.sA):`."""
        if not self._captured:
            self._capture()
        i = self.k % 2
        self.sA.wait_event(self.evB[i])                          # stage B of frame k-2 has released this set
        return self.bufs[i]

    def _run_b(self, frame_idx: int):
        i = frame_idx % 2
        with torch.cuda.stream(self.sB):
            self.sB.wait_event(self.evA[i])
            self.gB[i].replay()
            if self.graph_mode == "front":
                self._track_b(self.bufs[i])
            if self.on_result is not None:
                self.on_result(frame_idx)                        # e.g. enqueue the D2H copy of self.out on stream B
            self.evB[i].record(self.sB)
        self.done = frame_idx + 1

    on_result = None

    def submit(self):
        """Launch stage A of the frame just filled, and stage B of the previous frame."""
        i = self.k % 2
        with torch.cuda.stream(self.sA):
            self.gA[i].replay()
            self.evA[i].record(self.sA)
        if self.k >= 1 and self.done < self.k:
            self._run_b(self.k - 1)
        self.k += 1

    def flush(self):
        """Stage B of the last submitted frame; afterwards every result has been enqueued on stream B."""
        if self.k > self.done:
            self._run_b(self.k - 1)

    def step(self, track: bool = True):
        raise RuntimeError("use begin_frame()/submit()/flush() on an OverlappedPipeline")
