"""Command line of the reference script, kept: `--source ... --track --count` (yolo_multi_model.py:341-354).

What is kept is the per-stream loop (:244-339), the labels file (:34-39, :165-169), the class-count analytics
(:284-305) — as an incremental per-id majority-class counter instead of re-reading the whole CSV every frame
 — and, with `--save`, the annotated output (:58-162, :311-331): the drawing runs as an overlay kernel
on the device (overlay.py, N2) on the frame that is ALREADY there (the throughput path keeps a device copy of every frame
of a group: one upload and one download per frame) and the frames go to a sink (N3).  Frame sources (N3): `synthetic[:N]`,
a `.npy` stack [T,H,W,3], a directory of images (Pillow) and — when OpenCV is importable — a video file or a camera index
through `cv2.VideoCapture` (:252); sinks: `.npy`, raw BGR24, a PNG directory and — with OpenCV — `cv2.VideoWriter` (:256-260).
OpenCV is optional: this environment has none, there the video branches raise a clear error (no GUI: imshow is out of scope).
"""
from __future__ import annotations

import argparse
import os
import time
from collections import Counter, defaultdict
from typing import Iterator, Optional

import numpy as np


# ---- frame sources (N3) --------------------------------------------------------------------------------
def frame_source(spec: str, limit: Optional[int] = None) -> Iterator[np.ndarray]:
    if spec.startswith("synthetic"):
        from .synth import make_stream
        n = int(spec.split(":")[1]) if ":" in spec else 100
        st = make_stream(0, 640, 480, 8)
        for k in range(n if limit is None else min(n, limit)):
            yield st.frame_pixels(k).copy()
    elif spec.endswith(".npy"):
        arr = np.load(spec, mmap_mode="r")
        for k in range(len(arr) if limit is None else min(len(arr), limit)):
            yield np.ascontiguousarray(arr[k])
    elif os.path.isdir(spec):
        from PIL import Image
        names = sorted(f for f in os.listdir(spec) if f.lower().endswith((".jpg", ".jpeg", ".png", ".bmp")))
        for k, f in enumerate(names):
            if limit is not None and k >= limit:
                break
            yield np.asarray(Image.open(os.path.join(spec, f)).convert("RGB"))[:, :, ::-1].copy()   # BGR like cv2
    else:
        yield from _video_source(spec, limit)


VIDEO_EXT = (".mp4", ".avi", ".mov", ".mkv", ".m4v", ".webm", ".mpg", ".mpeg", ".wmv")


def _cv2():
    """OpenCV, if this interpreter has it (it is not a dependency: decoding / encoding video is the only use)."""
    try:
        import cv2
        return cv2
    except ImportError:
        return None


def _video_source(spec: str, limit: Optional[int]) -> Iterator[np.ndarray]:
    """A video file, a stream URL or a camera index through cv2.VideoCapture — the reference's only source kind
    (`cv2.VideoCapture(int(source) if source == '0' else source)`, yolo_multi_model.py:252; here every all-digit source is a
    camera index, App. C notes that the reference's '1' stays a path)."""
    is_cam = spec.isdigit()
    if not (is_cam or "://" in spec or spec.lower().endswith(VIDEO_EXT)):
        raise ValueError(f"cannot open source '{spec}' (synthetic[:N] | stack.npy | image directory | video file / camera index with OpenCV)")
    cv2 = _cv2()
    if cv2 is None:
        raise RuntimeError(f"source '{spec}' is a video / camera: that needs OpenCV (`import cv2` failed in this interpreter); "
                           f"decode it to a .npy stack or an image directory instead")
    cap = cv2.VideoCapture(int(spec) if is_cam else spec)
    try:
        if not cap.isOpened():                               # the reference checks this after creating its writer (:262)
            raise RuntimeError(f"cv2.VideoCapture could not open '{spec}'")
        k = 0
        while limit is None or k < limit:
            ok, frame = cap.read()                           # BGR uint8 [H,W,3], :272
            if not ok:
                break
            yield np.ascontiguousarray(frame)
            k += 1
    finally:
        cap.release()


# ---- frame sink (N3) ------------------------------------------------------------------------------------------
class FrameSink:
    """Where annotated frames go — the reference writes `output/<name>_output.mp4` with cv2.VideoWriter at 15 fps
    (yolo_multi_model.py:258-260, :331); there is no encoder in this environment, so the sink writes
      *.npy      one uint8 stack [T,H,W,3] (BGR), saved at close
      *.bgr      raw BGR24 frames appended as they arrive + `<path>.json` {width, height, fps, frames} (ffmpeg -f rawvideo
                 -pix_fmt bgr24 -s WxH -r fps turns it into the reference's mp4)
      directory  frame_000000.png ... (Pillow)."""

    def __init__(self, path: str, fps: int = 15):
        self.path, self.fps, self.n, self.shape = path, fps, 0, None
        self.kind = ("npy" if path.endswith(".npy") else "bgr" if path.endswith((".bgr", ".raw")) else
                     "video" if path.lower().endswith(VIDEO_EXT) else "dir")
        self._writer = None
        if self.kind == "video" and _cv2() is None:
            raise RuntimeError(f"sink '{path}' is a video file: that needs OpenCV (`import cv2` failed); use .bgr (raw BGR24 + .json), .npy or a directory")
        if self.kind == "dir":
            os.makedirs(path, exist_ok=True)
        else:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._frames = []
        self._f = open(path, "wb") if self.kind == "bgr" else None

    def write(self, frame: np.ndarray):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        self.shape = frame.shape
        if self.kind == "npy":
            self._frames.append(frame.copy())
        elif self.kind == "bgr":
            self._f.write(frame.tobytes())
        elif self.kind == "video":
            if self._writer is None:                         # cv2.VideoWriter(path, mp4v, 15 fps, (w, h)), yolo_multi_model.py:258-260
                cv2 = _cv2()
                self._writer = cv2.VideoWriter(self.path, cv2.VideoWriter_fourcc(*"mp4v"), self.fps, (frame.shape[1], frame.shape[0]))
            self._writer.write(frame)
        else:
            from PIL import Image
            Image.fromarray(frame[:, :, ::-1]).save(os.path.join(self.path, f"frame_{self.n:06d}.png"))
        self.n += 1

    def close(self):
        if self.kind == "video" and self._writer is not None:
            self._writer.release()
        if self.kind == "npy":
            np.save(self.path, np.stack(self._frames) if self._frames else np.zeros((0, 0, 0, 3), np.uint8))
        elif self.kind == "bgr":
            self._f.close()
            import json
            h, w = (self.shape or (0, 0, 3))[:2]
            json.dump({"width": int(w), "height": int(h), "fps": self.fps, "frames": self.n, "pix_fmt": "bgr24"}, open(self.path + ".json", "w"))


# ---- labels file + counting (N1) -------------------------------------------------------------------------
class LabelsWriter:
    """`frameId cls id conf x1 y1 x2 y2 -1 -1 -1 -1` per tracked box (yolo_multi_model.py:165-169).
    compat=True reproduces the reference's quirks (frameId always 0, :32; append mode, :39)."""

    def __init__(self, path: str, compat: bool = False):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        self.f = open(path, "a" if compat else "w")
        self.compat = compat

    def write(self, frame_id: int, results) -> int:
        n = 0
        for r in results:
            if r is None or r.boxes is None or r.boxes.id is None:
                continue
            for bbox in r.boxes:
                for conf, cls, xyxy, id_ in zip(bbox.conf, bbox.cls, bbox.xyxy, bbox.id):
                    x1, y1, x2, y2 = (int(v) for v in xyxy)                 # reference :166 int() box corners
                    fid = 0 if self.compat else frame_id
                    self.f.write(f"{fid} {int(cls)} {int(id_)} {round(float(conf), 3)} {x1} {y1} {x2} {y2} -1 -1 -1 -1\n")
                    n += 1
        self.f.flush()
        return n

    def close(self):
        self.f.close()


class ClassCounter:
    """Objects per class = number of track ids whose majority class is that class (yolo_multi_model.py:293-300),
    maintained incrementally: O(boxes) per frame instead of O(file)."""

    def __init__(self, names):
        self.names = names
        self.votes = defaultdict(Counter)

    def update(self, results):
        for r in results:
            if r is None or r.boxes is None or r.boxes.id is None:
                continue
            for cls, id_ in zip(r.boxes.cls, r.boxes.id):
                self.votes[int(id_)][int(cls)] += 1

    def counts(self) -> dict:
        c = Counter()
        for votes in self.votes.values():
            top = max(votes.values())
            c[min(k for k, v in votes.items() if v == top)] += 1     # ties -> lowest class id (pandas mode()[0])
        # the plate's order is the reference's: sorted by class NAME (yolo_multi_model.py:305; testing.jpg shows {'backpack': 1, 'bicycle': 4, ...})
        return dict(sorted(((self.names.get(k, str(k)), v) for k, v in c.items()), key=lambda item: item[0]))


DEFAULT_WEIGHTS = "yolo11n-pose.pt"       # the model file the reference loads (/root/reference/yolo_multi_model.py:17)


# ---- per-stream loop ------------------------------------------------------------------------------------------
def process_video(args: dict, model=None) -> dict:
    """One stream (yolo_multi_model.py:244-339).  Returns a summary instead of showing a window."""
    source, track, count = args["source"], args["track"], args["count"]
    if model is None:
        from .yolo import YOLO
        model = YOLO(args.get("weights", DEFAULT_WEIGHTS), random_init_ok=args.get("random_init", False), reid_weights=args.get("reid_weights"),
                     reid_fp32=not args.get("reid_f16", False), half=not args.get("fp32", False))
        model.overrides.update(conf=0.3, iou=0.4, agnostic_nms=False, max_det=1000)      # :18-21
    name = os.path.splitext(os.path.basename(str(source)))[0] or "stream"
    writer = LabelsWriter(os.path.join(args.get("outdir", "output"), f"{name}_labels.txt"), args.get("compat", False))
    counter = ClassCounter(model.names)
    frames, t0, fps = 0, time.time(), 0.0
    batch = int(args.get("batch", 16))
    src = frame_source(str(source), args.get("limit"))
    sink = FrameSink(args["save"]) if args.get("save") else None       # annotated output (N2 + N3), off by default
    overlay, fps_str = None, ""

    def emit(frame, res):
        """labels, counts and (with --save) the annotated frame of one processed frame; reference :284-331"""
        nonlocal frames, t0, fps, overlay, fps_str
        if track:
            writer.write(frames, res)
            if count:
                counter.update(res)
        frames += 1
        if frames % 10 == 0:                     # reference :321-326 (10-frame window)
            fps = 10 / max(time.time() - t0, 1e-9)
            fps_str = f"FPS: {fps:.2f}"
            t0 = time.time()
        if sink is not None:
            if overlay is None:
                overlay = model.overlay()
            cnt = counter.counts() if (count and track) else None
            dev_frame = getattr(res[0], "orig_img_device", None) if res else None
            if dev_frame is not None:                # the frame is still on the device: draw there, ONE download, no second upload
                sink.write(overlay.draw_resident(dev_frame, res, cnt, fps_str))
            else:
                sink.write(overlay.draw(frame, res, cnt, fps_str))

    if track and batch > 1 and hasattr(model, "track_stream"):
        # a file / synthetic source can supply frames ahead: groups of `batch` frames through the overlapped pipeline
        # (same rows as frame-by-frame model.track; --batch 1 keeps the reference's per-frame call, :41)
        kw = {"keep_device_frames": True} if sink is not None else {}
        for res in model.track_stream(src, batch=batch, device=args.get("device", 0), **kw):
            emit(res[0].orig_img, res)
        src = ()
    for frame in src:
        if track:
            res = model.track(frame, verbose=False, device=args.get("device", 0), persist=True, tracker="strongsort.yaml")
        else:
            res = model.predict(frame, verbose=False, device=args.get("device", 0))
            if count:                                # reference :280-282: counting needs tracking
                print("[INFO] count works only when objects are tracking.. so use both flags (--track --count)")
                frames += 1
                break
        emit(frame, res)
    if sink is not None:
        sink.close()
    writer.close()
    return {"source": str(source), "frames": frames, "fps": fps, "counts": counter.counts() if count and track else {}}


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--source", nargs="+", type=str, default=["synthetic:30"], help="frame sources")
    p.add_argument("--track", action="store_true")
    p.add_argument("--count", action="store_true")
    p.add_argument("--weights", default=DEFAULT_WEIGHTS, help="detector weights; the reference's default model file (yolo_multi_model.py:17)")
    p.add_argument("--reid-fp32", action="store_true", help="(the default since round 6) ReID crops + OSNet in fp32 on the fp32 kernels: float distances within 1e-4 of a CPU fp32 network")
    p.add_argument("--reid-f16", action="store_true", help="throughput mode: ReID crops + OSNet with f16 activations (1.7x the stream rate; appearance distances off by up to 3e-2)")
    p.add_argument("--fp32", action="store_true", help="every network operation in fp32 (the reference passes no half=): detector on the fp32 convolution kernels too — NMS keep lists equal the CPU fp32 network's; about 0.39x the f16 throughput")
    p.add_argument("--reid-weights", default=None, help="OSNet-x0.25 state_dict for the tracker's appearance features (required with --track unless --random-init)")
    p.add_argument("--limit", type=int, default=None)
    p.add_argument("--save", default=None, help="write annotated frames: stack.npy | video.bgr (raw BGR24 + .json) | directory of PNGs | video.mp4 (needs OpenCV)")
    p.add_argument("--batch", type=int, default=16, help="frames per group on the throughput path (1: per-frame model.track calls as in the reference)")
    p.add_argument("--random-init", action="store_true", help="run seeded random-init networks when the weights file is missing")
    a = p.parse_args(argv)
    jobs = [{"source": s, "track": a.track, "count": a.count, "weights": a.weights, "reid_weights": a.reid_weights, "limit": a.limit, "device": i, "random_init": a.random_init, "batch": a.batch, "reid_f16": a.reid_f16, "fp32": a.fp32,
             "save": (a.save if len(a.source) == 1 else f"{os.path.splitext(a.save)[0]}_{i}{os.path.splitext(a.save)[1]}") if a.save else None}
            for i, s in enumerate(a.source)]
    import torch
    ngpu = max(torch.cuda.device_count(), 1)
    for j in jobs:
        j["device"] %= ngpu                          # stream i -> GPU i mod N (the reference pins device 0, :41)
    if len(jobs) == 1:
        out = [process_video(jobs[0])]
    else:
        import torch.multiprocessing as mp
        with mp.get_context("spawn").Pool(processes=len(jobs)) as pool:      # :353-354
            out = pool.map(process_video, jobs)
    for o in out:
        print(o)
    return out


if __name__ == "__main__":
    main()
