"""labels writer, incremental class counter and the per-stream loop with a stub model (no GPU)."""
import numpy as np
import torch

from strongsort_yolo_amd.cli import ClassCounter, LabelsWriter, frame_source, process_video
from strongsort_yolo_amd.yolo import Boxes, Results


class StubModel:
    names = {0: "person", 2: "car"}

    def __init__(self):
        self.k = 0

    def track(self, frame, **kw):
        self.k += 1
        ids = torch.tensor([1., 2., 3.])
        cls = torch.tensor([0., 2., 0. if self.k < 3 else 2.])     # id 3 flips class: majority decides
        return [Results(frame, self.names, Boxes(torch.tensor([[1., 2., 3., 4.]] * 3), torch.tensor([.9, .8, .7]), cls, ids))]

    def predict(self, frame, **kw):
        return [Results(frame, self.names, Boxes(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0)))]


def test_labels_and_counts(tmp_path):
    out = process_video({"source": "synthetic:5", "track": True, "count": True, "outdir": str(tmp_path)}, StubModel())
    assert out["frames"] == 5
    assert out["counts"] == {"person": 1, "car": 2}               # id 3: class 2 in 3 of 5 frames
    # the plate is drawn as str(dict) with the classes in NAME order (yolo_multi_model.py:305, :313), not by count
    assert str(out["counts"]) == "{'car': 2, 'person': 1}"
    c = ClassCounter({0: "person", 1: "bicycle", 24: "backpack"})
    for tid, cls in ((1, 0), (2, 0), (3, 0), (4, 1), (5, 24), (6, 1)):
        c.votes[tid][cls] += 1
    assert str(c.counts()) == "{'backpack': 1, 'bicycle': 2, 'person': 3}"
    lines = open(tmp_path / "synthetic:5_labels.txt").read().strip().split("\n")
    assert len(lines) == 15
    f = lines[-1].split()
    assert len(f) == 12 and f[0] == "4" and f[1] in ("0", "2") and f[-4:] == ["-1"] * 4
    # compat mode reproduces the reference's frameId = 0 quirk (yolo_multi_model.py:32)
    w = LabelsWriter(str(tmp_path / "c_labels.txt"), compat=True)
    w.write(7, StubModel().track(np.zeros((4, 4, 3), np.uint8)))
    w.close()
    assert all(l.split()[0] == "0" for l in open(tmp_path / "c_labels.txt"))


def test_count_without_track_stops_after_one_frame(tmp_path):
    out = process_video({"source": "synthetic:5", "track": False, "count": True, "outdir": str(tmp_path)}, StubModel())
    assert out["frames"] == 1 and out["counts"] == {}


def test_frame_sources(tmp_path):
    stack = np.random.default_rng(0).integers(0, 255, (3, 8, 10, 3), dtype=np.uint8)
    np.save(tmp_path / "s.npy", stack)
    got = list(frame_source(str(tmp_path / "s.npy")))
    assert len(got) == 3 and np.array_equal(got[1], stack[1])
    assert len(list(frame_source("synthetic:4"))) == 4


class _StubCv2:
    """Just enough of OpenCV for the optional video branches (N3): VideoCapture over an in-memory clip, VideoWriter into a list."""
    written, opened = [], []

    class VideoCapture:
        def __init__(self, src):
            _StubCv2.opened.append(src)
            self.k, self.ok = 0, src != "missing.mp4"
            self.clip = np.random.default_rng(5).integers(0, 255, (4, 6, 8, 3), dtype=np.uint8)

        def isOpened(self):
            return self.ok

        def read(self):
            if self.k >= len(self.clip):
                return False, None
            self.k += 1
            return True, self.clip[self.k - 1]

        def release(self):
            _StubCv2.opened.append("released")

    class VideoWriter:
        def __init__(self, path, fourcc, fps, size):
            _StubCv2.written.append(("open", path, fourcc, fps, size))

        def write(self, frame):
            _StubCv2.written.append(frame.copy())

        def release(self):
            _StubCv2.written.append("released")

    @staticmethod
    def VideoWriter_fourcc(*c):
        return "".join(c)


def test_video_sources_and_sinks_go_through_opencv_when_it_is_there(tmp_path, monkeypatch):
    """yolo_multi_model.py:252 (VideoCapture of a file, or of camera '0'), :256-260 / :331 (VideoWriter mp4v 15 fps) with a stub cv2;
    without OpenCV the same specs raise an error that says what is missing (this image has no cv2)."""
    import sys
    import pytest
    from strongsort_yolo_amd.cli import FrameSink
    monkeypatch.setitem(sys.modules, "cv2", None)                       # `import cv2` fails
    with pytest.raises(RuntimeError, match="OpenCV"):
        list(frame_source("clip.mp4"))
    with pytest.raises(RuntimeError, match="OpenCV"):
        FrameSink(str(tmp_path / "out.mp4"))
    with pytest.raises(ValueError):
        list(frame_source("no_such_thing.xyz"))
    monkeypatch.setitem(sys.modules, "cv2", _StubCv2)
    _StubCv2.written.clear(); _StubCv2.opened.clear()
    got = list(frame_source("clip.mp4"))
    assert len(got) == 4 and got[0].shape == (6, 8, 3) and _StubCv2.opened == ["clip.mp4", "released"]
    assert len(list(frame_source("0", limit=2))) == 2 and _StubCv2.opened[2] == 0          # a camera index, two frames, released
    with pytest.raises(RuntimeError, match="could not open"):
        list(frame_source("missing.mp4"))
    sink = FrameSink(str(tmp_path / "o" / "out.mp4"))
    for f in got[:3]:
        sink.write(f)
    sink.close()
    assert _StubCv2.written[0] == ("open", str(tmp_path / "o" / "out.mp4"), "mp4v", 15, (8, 6))
    assert all(np.array_equal(a, b) for a, b in zip(_StubCv2.written[1:4], got[:3])) and _StubCv2.written[-1] == "released" and sink.n == 3


def test_video_source_drives_the_stream_loop(tmp_path, monkeypatch):
    import sys
    monkeypatch.setitem(sys.modules, "cv2", _StubCv2)
    out = process_video({"source": "clip.mp4", "track": True, "count": True, "outdir": str(tmp_path)}, StubModel())
    assert out["frames"] == 4 and (tmp_path / "clip_labels.txt").exists()


def test_result_rows_index_like_tensors():
    from strongsort_yolo_amd.yolo import Keypoints
    b = Boxes(torch.arange(12.).view(3, 4), torch.tensor([.9, .8, .7]), torch.zeros(3), torch.tensor([4., 5., 6.]))
    assert float(b[-1].conf) == float(b[2].conf) and int(b[-3].id) == 4 and len(b[-1]) == 1 and len(b[1:]) == 2
    k = Keypoints(torch.arange(3 * 17 * 3.).view(3, 17, 3))
    assert torch.equal(k[-1].data, k[2].data) and len(k[-2]) == 1
    import pytest
    with pytest.raises(IndexError):
        b[3]
    with pytest.raises(IndexError):
        b[-4]


def test_tracker_argument_is_checked_not_ignored():
    """`tracker=` (yolo_multi_model.py:41 passes "botsort.yaml"): Ultralytics configurations are answered by StrongSORT with one
    warning, anything else raises; the CLI's default weights are the reference's (`:17`)."""
    import warnings
    import pytest
    from strongsort_yolo_amd import cli
    from strongsort_yolo_amd.yolo import YOLO
    assert cli.DEFAULT_WEIGHTS == "yolo11n-pose.pt"
    m = YOLO("yolov8n.pt", random_init_ok=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m._check_tracker("botsort.yaml")
        m._check_tracker("botsort.yaml")
        m._check_tracker("strongsort.yaml")
    assert len([x for x in w if issubclass(x.category, RuntimeWarning)]) == 1
    with pytest.raises(ValueError):
        m._check_tracker("deepsort.yaml")
