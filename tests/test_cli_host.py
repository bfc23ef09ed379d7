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
