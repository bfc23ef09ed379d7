"""CPU-side pieces of bench.py (workload synthesis, oracle replay, CPU baseline plumbing) and tracker invariants."""
import importlib.util
import os

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import DetectConfig, StrongSortConfig
from strongsort_yolo_amd.engine import letterbox_geometry, scale_geometry
from strongsort_yolo_amd.synth import make_stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workload_and_oracle_replay():
    b = _bench()
    W, H = 640, 480
    g = letterbox_geometry(H, W)
    gs = scale_geometry(g, H, W)
    A = sum((g.out_h // s) * (g.out_w // s) for s in (8, 16, 32))
    wl = b.make_workload(3, W, H, 6, 8, gs, 80, A)
    assert wl["preds"].shape == (8, 84, A) and wl["agt"].shape == (8, A) and wl["pixels"].dtype == np.uint8
    rows = b.oracle_rows(wl, 8, W, H, gs, 80, StrongSortConfig(), DetectConfig())
    assert len(rows) == 8 and rows[-1].shape[1] == 8 and len(rows[-1]) > 0        # confirmed tracks by frame 3
    assert set(b.PRESETS) >= {"c2", "c3", "c4", "c5"} and b.PREFILL >= 103 and b.PREFILL % 16 == 0


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 10_000), st.integers(2, 14))
def test_tracker_invariants(seed, n_ids):
    s = make_stream(seed, 640, 480, n_ids, p_vanish=0.1, vanish_max=6)
    t = OracleStrongSort(StrongSortConfig(max_age=4), "c")
    last_next = 1
    for _ in range(25):
        f = s.next_frame()
        rows = t.update(f.dets, f.feats, (480, 640))
        ids = rows[:, 4].astype(int)
        assert len(set(ids.tolist())) == len(ids)                                  # one row per track
        di = rows[:, 7].astype(int)
        used = di[di >= 0]
        assert len(set(used.tolist())) == len(used) and (used < len(f.dets)).all()  # a detection feeds one track
        assert t.next_id >= last_next                                               # ids are never reused
        last_next = t.next_id
        assert all(tr.state in (1, 2) for tr in t.tracks)                           # deleted tracks are gone
        assert [tr.track_id for tr in t.tracks] == sorted(tr.track_id for tr in t.tracks)
        assert (rows[:, 0] >= 0).all() and (rows[:, 2] <= 639).all() and (rows[:, 3] <= 479).all()


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment must come back with n_gpus == 2
    (--gpus was parsed and ignored).  gloo + --dist-check: rendezvous, barrier and the max-over-ranks
    exchange only, no GPU work."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SS_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["dist_check"] is True and res["n_gpus"] == 2 and res["max_over_ranks"] == 2.0
    # under an external launcher the ranks it provides are used as they are
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_rank_affinity_follows_the_gpu_numa_node(tmp_path, monkeypatch):
    """bench.rank_cpus: a rank's cores come from the NUMA node its GPU hangs on (sysfs), shared by the ranks of that node; without
    the information a contiguous slice of the allowed cores."""
    import os
    import types
    import bench
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(16)), raising=False)
    assert bench.rank_cpus(1, 4) == ([4, 5, 6, 7], "contiguous slice")
    # four GPUs: 0, 1 on node 0 (cpus 0-7), 2, 3 on node 1 (cpus 8-15)
    import torch
    props = [types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x10 + d, pci_device_id=0) for d in range(4)]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props[d])
    for d in range(4):
        p = tmp_path / "bus" / "pci" / "devices" / f"0000:{0x10 + d:02x}:00.0"
        p.mkdir(parents=True)
        (p / "numa_node").write_text(f"{d // 2}\n")
    for k, cl in ((0, "0-7"), (1, "8-15")):
        p = tmp_path / "devices" / "system" / "node" / f"node{k}"
        p.mkdir(parents=True)
        (p / "cpulist").write_text(cl + "\n")
    assert bench.rank_cpus(0, 4, 0, sysfs=str(tmp_path)) == ([0, 1, 2, 3], "NUMA node 0 of GPU 0")
    assert bench.rank_cpus(3, 4, 3, sysfs=str(tmp_path)) == ([12, 13, 14, 15], "NUMA node 1 of GPU 3")
    (tmp_path / "bus" / "pci" / "devices" / "0000:12:00.0" / "numa_node").write_text("-1\n")
    assert bench.rank_cpus(2, 4, 2, sysfs=str(tmp_path))[1] == "contiguous slice"
