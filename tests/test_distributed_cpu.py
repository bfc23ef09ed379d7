"""N>1 path on CPU: world_size-2 gloo run shards independent streams over ranks and gathers the
per-stream results; every stream must equal the single-process run (no data-path collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.streams import assign_streams, max_over_ranks, run_sharded
from strongsort_yolo_amd.synth import make_stream

N_STREAMS, N_FRAMES = 5, 12


def _run_streams(stream_ids):
    out = {}
    for sid in stream_ids:
        st, trk = make_stream(sid, 640, 480, 6), OracleStrongSort(StrongSortConfig(), "c")
        rows = []
        for _ in range(N_FRAMES):
            f = st.next_frame()
            rows.append(trk.update(f.dets, f.feats, (480, 640)))
        out[sid] = np.concatenate([r[:, [4, 7]] for r in rows]) if rows else np.zeros((0, 2))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run_sharded(list(range(N_STREAMS)), _run_streams)
    t = max_over_ranks(0.5 + rank)
    dist.barrier()
    if rank == 0:
        q.put((sorted(res), {k: v.tolist() for k, v in res.items()}, t))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, res, t = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = _run_streams(list(range(N_STREAMS)))
    assert keys == list(range(N_STREAMS))
    for k in range(N_STREAMS):
        assert np.array_equal(np.asarray(res[k]).reshape(-1, 2), ref[k])
    assert t == 1.5                                       # MAX over ranks
    assert assign_streams(N_STREAMS, 2) == [[0, 2, 4], [1, 3]]
