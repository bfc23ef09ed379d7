"""Multi-GPU sharding: independent video streams, one process per GPU.

The reference fans out one OS process per `--source` (yolo_multi_model.py:351-354) and pins every one of
them to GPU 0 (`device=0`, :41).  Here stream i belongs to rank i % world; a rank batches all of its
streams into single launches.  There is no data-path collective: torch.distributed (RCCL on GPUs, gloo
in the CPU tests) only carries the start barrier, the max-over-ranks time and the result gather.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import torch
import torch.distributed as dist


def assign_streams(n_streams: int, world: int) -> List[List[int]]:
    """stream -> rank map: [[streams of rank 0], [streams of rank 1], ...] (round robin)."""
    return [list(range(r, n_streams, world)) for r in range(world)]


def run_sharded(stream_ids: Sequence[int], run_stream_batch: Callable[[Sequence[int]], Dict[int, object]]):
    """Run this rank's streams with `run_stream_batch` and gather {stream: result} on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = assign_streams(len(stream_ids), world)[rank]
    local = run_stream_batch([stream_ids[i] for i in mine])
    if world == 1:
        return dict(local)
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    out = {}
    for g in gathered:
        out.update(g)
    return out


def max_over_ranks(seconds: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
