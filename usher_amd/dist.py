"""Multi-GPU placement: query samples shard across ranks, the flattened MAT is
replicated, the placements are all-gathered (RCCL over xGMI when the process
group is `nccl`; `gloo` in the CPU tests).

The reference places samples one after another on one host
(usher_common.cpp:310); in the static-tree modes (-n / -p) samples are
independent, so sharding them needs no data-path collective other than the
final gather of fixed-size records (16 B per sample).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from .placement import RESULT_DTYPE, QueryBatch


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of samples owned by `rank`: sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def place_sharded(place_fn: Callable[[QueryBatch], np.ndarray], batch: QueryBatch, group=None,
                  device: Optional[str] = None) -> np.ndarray:
    """Every rank places its shard with `place_fn` (e.g. Placer.place) and receives the
    placements of all samples, in the batch's order.  `device` = where the gather buffers
    live ("cuda" for nccl, None/"cpu" for gloo)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return place_fn(batch)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(batch)
    lo, hi = shard_bounds(n, world, rank)
    local = place_fn(batch.slice(lo, hi)) if hi > lo else np.zeros(0, dtype=RESULT_DTYPE)
    cap = (n + world - 1) // world                       # largest shard
    buf = np.zeros((cap, 4), dtype=np.int32)
    if hi > lo:
        buf[: hi - lo] = local.view(np.int32).reshape(-1, 4)
    dev = torch.device(device) if device else torch.device("cpu")
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty((world * cap, 4), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    allr = recv.cpu().numpy()
    out = np.zeros(n, dtype=RESULT_DTYPE)
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        out[a:b] = np.ascontiguousarray(allr[r * cap: r * cap + (b - a)]).view(RESULT_DTYPE).reshape(-1)
    return out
