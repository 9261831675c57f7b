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


class ShardedPlacer:
    """One rank's end of the multi-GPU path as `bench.py` times it: the rank's shard of a batch lives in HBM as a query set
    (`Placer.upload`), `ugp_place_device` / `ugp_place_device_overlapped` write the 16-byte records straight into a device tensor,
    and that tensor is all-gathered ON THE DEVICE (`all_gather_into_tensor`; RCCL when the group is `nccl`) -- no host copy anywhere
    between the query rows and the gathered placements.

    `gather_on_host=True` is the hook for boxes with fewer GPUs than ranks (tests, BENCH_SHARE_DEVICE): the ranks share a device
    and the group is `gloo`, so the record tensor is copied to the host for the collective; everything else is the same code."""

    def __init__(self, placer, group=None, gather_on_host: bool = False, device_index: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.placer, self.group, self.on_host = placer, group, gather_on_host
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.dev = torch.device("cuda", torch.cuda.current_device() if device_index is None else device_index)
        self.depth = placer.pipeline_depth()
        self._outs, self._gathered, self._cap, self._k = [], None, 0, 0
        self.gather_host_s = 0.0   # host time spent issuing the collectives (they run behind the calls' completion)

    def upload(self, batch: QueryBatch):
        """This rank's contiguous shard of `batch` as a resident query set: (qset handle, total samples).  Collective-free."""
        lo, hi = shard_bounds(len(batch), self.world, self.rank)
        return (self.placer.upload(batch.slice(lo, hi)) if hi > lo else None), len(batch)

    def buffers(self, cap: int):
        """One output tensor per call that may be in flight (ugp_pipeline_depth) and the gather target, for shards of <= cap samples."""
        import torch
        if cap > self._cap:
            self._outs = [torch.zeros((cap, 4), dtype=torch.int32, device=self.dev) for _ in range(self.depth)]
            self._gathered = torch.zeros((self.world * cap, 4), dtype=torch.int32, device="cpu" if self.on_host else self.dev) if self.world > 1 else None
            self._cap = cap
        return self._outs, self._gathered

    def step(self, qset, cap: int, overlapped: bool = True, stream: Optional[int] = None):
        """Place the resident shard `qset` and gather: returns (this rank's record tensor, the gathered tensor or None when there is
        one rank).  Asynchronous: both are valid once the current stream has reached this point.  With `overlapped` consecutive
        calls share the device (include/usher_amd.h: a call is ordered behind what the stream held depth - 1 calls ago -- hence one
        output tensor per call in flight, which the all-gather queued right behind each call reads)."""
        import torch
        import torch.distributed as dist
        outs, gathered = self.buffers(cap)
        out = outs[self._k % self.depth]
        self._k += 1
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if qset is None:
            pass                                   # (more ranks than samples: this rank's shard is empty; it still joins the gather)
        elif overlapped:
            self.placer.place_device_overlapped(qset, out.data_ptr(), st)
        else:
            self.placer.place_device(qset, out.data_ptr(), st)
        if self.world > 1:
            import time
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(gathered, out.cpu() if self.on_host else out, group=self.group)
            self.gather_host_s += time.perf_counter() - t0
        return out, gathered

    def unpad(self, gathered, n: int, cap: int) -> np.ndarray:
        """The gathered (world x cap) records as the n placements in the batch's order."""
        allr = gathered.cpu().numpy()
        out = np.zeros(n, dtype=RESULT_DTYPE)
        for r in range(self.world):
            a, b = shard_bounds(n, self.world, r)
            out[a:b] = np.ascontiguousarray(allr[r * cap: r * cap + (b - a)]).view(RESULT_DTYPE).reshape(-1)
        return out


def place_sharded(place, batch: QueryBatch, group=None, device: Optional[str] = None) -> np.ndarray:
    """Every rank places its shard and receives the placements of all samples, in the batch's order.

    `place` = a `Placer` (the product path: device-resident shard, records written on the device, all-gather on the device --
    `ShardedPlacer`, the path `bench.py --gpus N` times) or any callable batch -> records (the CPU tests plug the oracle in under
    `gloo`).  `device` = where the gather buffers live for the callable form ("cuda" for nccl, None / "cpu" for gloo); for a
    `Placer`, `device="cpu"` selects the shared-device hook (gloo carries the gather)."""
    import torch
    import torch.distributed as dist

    if hasattr(place, "place_device"):
        sp = ShardedPlacer(place, group, gather_on_host=(device == "cpu"))
        n = len(batch)
        cap = (n + sp.world - 1) // sp.world
        qset, _ = sp.upload(batch)
        try:
            out, gathered = sp.step(qset, max(cap, 1), overlapped=False)
            torch.cuda.synchronize()
        finally:
            if qset is not None:
                place.free_qset(qset)
        if gathered is None:
            return np.ascontiguousarray(out.cpu().numpy()[:n]).view(RESULT_DTYPE).reshape(-1).copy()
        return sp.unpad(gathered, n, max(cap, 1))
    place_fn = place
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
