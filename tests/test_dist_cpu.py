"""N > 1 path on CPU: world_size-2 gloo, query sharding + all-gather of placements
(usher_amd/dist.py).  The per-rank placer here is the CPU oracle (test
infrastructure); on the GPU box the same code runs with Placer.place."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import synth
from usher_amd import QueryBatch
from usher_amd.dist import place_sharded, shard_bounds
from usher_amd.placement import RESULT_DTYPE


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _oracle_place_fn(arrays):
    from oracle import capi
    ot = capi.OracleTree(arrays)

    def fn(batch: QueryBatch):
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        for i in range(len(batch)):
            b, e = int(batch.ent_off[i]), int(batch.ent_off[i + 1])
            r = ot.place({"pos": batch.pos[b:e], "ref": batch.ref[b:e].astype(np.int8), "nuc": batch.nuc[b:e].astype(np.int8),
                          "is_missing": batch.is_missing[b:e].astype(np.int8)}, want_ties=False)
            out[i] = (r["best"], r["num_best"], r["best_j"], int(r["has_unique"]))
        return out
    return fn


def _worker(rank, world, port, n_queries, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arrays, queries = synth.make_case(77, n_leaves=120, n_queries=n_queries, n_sites=60)
        batch = QueryBatch(queries)
        got = place_sharded(_oracle_place_fn(arrays), batch)
        want = _oracle_place_fn(arrays)(batch)
        ret[rank] = bool((got.view(np.int32) == want.view(np.int32)).all()) and len(got) == n_queries
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_queries", [7, 1])
def test_place_sharded_world2_gloo(n_queries):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_queries, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
