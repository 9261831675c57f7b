"""The third pruning bound's per-batch tables on the device (ugp_bound3.hip: pair masks -> group sums -> scan -> group
tables) against the restatement in tests/stream_interp.py (b3_tables), block by block, through the test hook
ugp_debug_bound3_tables; and the placements of the same batches against the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import capi
from tests import stream_interp, synth
from usher_amd import FlatTreeView, Placer, QueryBatch, _lib

pytestmark = pytest.mark.gpu


def _tables(placer, tile):
    L = _lib.lib()
    n = C.c_uint64()
    assert L.ugp_debug_bound3_tables(placer._h, tile, None, None, 0, C.byref(n)) == 0
    if n.value == 0:
        return None, None
    over, under = np.zeros(n.value, np.uint16), np.zeros(n.value, np.uint16)
    assert L.ugp_debug_bound3_tables(placer._h, tile, over.ctypes.data_as(C.c_void_p), under.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
    return over.astype(np.int64), under.astype(np.int64)


@pytest.mark.parametrize("seed,n_leaves,ambig", [(11, 3000, (0, 0, 2, 6)), (12, 9000, (0, 0, 0, 0)), (13, 9000, (3, 8, 0, 3))])
def test_device_tables_equal_the_restatement(seed, n_leaves, ambig, monkeypatch):
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")   # (the bound serves the sorted main walk: needs the locality pre-pass)
    monkeypatch.setenv("UGP_BOUND3", "1")             # (pinned on: by itself the library leaves it off for plain batches on trees this small)
    monkeypatch.delenv("UGP_NO_BOUND3", raising=False)
    arrays, queries = synth.make_case(seed, n_leaves=n_leaves, n_queries=12, n_sites=400, n_ambig=ambig, p_masked=0.01)
    flat = FlatTreeView(arrays)
    placer = Placer(arrays)
    orc = capi.OracleTree(arrays)
    try:
        # (a) every sample of the batch is the same sample: whatever the locality sort does, each tile's tables are that sample's
        for q in queries[:3]:
            res = placer.place(QueryBatch([dict(q, name="c%d" % i) for i in range(700)]))
            want = orc.place(q)
            assert (int(res["best_set_difference"][0]), int(res["num_best"][0]), int(res["best_j"][0])) == (want["best"], want["num_best"], want["best_j"])
            assert (res == res[0]).all()
            b3 = stream_interp.b3_tables(flat, [stream_interp.sample_site_alleles(flat, q)[0]])
            for tile in (0, 1):
                over, under = _tables(placer, tile)
                assert over is not None, "the call did not build the tables"
                n = len(b3["over"])
                assert len(over) == n
                np.testing.assert_array_equal(over, np.minimum(b3["over"], 255))       # (one byte per block on the device: 255 = "255 or more")
                np.testing.assert_array_equal(under, np.minimum(b3["under"], 255))
        # (b) a mixed batch: each tile's tables lie between those of its least useful sample and those of the whole batch
        mixed = [dict(queries[i % len(queries)], name="m%d" % i) for i in range(1500)]
        res = placer.place(QueryBatch(mixed))
        for i in range(0, 1500, 97):
            w = orc.place(mixed[i])
            assert (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i])) == (w["best"], w["num_best"], w["best_j"])
        nibs = [stream_interp.sample_site_alleles(flat, q)[0] for q in queries]
        whole = stream_interp.b3_tables(flat, nibs)
        least = np.min([stream_interp.b3_tables(flat, [n])["over"] for n in nibs], axis=0)
        for tile in range(3):
            over, under = _tables(placer, tile)
            assert (over <= np.minimum(whole["over"], 255)).all() and (over >= np.minimum(least, 255)).all()
    finally:
        placer.close()
        del orc


@pytest.mark.parametrize("seed", [81, 83])
def test_long_branches_shared_by_the_queries(seed, monkeypatch):
    """nodes with 300-420 mutations, queries drawn from below them -- hundreds of useful events of one node.  The
    table kernel counts range starts in 8 bits per (tile, block): with every event of a node listed under its header's block that
    wrapped (cum_under 65535, subtree pruned, wrong placement); the events are now listed under their own word's block.  Device
    tables == the restatement block by block, placements == the oracle, with the third bound pinned on."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    monkeypatch.setenv("UGP_BOUND3", "1")
    monkeypatch.delenv("UGP_NO_BOUND3", raising=False)
    arrays, queries = synth.make_case(seed, n_leaves=6000, n_queries=40, genome_len=4000, n_sites=900, n_ambig=(0, 0, 2),
                                      mut_counts=(0, 1, 1, 1, 2, 3) * 8 + (300, 420))
    flat = FlatTreeView(arrays)
    placer = Placer(arrays)
    orc = capi.OracleTree(arrays)
    try:
        deep = [q for q in queries if stream_interp.b3_tables(flat, [stream_interp.sample_site_alleles(flat, q)[0]])["over"].max() >= 256]
        assert len(deep) >= 3, "no query below a long branch"
        for q in deep[:3]:
            res = placer.place(QueryBatch([dict(q, name="c%d" % i) for i in range(600)]))
            want = orc.place(q)
            assert (int(res["best_set_difference"][0]), int(res["num_best"][0]), int(res["best_j"][0])) == (want["best"], want["num_best"], want["best_j"])
            assert (res == res[0]).all()
            b3 = stream_interp.b3_tables(flat, [stream_interp.sample_site_alleles(flat, q)[0]])
            over, under = _tables(placer, 0)
            assert over is not None, "the call did not build the tables"
            # (counts of 255 and more are saturated on the device: "no bound" there -- what the 8-bit counters of round 5 turned into a wrong one)
            np.testing.assert_array_equal(over, np.minimum(b3["over"], 255))
            np.testing.assert_array_equal(under, np.minimum(b3["under"], 255))
            assert (under <= over).all()
        mixed = [dict(queries[i % len(queries)], name="m%d" % i) for i in range(2000)]
        res = placer.place(QueryBatch(mixed))
        for i in range(len(queries)):
            w = orc.place(mixed[i])
            assert (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i])) == (w["best"], w["num_best"], w["best_j"]), i
    finally:
        placer.close()
        del orc
