"""Multi-GPU paths on real devices (skipped on a one-GPU box; the CPU suite covers the sharding logic with gloo,
tests/test_dist_cpu.py): one process per GPU with the RCCL all-gather, the in-process MultiPlacer, and the
front end's --devices list.  `--devices 0` on one device must equal the default run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "usher_amd", "bin", "usher-amd")


def _n_devices():
    import torch
    return torch.cuda.device_count()


def test_place_sharded_on_two_devices_rccl():
    if _n_devices() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket() as sk:   # a free port, as bench.py's self-launch picks one
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py")], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "rank 0 OK" in r.stdout and "rank 1 OK" in r.stdout


@pytest.mark.parametrize("ranks", [2, 3])
def test_place_sharded_worker_on_a_shared_device(ranks):
    """The same worker on whatever devices there are (DIST_SHARE_DEVICE: ranks share a device, gloo carries the gather): the sharding,
    the padded gather buffers and the reassembly of usher_amd.dist.place_sharded with real device placements on every rank, an
    odd sample count and -- with three ranks -- shards of different sizes."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DIST_SHARE_DEVICE="1")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py")], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    for k in range(ranks):
        assert "rank %d OK" % k in r.stdout


def test_multiplacer_matches_single_device():
    from oracle import capi
    from usher_amd import MultiPlacer, Placer, QueryBatch, synth
    st = synth.SynthTree(300_000, n_sites=4000, seed=12)
    q = st.queries(3001, seed=5)
    batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    devs = list(range(min(_n_devices(), 4)))
    mp = MultiPlacer(st.arrays, devs + ([0] if len(devs) == 1 else []))   # on one GPU: two handles on device 0
    multi = mp.place(batch)
    mp.close()
    pl = Placer(st.arrays)
    single = pl.place(batch)
    pl.close()
    assert (multi.view(np.int32) == single.view(np.int32)).all()
    cf = capi.ClosedFormC(capi.OracleTree(st.arrays)).place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    assert (multi["best_set_difference"].astype(np.int64) == cf["best"]).all() and (multi["best_j"].astype(np.int64) == cf["best_j"]).all()


@pytest.mark.parametrize("flags", [["-n"], ["-u"], ["-p"]])
def test_cli_devices_list_equals_default(flags, tmp_path):
    """`--devices 0` (and `--devices 0,1,...` when the box has them) writes the same files as the default run."""
    pb = os.path.join(GOLD, "survey_ref", "syn", "tree.pb")
    vcf = os.path.join(GOLD, "survey_ref", "syn", "query.vcf")
    # "0,0": two replicas on one device, so the sharding threads run on a one-GPU box too
    lists = ["0", "0,0"] + (["0-%d" % (min(_n_devices(), 8) - 1)] if _n_devices() > 1 else [])
    outs = []
    for k, dev in enumerate([None] + lists):
        d = tmp_path / ("o%d" % k)
        d.mkdir()
        cmd = [EXE, "-i", pb, "-v", vcf, "-d", str(d)] + flags + (["--devices", dev] if dev else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, USHER_AMD_SHARD_MIN="7"))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({n: open(str(d / n)).read() for n in sorted(os.listdir(str(d)))})
    for o in outs[1:]:
        assert o == outs[0]


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_multi_rank_flow_on_one_box(ranks, tmp_path):
    """bench.py --gpus N starts its own ranks (torch.distributed.run child), the ranks share one generated tree through
    /dev/shm, shard the queries and print ONE JSON line with the aggregate.  On a box with fewer GPUs the BENCH_SHARE_DEVICE hook
    lets the ranks share the device(s) and gather through gloo -- the flow is what is tested, not the figure: 8 ranks as the
    driver's scaling run launches them, every rank flattening on its share of the host's cores (no rank's set-up far above the
    others'), the all-gather timed on its own."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if _n_devices() < ranks:
        env["BENCH_SHARE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--nodes", "300000", "--queries", "2400", "--steps", "2", "--warmup", "1",
                        "--strong", "--repeats", "1"], capture_output=True, text=True, timeout=1800, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["config"]["rccl_ranks"] == ranks and d["config"]["queries_total"] == 2400
    assert d["value"] > 0 and d["host_buffer_path"]["identical_to_device_path"] is True
    pr = d["per_rank"]
    assert len(pr["flatten_upload_s"]) == ranks and pr["all_gather_ms"] is not None
    assert pr["flattened_once_per_node"] is True          # (round 5: local rank 0 flattens, the others upload its file from /dev/shm)
    assert max(pr["flatten_upload_s"]) <= 1.2 * min(pr["flatten_upload_s"]) + 1.0, pr     # (+1 s: at this size the figure is mostly process noise)


def test_config4_strong_scaling_flow_every_shard_against_the_closed_form(tmp_path):
    """BASELINE config 4's workload as the driver's 8-GPU run launches it -- `bench.py --gpus 8 --strong --queries 1000000`: one global
    batch of 1 M queries, every rank owns a contiguous shard (125 000), places it with ugp_place_device_overlapped from HBM into a
    device tensor and joins the all-gather (usher_amd.dist.ShardedPlacer -- the package's exported path IS the benchmarked one).  On
    a box with fewer GPUs the ranks share the device (BENCH_SHARE_DEVICE: gloo carries the gather).  The gathered records of the
    timed region's last step -- all eight shards, 1 M records -- must equal one handle's answers for the whole batch, and 48 samples per
    shard the C closed form's."""
    import json
    from oracle import capi
    from usher_amd import synth
    from usher_amd.dist import shard_bounds
    ranks, nodes, total = 8, 2_000_000, 1_000_000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if _n_devices() < ranks:
        env["BENCH_SHARE_DEVICE"] = "1"
    dump = str(tmp_path / "gathered.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--nodes", str(nodes), "--queries", str(total), "--steps", "2",
                        "--warmup", "1", "--strong", "--repeats", "1", "--qsets", "2", "--no-extra", "--cpu-queries", "0", "--dump-gathered", dump],
                       capture_output=True, text=True, timeout=1800, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["config"]["queries_total"] == total
    z = np.load(dump)
    cap, world, last_set = int(z["cap"]), int(z["world"]), int(z["last_set"])
    assert world == ranks and cap == (total + ranks - 1) // ranks and z["records"].shape == (ranks * cap, 4)
    st = synth.SynthTree(nodes, n_sites=25000, seed=int(z["seed"]))
    q = st.queries(total, seed=int(z["seed"]) * 1000 + 17 + 7919 * last_set)        # (bench.py draw(): the global batch of a --strong run)
    # every one of the 1 M records against ONE handle placing the whole batch in this process (sharding, padding and gather lose or
    # move nothing), and 48 samples of every rank's shard -- first, last and spread -- against the C closed form (6 ms per sample here)
    from usher_amd import Placer, QueryBatch
    pl = Placer(st.arrays)
    one = pl.place(QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]))
    pl.close()
    cfc = capi.ClosedFormC(capi.OracleTree(st.arrays))
    for rk in range(ranks):
        lo, hi = shard_bounds(total, ranks, rk)
        rec = z["records"][rk * cap: rk * cap + (hi - lo)]
        assert (rec == one[lo:hi].view(np.int32).reshape(-1, 4)).all(), rk
        pick = np.unique(np.concatenate([[lo, hi - 1], np.linspace(lo, hi - 1, 46).astype(np.int64)]))
        off = q["ent_off"].astype(np.int64)
        idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in pick])
        sub_off = np.zeros(len(pick) + 1, np.uint64); sub_off[1:] = np.cumsum(off[pick + 1] - off[pick])
        cf = cfc.place_csr(sub_off, q["pos"][idx], q["ref"][idx], q["nuc"][idx], q["is_missing"][idx])
        got = z["records"][rk * cap + (pick - lo)]
        assert (got[:, 0].astype(np.int64) == cf["best"]).all() and (got[:, 1].astype(np.int64) == cf["num_best"]).all(), rk
        assert (got[:, 2].astype(np.int64) == cf["best_j"]).all() and (got[:, 3].astype(bool) == cf["has_unique"].astype(bool)).all(), rk


def test_flattening_saved_once_and_uploaded_from_a_file(tmp_path, monkeypatch):
    """ugp_flat_save / ugp_mat_create_from_flat (one flattening for the ranks of a node): a handle made from the file answers like
    one made from the arrays -- placements, tie lists, an extended search and the add-mode exclusion maps; a file written under other
    flattening switches, a truncated one and one that is no flattening are refused."""
    from oracle import capi
    from tests import synth
    from usher_amd import Placer, QueryBatch, UgpError
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    arrays, queries = synth.make_case(4242, n_leaves=2200, n_queries=700, n_sites=130, n_ambig=(0, 0, 2), p_masked=0.01)
    batch = QueryBatch(queries)
    path = str(tmp_path / "flat.bin")
    Placer.save_flat(arrays, path)
    a, b = Placer(arrays), Placer(arrays, flat_file=path)
    ra, rb = a.place(batch), b.place(batch)
    assert (ra.view(np.int32) == rb.view(np.int32)).all() and b.timing()["packed_path"] == 1
    ot = capi.OracleTree(arrays)
    for i in range(0, len(queries), 25):
        w = ot.place(queries[i])
        assert (int(rb["best_set_difference"][i]), int(rb["num_best"][i]), int(rb["best_j"][i])) == (w["best"], w["num_best"], w["best_j"])
    ta, tb = a.tied_nodes(batch, 64), b.tied_nodes(batch, 64)
    assert (ta[2] == tb[2]).all() and all(x.tolist() == y.tolist() for x, y in zip(ta[0], tb[0]))
    ea, eb = a.place_ex(batch, order="dfs"), b.place_ex(batch, order="dfs")
    assert (ea.view(np.int32) == eb.view(np.int32)).all()
    rec = [{"flat_j": int(ra["best_j"][0]), "leaf": False, "masked": False, "path": [], "own": []}] if int(ra["best_j"][0]) else []
    if rec:                                                    # (the update maps travel with the file: a node can be excluded)
        a.update(rec, []); b.update(rec, [])
        xa, xb = a.place(batch), b.place(batch)
        assert (xa.view(np.int32) == xb.view(np.int32)).all() and int(xb["best_j"][0]) != int(ra["best_j"][0])
    a.close(); b.close()
    raw = open(path, "rb").read()
    for bad in (raw[:len(raw) // 2], b"not a flattening" * 10, raw + b"x"):
        open(path, "wb").write(bad)
        with pytest.raises(UgpError):
            Placer(arrays, flat_file=path)
    monkeypatch.setenv("UGP_CHUNK_NODES", "77")                # written under other switches: refused, not mis-walked
    open(path, "wb").write(raw)
    with pytest.raises(UgpError):
        Placer(arrays, flat_file=path)
