#!/usr/bin/env python3
"""bench.py -- sample placements/sec of the MI355X placement engine.

One "step" = one pass of the hot path over one batch of synthetic query samples
(locality pre-pass + allele-tile build + k_best8 + phase 2 [+ RCCL all-gather of
the placements when N > 1]) with the flattened MAT and the query rows already
resident in HBM.  Workload (BASELINE.json metric): ~10M-node synthetic
SARS-CoV-2-scale MAT (L = 29,903, 25,000 variable sites), SARS-CoV-2-length
queries; `--nodes 100000 --sites 1500 --queries 1024` gives BASELINE config 2.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a torchrun environment launches the N ranks itself (as a
child `torch.distributed.run`, before anything touches a GPU) -- it never runs
fewer devices than it reports.  Queries shard across ranks, the MAT is
replicated, results are all-gathered over RCCL.  Default is weak scaling (each
rank places its own `--queries` samples); `--strong` shards a fixed total of
`--queries` samples (BASELINE config 4: `--strong --queries 1000000`).
Rank 0 prints ONE JSON line.
"""
import os as _os
# A process has four hardware queues by default (ROCm: GPU_MAX_HW_QUEUES), and streams that share one run their work one after the other.
# The handle's three streams and the caller's use them up; with N > 1 the RCCL stream of torch.distributed is a fifth, and which two then
# share a queue is the runtime's choice (DESIGN.md 4 "A side stream", tools/probe_context.py: a fifth stream cost a third of the rate).
# Eight queues for every N, so that the N = 1 and N > 1 lines are measured under the same setting (at N = 1 it changes nothing: 13.4 M/s
# either way, profiles/r06_sweep_hwq.txt).  Read by the HIP runtime when it initialises: set before torch is imported; a caller's value wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import glob
import re
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)    # (since round 6 nothing is tried out inside the caller's steps: the third bound is decided before the batch runs)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--sites", type=int, default=0, help="variable sites (default 25000 at >=1M nodes, else 1500)")
    ap.add_argument("--queries", type=int, default=16384, help="query samples per GPU per step (total samples with --strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --queries samples in total, sharded across the ranks")
    ap.add_argument("--genome", type=int, default=29903)
    ap.add_argument("--shape", choices=["random", "sars2"], default="random",
                    help="tree shape: random attachment (SURVEY 8d recipe) or SARS-CoV-2-like (shallow, polytomy-dominated)")
    ap.add_argument("--iupac-true", action="store_true", help="with --ambiguous: every IUPAC cell holds the sample's own base (default: any set of 2-3 bases)")
    ap.add_argument("--ambiguous", action="store_true", help="BASELINE config 5: 100-5000 N cells + 0-30 IUPAC cells per query")
    ap.add_argument("--cpu-queries", type=int, default=-1, help="queries timed on the CPU oracle (0 = skip; default: sized for ~10-30 s)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sort-by-source", action="store_true", help="experiment: hand the queries over already ordered by their generator source node")
    ap.add_argument("--no-overlap", action="store_true", help="time the stream-ordered ugp_place_device (one call at a time) instead of ugp_place_device_overlapped")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra keys (BASELINE configs 3 and 4 on one device, the end-to-end CLI run)")
    ap.add_argument("--depth", type=int, default=0, help="overlapped calls kept on the device at a time (2..4; 0 = the library's default)")
    ap.add_argument("--repeats", type=int, default=3, help="windows of --steps steps timed in all (`value` is the MEDIAN window; min / max are extra keys)")
    ap.add_argument("--dump-gathered", default="", help="test hook: rank 0 writes the records gathered by the first window's last step (all ranks' shards, padded) to this .npz")
    ap.add_argument("--qsets", type=int, default=4, help="distinct uploaded query sets the timed loop rotates through (no step places the samples of the step before it)")
    return ap.parse_args()


def self_launch(args) -> int:
    """--gpus N > 1 outside a torchrun environment: start the ranks as a child process and return its exit
    code.  Nothing in this process has initialised a GPU (device_count() does not)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (have and os.environ.get("BENCH_SHARE_DEVICE")):
        print("bench.py: --gpus %d requested but only %d device(s) are visible" % (args.gpus, have), file=sys.stderr)
        return 2
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def shared_tree(args, n_sites, world, local_rank):
    """The synthetic tree.  With several ranks on the node only local rank 0 generates it; the others map its
    arrays from /dev/shm (generation is single-threaded host work, 8 copies of it buy nothing)."""
    from usher_amd import synth
    import torch.distributed as dist
    make = lambda: synth.SynthTree(args.nodes, genome_len=args.genome, n_sites=n_sites, seed=args.seed, shape=args.shape)
    if world == 1:
        return make()
    tag = "/dev/shm/ugp_bench_tree_%s" % os.environ.get("MASTER_PORT", "0")
    keys = ("parent", "mut_off", "mut_pos", "mut_ref", "mut_par", "mut_nuc")
    st = None
    if local_rank == 0:
        st = make()
        for k in keys:
            np.save("%s_%s.npy" % (tag, k), st.arrays[k])
    dist.barrier()
    if local_rank != 0:
        st = synth.SynthTree.from_arrays({k: np.load("%s_%s.npy" % (tag, k)) for k in keys}, genome_len=args.genome, n_sites=n_sites, seed=args.seed, shape=args.shape)
    dist.barrier()
    if local_rank == 0:
        for k in keys:
            try:
                os.remove("%s_%s.npy" % (tag, k))
            except OSError:
                pass
    return st


def stored_profile(info, Q, packed):
    """HBM traffic and instruction-issue counters of the dominant kernel from the newest committed rocprofv3 PMC
    summary of this same workload (tools/profile_round.sh -> profiles/rNN_pmc_summary.json); None when absent."""
    try:
        def round_key(fn):   # rNN_ (a round's final profile) outranks its lettered intermediates rNNa_, rNNb_, ...
            m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(fn))
            return (int(m.group(1)), m.group(2) == "", m.group(2)) if m else (-1, False, "")
        for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=round_key, reverse=True):
            if not re.fullmatch(r"r\d+[a-z]*_pmc_summary\.json", os.path.basename(fn)):
                continue   # (rNN_dense_ / rNN_sars2_ / rNN_config5_: profiles of other workloads on the same tree)
            with open(fn) as f:
                ps = json.load(f)
            cfg = ps.get("bench", {}).get("config", {})
            want = "ugp::k_best8" if packed else "ugp::k_place<0>"
            # (two instantiations of the walk run per step -- coarse pass and main walk: the main walk is the one that moves more bytes)
            k = max((v for n, v in ps.get("kernels", {}).items() if n.startswith(want)), key=lambda v: v.get("hbm_read_bytes_per_dispatch_corrected", 0.0), default={})
            if cfg.get("nodes") == int(info["n_nodes"]) and cfg.get("queries_per_gpu") == Q and "hbm_read_bytes_per_dispatch_corrected" in k:
                # bytes of ALL kernels of one step: a kernel that runs n times per call has n x the dispatches of k_final (once per call);
                # kernels with fewer dispatches than calls (row checks at upload, host-buffer copies) are not part of a device-resident step
                ks = ps.get("kernels", {})
                calls = next((v.get("full_dispatches") for n, v in ks.items() if n.startswith("ugp::k_final") or n.startswith("ugp::k_merge")), None)
                total = None
                if calls:
                    total = 0.0
                    for n, v in ks.items():
                        d = v.get("full_dispatches") or 0
                        if d >= calls and "hbm_read_bytes_per_dispatch_corrected" in v:
                            total += (v["hbm_read_bytes_per_dispatch_corrected"] + v.get("hbm_write_bytes_per_dispatch", 0.0)) * (d / calls)
                return os.path.basename(fn), k, (int(total) if total else None)
    except Exception:
        pass
    return None, {}, None


VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: int32 lane-operations per second (~3.9e13)


def stored_dense_profile():
    """Counters of the UNPRUNED walk from the newest committed profile of `UGP_NO_PRUNE=1 bench.py --queries 2048`
    (tools/profile_round.sh rNN_dense "--queries 2048" -> profiles/rNN_dense_pmc_summary.json); {} when absent."""
    try:
        for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dense_pmc_summary.json")), reverse=True):
            with open(fn) as f:
                ps = json.load(f)
            k = max((v for n, v in ps.get("kernels", {}).items() if n.startswith("ugp::k_best8")), key=lambda v: v.get("avg_duration_ns_full_dispatch", 0.0), default={})
            if k:
                return os.path.basename(fn), k, ps.get("bench", {}).get("config", {})
    except Exception:
        pass
    return None, {}, {}


def dense_walk(pl, st, batch, res_pruned, info, dev, stream, nq=2048):
    """k_best8 with pruning OFF (UGP_NO_PRUNE: no bounds, no locality pre-pass, every tile walks every word of the packed stream) on
    the bench's own tree for the first `nq` queries of set 0: HIP-event time of the kernel, results equal to the pruned path's.
    THIS is the kernel SURVEY 7.1 prices (28 lane-ops per node evaluation, VALU-issue bound) and the one whose algorithmic bytes
    (one tree pass per tile) are really moved -- its fractions are the walk's roofline credit; the headline's are nominal."""
    import torch
    nq = min(nq, len(batch))
    sub = batch.slice(0, nq)
    out = torch.zeros((nq, 4), dtype=torch.int32, device=dev)
    os.environ["UGP_NO_PRUNE"] = "1"
    hq = None
    try:
        pl.reload_knobs()
        hq = pl.upload(sub)
        pl.place_device(hq, out.data_ptr(), stream)
        torch.cuda.synchronize()
        pl.timing_sum()
        t0 = time.perf_counter()
        n_rep = 3
        for _ in range(n_rep):
            pl.place_device(hq, out.data_ptr(), stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n_rep
        tmd = pl.timing_sum()
    finally:
        os.environ.pop("UGP_NO_PRUNE", None)
        pl.reload_knobs()
        if hq is not None:
            pl.free_qset(hq)
    k_ms = tmd["place_ms"] / max(1, tmd["calls"])
    tiles = (nq + 511) // 512
    same = bool((out.cpu().numpy()[:nq] == res_pruned[:nq]).all())
    algo = tiles * (info["algo_tree_bytes"] + 512 * info["algo_tile_bytes"])
    evals = float(nq) * (info["n_nodes"] + info["n_muts"])
    node_evals = float(nq) * info["n_nodes"]
    d = {"what": "k_best8 with UGP_NO_PRUNE=1 (no bounds, no pre-pass): every tile of 512 samples walks the whole packed stream",
         "samples": nq, "tiles": tiles, "packed_path": int(tmd["packed_path"]) and 1, "kernel_ms": round(k_ms, 3), "call_wall_ms": round(wall * 1e3, 3),
         "placements_per_s": round(nq / (k_ms * 1e-3), 1) if k_ms > 0 else None, "identical_to_pruned_path": same,
         "algo_bytes_per_launch": int(algo), "hbm_frac_algorithmic": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else None,
         "node_plus_mut_evals_per_s": round(evals / (k_ms * 1e-3), 1) if k_ms > 0 else None,
         # SURVEY 7.1's yardstick: lane-operations the chip could issue per (node, sample) evaluation at this rate -- its budget for a
         # straightforward kernel is 28 at 100 % VALU issue (= 140 k placements/s at 10 M nodes)
         "valu_lane_ops_available_per_node_eval": round(VALU_LANE_OPS_PER_S * (k_ms * 1e-3) / node_evals, 3) if k_ms > 0 else None,
         "survey_7_1": {"lane_ops_per_node_eval_budget": 28, "placements_per_s_at_full_issue": round(VALU_LANE_OPS_PER_S / 28 / info["n_nodes"], 1)}}
    name, k, cfg = stored_dense_profile()
    if k and cfg.get("nodes") == int(info["n_nodes"]):
        pq = cfg.get("queries_per_gpu") or nq
        p_ms = k.get("avg_duration_ns_full_dispatch", 0.0) * 1e-6
        p_tiles = (pq + 511) // 512
        p_algo = p_tiles * (info["algo_tree_bytes"] + 512 * info["algo_tile_bytes"])
        traffic = k.get("hbm_read_bytes_per_dispatch_corrected", 0.0) + k.get("hbm_write_bytes_per_dispatch", 0.0)
        valu = k.get("SQ_INSTS_VALU_per_dispatch")
        d["profile"] = {"source": name, "samples": pq, "kernel_ms": round(p_ms, 3),
                        "hbm_bytes_by_counters": int(traffic), "counter_over_algorithmic_bytes": round(traffic / p_algo, 3) if p_algo else None,
                        "measured_hbm_frac": round(traffic / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if p_ms > 0 else None,
                        "l2_hit_rate": k.get("l2_hit_rate"),
                        "valu_active_frac": k.get("valu_active_frac_measured"), "salu_busy_frac": k.get("salu_busy_frac_measured"),
                        "valu_issue_frac": k.get("valu_issue_frac"), "salu_issue_frac": k.get("salu_issue_frac"), "wave_wait_frac": k.get("wave_wait_frac"),
                        # wave64 VALU instructions x 64 lanes / (node, sample) evaluations: what the kernel SPENDS, against SURVEY 7.1's 28
                        "valu_lane_ops_per_node_eval": round(valu * 64.0 / (float(pq) * info["n_nodes"]), 3) if valu else None,
                        "valu_lane_ops_per_node_plus_mut_eval": round(valu * 64.0 / (float(pq) * (info["n_nodes"] + info["n_muts"])), 3) if valu else None}
    return d


def far_pruned_frac(st, dev_index, nq=16384):
    """Fraction of the packed stream's words the main walk jumps over for a batch of far queries and for a batch of the headline's
    queries (the instrumented build of the library, UGP_STATS=1: one call each on a handle of its own)."""
    import torch
    from usher_amd import Placer, QueryBatch
    os.environ["UGP_STATS"] = "1"
    pe = None
    try:
        pe = Placer(st.arrays, device=dev_index, experiments=True)
        out = {}
        for name, kw in (("far", dict(max_subst=200, min_subst=50, ref_every_8th=True)), ("headline", {})):
            qq = st.queries(nq, seed=9001, **kw)
            b_ = QueryBatch.from_csr(qq["ent_off"], qq["pos"], qq["ref"], qq["nuc"], qq["is_missing"])
            for _ in range(2):
                pe.place(b_)
            t_ = pe.timing()
            out[name] = round(t_["words_skipped"] / t_["words_total"], 5) if t_.get("words_total") else None
        return out
    finally:
        os.environ.pop("UGP_STATS", None)
        if pe is not None:
            pe.close()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the placement path has no CPU fallback)")
    # BENCH_SHARE_DEVICE=1 (test hook for one-GPU boxes): the ranks share the visible devices and gather through gloo, so that
    # the multi-rank flow (shared tree, sharding, gather, one JSON line) can be exercised without N GPUs; not a measurement
    share = world > 1 and bool(os.environ.get("BENCH_SHARE_DEVICE"))
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from usher_amd import Placer, QueryBatch, synth
    from usher_amd.dist import shard_bounds

    n_sites = args.sites or (25000 if args.nodes >= 1_000_000 else 1500)
    t0 = time.time()
    st = shared_tree(args, n_sites, world, local_rank)
    t_gen = time.time() - t0
    # several ranks on one host: each flattens the tree for itself -- on its share of the host's cores, not 32 threads each
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    auto_threads = local_world > 1 and "UGP_FLATTEN_THREADS" not in os.environ
    if auto_threads:
        os.environ["UGP_FLATTEN_THREADS"] = str(max(1, min(32, (os.cpu_count() or 1) // local_world)))
    if args.depth:
        os.environ["UGP_PIPELINE_DEPTH"] = str(args.depth)
    if world > 1:
        dist.barrier()   # (every rank starts its flattening at the same time: what an N-rank launch costs, not what a lone rank would)
    t0 = time.time()
    use_exp = bool(os.environ.get("UGP_STATS") or os.environ.get("BENCH_EXP_LIB"))   # (the build with the experiments, libusher_amd_exp.so)
    flat_file = None
    if world > 1 and not os.environ.get("BENCH_FLATTEN_PER_RANK"):
        # one flattening for the node: local rank 0 runs it (on all of the host's flattening threads) and leaves the result on
        # /dev/shm, the other ranks upload that file (ugp_flat_save / ugp_mat_create_from_flat)
        flat_file = "/dev/shm/ugp_bench_flat_%s" % os.environ.get("MASTER_PORT", "0")
        if local_rank == 0:
            if auto_threads:
                os.environ.pop("UGP_FLATTEN_THREADS", None)   # (the one flattening of the node gets the library's own thread count)
            Placer.save_flat(st.arrays, flat_file, experiments=use_exp)
        dist.barrier()
    pl = Placer(st.arrays, device=dev_index, experiments=use_exp, flat_file=flat_file)
    t_flat = time.time() - t0
    if flat_file:
        dist.barrier()
        if local_rank == 0:
            try:
                os.remove(flat_file)
            except OSError:
                pass
    info = pl.info()
    kw = dict(n_lo=100, n_hi=5000, iupac_hi=30) if args.ambiguous else {}
    if args.iupac_true:
        kw["iupac_true"] = True
    if args.shape == "sars2":
        kw["recent"] = True
    # The timed loop rotates through --qsets DISTINCT query sets (all uploaded before the clock starts): no step places the samples --
    # or walks the tree regions -- of the step before it.  Set k is drawn with its own seed; set 0 is the one the CPU baseline checks.
    n_sets = max(1, args.qsets)

    def draw(k):
        if args.strong:   # one global batch; this rank owns a contiguous shard of it
            qq = st.queries(args.queries, seed=args.seed * 1000 + 17 + 7919 * k, **kw)
            lo, hi = shard_bounds(args.queries, world, rank)
            e0, e1 = int(qq["ent_off"][lo]), int(qq["ent_off"][hi])
            return {"ent_off": qq["ent_off"][lo:hi + 1] - qq["ent_off"][lo], "pos": qq["pos"][e0:e1], "ref": qq["ref"][e0:e1], "nuc": qq["nuc"][e0:e1],
                    "is_missing": qq["is_missing"][e0:e1], "source": qq["source"][lo:hi]}
        return st.queries(args.queries, seed=args.seed * 1000 + 17 + rank + 7919 * k, **kw)
    q = draw(0)
    if args.strong:
        total_q = args.queries
        cap = (args.queries + world - 1) // world
    else:
        total_q = args.queries * world
        cap = args.queries
    if args.sort_by_source:
        from usher_amd import FlatTreeView
        d2b = FlatTreeView(st.arrays).dfs2bfs
        dfs_rank = np.empty(len(d2b), np.int64); dfs_rank[d2b] = np.arange(len(d2b))
        order = np.argsort(dfs_rank[q["source"]], kind="stable")
        lens = np.diff(q["ent_off"].astype(np.int64))
        starts = q["ent_off"].astype(np.int64)[:-1]
        idx = np.concatenate([np.arange(starts[i], starts[i] + lens[i]) for i in order]) if len(order) else np.zeros(0, np.int64)
        new_off = np.zeros(len(order) + 1, np.uint64); new_off[1:] = np.cumsum(lens[order])
        q = {"ent_off": new_off, "pos": q["pos"][idx], "ref": q["ref"][idx], "nuc": q["nuc"][idx], "is_missing": q["is_missing"][idx], "source": q["source"][order]}
    batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    qset = pl.upload(batch)
    Q = len(batch)
    qsets = [qset]
    for k in range(1, n_sets):
        qk = draw(k)
        qsets.append(pl.upload(QueryBatch.from_csr(qk["ent_off"], qk["pos"], qk["ref"], qk["nuc"], qk["is_missing"])))
        del qk
    # Two output buffers used alternately: consecutive ugp_place_device_overlapped calls share the device, and call k + 2 is
    # ordered behind whatever the caller's stream held when call k + 1 was made -- the all-gather that reads call k's buffer
    # included (include/usher_amd.h: one call of lag).  (shards differ by at most one sample: padded to `cap`)
    # The step IS the package's multi-GPU entry (usher_amd.dist.ShardedPlacer): the resident shard placed by
    # ugp_place_device_overlapped straight into a device tensor, that tensor all-gathered on the device (RCCL), one output tensor per
    # call in flight (include/usher_amd.h: a call is ordered behind what the stream held depth - 1 calls ago -- the all-gather that
    # reads call k's tensor included).  (shards differ by at most one sample: padded to `cap`)
    from usher_amd.dist import ShardedPlacer
    sp = ShardedPlacer(pl, gather_on_host=share, device_index=dev_index)
    depth = sp.depth
    outs, gathered = sp.buffers(cap)
    stream = torch.cuda.current_stream().cuda_stream
    n_step = [0]

    def step():
        qs_k = qsets[n_step[0] % n_sets]
        n_step[0] += 1
        sp.step(qs_k, cap, overlapped=not args.no_overlap, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pl.timing_sum()   # (reset the library's running totals: the timed region starts here)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()   # no synchronisation between steps: consecutive ugp_place_device calls overlap on the device
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    last_out = outs[(n_step[0] - 1) % depth]
    last_set = (n_step[0] - 1) % n_sets
    res_last = last_out.cpu().numpy()[:Q].copy()   # (results of the timed region's last step: query set `last_set`)
    if args.dump_gathered and rank == 0:   # (tests/test_multi_gpu.py checks every rank's shard of this against the closed form)
        g = (gathered if world > 1 else last_out).cpu().numpy()
        np.savez(args.dump_gathered, records=g, cap=cap, world=world, last_set=last_set, total=total_q, seed=args.seed, strong=int(args.strong))
    # HIP events recorded by the library on the streams its kernels ran on, summed over the timed steps
    tm = pl.timing_sum()
    assert tm["calls"] == args.steps, tm
    scale = args.steps / max(1, tm["calls"])
    place_ms, table_ms, merge_ms, coarse_ms = tm["place_ms"] * scale, tm["table_ms"] * scale, tm["merge_ms"] * scale, tm["coarse_ms"] * scale
    tiles, groups, packed = tm["n_tiles"], tm["n_groups"], tm["packed_path"]
    skipped, wtotal, nskips = tm["words_skipped"], tm["words_total"], tm["reserved"]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-rank set-up times and the all-gather on its own (N > 1): what the ranks cost each other outside the kernels
    flat_all, gather_ms = [round(t_flat, 3)], None
    if world > 1:
        tf = torch.tensor([t_flat], dtype=torch.float64, device="cpu" if share else dev)
        tfs = [torch.zeros_like(tf) for _ in range(world)]
        dist.all_gather(tfs, tf)
        flat_all = [round(float(x.item()), 3) for x in tfs]
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        for _ in range(5):
            dist.all_gather_into_tensor(gathered, last_out.cpu() if share else last_out)
        torch.cuda.synchronize()
        gather_ms = round((time.perf_counter() - tg) * 1e3 / 5, 4)
    # the same window again (--repeats - 1 times); `value` is the MEDIAN window, min / max beside it
    windows = [elapsed]
    for _ in range(max(0, args.repeats - 1)):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tw = time.perf_counter() - tw
        if world > 1:
            tt = torch.tensor([tw], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tw = float(tt.item())
        windows.append(tw)
    pl.timing_sum()
    out = outs[0]
    # every query set once through the stream-ordered entry point: set 0's results are what the host-buffer path and the CPU baseline
    # are compared with, and the timed region's last step must have produced exactly these for ITS set
    res = None
    rotation_exact = True
    for k in range(n_sets):
        pl.place_device(qsets[k], out.data_ptr(), stream)
        torch.cuda.synchronize()
        rk = out.cpu().numpy()[:Q].copy()
        if k == 0:
            res = rk
        if k == last_set:
            rotation_exact = bool((rk == res_last).all())
    # The same kernels one call at a time (a synchronisation after every step), outside the timed region: in the timed region
    # two consecutive calls share the device (ugp_place_device overlaps them), so a kernel's event-bracketed duration there is
    # the duration of a kernel that has company.  Reported beside it as kernel_ms_alone / frac_alone.
    alone = {"place_ms": 0.0, "coarse_ms": 0.0, "table_ms": 0.0, "merge_ms": 0.0}
    n_alone = max(2, min(args.steps, 5))
    pl.place_device(qset, out.data_ptr(), stream)   # (untimed: the call right behind a burst still takes the burst's half-size grid)
    torch.cuda.synchronize()
    pl.timing_sum()
    t_alone = time.perf_counter()
    for i_alone in range(n_alone):
        pl.place_device(qsets[i_alone % n_sets], out.data_ptr(), stream)
        torch.cuda.synchronize()
    t_alone = (time.perf_counter() - t_alone) / n_alone
    tma = pl.timing_sum()
    for k in alone:
        alone[k] = tma[k] / max(1, tma["calls"])
    # SURVEY 8(d)'s metric: wall time of ugp_place_batch -- host buffers in and out, i.e. query upload over PCIe,
    # row checks, the same kernels, result download.  Reported beside `value` (the contract keeps `value`
    # HBM-resident); timed on every rank the same way (barrier + max).
    host_path = None
    # (the collectives run on every rank, also on one whose shard is empty: only the place() calls depend on Q)
    if Q:
        pl.place(batch)   # warm
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    n_host = max(2, min(args.steps, 5))
    t1 = time.perf_counter()
    hres = None
    for _ in range(n_host):
        if Q:
            hres = pl.place(batch)
    dt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dt /= n_host
    # ... and with as many batches in flight as the handle keeps on the device (ugp_place_batch_async / ugp_job_wait, a ring of
    # ugp_pipeline_depth jobs): the same work from and to host buffers, the upload of one batch under the kernels of the others
    n_async = max(4, args.steps)
    dta, ares = None, None
    if Q:
        pl.job_wait(pl.place_async(batch))   # warm (pinned staging)
        torch.cuda.synchronize()
        n_ring = 2 if (Q > 32768 or args.ambiguous) else pl.pipeline_depth()
        t1 = time.perf_counter()
        ring = []
        for _ in range(n_async):
            if len(ring) == n_ring:
                ares = pl.job_wait(ring.pop(0))
            ring.append(pl.place_async(batch))
        while ring:
            ares = pl.job_wait(ring.pop(0))
        dta = time.perf_counter() - t1
    if world > 1:   # (every rank, also one with an empty shard: max over ranks, as for the other figures)
        tt = torch.tensor([dta or 0.0], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dta = float(tt.item()) or None
    if dta:
        dta /= n_async
    if Q:
        same = bool((np.stack([hres[k] for k in ("best_set_difference", "num_best", "best_j", "best_has_unique")], 1).astype(np.int64) == res.astype(np.int64)).all())
        host_path = {"metric": "SURVEY 8(d): Q / wall time of ugp_place_batch (query upload + kernels + result download)",
                     "placements_per_s": round(total_q / dt, 2), "ms_per_batch": round(dt * 1e3, 3), "identical_to_device_path": same,
                     "two_in_flight": {"entry": "ugp_place_batch_async / ugp_job_wait", "jobs_in_flight": n_ring, "placements_per_s": round(total_q / dta, 2), "ms_per_batch": round(dta * 1e3, 3),
                                       "identical_to_device_path": bool((np.stack([ares[k] for k in ("best_set_difference", "num_best", "best_j", "best_has_unique")], 1).astype(np.int64) == res.astype(np.int64)).all())} if (dta and ares is not None) else None}
    # ---- extra keys: BASELINE configs 3 and 4 on this one device, the end-to-end CLI (default headline run only)
    extra = None
    headline = (args.nodes == 10_000_000 and args.queries == 16384 and args.shape == "random" and not args.ambiguous and not args.strong and not args.sort_by_source)
    if world == 1 and headline and not args.no_extra:
        extra = {}

        def timed_config(placer, tree, nq, steps, warm, ties_cap=0, n_rot=3, **qkw):
            # (round 6) like the headline, the timed loop ROTATES through n_rot distinct uploaded query sets: no step places the samples --
            # or walks the tree regions -- of the step before it
            hqs, bb = [], None
            for r_ in range(max(1, n_rot)):
                qq = tree.queries(nq, seed=args.seed * 1000 + 4 + 104729 * r_, **qkw)
                b_ = QueryBatch.from_csr(qq["ent_off"], qq["pos"], qq["ref"], qq["nuc"], qq["is_missing"])
                hqs.append(placer.upload(b_))
                if r_ == 0:
                    bb = b_
                del qq
            dd = placer.pipeline_depth()
            oo = [torch.zeros((nq, 4), dtype=torch.int32, device=dev) for _ in range(dd)]
            for k in range(warm):
                placer.place_device_overlapped(hqs[k % len(hqs)], oo[k % dd].data_ptr(), stream)
            torch.cuda.synchronize()
            placer.timing_sum()
            tq = time.perf_counter()
            for k in range(steps):
                placer.place_device_overlapped(hqs[k % len(hqs)], oo[k % dd].data_ptr(), stream)
            torch.cuda.synchronize()
            tq = time.perf_counter() - tq
            tmq = placer.timing_sum()
            strict = torch.zeros((nq, 4), dtype=torch.int32, device=dev)
            placer.place_device(hqs[(steps - 1) % len(hqs)], strict.data_ptr(), stream)   # the stream-ordered entry point, one call alone, the last step's set
            torch.cuda.synchronize()
            same = bool((strict == oo[(steps - 1) % dd]).all().item())
            if (steps - 1) % len(hqs) != 0:   # (the tie lists below are of set 0)
                placer.place_device(hqs[0], strict.data_ptr(), stream)
                torch.cuda.synchronize()
            for h_ in hqs:
                placer.free_qset(h_)
            ties = None
            if ties_cap:   # tie reporting (-M / -D of the CLI): ugp_tied_nodes from and to host buffers, wall time of the second call
                import ctypes as C
                tj = np.zeros((nq, ties_cap), dtype=np.uint32); th = np.zeros((nq, ties_cap), dtype=np.uint8); tcnt = np.zeros(nq, dtype=np.uint32)
                ptr = lambda x: x.ctypes.data_as(C.c_void_p)
                for _ in range(2):   # (the second call is timed: buffers allocated)
                    tt0 = time.perf_counter()
                    placer._ck(placer._L.ugp_tied_nodes(placer._h, C.byref(bb.desc), ties_cap, ptr(tj), ptr(th), ptr(tcnt)))
                    tt1 = time.perf_counter()
                ties = {"entry": "ugp_tied_nodes (host buffers in, host buffers out)", "cap": ties_cap, "wall_ms": round((tt1 - tt0) * 1e3, 2),
                        "samples_with_ties": int((tcnt > 1).sum()), "counts_equal_num_best": bool((tcnt == strict[:, 1].cpu().numpy().astype(np.uint32)).all())}
            return {**({"tie_lists": ties} if ties else {}), "queries": nq, "steps": steps, "query_sets_rotated": len(hqs), "third_bound_steps": int(tmq.get("bound3", 0)), "placements_per_s": round(nq * steps / tq, 2), "ms_per_step": round(tq * 1e3 / steps, 3),
                    "k_best8_ms": round(tmq["place_ms"] / max(1, tmq["calls"]), 4), "sub_batches": int(tmq["place_launches"] // max(1, tmq["calls"])),
                    "identical_to_stream_ordered_call": same}

        # config 4's workload on one device: 1,000,000 queries on the 10M-node tree in one call (4 sub-batches of 262,144)
        try:   # (an extra key must never cost the bench line)
            c4 = timed_config(pl, st, 1_000_000, 4, 3, n_rot=2)   # (3 warm-up calls: every workspace set a long call cycles through is allocated before the clock starts)
            c4["workload"] = "BASELINE config 4 on one device: 1,000,000 queries on the %d-node MAT, one ugp_place_device_overlapped call per step" % info["n_nodes"]
            extra["config4_1m_queries_one_gpu"] = c4
        except Exception as ex:
            extra["config4_1m_queries_one_gpu"] = {"error": repr(ex)[:300]}
        # config 5's workload on one device: high-ambiguity queries (100-5,000 N cells + 0-30 IUPAC cells of any 2-3 bases each) and tie lists
        try:
            c5 = timed_config(pl, st, 16384, 12, 12, ties_cap=64, n_lo=100, n_hi=5000, iupac_hi=30)
            c5["workload"] = "BASELINE config 5 on one device: 16,384 queries with 100-5,000 N cells and 0-30 IUPAC cells each on the %d-node MAT, tie lists of up to 64 nodes" % info["n_nodes"]
            extra["config5_high_ambiguity_one_gpu"] = c5
        except Exception as ex:
            extra["config5_high_ambiguity_one_gpu"] = {"error": repr(ex)[:300]}
        # ---- (round 6) the walk WITHOUT pruning -- the dense path SURVEY 7.1 budgets -- and queries far from the tree
        try:
            extra["dense_walk"] = dense_walk(pl, st, batch, res, info, dev, stream)
        except Exception as ex:
            extra["dense_walk"] = {"error": repr(ex)[:300]}
        try:
            fq = timed_config(pl, st, 16384, 12, 12, n_rot=3, max_subst=200, min_subst=50, ref_every_8th=True)
            fq["workload"] = ("16,384 queries per step that are NOT near any node: a random node's genotype + 50-200 substitutions (70 %% at the tree's variable sites), "
                              "every 8th the all-reference sample; %d-node MAT" % info["n_nodes"])
            fq["pruned_frac"] = far_pruned_frac(st, dev_index)
            extra["far_queries"] = fq
        except Exception as ex:
            extra["far_queries"] = {"error": repr(ex)[:300]}
        # the drop-in CLI end to end: the same tree as parsimony.proto, 10,000 queries as a VCF, `usher-amd -i .. -v .. -n` (load the
        # MAT, read the VCF, flatten + upload, place, write placement_stats.tsv and the tree) -- wall time of the whole process
        try:
            import shutil
            import tempfile
            from tools.time_load import host_lib, write_workload
            dcli = tempfile.mkdtemp(prefix="ugp_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            nq_cli = 10_000
            qc = st.queries(nq_cli, seed=args.seed * 1000 + 5)
            tw = time.time()
            write_workload(host_lib(), st, qc, nq_cli, os.path.join(dcli, "base.pb"), os.path.join(dcli, "q.vcf"))
            tw = time.time() - tw
            exe = os.path.join(ROOT, "usher_amd", "bin", "usher-amd")

            def cli(pb_name, out_name):
                t_ = time.perf_counter()
                r_ = subprocess.run([exe, "-i", os.path.join(dcli, pb_name), "-v", os.path.join(dcli, "q.vcf"), "-n", "-d", os.path.join(dcli, out_name), "--device", str(dev_index)],
                                    capture_output=True, text=True, env=dict(os.environ, USHER_AMD_PROFILE="1"))
                return r_, time.perf_counter() - t_
            rr, tc = cli("base.pb", "out")
            lines = open(os.path.join(dcli, "out", "placement_stats.tsv")).read().splitlines() if rr.returncode == 0 else []
            # the same samples through the library (this process): score and number of optimal placements per sample
            bq = QueryBatch.from_csr(qc["ent_off"], qc["pos"], qc["ref"], qc["nuc"], qc["is_missing"])
            lib_res = pl.place(bq)
            same = len(lines) == nq_cli and all(l.split("\t")[1:3] == [str(int(lib_res["best_set_difference"][i])), str(int(lib_res["num_best"][i]))] for i, l in enumerate(lines))
            extra["cli_end_to_end"] = {"command": "usher-amd -i base.pb -v q.vcf -n -d out", "nodes": int(info["n_nodes"]), "queries": nq_cli, "exit_code": rr.returncode,
                                       "wall_s": round(tc, 3), "placements_per_s": round(nq_cli / tc, 1), "stats_equal_library_results": bool(same),
                                       "pb_bytes": os.path.getsize(os.path.join(dcli, "base.pb")), "vcf_bytes": os.path.getsize(os.path.join(dcli, "q.vcf")),
                                       "write_inputs_s": round(tw, 2), "profile": [l for l in rr.stderr.splitlines() if l.startswith("[usher-amd profile]") and ":" not in l.split("]", 1)[1][:12]][:10]}
            # ... and from the gzip-compressed MAT (the public SARS-CoV-2 MAT ships as .pb.gz, mutation_annotated_tree.cpp:522-547)
            try:
                tz = time.time()
                subprocess.run(["gzip", "-1", "-k", os.path.join(dcli, "base.pb")], check=True)
                tz = time.time() - tz
                rz, tcz = cli("base.pb.gz", "outz")
                same_z = rz.returncode == 0 and all(open(os.path.join(dcli, "outz", n)).read() == open(os.path.join(dcli, "out", n)).read() for n in ("placement_stats.tsv", "final-tree.nh"))
                extra["cli_end_to_end_gz"] = {"command": "usher-amd -i base.pb.gz -v q.vcf -n -d out", "exit_code": rz.returncode, "wall_s": round(tcz, 3), "over_plain": round(tcz / tc, 3) if tc > 0 else None,
                                              "pb_gz_bytes": os.path.getsize(os.path.join(dcli, "base.pb.gz")), "gzip_s": round(tz, 2), "outputs_equal_plain_run": bool(same_z),
                                              "profile": [l for l in rz.stderr.splitlines() if l.startswith("[usher-amd profile] load")][:8]}
            except Exception as ex:
                extra["cli_end_to_end_gz"] = {"error": repr(ex)[:300]}
            shutil.rmtree(dcli, ignore_errors=True)
        except Exception as ex:   # (the extra key must never cost the bench line)
            extra["cli_end_to_end"] = {"error": repr(ex)[:300]}

    result = None
    if rank == 0:
        # `value` = the MEDIAN of the --repeats windows of K steps each (the first window is `windows.ms_per_step[0]`)
        w_med = sorted(windows)[len(windows) // 2]
        ms_per_step = w_med * 1e3 / args.steps
        value = total_q * args.steps / w_med
        # roofline of the dominant kernel: algorithmic bytes per launch, SURVEY 8(d):
        # every T-sample tile makes one pass over the tree: B_tree + T * (L/2 + 16); T = 512 on the packed path
        T = 512 if packed else 64
        algo_bytes = tiles * (info["algo_tree_bytes"] + T * info["algo_tile_bytes"])
        # The dominant kernel's average launch duration: measured with the device to itself (one call at a time, right behind the timed
        # region, HIP events on the launch stream) -- `kernel_ms`.  Inside the timed region up to `depth` such launches share the
        # chip, each on its share of the resident wave slots, and a launch's event-bracketed duration there (`kernel_ms_in_region`) is
        # that of a kernel with a third of the machine: kept as the extra key.
        k_ms_region = place_ms / args.steps
        k_ms = alone["place_ms"] if alone["place_ms"] > 0 else k_ms_region
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        node_evals = float(Q) * (info["n_nodes"] + info["n_muts"])
        prof_name, prof, prof_all = stored_profile(info, Q, packed)
        traffic = int(prof["hbm_read_bytes_per_dispatch_corrected"] + prof.get("hbm_write_bytes_per_dispatch", 0)) if prof else None
        # What the counters of the stored profile say bounds the kernel: instruction issue when a pipe is busy most of the time, HBM
        # when the measured traffic is most of the peak, else the waves are waiting -- exposed memory latency.  The issue floor is
        # the time the kernel's own instructions need on the busier of the two pipes (VALU: 2 cycles per wave64 instruction per
        # SIMD; scalar unit: one instruction per cycle per CU), from the profile's instruction counts and ITS duration.
        prof_ms = prof["avg_duration_ns_full_dispatch"] * 1e-6 if prof.get("avg_duration_ns_full_dispatch") else None
        issue_floor_ms = round(max(prof.get("valu_issue_frac") or 0.0, prof.get("salu_issue_frac") or 0.0) * prof_ms, 4) if prof_ms else None
        prof_hbm_frac = round(traffic / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and prof_ms else None
        if not prof:
            bound = "hbm"   # (no stored counters for this workload: SURVEY 8d's declared roofline)
        elif max(prof.get("valu_active_frac_measured") or 0.0, prof.get("salu_busy_frac_measured") or 0.0) > 0.6:
            bound = "issue"
        elif (prof_hbm_frac or 0.0) > 0.5:
            bound = "hbm"
        else:
            bound = "latency"
        roofline = {"bound": bound, "bound_declared": "hbm (SURVEY 8d)", "bound_is": "what the stored profile's counters say limits the kernel: a pipe busy > 60 % = issue, measured traffic > 50 % of peak = hbm, else latency (waves waiting)",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    # `achieved` is NOMINAL: bytes a full tree pass per tile is entitled to read / kernel time; exact pruning skips
                    # most of the pass, so the bytes actually moved are `traffic` (from the stored PMC profile, not this run)
                    "achieved_is": "nominal (SURVEY 8d algorithmic bytes / measured kernel time)",
                    "traffic_source": prof_name,
                    # measured traffic / the PROFILE's own kernel duration / peak (both from the same rocprofv3 passes)
                    "measured_hbm_frac": prof_hbm_frac, "profile_kernel_ms": round(prof_ms, 4) if prof_ms else None,
                    "issue_floor_ms": issue_floor_ms,
                    "frac_of_issue_floor": round(issue_floor_ms / alone["place_ms"], 4) if issue_floor_ms and alone["place_ms"] > 0 else None,
                    # the kernel is bound by instruction issue, not HBM: fractions of the chip's VALU / scalar issue slots
                    # over the kernel's duration, from the same stored profile
                    "valu_issue_frac": prof.get("valu_issue_frac"), "salu_issue_frac": prof.get("salu_issue_frac"),
                    "valu_active_frac_measured": prof.get("valu_active_frac_measured"), "salu_busy_frac_measured": prof.get("salu_busy_frac_measured"),
                    "wave_wait_frac": prof.get("wave_wait_frac"),
                    "kernel": "k_best8" if packed else "k_place<0>", "tile_samples": T, "kernel_ms": round(k_ms, 4), "algo_bytes_per_launch": int(algo_bytes),
                    "kernel_ms_is": "average launch duration with the device to itself (HIP events on the launch stream, %d stream-ordered calls right behind the timed region)" % n_alone,
                    # one call at a time (see above): the kernel without a second batch on the device, and that step's wall time
                    "kernel_ms_alone": round(alone["place_ms"], 4),
                    "frac_alone": round(algo_bytes / (alone["place_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if alone["place_ms"] > 0 else None,
                    "ms_per_step_alone": round(t_alone * 1e3, 3),
                    # the same launch inside the timed region, where it shares the chip with the walks of the other batches in flight
                    "kernel_ms_in_region": round(k_ms_region, 4),
                    "frac_in_region": round(algo_bytes / (k_ms_region * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms_region > 0 else None,
                    # all kernels of a step: HBM bytes by the counters of the stored profile (every kernel that runs once or twice per step,
                    # per-dispatch bytes x dispatches per step) / this run's ms_per_step / peak
                    "hbm_bytes_all_kernels_per_step": prof_all,
                    "hbm_frac_all_kernels": round(prof_all / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if prof_all and ms_per_step > 0 else None,
                    # In the timed region several launches of this kernel share the chip (each on its share of the resident wave slots):
                    # a launch's duration there is that of a kernel with a third of the machine, so `frac` -- the contract's definition,
                    # bytes per launch / in-region duration -- falls when the pipeline gets deeper even as `value` rises.  Per second of
                    # chip time the kernel accounts for algo_bytes / ms_per_step:
                    "walks_on_device_avg": round(k_ms_region / ms_per_step, 2) if ms_per_step > 0 else None,
                    "frac_per_chip_second": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms_per_step > 0 else None,
                    "overlap": "off (--no-overlap: ugp_place_device, stream-ordered)" if args.no_overlap else "consecutive ugp_place_device_overlapped calls: %d on the device at a time (internal streams, workspace sets, output buffers)" % depth,
                    "node_plus_mut_evals_decided_per_s": round(node_evals / (k_ms * 1e-3), 1) if k_ms > 0 else 0.0,
                    "coarse_ms": round(coarse_ms / args.steps, 4), "table_ms": round(table_ms / args.steps, 4), "merge_ms": round(merge_ms / args.steps, 4),
                    # exact third pruning bound: the handle decides per block of six calls, by its own measured throughput, whether
                    # to build its per-batch tables (UGP_BOUND3=0/1 pins it); table_ms includes them when it did
                    "third_bound_steps": int(tm.get("bound3", 0)), "third_bound_steps_alone": int(tma.get("bound3", 0)) if tma else None,
                    # the two pruning counters exist only in the instrumented kernel variant (UGP_STATS=1)
                    "pruned_frac": (round(skipped / wtotal, 4) if wtotal else 0.0) if os.environ.get("UGP_STATS") else None,
                    "prune_skips": nskips if os.environ.get("UGP_STATS") else None}
        # ---- CPU baseline: the literal oracle (port of mapper2_body + driver), node-parallel on the host cores
        cpu = None
        n_cpu = args.cpu_queries
        if n_cpu != 0 and world == 1 and Q > 2:
            from oracle import capi
            cores = os.cpu_count() or 1
            ot = capi.OracleTree(st.arrays)
            # one sample on one thread (calibration against SURVEY 6's 12.8 placements/s/thread at 44k nodes)
            t1 = time.perf_counter(); r1 = ot.place_mt(synth.csr_sample(q, Q - 1), 1); dt1 = time.perf_counter() - t1
            bad = int((r1["best"], r1["num_best"], r1["best_j"]) != (int(res[Q - 1, 0]), int(res[Q - 1, 1]), int(res[Q - 1, 2])))
            ot.place_mt(synth.csr_sample(q, 0), cores)   # starts the pool
            if n_cpu < 0:
                t1 = time.perf_counter(); ot.place_mt(synth.csr_sample(q, 0), cores); dt = time.perf_counter() - t1
                n_cpu = int(max(1, min(Q - 1, 15.0 / max(dt, 1e-4))))
            t1 = time.perf_counter()
            for i in range(n_cpu):
                r = ot.place_mt(synth.csr_sample(q, i), cores)
                if (r["best"], r["num_best"], r["best_j"]) != (int(res[i, 0]), int(res[i, 1]), int(res[i, 2])):
                    bad += 1
            dt = time.perf_counter() - t1
            cpu = {"value": round(n_cpu / dt, 3), "unit": "placements/s", "cores": cores, "kind": "port",
                   "sample": "first %d queries of rank 0's batch on the same MAT, oracle/ugp_oracle.c orc_place_sample_pool (pass 1 of "
                             "usher_common.cpp:389-414, one sample at a time, persistent pool of %d threads pulling 2,048-node ranges)" % (n_cpu, cores),
                   "value_1thread": round(1.0 / dt1, 4), "sample_1thread": "1 query, 1 thread",
                   # BASELINE.md / SURVEY 6: the true reference measured 12.8 placements/s/thread on a 44,483-node tree; its cost per
                   # sample is O(N x depth), so at N nodes one thread of it would do at most 12.8 x 44,483 / N (depth grows too:
                   # an upper estimate).  calibration = this port's one-thread rate / that figure.
                   "calibration": {"reference_1thread_scaled": round(12.8 * 44483.0 / info["n_nodes"], 5), "port_over_reference": round((1.0 / dt1) / (12.8 * 44483.0 / info["n_nodes"]), 3),
                                   "basis": "BASELINE.md: 12.8 placements/s/thread at 44,483 nodes, scaled by 44,483 / N"},
                   "mismatches_vs_gpu": bad}
        result = {
            "metric": "sample placements/sec on 10M-node MAT; bit-exact parsimony score vs reference",
            "value": round(value, 2), "unit": "placements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "u16 packed (4-bit allele sets, 4-bit SWAR node counters, 16-bit D/cost)" if packed else "u32 (4-bit allele sets, int32 counters)", "data": "synthetic",
            "config": {"workload": "synthetic %s MAT %d nodes / %d mutations / %d variable sites, L=%d; %s%s"
                                   % (args.shape, info["n_nodes"], info["n_muts"], info["n_sites"], args.genome,
                                      ("%d queries in total, sharded" % total_q) if args.strong else ("%d queries per GPU per step" % Q),
                                      (" (100-5000 N + 0-30 IUPAC cells each%s)" % (", every code holding the sample's own base" if args.iupac_true else "")) if args.ambiguous else ""),
                       "nodes": int(info["n_nodes"]), "queries_per_gpu": Q, "queries_total": total_q, "tile": T, "tiles": tiles, "waves_per_tile": groups,
                       "query_sets_rotated": n_sets, "last_step_equals_stream_ordered_call": rotation_exact,
                       "parallelism": "queries sharded x%d, MAT replicated, RCCL all-gather of results" % world,
                       "rccl_ranks": world, "devices_visible": torch.cuda.device_count(),
                       "seed": args.seed, "gen_s": round(t_gen, 2), "flatten_upload_s": round(t_flat, 2)},
            # the K-step window --repeats times in all: `value` is the median window
            "windows": {"n": len(windows), "value_is": "median window", "ms_per_step": [round(w * 1e3 / args.steps, 3) for w in windows],
                        "value_first_window": round(total_q * args.steps / windows[0], 2),
                        "value_min": round(total_q * args.steps / max(windows), 2), "value_median": round(total_q * args.steps / sorted(windows)[len(windows) // 2], 2),
                        "value_max": round(total_q * args.steps / min(windows), 2)},
            # SURVEY 8(d) defines the metric with query upload and result download inside the clock: this is that figure (two
            # batches in flight from and to host buffers), next to `value` (rows resident in HBM, as the bench contract asks)
            "value_pcie_inclusive": (host_path or {}).get("two_in_flight", {}).get("placements_per_s") if host_path and host_path.get("two_in_flight") else None,
            "per_rank": {"flatten_upload_s": flat_all, "flatten_threads": os.environ.get("UGP_FLATTEN_THREADS") or "library default", "flattened_once_per_node": bool(flat_file),
                         "all_gather_ms": gather_ms, "all_gather_bytes": int(world * cap * 16) if world > 1 else None,
                         "gather_issue_ms_per_step": round(sp.gather_host_s * 1e3 / max(1, n_step[0]), 4) if world > 1 else None},
            "other_configs": extra,
            "roofline": roofline, "cpu_baseline": cpu,
            # `value` has the query rows resident in HBM when the timed region starts (bench contract); the same batch through
            # the host-buffer entry point -- SURVEY 8(d)'s definition of the metric -- is this:
            "host_buffer_path": host_path,
        }
    for h in qsets:
        pl.free_qset(h)
    pl.close()
    if extra is not None:
        # config 3's size: a 15M-node SARS-CoV-2-shaped tree (the public MAT is not in the image), 10,000 queries
        try:
            t0 = time.time()
            st3 = synth.SynthTree(15_000_000, genome_len=args.genome, n_sites=25000, seed=args.seed, shape="sars2")
            t_gen3 = time.time() - t0
            t0 = time.time()
            pl3 = Placer(st3.arrays, device=dev_index)
            t_flat3 = time.time() - t0
            free_b, total_b = torch.cuda.mem_get_info(dev)
            c3 = timed_config(pl3, st3, 10_000, 12, 24, recent=True)   # (24 warm-up calls: a new handle -- its workspace sets are allocated and first touched in the first calls)
            i3 = pl3.info()
            c3.update({"workload": "BASELINE config 3's size: synthetic sars2-shaped MAT %d nodes / %d mutations, 10,000 queries per step" % (i3["n_nodes"], i3["n_muts"]),
                       "gen_s": round(t_gen3, 2), "flatten_upload_s": round(t_flat3, 2), "device_bytes_in_use": int(total_b - free_b)})
            extra["config3_15m_nodes_10k_queries"] = c3
            pl3.close()
            del st3
        except Exception as ex:
            extra["config3_15m_nodes_10k_queries"] = {"error": repr(ex)[:300]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
