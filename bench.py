#!/usr/bin/env python3
"""bench.py -- sample placements/sec of the MI355X placement engine.

One "step" = one pass of the hot path over one batch of synthetic query samples
(allele-tile build + ugp_place kernel + partial merge [+ RCCL all-gather of the
placements when N > 1]) with the flattened MAT and the query rows already
resident in HBM.  Workload (BASELINE.json metric): ~10M-node synthetic
SARS-CoV-2-scale MAT (L = 29,903, 25,000 variable sites), SARS-CoV-2-length
queries; `--nodes 100000 --sites 1500 --queries 1024` gives BASELINE config 1.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Queries shard across ranks (each rank places its own `--queries` samples: weak
scaling), the MAT is replicated, results are all-gathered over RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--sites", type=int, default=0, help="variable sites (default 25000 at >=1M nodes, else 1500)")
    ap.add_argument("--queries", type=int, default=16384, help="query samples per GPU per step")
    ap.add_argument("--genome", type=int, default=29903)
    ap.add_argument("--ambiguous", action="store_true", help="BASELINE config 5: 100-5000 N cells + 0-30 IUPAC cells per query")
    ap.add_argument("--cpu-queries", type=int, default=-1, help="queries timed on the CPU oracle (0 = skip; default: sized for ~10-30 s)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sort-by-source", action="store_true", help="experiment: hand the queries over already ordered by their generator source node")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the placement path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from usher_amd import Placer, QueryBatch, synth

    n_sites = args.sites or (25000 if args.nodes >= 1_000_000 else 1500)
    t0 = time.time()
    st = synth.SynthTree(args.nodes, genome_len=args.genome, n_sites=n_sites, seed=args.seed)
    t_gen = time.time() - t0
    t0 = time.time()
    pl = Placer(st.arrays, device=local_rank)
    t_flat = time.time() - t0
    info = pl.info()
    kw = dict(n_lo=100, n_hi=5000, iupac_hi=30) if args.ambiguous else {}
    q = st.queries(args.queries, seed=args.seed * 1000 + 17 + rank, **kw)
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
    out = torch.zeros((Q, 4), dtype=torch.int32, device=dev)
    gathered = torch.zeros((world * Q, 4), dtype=torch.int32, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        pl.place_device(qset, out.data_ptr(), stream)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    place_ms = table_ms = merge_ms = 0.0
    tiles = groups = packed = skipped = wtotal = nskips = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = pl.timing()   # HIP events recorded by the library on `stream` around each kernel of this step
        place_ms += tm["place_ms"]; table_ms += tm["table_ms"]; merge_ms += tm["merge_ms"]
        tiles, groups, packed = tm["n_tiles"], tm["n_groups"], tm["packed_path"]
        skipped, wtotal, nskips = tm["words_skipped"], tm["words_total"], tm["reserved"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    res = out.cpu().numpy()
    host_path = None
    if rank == 0 and world == 1:
        pl.place(batch)   # warm
        t0 = time.perf_counter()
        for _ in range(2):
            hres = pl.place(batch)
        dt = (time.perf_counter() - t0) / 2
        host_path = {"placements_per_s": round(Q / dt, 2), "ms_per_batch": round(dt * 1e3, 3),
                     "identical_to_device_path": bool((np.stack([hres[k] for k in ("best_set_difference", "num_best", "best_j", "best_has_unique")], 1).astype(np.int64) == res.astype(np.int64)).all())}
    result = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * Q * args.steps / elapsed
        # roofline of the dominant kernel (k_place): algorithmic bytes per launch, SURVEY 8(d):
        # every T-sample tile makes one pass over the tree: B_tree + T * (L/2 + 16); T = 512 on the packed path
        T = 512 if packed else 64
        algo_bytes = tiles * (info["algo_tree_bytes"] + T * info["algo_tile_bytes"])
        k_ms = place_ms / args.steps
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        node_evals = float(Q) * (info["n_nodes"] + info["n_muts"])
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass of this same
        # workload (tools/profile_round.sh -> profiles/rNN_pmc_summary.json; FETCH_SIZE doubled as the
        # gfx950 note in MI355X_MICROARCH.md prescribes); null when the workload differs.
        traffic = None
        try:
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
                with open(fn) as f:
                    ps = json.load(f)
                cfg = ps.get("bench", {}).get("config", {})
                want = "ugp::k_best8" if packed else "ugp::k_place<0>"
                k = next((v for n, v in ps.get("kernels", {}).items() if n.startswith(want)), {})
                if cfg.get("nodes") == int(info["n_nodes"]) and cfg.get("queries_per_gpu") == Q and "hbm_read_bytes_per_dispatch_corrected" in k:
                    traffic = int(k["hbm_read_bytes_per_dispatch_corrected"] + k.get("hbm_write_bytes_per_dispatch", 0))
                    break
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "k_best8" if packed else "k_place<0>", "tile_samples": T, "kernel_ms": round(k_ms, 4), "algo_bytes_per_launch": int(algo_bytes),
                    "node_plus_mut_evals_per_s": round(node_evals / (k_ms * 1e-3), 1) if k_ms > 0 else 0.0,
                    "table_ms": round(table_ms / args.steps, 4), "merge_ms": round(merge_ms / args.steps, 4),
                    # the two pruning counters exist only in the instrumented kernel variant (UGP_STATS=1)
                    "pruned_frac": (round(skipped / wtotal, 4) if wtotal else 0.0) if os.environ.get("UGP_STATS") else None,
                    "prune_skips": nskips if os.environ.get("UGP_STATS") else None}
        # ---- CPU baseline: the literal oracle (port of mapper2_body + driver), node-parallel on the host cores
        cpu = None
        n_cpu = args.cpu_queries
        if n_cpu != 0:
            from oracle import capi
            cores = os.cpu_count() or 1
            ot = capi.OracleTree(st.arrays)
            if n_cpu < 0:
                s0 = synth.csr_sample(q, 0)
                t1 = time.perf_counter(); r0 = ot.place_mt(s0, cores); dt = time.perf_counter() - t1
                n_cpu = int(max(1, min(Q - 1, 15.0 / max(dt, 1e-4))))
            t1 = time.perf_counter()
            bad = 0
            for i in range(n_cpu):
                r = ot.place_mt(synth.csr_sample(q, i), cores)
                if (r["best"], r["num_best"], r["best_j"]) != (int(res[i, 0]), int(res[i, 1]), int(res[i, 2])):
                    bad += 1
            dt = time.perf_counter() - t1
            cpu = {"value": round(n_cpu / dt, 3), "unit": "placements/s", "cores": cores, "kind": "port",
                   "sample": "first %d queries of rank 0's batch on the same MAT, oracle/ugp_oracle.c orc_place_sample_mt (pass 1 of "
                             "usher_common.cpp:389-414, one sample at a time, %d threads over nodes)" % (n_cpu, cores),
                   "mismatches_vs_gpu": bad}
        result = {
            "metric": "sample placements/sec on 10M-node MAT; bit-exact parsimony score vs reference",
            "value": round(value, 2), "unit": "placements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 packed (4-bit allele sets, 4-bit SWAR node counters, 16-bit D/cost)" if packed else "u32 (4-bit allele sets, int32 counters)", "data": "synthetic",
            "config": {"workload": "synthetic MAT %d nodes / %d mutations / %d variable sites, L=%d; %d queries per GPU per step%s"
                                   % (info["n_nodes"], info["n_muts"], info["n_sites"], args.genome, Q,
                                      " (100-5000 N + 0-30 IUPAC cells each)" if args.ambiguous else ""),
                       "nodes": int(info["n_nodes"]), "queries_per_gpu": Q, "tile": T, "tiles": tiles, "waves_per_tile": groups,
                       "parallelism": "queries sharded x%d, MAT replicated, RCCL all-gather of results" % world,
                       "seed": args.seed, "gen_s": round(t_gen, 2), "flatten_upload_s": round(t_flat, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
            # the same batch through the host-buffer entry point (ugp_place_batch: query upload over PCIe, tile
            # build, kernels, result download), outside the timed region; never `value`
            "host_buffer_path": host_path,
        }
    pl.free_qset(qset)
    pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
