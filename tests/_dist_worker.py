"""Worker of tests/test_multi_gpu.py: one rank per GPU (torchrun), queries sharded through
usher_amd.dist.place_sharded with the RCCL all-gather, checked on every rank against the closed form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import capi
    from usher_amd import Placer, QueryBatch, synth
    from usher_amd.dist import place_sharded
    local = int(os.environ["LOCAL_RANK"])
    # DIST_SHARE_DEVICE=1 (a one-GPU box): the ranks share the visible device(s) and gather through gloo -- every line of this
    # worker and of usher_amd.dist.place_sharded runs as on N GPUs except the collective's backend
    share = bool(os.environ.get("DIST_SHARE_DEVICE"))
    dev_index = local % torch.cuda.device_count() if share else local
    torch.cuda.set_device(dev_index)
    if share:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    st = synth.SynthTree(300_000, n_sites=4000, seed=12)
    q = st.queries(5001, seed=99, n_lo=0, n_hi=30, iupac_hi=3)      # odd count: shards differ in size
    batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    pl = Placer(st.arrays, device=dev_index)
    # the Placer itself: the device path (resident shard -> ugp_place_device into a device tensor -> all-gather of that tensor)
    res = place_sharded(pl, batch, device="cpu" if share else None)
    # more ranks than samples: the empty shards still join the gather
    tiny = place_sharded(pl, batch.slice(0, 1), device="cpu" if share else None)
    assert len(tiny) == 1 and (tiny.view(np.int32) == res[:1].view(np.int32)).all()
    pl.close()
    cf = capi.ClosedFormC(capi.OracleTree(st.arrays)).place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    ok = (res["best_set_difference"].astype(np.int64) == cf["best"]).all() and (res["num_best"].astype(np.int64) == cf["num_best"]).all() \
        and (res["best_j"].astype(np.int64) == cf["best_j"]).all() and (res["best_has_unique"].astype(bool) == cf["has_unique"]).all()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d %s" % (int(os.environ["RANK"]), "OK" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
