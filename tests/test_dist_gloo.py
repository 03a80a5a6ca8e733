"""N > 1 path on CPUs: world_size-2 gloo run of the candidate-hit all-gather (folddisco_amd/dist.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from folddisco_amd import dist as fdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 1001
    lo, hi = fdist.shard_range(rank, world, S)
    rng = np.random.Generator(np.random.PCG64(5))
    allrec = np.zeros(S, fdist.REC_DTYPE)
    allrec["nid"] = np.arange(S)
    allrec["idf"] = np.round(rng.uniform(0, 3, S), 1).astype(np.float32)   # many ties
    allrec["total_match_count"] = rng.integers(1, 9, S)
    touched = rng.uniform(size=S) < 0.4
    local = allrec[lo:hi][touched[lo:hi]]
    got_all = fdist.allgather_hits(local, torch.device("cpu"))
    got_top = fdist.allgather_hits(local, torch.device("cpu"), top_n=37)
    want_all = fdist.rank_hits(allrec[touched])
    ok = np.array_equal(got_all, want_all) and np.array_equal(got_top, want_all[:37])
    # empty shard contribution
    e = fdist.allgather_hits(local[:0] if rank == 1 else local, torch.device("cpu"), top_n=5)
    ok = ok and len(e) == min(5, int(touched[: fdist.shard_range(0, world, S)[1]].sum()))
    # posting lengths of the shards add up to the lengths over the whole database (the idf denominator of a sharded query)
    per_rank = [np.array([3, 0, 7, 2 ** 33], np.uint64), np.array([1, 0, 0, 5], np.uint64)]
    tot = fdist.reduce_lengths(per_rank[rank])
    ok = ok and tot.dtype == np.uint64 and tot.tolist() == [4, 0, 7, 2 ** 33 + 5]
    # batched form: many queries, two collectives; generic fixed-size records; scalar sum
    many_local = [local, local[:0], local[: len(local) // 2]]
    many = fdist.allgather_hits_many(many_local, torch.device("cpu"), top_n=50)
    for got, loc in zip(many, many_local):
        ok = ok and np.array_equal(got, fdist.allgather_hits(loc, torch.device("cpu"), top_n=50))
    dt = np.dtype([("a", np.uint32), ("v", np.float32, (3,))])
    mine = np.zeros(3 + 2 * rank, dt)
    mine["a"] = 100 * rank + np.arange(len(mine))
    cat = fdist.allgather_array(mine, torch.device("cpu"))
    ok = ok and cat["a"].tolist() == [0, 1, 2, 100, 101, 102, 103, 104]
    ok = ok and fdist.allreduce_sum(rank + 5) == 11
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_hits_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
