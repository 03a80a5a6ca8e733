"""Motif-query leg of bench.py: Q planted motif queries scored against the resident shard(s).

A query = 4 residues of a shard structure that lie within 10 A of each other (so it has hits by construction),
default expansion (-d 0.5 -a 5).  Per query every rank runs make_query_map (features / hashes / posting lengths on
the GPU), count_query on its shard, keeps its top-N candidates and the ranks all-gather the candidate records
(RCCL).  queries/s = Q / wall time of the loop (max over ranks).  With match=True the top candidates of the local
shard additionally go through the pair scan + graph + Kabsch (retrieval)."""
from __future__ import annotations

import time

import numpy as np
import torch

from . import dist as fdist
from .api import PackedStructures, count_query, count_query_batch, idf_of_lengths, length_penalty
from .query import make_query_map, make_query_maps, retrieve, retrieve_batch


def _pick_queries(d, S, n_queries, seed, k=4):
    rng = np.random.Generator(np.random.PCG64(seed))
    off = d["res_off"].cpu().numpy()
    out = []
    tries = 0
    while len(out) < n_queries and tries < 50 * n_queries:
        tries += 1
        s = int(rng.integers(0, S))
        a, b = int(off[s]), int(off[s + 1])
        if b - a < 30:
            continue
        ca = d["ca_xyz"][a:b].cpu().numpy()
        c = int(rng.integers(0, b - a))
        near = np.nonzero(np.linalg.norm(ca - ca[c], axis=1) < 10.0)[0]
        if len(near) < k:
            continue
        idx = np.sort(rng.choice(near, size=k, replace=False))
        item = dict(n_xyz=d["n_xyz"][a:b].cpu().numpy(), ca_xyz=ca, cb_xyz=d["cb_xyz"][a:b].cpu().numpy(), aa=d["aa"][a:b].cpu().numpy())
        out.append((s, idx.astype(np.uint32), item))
    return out


def run(ctx, batch, ix, d, S, world, rank, dist, dev, n_queries=64, top_n=1000, match_top=32, seed=4242):
    # rank 0 cuts the motifs out of its shard and broadcasts them: every rank scores the SAME queries against its own shard
    queries = _pick_queries(d, S, n_queries, seed) if rank == 0 else None
    if dist is not None:
        box = [queries]
        dist.broadcast_object_list(box, src=0, device=dev if dist.get_backend() == "nccl" else None)
        queries = box[0]
    nres = np.diff(d["res_off"].cpu().numpy()).astype(np.uint64)
    pen = length_penalty(nres, 0.5)
    S_total = S * world
    sharded = dist is not None          # every rank holds the postings of its own structures only
    qbatches = [ctx.upload(PackedStructures.concat([it])) for _, _, it in queries]
    qall = ctx.upload(PackedStructures.concat([it for _, _, it in queries]))    # every query structure in one batch (batched legs)

    def one(k, match):
        s, idx, _ = queries[k]
        qm = make_query_map(ctx, qbatches[k], idx, None, None if sharded else ix, float(S_total))
        lens = fdist.global_posting_lengths(ix, qm.hash, dev) if sharded else None      # idf over the whole database
        if sharded and match:
            pl = fdist.global_posting_lengths(ix, qm.primary_hash, dev)
            qm.set_idf(np.where(pl > 0, idf_of_lengths(np.maximum(pl, 1), S_total), 0.0).astype(np.float32))
        recs = count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S_total, as_array=True, lengths=lens)
        glob = fdist.allgather_hits(recs, dev, top_n=top_n)
        n_match = 0
        if match and len(recs):
            local_top = fdist.rank_hits(recs, match_top)
            cand = (local_top["nid"] - ix.first_id).astype(np.uint32)
            ms = retrieve(ctx, batch, None, cand, qm, qbatches[k])
            n_match = len(ms)
        return len(glob), len(qm.hash), n_match

    def timed(match):
        for k in range(min(8, len(queries))):   # warm: scratch buffers reach their steady-state sizes
            one(k, match)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        tot_hits = tot_hashes = tot_m = 0
        for k in range(len(queries)):
            h, q, m = one(k, match)
            tot_hits += h; tot_hashes += q; tot_m += m
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, tot_hits, tot_hashes, tot_m

    def batched(chunk=32, match=False):
        """throughput mode: query maps per query, then ONE posting-length launch + ONE scoring pass per chunk of queries;
        with match=True the local top candidates of every query additionally go through retrieval"""
        def go():
            tot = 0
            for c0 in range(0, len(queries), chunk):
                ks = range(c0, min(c0 + chunk, len(queries)))
                qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], ix if (match and not sharded) else None, float(S_total))
                if match and sharded:
                    for qm in qms:
                        pl = fdist.global_posting_lengths(ix, qm.primary_hash, dev)
                        qm.set_idf(np.where(pl > 0, idf_of_lengths(np.maximum(pl, 1), S_total), 0.0).astype(np.float32))
                recs = count_query_batch(ctx, ix, [(qm.hash, qm.qi, qm.qj) for qm in qms], pen, total_structures=S_total, top_n=top_n,
                                         lengths_fn=(lambda l: fdist.reduce_lengths(l, dev)) if sharded else None)
                cl = []
                for k, qm, r in zip(ks, qms, recs):
                    n = len(fdist.allgather_hits(r, dev, top_n=top_n))
                    if match:
                        cl.append((fdist.rank_hits(r, match_top)["nid"] - ix.first_id).astype(np.uint32) if len(r) else np.zeros(0, np.uint32))
                    else:
                        tot += n
                if match:   # one pair scan / gather / Kabsch launch for the whole chunk of queries
                    tot += len(retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0])
            return tot
        go()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        tot = go()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, tot

    dt1, hits, hashes, _ = timed(False)
    dtb, hits_b = batched()
    dtbm, nm_b = batched(match=True)
    dt2, _, _, nm = timed(True)
    # roofline of the scoring kernel for the last query (HIP events on the context's stream)
    ctx.enable_timing(True)
    s, idx, _ = queries[-1]
    qm = make_query_map(ctx, qbatches[-1], idx, None, ix, float(S_total))
    lens = ix.posting_lengths(qm.hash)
    rows = count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S_total)
    ctx.synchronize()
    st = {n: ms for n, ms, _ in ctx.last_timings()}
    ctx.enable_timing(False)
    return {
        # headline = the reference's default query (prefilter + candidate selection + matching + RMSD), 32 queries per launch set
        "metric": "motif queries/sec", "value": len(queries) / dtbm, "unit": "queries/s", "n_queries": len(queries),
        "mode": "full query (make_query_map, count_query, all-gather + top-N, retrieval of the top %d candidates, Kabsch, metrics), "
                "batches of 32 queries" % match_top,
        "ms_per_query": dtbm / len(queries) * 1e3,
        "batched_with_matching": {"value": len(queries) / dtbm, "ms_per_query": dtbm / len(queries) * 1e3, "matches": nm_b, "match_top": match_top},
        "batched": {"value": len(queries) / dtb, "ms_per_query": dtb / len(queries) * 1e3, "chunk": 32, "avg_hits": hits_b / len(queries),
                    "mode": "prefilter only: make_query_map_batch + count_query_batch_top + all-gather, eight launches per 32 queries"},
        "single": {"value": len(queries) / dt1, "ms_per_query": dt1 / len(queries) * 1e3, "mode": "prefilter only, one query per call"},
        "with_matching": {"value": len(queries) / dt2, "ms_per_query": dt2 / len(queries) * 1e3, "matches": nm, "match_top": match_top,
                          "mode": "full query, one query per call"},
        "avg_query_hashes": hashes / len(queries), "avg_hits": hits / len(queries),
        "last_query": {"hashes": int(len(qm.hash)), "postings_decoded": int(lens.sum()), "touched": len(rows), "stages_ms": st},
    }
