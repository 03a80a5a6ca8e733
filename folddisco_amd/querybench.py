"""Motif-query leg of bench.py: Q planted motif queries scored against the resident index of the rank's shard.

A query = 4 residues of a database structure that lie within 10 A of each other (so it has hits by construction),
default expansion (-d 0.5 -a 5).  The index and the coordinates are sharded by structure id over the ranks;
per batch of 32 queries every rank runs make_query_map_batch (features / hashes on the GPU), all-reduces the posting
lengths (idf over the WHOLE database), scores its shard (count_query_batch_top), the ranks all-gather the candidate
records (RCCL) and rank them globally (idf descending, top_n = 1000).  With matching, the first match_top candidates of
the GLOBAL ranking are retrieved on the rank that owns them (pair scan + components + Kabsch) and the match records are
all-gathered.  queries/s = Q / wall time of the loop (max over ranks).

Roofline: B_q = sum of posting bytes of the query's hashes + 8 T (touched structures) + (nodes + edges) S / 8,
over the HIP-event time of the scoring stage (k_cq_seg + k_cq_bounds + k_cq_rows_finalize + compaction scan).
cpu_baseline: supplied by bench.py as a callback (the product package never touches oracle/)."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import dist as fdist
from .api import PackedStructures, count_query, count_query_batch, count_query_maps, idf_of_lengths, length_penalty
from .query import MATCH_DTYPE, make_query_map, make_query_maps, query_batch, query_batch_submit, retrieve, retrieve_batch


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


def _pmc_query_traffic(S_total, world, batch=32):
    """HBM bytes per batch of the prefilter kernels from the committed rocprofv3 PMC passes of the batched query (profiles/*pmc_query_traffic_*.json,
    tools/pmc_query_traffic.sh): sum over the kernels of 2 x FETCH_SIZE + WRITE_SIZE (gfx950 corrections as in bench.py)"""
    import glob
    import json
    tag = "S%d" % S_total if world == 1 else "S%d_N%d" % (S_total, world)
    if batch != 32:
        tag += "_B%d" % batch      # passes over batches of another size (tools/profile_round5.sh: _B128 = the headline's batch)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "*pmc_query_traffic_%s.json" % tag)), reverse=True):
        try:
            dd = json.load(open(f))
            # FETCH_SIZE counts a read request made for a 16-byte-per-lane access at half its size on gfx950: every kernel carries its own
            # correction (1 + the share of its read bytes that such loads fetch; files of rounds 1-3: 2 for every kernel)
            lo = mid = hi = 0.0
            for v in dd["kernels"].values():
                n = v.get("launches_per_batch", 1)
                lo += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * n
                mid += (v["fetch_bytes_per_launch"] * v.get("fetch_correction", 2.0) + v["write_bytes_per_launch"]) * n
                hi += (2.0 * v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * n
            return {"bytes_per_launch": mid, "bytes_per_launch_raw_counters": lo, "bytes_per_launch_all_reads_doubled": hi, "kernels": sorted(dd["kernels"]),
                    "source": "committed PMC pass: profiles/" + os.path.basename(f), "measured_in_this_run": False}
        except Exception:
            continue
    return None


def run(ctx, batch, ix, d, S, world, rank, dist, dev, n_queries=64, top_n=1000, match_top=32, seed=4242, lo=0, S_total=None,
        cpu_baseline_fn=None, hbm_peak_gbs=8000.0):
    S_total = S * world if S_total is None else S_total
    sharded = dist is not None          # every rank holds the postings of its own structures only
    # rank 0 cuts the motifs out of its shard and broadcasts them: every rank scores the SAME queries against its own shard
    queries = _pick_queries(d, S, n_queries, seed) if rank == 0 else None
    if sharded:
        box = [queries]
        dist.broadcast_object_list(box, src=0, device=dev if dist.get_backend() == "nccl" else None)
        queries = box[0]
    res_off_h = d["res_off"].cpu().numpy()
    nres = np.diff(res_off_h).astype(np.uint64)
    pen = length_penalty(nres, 0.5)
    ix.set_penalty(pen)
    qbatches = [ctx.upload(PackedStructures.concat([it])) for _, _, it in queries]
    qall = ctx.upload(PackedStructures.concat([it for _, _, it in queries]))    # every query structure in one batch (batched legs)
    first = ix.first_id
    # the exchange: libfdgpu's own RCCL communicator when every rank has its GPU (backend nccl), torch.distributed (gloo) otherwise —
    # both go through the same fused library entry points (dist.sharded_count_query_maps)
    comm = None
    if sharded and dist.get_backend() == "nccl" and os.environ.get("FD_BENCH_COMM", "rccl") == "rccl":
        comm = fdist.Comm(ctx, rank, world)

    def sharded_prefilter(qms):
        globs = fdist.sharded_count_query_maps(ctx, ix, qms, None, S_total, top_n, dev, comm)
        for q in qms:
            q._cache.pop("idf", None)       # rewritten inside the library from the global posting lengths
        return globs

    def sharded_matches(globs, qms, ks):
        """matches of the first match_top candidates of every query's GLOBAL ranking, each on the rank that owns it -> number of
        match records over all ranks (the records themselves are all-gathered)"""
        if comm is not None:
            return len(comm.sharded_retrieve(batch, first, None, [g["nid"][:match_top] for g in globs], qms, qall, list(ks))[0])
        cl = [owned(g, match_top) for g in globs]
        marr = retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0]
        return len(fdist.allgather_array(marr, dev))

    def top_cands(recs, off, n):
        """[T, n] candidate structures = the first n records of every query's ranking, straight from the library's flat output
        (None when a query has fewer than n records: the caller takes the per-query path)"""
        if len(off) < 2 or int((off[1:] - off[:-1]).min()) < n:
            return None
        c = recs["nid"][(off[:-1, None] + np.arange(n, dtype=np.uint64)[None, :]).astype(np.int64)]
        return c - np.uint32(first) if first else c

    def owned(glob, n):
        """candidate slots (local structure indices) of the first n records of the global ranking that this rank owns"""
        nid = glob["nid"][:n].astype(np.int64)
        mine = nid[(nid >= first) & (nid < first + S)]
        return (mine - first).astype(np.uint32)

    def one(k, match):
        s, idx, _ = queries[k]
        if sharded:     # lengths all-reduced, local selection, all-gather, global ranking; matching on the owning ranks
            qa = make_query_maps(ctx, qall, [(k, idx)], None, float(S_total))
            glob = sharded_prefilter(qa)[0]
            n_match = sharded_matches([glob], qa, [k]) if match else 0
            return len(glob), len(qa[0].hash), n_match
        qm = make_query_map(ctx, qbatches[k], idx, None, ix, float(S_total))
        glob = count_query_maps(ctx, ix, [qm], None, total_structures=S_total, top_n=top_n)[0]     # posting lengths, idf, scoring, top_n ranked on the device
        n_match = 0
        if match and len(glob):
            cand = owned(glob, match_top)
            n_match = len(retrieve(ctx, batch, None, cand, qm, qbatches[k])) if len(cand) else 0
        return len(glob), len(qm.hash), n_match

    def timed(fn):
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        dt = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    def loop(match, ks):
        def go():
            th = tq = tm = 0
            for k in ks:
                h, q, m = one(k, match)
                th += h; tq += q; tm += m
            return th, tq, tm
        return go

    def batched(chunk=32, match=False, reps=1):
        """throughput mode: query maps per chunk of queries, ONE posting-length launch + ONE scoring pass per chunk; with
        match=True the first match_top candidates of every query's GLOBAL ranking additionally go through retrieval on
        their owning rank (one pair scan / gather / Kabsch launch per chunk) and the match records are all-gathered"""
        def go():
            tot = 0
            for c0 in range(0, len(queries), chunk):
                ks = range(c0, min(c0 + chunk, len(queries)))
                qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], ix if (match and not sharded) else None, float(S_total))
                if sharded:     # the same fused entry points as the single-index leg, with the exchange in between
                    globs = sharded_prefilter(qms)
                else:           # the query maps go back into the library as they are; the penalty is resident (set_penalty)
                    recs, off = count_query_maps(ctx, ix, qms, None, total_structures=S_total, top_n=top_n, flat=True)
                if match:   # one pair scan / gather / Kabsch launch for the whole chunk of queries
                    if sharded:
                        tot += sharded_matches(globs, qms, ks)
                    else:       # one rank owns every structure: the candidates are the first match_top records of every ranking
                        cl = top_cands(recs, off, match_top)
                        if cl is None:
                            cl = [recs["nid"][int(off[t]): int(off[t]) + min(match_top, int(off[t + 1] - off[t]))] - np.uint32(first) for t in range(len(qms))]
                        tot += len(retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0])
                else:
                    tot += sum(len(g) for g in globs) if sharded else len(recs)
            return tot
        go()
        if reps > 1:       # the headline leg: the median of several passes (a pass over 128 queries is a few milliseconds)
            runs = sorted((timed(go) for _ in range(reps)), key=lambda x: x[0])
            return runs[len(runs) // 2]
        return timed(go)

    MT_REPS = 12
    PIPE_SHAPE = (6, 10)    # lanes, batches in flight of the headline leg (the library's default lane count; swept in round 5: 4x6 150 k, 4x12 154 k, 6x10 159 k, 8x16 161 k queries/s)
    PIPE_REPS = 48          # passes over the query set per timed run of the pipelined leg (batches of 128 -> 24 batches in the pipeline)

    def batched_mt(workers=2, chunk=32):
        """the full batched query from `workers` host threads, one context (stream + workspaces) each, sharing the resident index and
        coordinates — how a rayon host drives the C ABI (query_pdb.rs:348: queries.into_par_iter()): one thread's table building and
        copies overlap the other's kernels.  Single-rank only."""
        from concurrent.futures import ThreadPoolExecutor
        from .api import Context
        ctxs = [ctx] + [Context(torch.cuda.current_device()) for _ in range(workers - 1)]
        starts = list(range(0, len(queries), chunk)) * MT_REPS      # the query set MT_REPS times over: several batches per thread

        def work(w):
            cx, tot = ctxs[w], 0
            for c0 in starts[w::workers]:
                ks = range(c0, min(c0 + chunk, len(queries)))
                if first == 0:      # one fused library call per batch (fdgpu_query_batch)
                    tot += len(query_batch(cx, ix, batch, qall, [(k, queries[k][1]) for k in ks], float(S_total), top_n, match_top)[2][0])
                    continue
                qms = make_query_maps(cx, qall, [(k, queries[k][1]) for k in ks], ix, float(S_total))
                recs, off = count_query_maps(cx, ix, qms, None, total_structures=S_total, top_n=top_n, flat=True)
                cl = top_cands(recs, off, match_top)
                if cl is None:
                    cl = [recs["nid"][int(off[t]): int(off[t]) + min(match_top, int(off[t + 1] - off[t]))] - np.uint32(first) for t in range(len(qms))]
                tot += len(retrieve_batch(cx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0])
                del qms
            cx.synchronize()
            return tot

        def go():
            with ThreadPoolExecutor(workers) as ex:
                return sum(ex.map(work, range(workers)))
        go()
        out = timed(go)
        for cx in ctxs[1:]:
            cx.close()
        return out

    pipe_runs = {}      # queries/s of the timed runs of every shape (the median is the leg's value)

    def run_pipe():
        """{(lanes, in flight): queries/s}, error — batches of 128 through the context's query lanes, one host thread"""
        out = {}
        if not (big_ok and not sharded and first == 0):
            return out, None
        try:
            from collections import deque
            ks1 = list(range(len(queries)))
            chunks = [[(k, queries[k][1]) for k in ks1[c0:c0 + 128]] for c0 in range(0, len(ks1), 128)]

            def go_pipe(reps, depth):
                pend, tot = deque(), 0
                for _ in range(reps):
                    for qs in chunks:
                        pend.append(query_batch_submit(ctx, ix, batch, qall, qs, float(S_total), top_n, match_top))
                        if len(pend) >= depth:
                            tot += len(pend.popleft().wait()[2][0])
                while pend:
                    tot += len(pend.popleft().wait()[2][0])
                return tot
            nm1 = go_pipe(1, 1)
            for lanes, depth in [tuple(int(y) for y in x.split(":")) for x in os.environ.get("FD_BENCH_PIPE", "%d:%d" % PIPE_SHAPE).split(",")]:
                n_l = ctx.L.fdgpu_query_lanes(ctx.h, lanes)
                assert n_l >= lanes, ctx.L.fdgpu_last_error(ctx.h)
                # warm-up AT THE SHAPE that is timed: every lane allocates its scratch, and the result blocks of `depth` batches in flight are page-locked
                # once (warmed with fewer in flight than timed, the first shape of a run measured 12-18 % below the same shape measured later)
                # ... and LONG enough: tools/pipe_variance.py, ten timed runs of 48 batches in one process behind a warm-up of 24 batches: 120, 133, 164, 161,
                # 163, 161, 162, 162 k queries/s — the lanes reach their pace after ~100 batches (the same on either NUMA node), and a median of three
                # runs behind a short warm-up came out at 135 k or at 160 k from one bench run to the next
                go_pipe(2 * n_l, n_l)
                go_pipe(2 * PIPE_REPS, depth)
                # The long run that showed up in one bench line in five was THIS harness: a generation-2 pass of Python's cyclic collector over the
                # result views of the batches in flight — one gap of 19 ms between two completed batches in one run of ten with the collector on, ten
                # runs within +-2.7 % with it off (profiles/round6_pipelined_leg_gc_ab_S542000.txt, tools/pipe_variance.py --gc-ab).  A Rust or C++
                # caller has no such pause: collected once, then off for the timed runs.  The value stays the median of seven; the runs are in the line.
                import gc
                gc.collect()
                gc.disable()
                try:
                    raw = [timed(lambda: go_pipe(PIPE_REPS, depth)) for _ in range(7)]
                finally:
                    gc.enable()
                runs = sorted(raw, key=lambda x: x[0])
                assert runs[3][1] == PIPE_REPS * nm1
                out[(n_l, depth)] = len(queries) * PIPE_REPS / runs[3][0]
                pipe_runs[(n_l, depth)] = [len(queries) * PIPE_REPS / r[0] for r in raw]      # in the order they ran
            return out, None
        except Exception as e:
            return out, repr(e)[:300]

    def progress(msg):
        if os.environ.get("FD_BENCH_TRACE"):
            print("[querybench %s r%d] %s" % (time.strftime("%H:%M:%S"), rank, msg), file=sys.stderr, flush=True)
    warm = range(min(8, len(queries)))
    progress("queries picked")
    loop(False, warm)()
    progress("warm-up done")
    big_ok = len(queries) >= 128
    pipe, err_pipe = {}, None
    if os.environ.get("FD_BENCH_PIPE_EARLY"):      # measurement aid: the pipelined leg before the other legs
        pipe, err_pipe = run_pipe()
        progress("pipelined leg (early): %r %r" % (pipe, err_pipe))
    dt1, (hits, hashes, _) = timed(loop(False, range(len(queries))))
    progress("single prefilter leg done")
    dtb, hits_b = batched()
    progress("batched prefilter leg done")
    dtbm, nm_b = batched(match=True)
    progress("batched full leg done")
    big = 128 if len(queries) >= 128 else None          # the same full query in batches of 128 (one host thread): the per-batch synchronisations amortise
    dtbm_big = batched(chunk=big, match=True, reps=7)[0] if big else None
    # ... and 512 per batch: the 128 queries four times over in ONE batch (every copy is scored and retrieved like any other query)
    dtbm_512, err_512 = None, None
    if big and not sharded:
        try:
            ks4 = list(range(len(queries))) * 4

            def go512():
                qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks4], ix, float(S_total))
                recs, off = count_query_maps(ctx, ix, qms, None, total_structures=S_total, top_n=top_n, flat=True)
                cl = top_cands(recs, off, match_top)
                if cl is None:
                    cl = [recs["nid"][int(off[t]): int(off[t]) + min(match_top, int(off[t + 1] - off[t]))] - np.uint32(first) for t in range(len(qms))]
                return len(retrieve_batch(ctx, batch, None, cl, qms, qall, ks4, as_arrays=True)[0])
            go512()
            dtbm_512, nm_512 = timed(go512)
            assert nm_512 == 4 * nm_b, (nm_512, nm_b)
        except Exception as e:      # the other legs stand on their own
            dtbm_512, err_512 = None, repr(e)[:300]
    # the same three stages as ONE library call per batch (fdgpu_query_batch: the retrieval's tables are built while the scoring kernels run, the
    # ranked records cross the bus while the retrieval runs) — single index only (the sharded form has an exchange between the stages)
    dt_fused, dt_fused_512, err_fused = None, None, None
    if big and not sharded and first == 0:
        try:
            def go_fused(ks_all, chunk):
                tot = 0
                for c0 in range(0, len(ks_all), chunk):
                    ks = ks_all[c0:c0 + chunk]
                    _, _, (marr, _, _, _) = query_batch(ctx, ix, batch, qall, [(k, queries[k][1]) for k in ks], float(S_total), top_n, match_top)
                    tot += len(marr)
                return tot
            ks1 = list(range(len(queries)))
            assert go_fused(ks1, big) == nm_b, "fused call: match count differs from the three calls"
            runs = sorted((timed(lambda: go_fused(ks1, big)) for _ in range(7)), key=lambda x: x[0])
            dt_fused = runs[len(runs) // 2][0]
            ks4 = ks1 * 4
            go_fused(ks4, len(ks4))
            dt_fused_512 = timed(lambda: go_fused(ks4, len(ks4)))[0]
        except Exception as e:
            dt_fused, err_fused = None, repr(e)[:300]
    # ... and NON-BLOCKING: the same batches handed to the context's query lanes (fdgpu_query_batch_submit / _wait), ONE host thread keeping
    # `depth` batches in flight — a batch's host-side steps (tables, waits, result copies) are covered by the other lanes' kernels
    if not pipe and not err_pipe:
        pipe, err_pipe = run_pipe()
    progress("big batch legs done")
    dtb2, mt_workers, mt_chunk, mt_all = None, 0, 32, {}
    if not sharded and len(queries) >= 64:
        for wk, ch in ((2, 32), (3, 32)) + (((3, 128),) if len(queries) >= 128 else ()):
            try:
                t_w, nm_w = batched_mt(wk, ch)
                assert nm_w == nm_b * MT_REPS, (nm_w, nm_b)
                mt_all["%dx%d" % (wk, ch)] = len(queries) * MT_REPS / t_w
                if dtb2 is None or t_w < dtb2:
                    dtb2, mt_workers, mt_chunk = t_w, wk, ch
            except Exception as e:  # noqa: BLE001
                print("[querybench] %d-context leg (batches of %d) failed: %r" % (wk, ch, e), file=sys.stderr)
    progress("multi-context legs done")
    loop(True, warm)()
    dt2, (_, _, nm) = timed(loop(True, range(len(queries))))
    progress("single full leg done")

    # ---- roofline of the prefilter the headline runs: one batch of 32 queries through the fused path (posting-length pass, plan, segment
    # sums + bounds + scoring = "cq_batch"; ranking keys, radix select, records of the survivors, bitonic sort = "cq_topn"), HIP events on
    # the context's stream.  B_q is SURVEY §8(d)'s figure for the same batch; PMC traffic from the committed profile of this command.
    def prefilter_roofline(ks, traffic):
        qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], None if sharded else ix, float(S_total))
        st_score = {}
        for _ in range(3):      # the third launch is the one read (the first of a new batch size sizes the pooled scratch)
            ctx.enable_timing(True)
            if sharded:
                top = sharded_prefilter(qms)
            else:
                top = count_query_maps(ctx, ix, qms, None, total_structures=S_total, top_n=top_n)
            ctx.synchronize()
            st_score = {n: ms for n, ms, _ in ctx.last_timings()}
            ctx.enable_timing(False)
        lens_fn = (lambda l: fdist.reduce_lengths(l, dev)) if sharded and comm is None else ((lambda l: comm.allreduce_lengths(l)) if sharded else None)
        recs = count_query_batch(ctx, ix, [(qm.hash, qm.qi, qm.qj) for qm in qms], pen, total_structures=S_total, top_n=0, lengths_fn=lens_fn)    # every touched structure: T of B_q
        ctx.enable_timing(True)
        if sharded:
            cl = [owned(g, match_top) for g in top]
        else:
            cl = [(g["nid"][:match_top].astype(np.int64) - first).astype(np.uint32) for g in top]
        marr = retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0]
        ctx.synchronize()
        st_match = {n: ms for n, ms, _ in ctx.last_timings()}
        ctx.enable_timing(False)
        post_bytes = int(sum(int(ix.posting_bytes(qm.hash).sum()) for qm in qms))
        touched = int(sum(len(r) for r in recs))
        n_rows = int(sum(len(np.unique(qm.qi)) + len(np.unique(qm.qi.astype(np.uint64) << np.uint64(32) | qm.qj.astype(np.uint64))) for qm in qms))
        b_q = post_bytes + 8 * touched + n_rows * ((S + 31) // 32) * 4
        t_score = st_score.get("cq_batch", 0.0) + st_score.get("cq_topn", 0.0)
        cand_res = int(sum(int(nres[c].sum()) for c in cl))
        t_match = st_match.get("match_pairs", 0.0)
        return {
            "bound": "hbm", "kernel": "prefilter of the batched full query: cq_batch (k_qt_layout, k_qt_bases, k_qt_score32<pass A: 32-bit idf sums per tile of structures in LDS>) + cq_topn "
                                      "(k_qt_thr, k_qt_rows: records of the survivors from pass A's decoded stream, k_qt_sort)",
            "queries_per_launch": len(ks), "top_n": top_n,
            "algorithmic_bytes_per_launch": b_q, "posting_bytes": post_bytes, "touched_structures": touched, "occupancy_rows": n_rows,
            "avg_ms": t_score, "stages_ms": {k: round(v, 4) for k, v in st_score.items()},
            "achieved": b_q / (t_score * 1e-3) / 1e9 if t_score > 0 else None, "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": b_q / (t_score * 1e-3) / 1e9 / hbm_peak_gbs if t_score > 0 else None,
            "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
            "note": "~%d KB of postings per query, decoded once (scores; the survivors' rows come from the decoded stream) by workgroups of one (query, 16,384-structure tile) each; "
                    "VALU-issue and latency bound, not bandwidth bound (DESIGN §4)" % (post_bytes // max(len(ks), 1) // 1024),
            "match_pairs": {"algorithmic_bytes_per_launch": 37 * cand_res + 16 * len(marr), "candidates": int(sum(len(c) for c in cl)), "avg_ms": t_match,
                            "achieved": (37 * cand_res + 16 * len(marr)) / (t_match * 1e-3) / 1e9 if t_match > 0 else None, "unit": "GB/s",
                            "note": "pair scan of the top %d candidates of %d queries; instruction / latency bound like the index build's pair kernel" % (match_top, len(ks))},
        }, st_score, st_match

    # the figure is quoted at the batch size `query.value` runs (128 queries per launch) when the query set has that many; the 32-query launch
    # of rounds 1-4 (whose PMC pass is the committed one) stays beside it as `per_32_queries`
    r32, st_score, st_match = prefilter_roofline(range(min(32, len(queries))), _pmc_query_traffic(S_total, world))
    roofline = r32
    if big and not sharded:
        try:
            roofline = prefilter_roofline(range(big), _pmc_query_traffic(S_total, world, big))[0]
            roofline["per_32_queries"] = {k: r32[k] for k in ("queries_per_launch", "algorithmic_bytes_per_launch", "avg_ms", "stages_ms", "achieved", "frac", "traffic", "traffic_detail", "match_pairs")}
        except Exception as e:
            roofline = r32
            roofline["batch_%d_error" % big] = repr(e)[:200]

    # ---- whole-structure query mode (no -q, BASELINE configs[4]): every residue of a ~300-residue database structure is a query
    # residue (~90 k hashes, every structure touched); prefilter, then retrieval of the top 20 (rank 0's structure, local shard)
    whole = None
    try:
        lens_all = nres.astype(np.int64)
        cand_s = np.nonzero((lens_all >= 295) & (lens_all <= 305))[0]
        if len(cand_s) and not sharded:
            s = int(cand_s[0])
            a, b = int(res_off_h[s]), int(res_off_h[s + 1])
            item = dict(n_xyz=d["n_xyz"][a:b].cpu().numpy(), ca_xyz=d["ca_xyz"][a:b].cpu().numpy(), cb_xyz=d["cb_xyz"][a:b].cpu().numpy(), aa=d["aa"][a:b].cpu().numpy())
            qb = ctx.upload(PackedStructures.concat([item]))
            allres = np.arange(b - a, dtype=np.uint32)

            def wq(match):
                qm = make_query_map(ctx, qb, allres, None, ix, float(S_total))
                top = count_query_maps(ctx, ix, [qm], None, total_structures=S_total, top_n=top_n)[0]     # ranked on the device
                n_m = 0
                if match:
                    # (the array form of the result: building a Python dict per match — 184 matches x 2 x 297 residue ints — is 1.5 ms of binding work, not library time)
                    n_m = len(retrieve_batch(ctx, batch, None, [(top["nid"][:20].astype(np.int64) - first).astype(np.uint32)], [qm], qb, [0], as_arrays=True)[0])
                return len(qm.hash), n_m, qm
            wq(True)
            # three runs each, the median reported (one query per run: a single sample catches allocator and page-fault noise of 10+ ms)
            pre = sorted((timed(lambda: wq(False)) for _ in range(3)), key=lambda x: x[0])
            full = sorted((timed(lambda: wq(True)) for _ in range(3)), key=lambda x: x[0])
            t_pre, (nh, _, qm) = pre[1]
            t_full, (_, n_m, _) = full[1]
            nt = len(count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S_total, as_array=True))      # T of B_q: every touched structure
            ctx.enable_timing(True)
            count_query_maps(ctx, ix, [qm], None, total_structures=S_total, top_n=top_n)       # the scoring + selection the timed legs run
            ctx.synchronize()
            stw = {n: ms for n, ms, _ in ctx.last_timings()}
            ctx.enable_timing(False)
            pbytes = int(ix.posting_bytes(qm.hash).sum())
            rows_w = len(np.unique(qm.qi)) + len(np.unique(qm.qi.astype(np.uint64) << np.uint64(32) | qm.qj.astype(np.uint64)))
            bq = pbytes + 8 * nt + rows_w * ((S + 31) // 32) * 4
            t_sc = stw.get("cq_batch", 0.0) + stw.get("cq_topn", 0.0)
            # ... and the same mode through the fused call with several queries in flight on the context's lanes (what query.value is for motif batches):
            # distinct ~300-residue structures as queries, one query per fdgpu_query_batch_submit; a query's host glue (map assembly, the plan pass of its
            # > 64-node graphs) runs under the other queries' kernels.  Results byte-identical to the blocking call's (tools/whole_pipe.py checks).
            whole_pipe = None
            try:
                from collections import deque
                import gc
                picks = cand_s[:24]
                w_items = []
                for s2 in picks:
                    a2, b2 = int(res_off_h[s2]), int(res_off_h[s2 + 1])
                    w_items.append(dict(n_xyz=d["n_xyz"][a2:b2].cpu().numpy(), ca_xyz=d["ca_xyz"][a2:b2].cpu().numpy(), cb_xyz=d["cb_xyz"][a2:b2].cpu().numpy(), aa=d["aa"][a2:b2].cpu().numpy()))
                wq_all = ctx.upload(PackedStructures.concat(w_items))
                wqs = [[(k, np.arange(int(lens_all[s2]), dtype=np.uint32))] for k, s2 in enumerate(picks)]
                n_lw = ctx.L.fdgpu_query_lanes(ctx.h, 0)

                def w_block():
                    return sum(len(query_batch(ctx, ix, batch, wq_all, q1, float(S_total), top_n, 20)[2][0]) for q1 in wqs)

                def w_pipe(depth):
                    pend, tot = deque(), 0
                    for q1 in wqs:
                        pend.append(query_batch_submit(ctx, ix, batch, wq_all, q1, float(S_total), top_n, 20))
                        if len(pend) >= depth:
                            tot += len(pend.popleft().wait()[2][0])
                    while pend:
                        tot += len(pend.popleft().wait()[2][0])
                    return tot
                if n_lw >= 2 and len(wqs) >= 4:
                    nm_b = w_block()
                    w_pipe(n_lw)
                    gc.collect(); gc.disable()
                    try:
                        tb = sorted(timed(w_block)[0] for _ in range(3))[1]
                        runs_p = [timed(lambda: w_pipe(n_lw + 2)) for _ in range(5)]
                    finally:
                        gc.enable()
                    assert all(r[1] == nm_b for r in runs_p)
                    tp = sorted(r[0] for r in runs_p)[2]
                    whole_pipe = {"value": len(wqs) / tp, "unit": "queries/s", "ms_per_query": tp / len(wqs) * 1e3, "queries": len(wqs), "lanes": int(n_lw), "in_flight": int(n_lw + 2),
                                  "blocking_fused_queries_per_s": len(wqs) / tb, "runs_queries_per_s": [round(len(wqs) / r[0], 1) for r in runs_p], "matches": int(nm_b),
                                  "mode": "fdgpu_query_batch_submit / _wait, ONE whole-structure query per call (%d distinct structures of 295-305 residues, top 1000 ranked, "
                                          "retrieval of the top 20), one host thread; median of 5 passes" % len(wqs)}
            except Exception as e:  # noqa: BLE001
                whole_pipe = {"error": repr(e)[:300]}
            whole = {"query_residues": b - a, "query_hashes": nh, "touched_structures": nt, "prefilter_ms": t_pre * 1e3, "full_ms": t_full * 1e3, "pipelined": whole_pipe,
                     "runs_ms": {"prefilter": [round(x[0] * 1e3, 2) for x in pre], "full": [round(x[0] * 1e3, 2) for x in full]},
                     "queries_per_s": 1.0 / t_full, "matches_top20": n_m, "stages_ms": stw,
                     "roofline": {"bound": "hbm", "kernel": "scoring + selection of the whole-structure query: cq_batch (k_qt_plan, k_qt_score<pass A, rows in slices>, "
                                                            "k_qd_reduce) + cq_topn (k_qt_thr, k_qd_surv, k_qt_score<pass B>, k_qd_records, k_qt_sort)",
                                  "algorithmic_bytes_per_launch": bq, "posting_bytes": pbytes,
                                  "occupancy_rows": rows_w, "avg_ms": t_sc, "achieved": bq / (t_sc * 1e-3) / 1e9 if t_sc > 0 else None,
                                  "peak": hbm_peak_gbs, "unit": "GB/s", "frac": bq / (t_sc * 1e-3) / 1e9 / hbm_peak_gbs if t_sc > 0 else None}}
    except Exception as e:  # noqa: BLE001
        whole = {"error": repr(e)}

    cpu = None
    if cpu_baseline_fn is not None:     # bench.py's cpu_baseline leg (the only place that may use oracle/)
        try:
            cpu = cpu_baseline_fn(ix, d, nres, res_off_h, [(s, idx) for s, idx, _ in queries], top_n, match_top, S)
        except Exception as e:  # noqa: BLE001 — the bench line must still be printed
            cpu = {"error": repr(e)}

    # ONE definition of the headline: the library's default 6 lanes, 10 batches in flight (FD_BENCH_PIPE adds other shapes beside it for measurements)
    pipe_key = PIPE_SHAPE if PIPE_SHAPE in pipe else (max(pipe, key=lambda k: pipe[k]) if pipe else None)
    pipe_best = pipe.get(pipe_key) if pipe_key else None
    pipe_depth = pipe_key[1] if pipe_key else None
    return {
        # headline = the reference's default query (prefilter + candidate selection + matching + RMSD), 32 queries per launch set
        # headline = ONE host thread, batches of 128 full queries (what one Rust host thread per GPU drives through the ABI; fixed definition
        # since round 4 — rounds 1-3 took the best of four legs).  Several host threads with a context each are reported beside it (_mt).
        "metric": "motif queries/sec", "value": pipe_best if pipe_best else len(queries) / (dt_fused if dt_fused else dtbm_big if dtbm_big else dtbm), "unit": "queries/s",
        "n_queries": len(queries),
        "structures": S_total, "structures_per_gpu": S,
        "mode": "full query (make_query_map, count_query, all-gather + global top-%d, retrieval of the global top %d candidates on their owning "
                "rank, Kabsch, metrics); value = batches of %d from ONE host thread, %s; batches of 32 / 512 and several host "
                "threads with one context each are reported beside it (batched_with_matching, _512, _mt)"
                % (top_n, match_top, big if dtbm_big else 32,
                   ("fdgpu_query_batch_submit / _wait with %d batches in flight on the context's %d query lanes (pipelined_128; the blocking call: fused_128; the three "
                    "separate calls: batched_with_matching_128)" % (pipe_depth, pipe_key[0])) if pipe_best else
                   "one fdgpu_query_batch call per batch (fused_128; the three separate calls: batched_with_matching_128)"
                   if dt_fused else "three library calls per batch (batched_with_matching_128)"),
        "ms_per_query": 1e3 / pipe_best if pipe_best else (dt_fused if dt_fused else dtbm_big if dtbm_big else dtbm) / len(queries) * 1e3,
        "pipelined_128": ({"error": err_pipe} if err_pipe else None) if not pipe_best else {
            "value": pipe_best, "ms_per_query": 1e3 / pipe_best, "chunk": big, "host_threads": 1, "lanes": pipe_key[0], "in_flight": pipe_depth, "queries": len(queries) * PIPE_REPS,
            "queries_per_s_by_lanes_x_in_flight": {"%dx%d" % k: v for k, v in pipe.items()},
            "runs_queries_per_s": {"%dx%d" % k: [round(x, 1) for x in v] for k, v in pipe_runs.items()},
            "mode": "fdgpu_query_batch_submit / fdgpu_query_batch_wait: ONE host thread keeps %d batches of 128 in flight on %d library-owned lanes (sibling contexts: "
                    "own stream + scratch, each runs fdgpu_query_batch itself); %d passes over the %d queries per timed run, median of 7 behind a warm-up of two such runs" % (pipe_depth, pipe_key[0], PIPE_REPS, len(queries))},
        "fused_128": ({"error": err_fused} if err_fused else None) if not dt_fused else {
            "value": len(queries) / dt_fused, "ms_per_query": dt_fused / len(queries) * 1e3, "chunk": big, "host_threads": 1,
            "mode": "fdgpu_query_batch: query maps, scoring + ranked top %d, retrieval of the top %d in ONE call per batch of 128 (median of 7 passes)" % (top_n, match_top)},
        "fused_512": None if not dt_fused_512 else {"value": 4 * len(queries) / dt_fused_512, "ms_per_query": dt_fused_512 / (4 * len(queries)) * 1e3,
                                                    "chunk": 4 * len(queries), "host_threads": 1},
        "batched_with_matching": {"value": len(queries) / dtbm, "ms_per_query": dtbm / len(queries) * 1e3, "matches": int(nm_b), "match_top": match_top, "chunk": 32,
                                  "host_threads": 1},
        "batched_with_matching_512": ({"error": err_512} if err_512 else None) if not dtbm_512 else {
            "value": 4 * len(queries) / dtbm_512, "ms_per_query": dtbm_512 / (4 * len(queries)) * 1e3, "chunk": 4 * len(queries), "host_threads": 1,
            "mode": "the same full query, the 128 queries four times over in ONE batch of 512, one host thread"},
        "batched_with_matching_128": None if not dtbm_big else {"value": len(queries) / dtbm_big, "ms_per_query": dtbm_big / len(queries) * 1e3, "chunk": big,
                                                                "host_threads": 1, "mode": "the same full query, 128 queries per batch, one host thread"},
        "batched_with_matching_mt": None if not dtb2 else {
            "value": len(queries) * MT_REPS / dtb2, "ms_per_query": dtb2 / (len(queries) * MT_REPS) * 1e3, "host_threads": mt_workers, "chunk": mt_chunk,
            "queries": len(queries) * MT_REPS, "queries_per_s_by_threads_x_batch": {str(k): v for k, v in mt_all.items()},
            "mode": "the same full batched query (one fdgpu_query_batch call per batch) driven by several host threads with one context (stream + workspaces) each, sharing the resident index"},
        "batched": {"value": len(queries) / dtb, "ms_per_query": dtb / len(queries) * 1e3, "chunk": 32, "avg_hits": hits_b / len(queries),
                    "mode": "prefilter only: make_query_map_batch + count_query_batch_top + all-gather"},
        "single": {"value": len(queries) / dt1, "ms_per_query": dt1 / len(queries) * 1e3, "mode": "prefilter only, one query per call"},
        "with_matching": {"value": len(queries) / dt2, "ms_per_query": dt2 / len(queries) * 1e3, "matches": int(nm), "match_top": match_top,
                          "mode": "full query, one query per call"},
        "avg_query_hashes": hashes / len(queries), "avg_hits": hits / len(queries),
        "roofline": roofline, "whole_structure": whole, "cpu_baseline": cpu,
        "stages_ms_per_32_queries": {**st_score, **st_match},
        "exchange": None if not sharded else {
            "transport": "libfdgpu RCCL communicator (fdgpu_sharded_count_query_maps + fdgpu_sharded_retrieve)" if comm is not None else
                         "torch.distributed %s around fdgpu_query_maps_lengths / fdgpu_count_query_maps_top_global" % dist.get_backend(),
            "collectives": None if comm is None else dict(zip(("allreduce", "allgather"), comm.stats()))},
    }


def run_replicas(ctx, batch, ix, d, S_total, world, rank, dist, dev, n_queries=64, top_n=1000, match_top=32, seed=4242, chunk=128, reps=0):
    """SURVEY §8e, last row: the index (26 GB at Swiss-Prot scale, of 288) and the coordinates REPLICATED on every GPU, the queries sharded —
    batch b of `chunk` queries runs on rank b % world through the single-index fused path (no data-path collective), the outputs are
    gathered at the end (here: their counts).  queries/s = all queries / max-over-ranks wall time."""
    queries = _pick_queries(d, S_total, n_queries, seed) if rank == 0 else None
    if dist is not None:
        box = [queries]
        dist.broadcast_object_list(box, src=0, device=dev if dist.get_backend() == "nccl" else None)
        queries = box[0]
    nres = np.diff(d["res_off"].cpu().numpy()).astype(np.uint64)
    ix.set_penalty(length_penalty(nres, 0.5))
    qall = ctx.upload(PackedStructures.concat([it for _, _, it in queries]))
    first = ix.first_id
    chunk = min(chunk, len(queries))
    # the shape of the single-GPU headline leg on every rank: batches of 128, ten in flight on the rank's six lanes, 48 batches per rank in the timed run
    # (twelve batches of 32 dealt to eight ranks was a 2 ms timed region)
    reps = reps or max(1, (48 * world * chunk) // max(len(queries), 1))
    starts = list(range(0, len(queries), chunk)) * reps
    LANES = 10

    def go():
        tot = 0
        if first == 0:      # the replica holds the whole database: the rank's batches through its query lanes (fdgpu_query_batch_submit / _wait), LANES in flight
            from collections import deque
            pend = deque()
            for c0 in starts[rank::world]:
                ks = range(c0, min(c0 + chunk, len(queries)))
                pend.append(query_batch_submit(ctx, ix, batch, qall, [(k, queries[k][1]) for k in ks], float(S_total), top_n, match_top))
                if len(pend) >= LANES:
                    tot += len(pend.popleft().wait()[2][0])
            while pend:
                tot += len(pend.popleft().wait()[2][0])
            return tot
        for c0 in starts[rank::world]:
            ks = range(c0, min(c0 + chunk, len(queries)))
            qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], ix, float(S_total))
            recs = count_query_maps(ctx, ix, qms, None, total_structures=S_total, top_n=top_n)
            cl = [(g["nid"][:match_top].astype(np.int64) - first).astype(np.uint32) for g in recs]
            tot += len(retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)[0])
        ctx.synchronize()
        return tot
    for _ in range(3 if first == 0 else 1):      # the lanes reach their pace after ~100 batches (tools/pipe_variance.py)
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
        dv = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dv)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        m = torch.tensor([tot], dtype=torch.int64, device=dv)
        dist.all_reduce(m, op=dist.ReduceOp.SUM)     # the gather of outputs, reduced to their count here
        tot = int(m.item())
    nq = len(queries) * reps
    return {"value": nq / dt, "unit": "queries/s", "queries": nq, "ms_per_query": dt / nq * 1e3, "matches": tot, "replicas": world,
            "mode": "index + coordinates replicated on every GPU, batches of %d queries dealt round-robin to the ranks, full query (prefilter top %d, "
                    "retrieval of the top %d), one host thread per rank keeping %d batches in flight on its context's query lanes (fdgpu_query_batch_submit / _wait), "
                    "no data-path collective" % (chunk, top_n, match_top, LANES)}
