#!/usr/bin/env python
"""bench.py — structures/sec indexed (+ motif queries/sec) on MI355X, BASELINE.json's metric.

One step = one pass of the index-build hot path (pair enumeration + PDBTrRosetta hash -> stable radix
sort -> delta/varint posting encode) over one shard of synthetic AFDB-shaped structures that is already
resident in HBM.  Weak scaling: every rank indexes its own shard of --structures structures (default
67,750 = Swiss-Prot 542k / 8, so that --gpus 8 is Swiss-Prot scale) with ids offset by rank; the build
needs no collective (index build shards by structure).  After the timed build the ranks
score a batch of motif queries against their shard and all-gather the candidate hits (RCCL), reported
as queries/s in the "query" object.

Contract: python bench.py --gpus N --steps K --warmup W  -> one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--structures", type=int, default=67750, help="structures per GPU (shard)")
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="structures timed on the host for cpu_baseline")
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-query", action="store_true")
    ap.add_argument("--chunk", type=int, default=100000, help="structures per fdgpu_index_build call; a larger shard is held as a "
                    "FolddiscoIndexSet (one resident sub-index per chunk)")
    ap.add_argument("--pipeline", type=int, default=0, help="extra leg: builds issued from this many host threads / HIP streams (0 = skip)")
    return ap.parse_args()


def pmc_traffic(stage, S):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*pmc_traffic_S<structures>.json, tools/profile_bench.sh): 2 x FETCH_SIZE (gfx950 reports half of a
    coalesced stream, MI355X_MICROARCH.md; checked on k_enc_sizes / k_rs_hist whose read bytes are known) +
    WRITE_SIZE (matches the known write bytes of k_frames and k_pair_emit2).  None when no profile of this
    workload size is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic_S{S}.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.get("kernels", {}).items():
            if k.startswith("k_" + stage):
                return {"bytes_per_launch": 2.0 * v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"], "kernel": k,
                        "rocprof_avg_us": v.get("avg_us"), "source": os.path.basename(f)}
    return None


def cpu_baseline(ps_sample, n_threads):
    """CPU restatement of the reference path (oracle/, OpenMP over structures for the hash stage like the
    reference's rayon par_iter, serial dense-table count+fill), timed on the host cores of this box."""
    import oracle
    from tests.helpers import packed_to_oracle_structs
    os.environ["OMP_NUM_THREADS"] = str(n_threads)
    structs = packed_to_oracle_structs(ps_sample)
    t0 = time.perf_counter()
    # the reference hashes every structure twice (count pass + fill pass, controller/mod.rs:274-441)
    h, off = oracle.hash_batch(structs)
    h2, off2 = oracle.hash_batch(structs)
    ix = oracle.build_index_from_lists_mt(h, off, n_threads)
    dt = time.perf_counter() - t0
    return len(structs) / dt, dt, ix


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("FD_BENCH_BACKEND", "nccl")   # "gloo": several ranks on ONE GPU (plumbing smoke test)
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import folddisco_amd as fd
    from folddisco_amd import synth

    S = args.structures
    # ---- synthetic shard generated directly in HBM
    d = synth.generate(S, seed=args.seed + 1000 * rank, device=dev)
    res_off = d["res_off"].contiguous()
    R = int(res_off[-1].item())
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev)
    ctx = fd.Context(local_rank, stream=stream.cuda_stream)
    keep = (res_off, d["n_xyz"], d["ca_xyz"], d["cb_xyz"], d["aa"])
    batch = ctx.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(),
                            d["aa"].data_ptr(), None, keepalive=keep)
    first_id = rank * S
    # shards beyond --chunk structures (more than 2^32 residue pairs): one build call and one resident sub-index per chunk
    chunked = S > args.chunk
    chunk_batches = []
    if chunked:
        off_cpu = res_off.cpu()
        for a in range(0, S, args.chunk):
            b = min(a + args.chunk, S)
            r0, r1 = int(off_cpu[a]), int(off_cpu[b])
            ro = (res_off[a:b + 1] - res_off[a]).contiguous()
            parts = (ro, d["n_xyz"][r0:r1], d["ca_xyz"][r0:r1], d["cb_xyz"][r0:r1], d["aa"][r0:r1])
            chunk_batches.append(ctx.wrap_device(b - a, r1 - r0, ro.data_ptr(), parts[1].data_ptr(), parts[2].data_ptr(), parts[3].data_ptr(),
                                                 parts[4].data_ptr(), None, keepalive=parts))

    def build_shard():
        if not chunked:
            return fd.FolddiscoIndex.build(ctx, batch, first_id=first_id)
        return fd.FolddiscoIndexSet.build(ctx, chunk_batches, first_id=first_id)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    ix = None
    for _ in range(args.warmup):
        ix = None
        ix = build_shard()
    ctx.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ix = None  # release the previous index before building the next one
        ix = build_shard()
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * S * args.steps / dt
    if chunked:
        n_post, n_hash, vlen = ix.num_postings, sum(p.num_hashes for p in ix.parts), sum(p.value_len for p in ix.parts)
    else:
        n_post, n_hash, vlen = ix.num_postings, ix.num_hashes, ix.value_len

    # ---- extra leg: the same K builds issued from P host threads, each with its own context / stream / workspace, so that the
    # VALU-bound pair kernel of one build overlaps the HBM-bound sort/encode of another (how a multi-shard job would run)
    pipelined = None
    if args.pipeline > 1:
        import threading
        P = args.pipeline
        ctxs = [ctx] + [fd.Context(local_rank) for _ in range(P - 1)]
        bts = [batch] + [c.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(),
                                       d["aa"].data_ptr(), None, keepalive=keep) for c in ctxs[1:]]
        per = [(args.steps + P - 1 - t) // P for t in range(P)]

        def worker(t, n):
            torch.cuda.set_device(local_rank)   # the current HIP device is per host thread
            x = None
            for _ in range(n):
                x = None
                x = fd.FolddiscoIndex.build(ctxs[t], bts[t], first_id=first_id)
            ctxs[t].synchronize()

        def run_all(counts):
            th = [threading.Thread(target=worker, args=(t, counts[t])) for t in range(P)]
            for x in th: x.start()
            for x in th: x.join()
        ix = None
        run_all([1] * P)   # warm every context (workspace allocation)
        barrier()
        t0p = time.perf_counter()
        run_all(per)
        barrier()
        dtp = time.perf_counter() - t0p
        if dist is not None:
            t = torch.tensor([dtp], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtp = float(t.item())
        pipelined = {"value": world * S * args.steps / dtp, "unit": "structures/s", "streams": P, "ms_per_step": dtp / args.steps * 1e3,
                     "note": "same K builds, issued concurrently from %d host threads on %d HIP streams" % (P, P)}
        del bts
        for c in ctxs[1:]:
            c.close()

    # ---- per-kernel timings of one more (untimed) step with HIP events on the build stream -> roofline
    ctx.enable_timing(True)
    ix = None
    if chunked:
        stages, parts, fid = [], [], first_id
        for cb in chunk_batches:
            parts.append(fd.FolddiscoIndex.build(ctx, cb, first_id=fid))
            ctx.synchronize()
            stages += ctx.last_timings()
            fid += cb.n_struct
        ix = fd.FolddiscoIndexSet(parts)
    else:
        ix = fd.FolddiscoIndex.build(ctx, batch, first_id=first_id)
        ctx.synchronize()
        stages = ctx.last_timings()
    ctx.enable_timing(False)
    agg = {}
    for name, ms, by in stages:
        a = agg.setdefault(name, [0.0, 0, 0])
        a[0] += ms; a[1] += by; a[2] += 1
    dom = max(agg.items(), key=lambda kv: kv[1][0])
    dom_name, (dom_ms, dom_bytes, dom_n) = dom
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom_name, "launches_per_step": dom_n, "avg_ms": dom_ms / max(dom_n, 1),
                "algorithmic_bytes_per_launch": dom_bytes / max(dom_n, 1),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes per launch from the committed PMC passes of this same command (number, or null when no profile of this
                # workload size is committed); provenance in traffic_detail
                "traffic": (lambda t: t["bytes_per_launch"] if t else None)(pmc_traffic(dom_name, S)),
                "traffic_detail": pmc_traffic(dom_name, S),
                # SURVEY §8(d) end-to-end figure: B_idx = 37 R + 4 U + 4 U + v U + 12 H/S bytes per structure
                "end_to_end": (lambda b: {"algorithmic_bytes_per_structure": b, "achieved": value / world * b / 1e9, "unit": "GB/s",
                                          "frac": value / world * b / 1e9 / HBM_PEAK_GBS})(
                    37.0 * R / S + 8.0 * n_post / S + vlen / S + 12.0 * n_hash / S),
                "stages_ms": {k: round(v[0], 3) for k, v in agg.items()},
                "stages_gbs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0 for k, v in agg.items()}}

    # ---- motif queries against the resident shard
    query = None
    if chunked and not args.no_query:
        query = {"note": "query leg runs on single-part shards only (use folddisco_amd.count_query_set for a FolddiscoIndexSet)"}
    elif not args.no_query:
        try:
            from folddisco_amd import querybench
            if os.environ.get("FD_PROFILE_QUERY"):
                import cProfile, pstats
                pr = cProfile.Profile(); pr.enable()
                query = querybench.run(ctx, batch, ix, d, S, world, rank, dist, dev, n_queries=args.queries)
                pr.disable(); pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
            else:
                query = querybench.run(ctx, batch, ix, d, S, world, rank, dist, dev, n_queries=args.queries)
        except Exception as e:  # the index-build line must still be printed
            query = {"error": repr(e)}

    out = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample, S)
            off = res_off[: ns + 1].cpu().numpy().astype(np.uint64)
            r_s = int(off[-1])
            ps = fd.PackedStructures(off, d["n_xyz"][:r_s].cpu().numpy(), d["ca_xyz"][:r_s].cpu().numpy(), d["cb_xyz"][:r_s].cpu().numpy(),
                                     d["aa"][:r_s].cpu().numpy())
            cores = os.cpu_count() or 1
            v, secs, _ = cpu_baseline(ps, cores)
            cpu = {"value": v, "unit": "structures/s", "cores": cores, "kind": "port",
                   "sample": f"first {ns} structures of the shard ({r_s} residues): 2x hash+sort+dedup (OpenMP over structures) + "
                             f"count/fill table build with the reference's ownership partition over {cores} threads, {secs:.1f} s"}
        out = {
            "metric": "structures/sec indexed", "value": value, "unit": "structures/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+u32", "data": "synthetic",
            "config": {"workload": f"Swiss-Prot/8 shard per GPU: {S} synthetic AFDB-shaped structures ({R} residues, "
                                   f"{n_post} postings, {n_hash} distinct hashes, {vlen} value bytes) index build, PDBTrRosetta default",
                       "structures_per_gpu": S, "residues_per_gpu": R, "postings_per_gpu": n_post, "parallelism": f"shard-by-structure x{world}"},
            "roofline": roofline, "pipelined": pipelined, "cpu_baseline": cpu, "query": query,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
