"""Where a batch of 32 full motif queries spends its wall time (host Python, host C++ glue, GPU): cProfile over the bench's batched
full-query loop at --structures S (default 67,750), with FDGPU_TRACE=1 printing the stage split inside fdgpu_retrieve_batch.

    python tools/profile_query_host.py [--structures N] > gpurun_out/query_host_profile.txt
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structures", type=int, default=67750)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--chunk", default="32", help="queries per batch (a comma list measures each in turn)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--fused", action="store_true", help="one fdgpu_query_batch call per batch instead of the three calls")
    a = ap.parse_args()
    chunks = [int(x) for x in str(a.chunk).split(",")]
    a.chunk = chunks[0]
    import numpy as np
    import torch
    import folddisco_amd as fd
    from folddisco_amd import synth
    from folddisco_amd import dist as fdist
    from folddisco_amd.api import PackedStructures, count_query_batch, count_query_maps, length_penalty
    from folddisco_amd.query import make_query_maps, query_batch, retrieve_batch
    from folddisco_amd.querybench import _pick_queries
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    S = a.structures
    ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    # blocks of <= 67,750 structures (one build call each), merged on the device into one resident index — like bench.py
    BLK = 67750
    ds, parts, wraps = [], [], []
    for b0 in range(0, S, BLK):
        n = min(BLK, S - b0)
        db = synth.generate(n, seed=7 + 1000 * (b0 // BLK), device=dev)
        rb = db["res_off"].contiguous()
        wb = ctx.wrap_device(n, int(rb[-1].item()), rb.data_ptr(), db["n_xyz"].data_ptr(), db["ca_xyz"].data_ptr(), db["cb_xyz"].data_ptr(),
                             db["aa"].data_ptr(), None, keepalive=(rb, db))
        parts.append(fd.FolddiscoIndex.build(ctx, wb, first_id=b0))
        ds.append(db); wraps.append(wb)
    ix = parts[0] if len(parts) == 1 else fd.FolddiscoIndexSet(parts).merge()
    del parts
    # one batch over all the coordinates (candidates are addressed by global structure index)
    offs = [0]
    for db in ds:
        offs.append(offs[-1] + int(db["res_off"][-1].item()))
    d = dict(res_off=torch.cat([db["res_off"][(1 if k else 0):] + offs[k] for k, db in enumerate(ds)]),
             n_xyz=torch.cat([db["n_xyz"] for db in ds]), ca_xyz=torch.cat([db["ca_xyz"] for db in ds]), cb_xyz=torch.cat([db["cb_xyz"] for db in ds]),
             aa=torch.cat([db["aa"] for db in ds]))
    del ds, wraps
    ro = d["res_off"].contiguous()
    batch = ctx.wrap_device(S, int(ro[-1].item()), ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(),
                            d["aa"].data_ptr(), None, keepalive=(ro, d))
    queries = _pick_queries(d, S, a.queries, 4242)
    nres = np.diff(ro.cpu().numpy()).astype(np.uint64)
    pen = length_penalty(nres, 0.5)
    ix.set_penalty(pen)
    qall = ctx.upload(PackedStructures.concat([it for _, _, it in queries]))
    T = {}

    def go(match, trace=False):
        for c0 in range(0, len(queries), a.chunk):
            ks = range(c0, min(c0 + a.chunk, len(queries)))
            t0 = time.perf_counter()
            if a.fused:
                query_batch(ctx, ix, batch, qall, [(k, queries[k][1]) for k in ks], float(S), 1000, 32)
                T["query_batch"] = T.get("query_batch", 0.0) + time.perf_counter() - t0
                continue
            qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], ix, float(S))
            t1 = time.perf_counter()
            recs = count_query_maps(ctx, ix, qms, None, total_structures=S, top_n=1000)
            t2 = time.perf_counter()
            globs = fdist.allgather_hits_many(recs, None, top_n=1000, ranked=True)
            t3 = time.perf_counter()
            if match:
                cl = [g["nid"][:32].astype(np.uint32) for g in globs]
                retrieve_batch(ctx, batch, None, cl, qms, qall, list(ks), as_arrays=True)
            t4 = time.perf_counter()
            for k, v in (("make_query_maps", t1 - t0), ("count_query_batch", t2 - t1), ("rank", t3 - t2), ("retrieve_batch", t4 - t3)):
                T[k] = T.get(k, 0.0) + v

    n_rep = a.reps
    for ch in chunks:
        a.chunk = ch
        go(True)
        T.clear()
        t0 = time.perf_counter()
        for _ in range(n_rep):
            go(True)
        dt = time.perf_counter() - t0
        print(f"full batched: {a.queries * n_rep / dt:.0f} q/s; per {ch}-query batch: " +
              ", ".join(f"{k} {v / n_rep / (len(queries) / a.chunk) * 1e3:.3f} ms" for k, v in T.items()), flush=True)
    a.chunk = chunks[0]
    if a.no_profile:
        return
    os.environ["FDGPU_TRACE"] = "1"
    go(True)
    del os.environ["FDGPU_TRACE"]
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        go(True)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())


if __name__ == "__main__":
    main()
