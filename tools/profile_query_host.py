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
    a = ap.parse_args()
    import numpy as np
    import torch
    import folddisco_amd as fd
    from folddisco_amd import synth
    from folddisco_amd import dist as fdist
    from folddisco_amd.api import PackedStructures, count_query_batch, count_query_maps, length_penalty
    from folddisco_amd.query import make_query_maps, retrieve_batch
    from folddisco_amd.querybench import _pick_queries
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    S = a.structures
    d = synth.generate(S, seed=7, device=dev)
    ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ro = d["res_off"].contiguous()
    batch = ctx.wrap_device(S, int(ro[-1].item()), ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(),
                            d["aa"].data_ptr(), None, keepalive=(ro, d))
    ix = fd.FolddiscoIndex.build(ctx, batch)
    queries = _pick_queries(d, S, a.queries, 4242)
    nres = np.diff(ro.cpu().numpy()).astype(np.uint64)
    pen = length_penalty(nres, 0.5)
    ix.set_penalty(pen)
    qall = ctx.upload(PackedStructures.concat([it for _, _, it in queries]))
    T = {}

    def go(match, trace=False):
        for c0 in range(0, len(queries), 32):
            ks = range(c0, min(c0 + 32, len(queries)))
            t0 = time.perf_counter()
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

    go(True)
    T.clear()
    n_rep = 10
    t0 = time.perf_counter()
    for _ in range(n_rep):
        go(True)
    dt = time.perf_counter() - t0
    print(f"full batched: {a.queries * n_rep / dt:.0f} q/s; per 32-query batch: " + ", ".join(f"{k} {v / n_rep / (len(queries) / 32) * 1e3:.3f} ms" for k, v in T.items()))
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
