"""One host thread, batches of 128 full motif queries at --structures S: the blocking fused call against fdgpu_query_batch_submit / _wait with
2, 3, 4 ... batches in flight on the context's query lanes (fd_lanes.hip).  Prints queries/s per form and checks that the pipelined results
are byte-identical to the blocking call's.

    python tools/query_pipe.py [--structures 542000] [--lanes 2,3,4] [--reps 24]
"""
import argparse
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structures", type=int, default=542000)
    ap.add_argument("--queries", type=int, default=128)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--lanes", default="2,3,4")
    ap.add_argument("--profile", default="", help="'blocking' or 'LANESxDEPTH' (e.g. 4x6): warm up, sleep 1 s, run ONLY that loop once (--reps passes), sleep 1 s, exit — "
                    "for rocprofv3 --kernel-trace (tools/profile_round5_pipe.sh cuts the trace at the gaps)")
    ap.add_argument("--distinct", type=int, default=1, help="number of distinct query sets cycled through (1 = the bench's: the same 128 queries every pass)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from _resident import build_resident
    from folddisco_amd.api import PackedStructures, length_penalty
    from folddisco_amd.query import query_batch, query_batch_submit
    from folddisco_amd.querybench import _pick_queries
    S = a.structures
    t0 = time.perf_counter()
    ctx, batch, ix, d, ro = build_resident(S)
    print("resident database of %d structures in %.1f s" % (S, time.perf_counter() - t0), flush=True)
    sets = []
    for k in range(a.distinct):
        qs = _pick_queries(d, S, a.queries, 4242 + 17 * k)
        qall = ctx.upload(PackedStructures.concat([it for _, _, it in qs]))
        chunks = [[(t, qs[t][1]) for t in range(c0, min(c0 + a.chunk, len(qs)))] for c0 in range(0, len(qs), a.chunk)]
        sets.append((qall, chunks))
    nres = np.diff(ro.cpu().numpy()).astype(np.uint64)
    ix.set_penalty(length_penalty(nres, 0.5))
    n_q = a.queries

    def blocking(reps):
        tot = 0
        for r in range(reps):
            qall, chunks = sets[r % len(sets)]
            for qs in chunks:
                tot += len(query_batch(ctx, ix, batch, qall, qs, float(S), 1000, 32)[2][0])
        return tot

    def piped(reps, depth):
        pend, tot = deque(), 0
        for r in range(reps):
            qall, chunks = sets[r % len(sets)]
            for qs in chunks:
                pend.append(query_batch_submit(ctx, ix, batch, qall, qs, float(S), 1000, 32))
                if len(pend) >= depth:
                    tot += len(pend.popleft().wait()[2][0])
        while pend:
            tot += len(pend.popleft().wait()[2][0])
        return tot

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r
    if a.profile:
        if a.profile == "blocking":
            blocking(4)
            fn = lambda: blocking(a.reps)
        else:
            lanes, depth = (int(x) for x in a.profile.split("x"))
            assert ctx.L.fdgpu_query_lanes(ctx.h, lanes) >= lanes
            piped(3 * lanes, lanes)
            fn = lambda: piped(a.reps, depth)
        torch.cuda.synchronize()
        time.sleep(1.0)
        dt, tot = timed(fn)
        time.sleep(1.0)
        print("PROFILE %s: %d batches of %d in %.3f ms = %.0f queries/s (%.3f ms per batch), %d matches" %
              (a.profile, a.reps * len(sets[0][1]), a.chunk, dt * 1e3, n_q * a.reps / dt, dt / a.reps / len(sets[0][1]) * 1e3, tot), flush=True)
        return
    want = blocking(max(2, len(sets)))
    runs = sorted(timed(lambda: blocking(a.reps))[0] for _ in range(3))
    print("blocking fdgpu_query_batch, one host thread       : %8.0f queries/s (%.3f ms per batch of %d)" % (n_q * a.reps / runs[1], runs[1] / a.reps / len(sets[0][1]) * 1e3, a.chunk), flush=True)
    # byte-identical results
    ref = [query_batch(ctx, ix, batch, qall, qs, float(S), 1000, 32) for qall, chunks in sets for qs in chunks]
    for depth in [int(x) for x in a.lanes.split(",")]:
        n_l = ctx.L.fdgpu_query_lanes(ctx.h, depth)
        assert n_l >= depth, n_l
        piped(2 * depth, depth)
        jobs = [query_batch_submit(ctx, ix, batch, qall, qs, float(S), 1000, 32) for qall, chunks in sets for qs in chunks]
        fields = ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi", "primary_hash")
        for (m0, (r0, o0), t0_), j in zip(ref, jobs):
            m1, (r1, o1), t1 = j.wait()
            assert all(getattr(x, f).tobytes() == getattr(y, f).tobytes() for x, y in zip(m0, m1) for f in fields)
            assert r0.tobytes() == r1.tobytes() and o0.tobytes() == o1.tobytes() and all(x.tobytes() == y.tobytes() for x, y in zip(t0_, t1))
        runs = sorted(timed(lambda: piped(a.reps, depth))[0] for _ in range(3))
        print("submit / wait, %d batches in flight (%d lanes)        : %8.0f queries/s (%.3f ms per batch); results byte-identical to the blocking call" %
              (depth, n_l, n_q * a.reps / runs[1], runs[1] / a.reps / len(sets[0][1]) * 1e3), flush=True)
    # more batches in flight than lanes: the queue absorbs them
    n_l = ctx.L.fdgpu_query_lanes(ctx.h, 0)
    runs = sorted(timed(lambda: piped(a.reps, n_l + 2))[0] for _ in range(3))
    print("submit / wait, %d in flight on %d lanes                : %8.0f queries/s" % (n_l + 2, n_l, n_q * a.reps / runs[1]), flush=True)


if __name__ == "__main__":
    main()
