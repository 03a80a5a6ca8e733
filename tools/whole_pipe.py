"""Whole-structure queries (every residue of a ~300-residue database structure is a query residue: BASELINE configs[4]) through the fused call:
one blocking fdgpu_query_batch per query against fdgpu_query_batch_submit / _wait with several queries in flight on the context's lanes.
Prints queries/s per form and checks that the pipelined results are byte-identical to the blocking call's.

    python tools/whole_pipe.py [--structures 542000] [--queries 24] [--lanes 6] [--depth 8]
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
    ap.add_argument("--queries", type=int, default=24)
    ap.add_argument("--lanes", type=int, default=6)
    ap.add_argument("--depth", type=int, default=8)
    a = ap.parse_args()
    import gc
    import numpy as np
    import torch
    from _resident import build_resident
    from folddisco_amd.api import PackedStructures, length_penalty
    from folddisco_amd.query import query_batch, query_batch_submit
    S = a.structures
    ctx, batch, ix, d, ro = build_resident(S)
    roh = ro.cpu().numpy()
    nres = np.diff(roh).astype(np.uint64)
    ix.set_penalty(length_penalty(nres, 0.5))
    picks = np.nonzero((nres >= 280) & (nres <= 320))[0][: a.queries]
    items = []
    for s in picks:
        x, y = int(roh[s]), int(roh[s + 1])
        items.append(dict(n_xyz=d["n_xyz"][x:y].cpu().numpy(), ca_xyz=d["ca_xyz"][x:y].cpu().numpy(), cb_xyz=d["cb_xyz"][x:y].cpu().numpy(), aa=d["aa"][x:y].cpu().numpy()))
    qall = ctx.upload(PackedStructures.concat(items))
    qs = [[(k, np.arange(int(nres[s]), dtype=np.uint32))] for k, s in enumerate(picks)]

    def blocking():
        return [query_batch(ctx, ix, batch, qall, q, float(S), 1000, 20) for q in qs]

    def piped(depth):
        pend, out = deque(), []
        for q in qs:
            pend.append(query_batch_submit(ctx, ix, batch, qall, q, float(S), 1000, 20))
            if len(pend) >= depth:
                out.append(pend.popleft().wait())
        while pend:
            out.append(pend.popleft().wait())
        return out

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r
    ref = blocking()
    print("%d whole-structure queries of %d..%d residues, %d..%d hashes, matches of the top 20: %s" % (len(qs), int(nres[picks].min()), int(nres[picks].max()),
          min(len(r[0][0].hash) for r in ref), max(len(r[0][0].hash) for r in ref), [len(r[2][0]) for r in ref][:8]), flush=True)
    gc.collect(); gc.disable()
    tb = sorted(timed(blocking)[0] for _ in range(3))[1]
    print("blocking fdgpu_query_batch, one query per call : %7.1f queries/s (%.2f ms per query)" % (len(qs) / tb, tb / len(qs) * 1e3), flush=True)
    assert ctx.L.fdgpu_query_lanes(ctx.h, a.lanes) >= a.lanes
    piped(a.lanes)
    got = piped(a.depth)
    for (m0, (r0, o0), t0_), (m1, (r1, o1), t1) in zip(ref, got):
        assert r0.tobytes() == r1.tobytes() and o0.tobytes() == o1.tobytes() and all(x.tobytes() == y.tobytes() for x, y in zip(t0_, t1))
        assert all(getattr(x, f).tobytes() == getattr(y, f).tobytes() for x, y in zip(m0, m1) for f in ("hash", "qi", "qj", "idf"))
    for depth in sorted({a.lanes, a.depth, 2 * a.lanes}):
        tp = sorted(timed(lambda: piped(depth))[0] for _ in range(3))[1]
        print("submit / wait, %2d in flight on %d lanes            : %7.1f queries/s (%.2f ms per query); results byte-identical to the blocking call" %
              (depth, a.lanes, len(qs) / tp, tp / len(qs) * 1e3), flush=True)


if __name__ == "__main__":
    main()
