"""Stage split of a whole-structure query (every residue of a ~300-residue database structure is a query residue): query map, scoring,
retrieval of the top 20 with FDGPU_TRACE=1 (stage timings of fdgpu_retrieve_batch on stderr).

    python tools/profile_whole_query.py [--structures N]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structures", type=int, default=67750)
    a = ap.parse_args()
    import numpy as np
    import folddisco_amd as fd  # noqa: F401
    from _resident import build_resident
    from folddisco_amd.api import PackedStructures, count_query_maps, length_penalty
    from folddisco_amd.query import make_query_map, retrieve_batch
    S = a.structures
    ctx, batch, ix, d, ro = build_resident(S)
    roh = ro.cpu().numpy()
    nres = np.diff(roh).astype(np.uint64)
    ix.set_penalty(length_penalty(nres, 0.5))
    s = int(np.nonzero((nres >= 295) & (nres <= 305))[0][0])
    x, y = int(roh[s]), int(roh[s + 1])
    item = dict(n_xyz=d["n_xyz"][x:y].cpu().numpy(), ca_xyz=d["ca_xyz"][x:y].cpu().numpy(), cb_xyz=d["cb_xyz"][x:y].cpu().numpy(), aa=d["aa"][x:y].cpu().numpy())
    qb = ctx.upload(PackedStructures.concat([item]))
    allres = np.arange(y - x, dtype=np.uint32)

    def go(trace):
        if trace:
            os.environ["FDGPU_TRACE"] = "1"
        t0 = time.perf_counter()
        qm = make_query_map(ctx, qb, allres, None, ix, float(S))
        os.environ.pop("FDGPU_TRACE", None)
        t1 = time.perf_counter()
        top = count_query_maps(ctx, ix, [qm], None, total_structures=S, top_n=1000)[0]
        t2 = time.perf_counter()
        if trace:
            os.environ["FDGPU_TRACE"] = "1"
        m = retrieve_batch(ctx, batch, None, [top["nid"][:20].astype(np.uint32)], [qm], qb, [0], as_arrays=True)[0]
        os.environ.pop("FDGPU_TRACE", None)
        t3 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(m)

    go(False)
    ctx.enable_timing(True)
    go(False)
    ctx.synchronize()
    print("kernel stages of the last library call (ms):", {n: round(ms, 3) for n, ms, _ in ctx.last_timings()})
    ctx.enable_timing(False)
    print("query map %.1f ms, scoring + top 1000 %.1f ms, retrieval of the top 20 %.1f ms, %d matches" % go(False))
    print("traced run: query map %.1f ms, scoring + top 1000 %.1f ms, retrieval of the top 20 %.1f ms, %d matches" % go(True))


if __name__ == "__main__":
    main()
