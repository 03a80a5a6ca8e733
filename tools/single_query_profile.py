"""Stage times of one motif query at a time (the `with_matching` leg of bench.py's query section)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, querybench, dist as fdist
from folddisco_amd.query import make_query_map, retrieve, retrieve_batch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=20260927, device=dev)
ro = d["res_off"].contiguous(); R = int(ro[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = ctx.wrap_device(S, R, ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
ix = fd.FolddiscoIndex.build(ctx, batch)
queries = querybench._pick_queries(d, S, 64, 4242)
pen = fd.length_penalty(np.diff(ro.cpu().numpy()).astype(np.uint64), 0.5)
qb = [ctx.upload(fd.PackedStructures.concat([it])) for _, _, it in queries]
T = np.zeros(5)
for rep in range(2):
    T[:] = 0
    for k, (s, idx, _) in enumerate(queries):
        t0 = time.perf_counter(); qm = make_query_map(ctx, qb[k], idx, None, ix, float(S))
        t1 = time.perf_counter(); recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
        t2 = time.perf_counter(); top = fdist.rank_hits(recs, 32); cand = top["nid"].astype(np.uint32)
        t3 = time.perf_counter(); ms = retrieve(ctx, batch, None, cand, qm, qb[k])
        t4 = time.perf_counter(); arr = retrieve_batch(ctx, batch, None, [cand], [qm], qb[k], [0], as_arrays=True)
        t5 = time.perf_counter()
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
print("per query (ms): make_query_map %.3f  count_query %.3f  rank %.3f  retrieve(dicts) %.3f  retrieve(arrays) %.3f" % tuple(1e3 * T / len(queries)))
