"""cProfile of retrieval (pair scan + graph + Kabsch) for the bench's planted motif queries."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, querybench, dist as fdist
from folddisco_amd.query import make_query_map, retrieve
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=20260927, device=dev)
res_off = d["res_off"].contiguous(); R = int(res_off[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = ctx.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
ix = fd.FolddiscoIndex.build(ctx, batch)
queries = querybench._pick_queries(d, S, 64, 4242)
nres = np.diff(res_off.cpu().numpy()).astype(np.uint64)
pen = fd.length_penalty(nres, 0.5)
qb = [ctx.upload(fd.PackedStructures.concat([it])) for _, _, it in queries]
qms = [make_query_map(ctx, qb[k], queries[k][1], None, ix, float(S)) for k in range(64)]
recs = fd.count_query_batch(ctx, ix, [(q.hash, q.qi, q.qj) for q in qms], pen, total_structures=S)
cands = [(fdist.rank_hits(r, 32)["nid"] - ix.first_id).astype(np.uint32) for r in recs]
def go():
    n = 0
    for k in range(64):
        n += len(retrieve(ctx, batch, None, cands[k], qms[k], qb[k]))
    return n
go()
t = time.perf_counter(); n = go(); print("retrieve ms/query", (time.perf_counter() - t) / 64 * 1e3, "matches", n)
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
ctx.enable_timing(True)
retrieve(ctx, batch, None, cands[0], qms[0], qb[0])
print(ctx.last_timings())
