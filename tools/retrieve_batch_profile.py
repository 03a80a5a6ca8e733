import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, querybench, dist as fdist
from folddisco_amd.query import make_query_maps, retrieve_batch
S = 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=20260927, device=dev)
res_off = d["res_off"].contiguous(); R = int(res_off[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = ctx.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
ix = fd.FolddiscoIndex.build(ctx, batch)
queries = querybench._pick_queries(d, S, 64, 4242)
nres = np.diff(res_off.cpu().numpy()).astype(np.uint64)
pen = fd.length_penalty(nres, 0.5)
qall = ctx.upload(fd.PackedStructures.concat([it for _, _, it in queries]))
def go():
    n = 0
    for c0 in range(0, 64, 32):
        ks = list(range(c0, c0 + 32))
        qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in ks], ix, float(S))
        recs = fd.count_query_batch(ctx, ix, [(q.hash, q.qi, q.qj) for q in qms], pen, total_structures=S, top_n=1000)
        cl = [(fdist.rank_hits(r, 32)["nid"]).astype(np.uint32) for r in recs]
        n += sum(len(m) for m in retrieve_batch(ctx, batch, None, cl, qms, qall, ks))
    return n
go()
t = time.perf_counter(); n = go(); print("ms/query", (time.perf_counter() - t) / 64 * 1e3, n)
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
os.environ["FDGPU_TRACE"] = "1"
go()
