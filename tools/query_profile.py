"""cProfile of the batched query loop of bench.py's query leg (host-side cost breakdown)."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, querybench, dist as fdist
from folddisco_amd.query import make_query_map, make_query_maps
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
qall = ctx.upload(fd.PackedStructures.concat([it for _, _, it in queries]))
def go():
    for c0 in range(0, 64, 32):
        qms = make_query_maps(ctx, qall, [(k, queries[k][1]) for k in range(c0, c0 + 32)], None, float(S))
        recs = fd.count_query_batch(ctx, ix, [(q.hash, q.qi, q.qj) for q in qms], pen, total_structures=S)
        for r in recs:
            fdist.allgather_hits(r, dev, top_n=1000)
go()
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
