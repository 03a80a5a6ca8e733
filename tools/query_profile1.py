"""cProfile of the one-query-at-a-time prefilter path."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, querybench, dist as fdist
from folddisco_amd.query import make_query_map
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=20260927, device=dev)
res_off = d["res_off"].contiguous(); R = int(res_off[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = ctx.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
ix = fd.FolddiscoIndex.build(ctx, batch)
for _ in range(int(os.environ.get('PRE_BUILDS', '0'))):
    ix = None
    ix = fd.FolddiscoIndex.build(ctx, batch)
if os.environ.get('PRE_TIMING'):
    ctx.enable_timing(True); ix = None; ix = fd.FolddiscoIndex.build(ctx, batch); ctx.synchronize(); ctx.last_timings(); ctx.enable_timing(False)
queries = querybench._pick_queries(d, S, 64, 4242)
nres = np.diff(res_off.cpu().numpy()).astype(np.uint64)
pen = fd.length_penalty(nres, 0.5)
qb = [ctx.upload(fd.PackedStructures.concat([it])) for _, _, it in queries]
def go():
    for k in range(64):
        qm = make_query_map(ctx, qb[k], queries[k][1], None, ix, float(S))
        recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
        fdist.allgather_hits(recs, dev, top_n=1000)
go()
t0=time.perf_counter(); go(); print('ms/query', (time.perf_counter()-t0)/64*1e3)
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
ctx.enable_timing(True)
qm = make_query_map(ctx, qb[0], queries[0][1], None, ix, float(S))
fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
print(ctx.last_timings())
