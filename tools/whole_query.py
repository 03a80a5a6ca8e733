"""Whole-structure query mode (no -q): every residue of a ~300-residue structure is a query node (BASELINE configs[4])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, dist as fdist
from folddisco_amd.query import make_query_map, retrieve
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=20260927, device=dev)
res_off = d["res_off"].contiguous(); R = int(res_off[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = ctx.wrap_device(S, R, res_off.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
ix = fd.FolddiscoIndex.build(ctx, batch)
off = res_off.cpu().numpy()
nres = np.diff(off).astype(np.uint64)
pen = fd.length_penalty(nres, 0.5)
s = int(np.argmin(np.abs(nres.astype(np.int64) - 300)))
a, b = int(off[s]), int(off[s + 1])
item = dict(n_xyz=d["n_xyz"][a:b].cpu().numpy(), ca_xyz=d["ca_xyz"][a:b].cpu().numpy(), cb_xyz=d["cb_xyz"][a:b].cpu().numpy(), aa=d["aa"][a:b].cpu().numpy())
qb = ctx.upload(fd.PackedStructures.concat([item]))
idx = np.arange(b - a, dtype=np.uint32)
for rep in range(2):
    t0 = time.perf_counter(); qm = make_query_map(ctx, qb, idx, None, ix, float(S)); t1 = time.perf_counter()
    recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True); t2 = time.perf_counter()
    top = fdist.rank_hits(recs, 100); t3 = time.perf_counter()
    print(f"residues {b-a} hashes {len(qm.hash)} touched {len(recs)} | make_query_map {1e3*(t1-t0):.1f} ms, count_query {1e3*(t2-t1):.1f} ms, rank {1e3*(t3-t2):.1f} ms; top nid {int(top['nid'][0])} (self {s}) idf {float(top['idf'][0]):.3f}")
ctx.enable_timing(True)
fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
print(ctx.last_timings())
for rep in range(2):
    t0 = time.perf_counter(); ms = retrieve(ctx, batch, None, (top["nid"][:20]).astype(np.uint32), qm, qb); print("retrieve top20: %.1f ms, %d matches" % (1e3 * (time.perf_counter() - t0), len(ms)))
