"""Worker of tests/test_gpu_scale.py::test_sharded_query_equals_single_index_at_ecoli_scale: two ranks (gloo) on one GPU, an E. coli-scale
synthetic database (4,400 structures) sharded by structure id; planted motif queries through dist.sharded_query — idf from all-reduced
posting lengths, all-gather of the candidate records, retrieval on the owning rank, all-gather of the matches — against the single
index on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import folddisco_amd as fd  # noqa: E402
from folddisco_amd import dist as fdist  # noqa: E402
from folddisco_amd import query as fq  # noqa: E402
from folddisco_amd import querybench, synth  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ctx = fd.Context(0)
S = 4400
d = synth.generate(S, seed=4400)
ps = synth.to_packed(d)
lo, hi = fdist.shard_range(rank, world, S)
nres = np.diff(ps.res_off).astype(np.uint64)
sl = slice(int(ps.res_off[lo]), int(ps.res_off[hi]))
shard_ps = fd.PackedStructures((ps.res_off[lo:hi + 1] - ps.res_off[lo]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl])
shard = ctx.upload(shard_ps)
ix = fd.FolddiscoIndex.build(ctx, shard, first_id=lo)
pen_shard = fd.length_penalty(nres[lo:hi], 0.5)
queries = querybench._pick_queries(d, S, 6, seed=12)
ok = True
if rank == 0:
    full = ctx.upload(ps)
    fix = fd.FolddiscoIndex.build(ctx, full)
    pen = fd.length_penalty(nres, 0.5)
for s, idx, item in queries:
    qb = ctx.upload(fd.PackedStructures.concat([item]))
    recs, matches = fdist.sharded_query(ctx, ix, lo, shard, qb, idx, None, pen_shard, S, top_n=200)
    if rank == 0:
        qm = fq.make_query_map(ctx, qb, idx, None, fix, float(S))
        want = fdist.rank_hits(fd.count_query(ctx, fix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True), 200)
        same_recs = recs.tobytes() == want.tobytes()
        wm = fq.retrieve(ctx, full, None, want["nid"].astype(np.uint32), qm, qb)
        key = lambda m, nid: (nid, tuple(m["processed"]), tuple(m["from_hash"]), round(m["rmsd"], 5), round(m["idf"], 5))
        a = [key(m, m["nid"]) for m in matches]
        b = [key(m, int(want["nid"][m["cand"]])) for m in wm]
        both = lo < int(want["nid"].min()) + 1 and any(n >= hi for n in want["nid"]) and any(n < hi for n in want["nid"])
        print("QUERY", s, "records", len(recs), "same" if same_recs else "DIFFERENT", "matches", len(a), "same" if a == b else "DIFFERENT",
              "both-shards" if both else "one-shard", flush=True)
        ok = ok and same_recs and a == b and len(recs) == 200 and len(a) >= 1
# the fused batched form (what bench.py's sharded leg runs under gloo): query maps without an index -> fdgpu_query_maps_lengths -> all-reduce
# -> fdgpu_count_query_maps_top_global (device selection of the shard's top_n) -> all-gather -> ranking, then retrieval on the owning rank
qall = ctx.upload(fd.PackedStructures.concat([it for _, _, it in queries]))
qlist = [(k, queries[k][1]) for k in range(len(queries))]
maps = fq.make_query_maps(ctx, qall, qlist, None, float(S))
ix.set_penalty(pen_shard)
globs = fdist.sharded_count_query_maps(ctx, ix, maps, None, S, 200, None, None)
for m in maps:
    m._cache.pop("idf", None)
cl = [(g["nid"][:20][(g["nid"][:20] >= lo) & (g["nid"][:20] < hi)] - lo).astype(np.uint32) for g in globs]
marr = fq.retrieve_batch(ctx, shard, None, cl, maps, qall, list(range(len(maps))), as_arrays=True)
n_all = len(fdist.allgather_array(marr[0]))
if rank == 0:
    fix.set_penalty(pen)
    ref_maps = fq.make_query_maps(ctx, qall, qlist, fix, float(S))
    want = fd.api.count_query_maps(ctx, fix, ref_maps, None, total_structures=S, top_n=200)
    same = all(g.tobytes() == w.tobytes() for g, w in zip(globs, want)) and all(np.array_equal(m.idf, r.idf) for m, r in zip(maps, ref_maps))
    wm = fq.retrieve_batch(ctx, full, None, [w["nid"][:20] for w in want], ref_maps, qall, list(range(len(maps))), as_arrays=True)
    print("FUSED", "same" if same else "DIFFERENT", "matches", n_all, "same" if n_all == len(wm[0]) else "DIFFERENT", flush=True)
    ok = ok and same and n_all == len(wm[0]) and n_all > 0
dist.barrier()
if rank == 0:
    print("SHARDED_OK" if ok else "SHARDED_MISMATCH", flush=True)
dist.destroy_process_group()
