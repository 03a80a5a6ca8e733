"""Worker of tests/test_gpu_e2e.py::test_sharded_query_equals_single_index: two ranks (gloo) on one GPU, each holding the index
and coordinates of its own structures; rank 0 checks the sharded result against the single-index one."""
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
from folddisco_amd import structure as st  # noqa: E402
from tests.helpers import Q1G2F, Q4CHA, SER  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ctx = fd.Context(0)
structs = [st.read_compact_structure(p) for p in SER]
S = len(structs)
lo, hi = fdist.shard_range(rank, world, S)
nres = np.array([s.n for s in structs], np.uint64)
shard = ctx.upload(fd.PackedStructures.concat([s.as_item() for s in structs[lo:hi]]))
ix = fd.FolddiscoIndex.build(ctx, shard, first_id=lo)
std_shard = np.concatenate([s.resname_std() for s in structs[lo:hi]])
pen_shard = fd.length_penalty(nres[lo:hi], 0.5)
ok = True
for qpath, qstr in ((Q4CHA, "B57,B102,C195"), (Q1G2F, "F207,F212,F225,F229"), (Q4CHA, "B57:HKR,B102,C195:ST")):
    q = st.read_compact_structure(qpath)
    res = fq.parse_query_string(qstr, q.chains[0])
    pairs = [(q.get_index(c, r), s) for c, r, s in res]
    pairs = [(i, s) for i, s in pairs if i is not None]
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    recs, matches = fdist.sharded_query(ctx, ix, lo, shard, qb, [i for i, _ in pairs], [s for _, s in pairs], pen_shard, S,
                                        resname_std_shard=std_shard)
    if rank == 0:
        full = ctx.upload(fd.PackedStructures.concat([s.as_item() for s in structs]))
        fix = fd.FolddiscoIndex.build(ctx, full)
        qm = fq.make_query_map(ctx, qb, [i for i, _ in pairs], [s for _, s in pairs], fix, float(S))
        want = fdist.rank_hits(fd.count_query(ctx, fix, qm.hash, qm.qi, qm.qj, fd.length_penalty(nres, 0.5), total_structures=S, as_array=True))
        same_recs = recs.tobytes() == want.tobytes()
        wm = fq.retrieve(ctx, full, np.concatenate([s.resname_std() for s in structs]), want["nid"].astype(np.uint32), qm, qb)
        key = lambda m, nid: (nid, tuple(m["processed"]), tuple(m["from_hash"]), round(m["rmsd"], 5), round(m["idf"], 5))
        a = [key(m, m["nid"]) for m in matches]
        b = [key(m, int(want["nid"][m["cand"]])) for m in wm]
        print("QUERY", qstr, "records", len(recs), "same" if same_recs else "DIFFERENT", "matches", len(a), "same" if a == b else "DIFFERENT", flush=True)
        ok = ok and same_recs and a == b and (len(recs) > 0 or qpath == Q1G2F)   # no zinc finger among the serine peptidases
dist.barrier()
if rank == 0:
    print("SHARDED_OK" if ok else "SHARDED_MISMATCH", flush=True)
dist.destroy_process_group()
