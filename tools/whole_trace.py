"""Whole-structure query (config 5) at --structures S: the bench's leg (query map, scoring + top 1000 on the device, retrieval of the top 20) timed per stage,
once more with FDGPU_TRACE=1 (the library's own stage stamps on stderr).

    python tools/whole_trace.py [S]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from _resident import build_resident
import folddisco_amd as fd
from folddisco_amd.api import PackedStructures, count_query_maps, length_penalty
from folddisco_amd.query import make_query_map, retrieve, retrieve_batch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 542000
ctx, batch, ix, d, ro = build_resident(S)
res_off = ro.cpu().numpy()
nres = np.diff(res_off).astype(np.uint64)
ix.set_penalty(length_penalty(nres, 0.5))
cand_s = np.nonzero((nres.astype(np.int64) >= 295) & (nres.astype(np.int64) <= 305))[0]
s = int(cand_s[0])
a, b = int(res_off[s]), int(res_off[s + 1])
item = dict(n_xyz=d["n_xyz"][a:b].cpu().numpy(), ca_xyz=d["ca_xyz"][a:b].cpu().numpy(), cb_xyz=d["cb_xyz"][a:b].cpu().numpy(), aa=d["aa"][a:b].cpu().numpy())
qb = ctx.upload(PackedStructures.concat([item]))
allres = np.arange(b - a, dtype=np.uint32)


def once():
    t0 = time.perf_counter()
    qm = make_query_map(ctx, qb, allres, None, ix, float(S))
    t1 = time.perf_counter()
    top = count_query_maps(ctx, ix, [qm], None, total_structures=S, top_n=1000)[0]
    t2 = time.perf_counter()
    m = retrieve_batch(ctx, batch, None, [(top["nid"][:20].astype(np.int64)).astype(np.uint32)], [qm], qb, [0], as_arrays=True)[0]
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(qm.hash), len(m)


once(); once()
runs = [once() for _ in range(7)]
med = lambda k: sorted(r[k] for r in runs)[len(runs) // 2]
print("whole-structure query, %d residues, %d hashes, %d matches of the top 20: query map %.2f ms, scoring + top 1000 %.2f ms, retrieval %.2f ms, total %.2f ms (medians of 7)" %
      (b - a, runs[0][3], runs[0][4], med(0), med(1), med(2), sorted(r[0] + r[1] + r[2] for r in runs)[3]), flush=True)
os.environ["FDGPU_TRACE"] = "1"
once()
