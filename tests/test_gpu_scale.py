"""Full-size parity through size-independent properties (BASELINE.json configs[1]: E. coli-scale index build + motif
queries on one GPU).  The oracle needs minutes at this size, so the checks are:

* the index exported by fdgpu_index_build (unordered pairs + frames + tables + speculative torsions + radix sort + varint
  encode) is byte-identical to the HOST inversion of the per-structure sorted-unique hash lists returned by
  fdgpu_hash_batch — the S1 entry point, an independent kernel (row-major ordered pairs, generic exact libm chain) that
  the small-size tests pin to the oracle;
* encode -> decode round trip: the decoded posting lists are strictly ascending, inside the shard's id range, and their
  multiset equals the (hash, id) pairs;
* count_query on planted motifs equals a numpy recount over the decoded postings (match / node / edge counts exact, idf
  relative 1e-5 — the tolerance DESIGN.md states for the f32 sum).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ECOLI = 4400


def _host_inversion(h, off, first_id):
    """(hashes, offsets, value bytes) of the reference's index layout from per-structure sorted-unique
    hash lists: stable sort by hash keeps ids ascending; deltas as LEB128."""
    ids = np.repeat(np.arange(len(off) - 1, dtype=np.uint64) + np.uint64(first_id), np.diff(off).astype(np.int64))
    order = np.argsort(h, kind="stable")
    hs, idsrt = h[order], ids[order]
    first = np.ones(len(hs), bool)
    first[1:] = hs[1:] != hs[:-1]
    delta = idsrt.copy()
    delta[1:] = np.where(first[1:], idsrt[1:], idsrt[1:] - idsrt[:-1])
    nb = np.ones(len(delta), np.int64)
    for k in range(1, 10):
        nb += (delta >= (np.uint64(1) << np.uint64(7 * k))).astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(nb)])
    value = np.zeros(pos[-1], np.uint8)
    rem = delta.copy()
    for k in range(int(nb.max())):
        sel = nb > k
        byte = (rem[sel] & np.uint64(0x7f)).astype(np.uint8)
        more = (nb[sel] > k + 1)
        value[pos[:-1][sel] + k] = byte | (more.astype(np.uint8) << 7)
        rem[sel] >>= np.uint64(7)
    uh = hs[first]
    starts = np.nonzero(first)[0]
    offsets = np.concatenate([pos[starts], [pos[-1]]]).astype(np.uint64)
    return uh, offsets, value, hs, idsrt


@pytest.fixture(scope="module")
def ecoli():
    import folddisco_amd as fd
    from folddisco_amd import synth
    ctx = fd.Context(0)
    d = synth.generate(ECOLI, seed=4400)
    ps = synth.to_packed(d)
    batch = ctx.upload(ps)
    first_id = 1000                      # a shard that does not start at 0: absolute first ids take two varint bytes
    ix = fd.FolddiscoIndex.build(ctx, batch, first_id=first_id)
    return ctx, ps, batch, ix, first_id


def test_index_equals_host_inversion_of_s1_hash_lists(ecoli):
    import folddisco_amd as fd
    ctx, ps, batch, ix, first_id = ecoli
    h, off = fd.get_geometric_hash_as_u32(ctx, batch)          # sorted unique per structure
    for s in (0, 1, ECOLI // 2, ECOLI - 1):
        seg = h[int(off[s]):int(off[s + 1])]
        assert np.all(seg[1:] > seg[:-1])
    uh, offsets, value, _, _ = _host_inversion(h, off, first_id)
    v, hh, oo = ix.export()
    assert ix.num_hashes == len(uh) and ix.value_len == len(value)
    assert np.array_equal(hh, uh) and np.array_equal(oo, offsets) and np.array_equal(v, value)


def test_round_trip_and_count_query_at_scale(ecoli):
    import folddisco_amd as fd
    from folddisco_amd import querybench
    from folddisco_amd.query import make_query_map
    ctx, ps, batch, ix, first_id = ecoli
    v, hh, oo = ix.export()
    # decode every posting list on the host
    term = (v & 0x80) == 0
    assert term[-1] and term[oo[1:].astype(np.int64) - 1].all()          # every list ends on a terminator
    tpos = np.nonzero(term)[0]
    L = np.diff(np.concatenate([[-1], tpos]))                              # bytes per varint
    start = tpos - L + 1
    vals = np.zeros(len(tpos), np.uint64)
    for k in range(int(L.max())):
        sel = L > k
        vals[sel] |= (v[start[sel] + k] & 0x7f).astype(np.uint64) << np.uint64(7 * k)
    list_of_val = np.searchsorted(oo, tpos, side="right") - 1
    starts = np.searchsorted(list_of_val, np.arange(len(hh)))
    csum = np.cumsum(vals)
    base = np.where(starts > 0, csum[np.maximum(starts, 1) - 1], np.uint64(0))
    ids = csum - base[list_of_val]
    assert ids.min() >= first_id and ids.max() < first_id + ECOLI
    same = list_of_val[1:] == list_of_val[:-1]
    assert np.all(ids[1:][same] > ids[:-1][same])                          # strictly ascending inside a list
    h1, off1 = fd.get_geometric_hash_as_u32(ctx, batch)
    assert len(ids) == len(h1)                                             # one posting per (structure, distinct hash)
    # planted motifs: GPU scoring vs a numpy recount over the decoded postings
    import torch
    d = dict(res_off=torch.from_numpy(ps.res_off.astype(np.int64)), n_xyz=torch.from_numpy(ps.n_xyz), ca_xyz=torch.from_numpy(ps.ca_xyz),
             cb_xyz=torch.from_numpy(ps.cb_xyz), aa=torch.from_numpy(ps.aa))
    queries = querybench._pick_queries(d, ECOLI, 6, seed=7)
    nres = np.diff(ps.res_off).astype(np.uint64)
    pen = fd.length_penalty(nres, 0.5)
    for s, idx, item in queries:
        qb = ctx.upload(fd.PackedStructures.concat([item]))
        qm = make_query_map(ctx, qb, idx, None, ix, float(ECOLI))
        recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=ECOLI, as_array=True)
        match = np.zeros(ECOLI, np.int64)
        idf = np.zeros(ECOLI, np.float64)
        nodes, edges = {}, {}
        for qh, qi, qj in zip(qm.hash, qm.qi, qm.qj):
            k = np.searchsorted(hh, qh)
            if k >= len(hh) or hh[k] != qh:
                continue
            a, b = starts[k], (starts[k + 1] if k + 1 < len(hh) else len(ids))
            loc = (ids[a:b] - first_id).astype(np.int64)
            match[loc] += 1
            idf[loc] += float(np.log2(np.float32(ECOLI) / np.float32(len(loc))))
            nodes.setdefault(int(qi), np.zeros(ECOLI, bool))[loc] = True
            edges.setdefault((int(qi), int(qj)), np.zeros(ECOLI, bool))[loc] = True
        touched = np.nonzero(match)[0]
        assert np.array_equal(recs["nid"].astype(np.int64) - first_id, touched)
        assert np.array_equal(recs["total_match_count"], match[touched])
        assert np.array_equal(recs["node_count"], sum(x.astype(np.int64) for x in nodes.values())[touched])
        assert np.array_equal(recs["edge_count"], sum(x.astype(np.int64) for x in edges.values())[touched])
        want = idf[touched] * pen[touched].astype(np.float64)
        assert np.allclose(recs["idf"], want, rtol=1e-5, atol=1e-6)
        assert s in touched                                                   # the structure the motif was cut from is a hit


def test_sharded_query_equals_single_index_at_ecoli_scale():
    """SURVEY §8e at configs[1] size: two ranks (gloo, one GPU), 4,400 structures sharded by id, six planted motif queries through
    dist.sharded_query == the single-index query (records byte-identical, matches identical, candidates from both shards)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29755", os.path.join(root, "tests", "shard_query_worker_synth.py")],
                         cwd=root, capture_output=True, text=True, timeout=900)
    assert "SHARDED_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("QUERY") == 6 and "DIFFERENT" not in out.stdout and "both-shards" in out.stdout


def test_c_abi_communicator_world_1(ecoli):
    """fdgpu_comm_* / fdgpu_sharded_count_query[_maps] / fdgpu_sharded_retrieve (csrc/fd_comm.hip: RCCL bound with dlopen) with one rank:
    the collectives are NOT short-cut — every call below issues its ncclAllReduce / ncclAllGather on this GPU (fdgpu_comm_stats counts
    them) and unpacks through the same stride / per-rank code the N-rank path runs — and the results equal the single-index calls."""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import querybench
    from folddisco_amd.query import make_query_map, make_query_maps, retrieve_batch
    ctx, ps, batch, ix, first_id = ecoli
    comm = fdist.Comm(ctx, 0, 1)
    assert len(comm.unique_id) == 128 and any(comm.unique_id)
    assert comm.stats() == (0, 0)
    lens = np.array([3, 0, 2 ** 40 + 7], np.uint64)
    assert np.array_equal(comm.allreduce_lengths(lens), lens)
    assert comm.stats() == (1, 0)                                          # ncclAllReduce ran
    import torch
    d = dict(res_off=torch.from_numpy(ps.res_off.astype(np.int64)), n_xyz=torch.from_numpy(ps.n_xyz), ca_xyz=torch.from_numpy(ps.ca_xyz),
             cb_xyz=torch.from_numpy(ps.cb_xyz), aa=torch.from_numpy(ps.aa))
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    picked = querybench._pick_queries(d, ECOLI, 5, seed=2)
    qs = []
    for s, idx, item in picked:
        qm = make_query_map(ctx, ctx.upload(fd.PackedStructures.concat([item])), idx, None, ix, float(ECOLI))
        qs.append((qm.hash, qm.qi, qm.qj))
    qs.append((np.array([0x3ffffff0], np.uint32), np.zeros(1, np.uint32), np.ones(1, np.uint32)))      # a query without hits
    for top_n, gathers in ((0, 2), (50, 1), (4000, 2)):        # lists: counts + payload; device path: ONE gather of state + ranked records
        a0, g0 = comm.stats()
        got = comm.sharded_count_query(ix, qs, pen, ECOLI, top_n=top_n)
        assert comm.stats() == (a0 + 1, g0 + gathers)
        for g, (qh, qi, qj) in zip(got, qs):
            want = fdist.rank_hits(fd.count_query(ctx, ix, qh, qi, qj, pen, total_structures=ECOLI, as_array=True), top_n or None)
            assert g.tobytes() == want.tobytes()
        assert len(got[-1]) == 0 and len(got[0]) > 0
    # query maps straight into the sharded call (made WITHOUT an index: a shard's lengths mean nothing) == the single-index fused call;
    # the maps' idf is rewritten from the all-reduced lengths of primary_hash
    qall = ctx.upload(fd.PackedStructures.concat([it for _, _, it in picked]))
    qlist = [(k, picked[k][1]) for k in range(len(picked))]
    ix.set_penalty(pen)
    ref_maps = make_query_maps(ctx, qall, qlist, ix, float(ECOLI))
    want = fd.api.count_query_maps(ctx, ix, ref_maps, None, total_structures=ECOLI, top_n=100)
    maps = make_query_maps(ctx, qall, qlist, None, float(ECOLI))
    assert not np.any(maps[0].idf)
    a0, g0 = comm.stats()
    got = comm.sharded_count_query_maps(ix, maps, None, ECOLI, top_n=100)
    assert comm.stats() == (a0 + 1, g0 + 1)
    for m in maps:
        m._cache.pop("idf", None)
    for g, w, m, r in zip(got, want, maps, ref_maps):
        assert g.tobytes() == w.tobytes() and len(g) == 100
        assert np.array_equal(m.idf, r.idf) and np.any(m.idf)
    # the gloo form of the same call (both local halves through the C ABI, no process group = no exchange) agrees too
    maps2 = make_query_maps(ctx, qall, qlist, None, float(ECOLI))
    got2 = fdist.sharded_count_query_maps(ctx, ix, maps2, None, ECOLI, 100, None, None)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(got2, want))
    # sharded retrieval: global candidate ids in, matches of the owning rank gathered and merged == retrieve_batch on the database
    cand_nids = [g["nid"][:12] for g in got]
    cand_nids[1] = np.concatenate([cand_nids[1][:5], np.array([first_id + ECOLI + 7, 3], np.uint32)])      # ids no rank owns: no matches, slots kept
    a0, g0 = comm.stats()
    marr, moff, rarr, roff = comm.sharded_retrieve(batch, first_id, None, cand_nids, maps, qall, list(range(len(maps))))
    assert comm.stats() == (a0, g0 + 2)
    local = [(c[(c >= first_id) & (c < first_id + ECOLI)] - first_id).astype(np.uint32) for c in cand_nids]
    wm, wmo, wr, wro = retrieve_batch(ctx, batch, None, local, ref_maps, qall, list(range(len(maps))), as_arrays=True)
    assert np.array_equal(moff, wmo) and np.array_equal(roff, wro) and np.array_equal(rarr, wr) and len(marr) > 0
    for t in range(len(maps)):
        own = np.nonzero((cand_nids[t] >= first_id) & (cand_nids[t] < first_id + ECOLI))[0]
        a, b = int(moff[t]), int(moff[t + 1])
        assert np.array_equal(marr["cand"][a:b], own[wm["cand"][a:b]])                     # slot in the GLOBAL list
    for f in marr.dtype.names:
        if f != "cand":
            assert np.array_equal(marr[f], wm[f]), f
    comm.close()


def test_device_merge_of_gathered_messages():
    """What every rank runs after the all-gather (fd_comm.hip: k_comm_plan / k_comm_pack, the radix select and bitonic sort of the union,
    all on the device) driven with hand-made contributions of 3 and 8 ranks: different lengths per rank and query, empty contributions,
    idf ties across ranks (broken by ascending nid), fewer records than top_n, more ties at the cut-off than the selection holds (the host
    ranks that call), a rank reporting an error."""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd.api import REC_DTYPE, FdgpuError
    ctx = fd.Context(0)
    rng = np.random.Generator(np.random.PCG64(99))

    def contribution(n, lo, hi, tie=None):
        r = np.zeros(n, REC_DTYPE)
        r["nid"] = np.sort(rng.choice(np.arange(lo, hi), size=n, replace=False)).astype(np.uint32)
        r["total_match_count"] = rng.integers(1, 50, n)
        r["node_count"] = rng.integers(1, 5, n)
        r["edge_count"] = rng.integers(1, 9, n)
        r["idf"] = (rng.integers(0, 40, n) / 8.0 if tie else rng.random(n) * 30.0).astype(np.float32)
        return fdist.rank_hits(r)

    for W, T, top_n in ((3, 5, 64), (8, 33, 1000), (2, 1, 3072)):
        per_rank = []
        for r in range(W):
            lists = []
            for t in range(T):
                n = int(rng.integers(0, top_n + 1)) if (r + t) % 4 else 0          # some empty contributions
                if t == 2:
                    n = min(top_n, 7)                                             # fewer than top_n over all ranks
                lists.append(contribution(n, r * 100000, (r + 1) * 100000, tie=(t % 2 == 0)))
            per_rank.append(lists)
        msgs = np.concatenate([fdist.build_message(ctx, per_rank[r], top_n) for r in range(W)])
        assert len(msgs) == W * int(ctx.L.fdgpu_comm_message_bytes(T, top_n))
        got = fdist.merge_gathered(ctx, msgs, W, T, top_n)
        for t in range(T):
            want = fdist.rank_hits(np.concatenate([per_rank[r][t] for r in range(W)]), top_n)
            assert got[t].tobytes() == want.tobytes(), (W, T, top_n, t)
    # more than top_n + 1024 records tie at the cut-off: the device selection overflows and the packed lists are ranked on the host
    W, T, top_n = 4, 2, 600
    per_rank = []
    for r in range(W):
        a = contribution(600, r * 1000, (r + 1) * 1000)
        a["idf"] = np.float32(5.0)
        per_rank.append([a, contribution(10, r * 1000, (r + 1) * 1000)])
    msgs = np.concatenate([fdist.build_message(ctx, per_rank[r], top_n) for r in range(W)])
    got = fdist.merge_gathered(ctx, msgs, W, T, top_n)
    for t in range(T):
        assert got[t].tobytes() == fdist.rank_hits(np.concatenate([per_rank[r][t] for r in range(W)]), top_n).tobytes()
    # a rank whose local step failed: every rank sees the status and fails the call
    bad = np.concatenate([fdist.build_message(ctx, per_rank[0], top_n), fdist.build_message(ctx, [per_rank[1][0][:0]] * T, top_n, status=2)])
    with pytest.raises(FdgpuError, match="rank 1 failed"):
        fdist.merge_gathered(ctx, bad, 2, T, top_n)


def test_config2_ecoli_zinc_finger_against_oracle_outright():
    """BASELINE.json configs[1] as written — "E. coli proteome index build + query/zinc_finger.txt on 1 MI355X" — against the oracle OUTRIGHT:
    4,400 synthetic structures with the reference's zinc-finger motif (query/zinc_finger.txt) planted into 24 of them.  The oracle hashes
    every structure itself and builds its own index (its dense two-pass table build); the GPU index must equal it byte for byte, the query
    map bit for bit, count_query over ALL structures (counts exact, idf 1e-5), the device-ranked top N, and the matches of the top 32
    candidates (residues exact, RMSD 1e-4) — no recount, no property check."""
    import folddisco_amd as fd
    import oracle
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from folddisco_amd import synth
    from tests.test_gpu_configs import _check_matches, _ostruct, _parse_query_file, _rot
    ctx = fd.Context(0)
    ps = synth.to_packed(synth.generate(ECOLI, seed=1729))
    rng = np.random.Generator(np.random.PCG64(11))
    path, qstr = _parse_query_file("zinc_finger.txt")
    q = st.read_compact_structure(path)
    res = fq.parse_query_string(qstr, q.chains[0])
    pairs = [(q.get_index(c, r), s) for c, r, s in res]
    idx = np.array([i for i, _ in pairs])
    planted = np.sort(rng.permutation(ECOLI)[:24])
    extra = np.zeros(ECOLI, np.int64)
    ins_pos, ins = [], dict(n_xyz=[], ca_xyz=[], cb_xyz=[], aa=[])
    for t, sid in enumerate(planted):
        R, tr = _rot(rng), rng.normal(0, 30, 3) + 60.0
        noise = 0.0 if t % 2 == 0 else 0.12
        for key, src in (("n_xyz", q.n_xyz), ("ca_xyz", q.ca_xyz), ("cb_xyz", q.cb_xyz)):
            x = src[idx].astype(np.float64) @ R.T + tr + rng.normal(0, 1, (len(idx), 3)) * noise
            ins[key].append(np.round(x, 3).astype(np.float32))
        ins["aa"].append(q.aa[idx])
        ins_pos.append(np.full(len(idx), int(ps.res_off[sid + 1])))
        extra[sid] += len(idx)
    pos = np.concatenate(ins_pos)
    order = np.argsort(pos, kind="stable")
    cat = {k: np.concatenate(v)[order] for k, v in ins.items()}
    new_off = ps.res_off.astype(np.int64).copy()
    new_off[1:] += np.cumsum(extra)
    ps = fd.PackedStructures(new_off.astype(np.uint64), np.insert(ps.n_xyz, pos[order], cat["n_xyz"], axis=0), np.insert(ps.ca_xyz, pos[order], cat["ca_xyz"], axis=0),
                             np.insert(ps.cb_xyz, pos[order], cat["cb_xyz"], axis=0), np.insert(ps.aa, pos[order], cat["aa"]))
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    # --- the oracle's own index of the same structures
    structs = [_ostruct(ps, s) for s in range(ECOLI)]
    oh, ooff = oracle.hash_batch(structs)
    oix = oracle.build_index_from_lists_mt(oh, ooff, 16)
    v, h, o = ix.export()
    assert np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
    nres = np.diff(ps.res_off).astype(np.uint64)
    pen = fd.length_penalty(nres, 0.5)
    # --- query map, prefilter over all structures, ranked selection
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    qm = fq.make_query_map(ctx, qb, idx.astype(np.uint32), [s for _, s in pairs], ix, float(ECOLI))
    oq = oracle.read_pdb(path)
    om = oracle.make_query_map(oq, qstr, oix, float(ECOLI))
    oa = om.arrays()
    assert np.array_equal(qm.hash, oa["hash"]) and np.array_equal(qm.qi, oa["qi"]) and np.array_equal(qm.qj, oa["qj"])
    assert np.array_equal(qm.idf.view(np.uint32), oa["idf"].view(np.uint32))
    recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=ECOLI, as_array=True)
    want = oracle.count_query(om, oix, nres)
    assert [(int(r["nid"]), int(r["total_match_count"]), int(r["node_count"]), int(r["edge_count"])) for r in recs] == \
           [(w["nid"], w["total_match_count"], w["node_count"], w["edge_count"]) for w in want]
    assert np.allclose(recs["idf"], [w["idf"] for w in want], rtol=1e-5, atol=0)
    assert set(planted.tolist()) <= set(int(r["nid"]) for r in recs)
    top = fd.count_query_maps(ctx, ix, [qm], pen, total_structures=ECOLI, top_n=1000)[0]
    assert top.tobytes() == fdist.rank_hits(recs, 1000).tobytes()
    # --- retrieval of the top 32 candidates
    cand = top["nid"][:32].astype(np.uint32)
    got = fq.retrieve(ctx, batch, None, cand, qm, qb)
    n = _check_matches(got, cand, ps, oq, om)
    full = sum(1 for g in got if all(x >= 0 for x in g["processed"]))
    assert n >= 12 and full >= 12          # the noise-free plants, at least, are recovered whole


def test_merge_of_gathered_retrieval_payloads():
    """What every rank runs after fdgpu_sharded_retrieve's second all-gather (padded payloads of match records + residue lists, merged per query
    by candidate slot) driven with hand-made contributions of 2, 3 and 8 ranks: ragged counts, ranks without matches, queries nobody matched,
    several components per slot (their order inside a rank must survive), a rank reporting an error."""
    import ctypes as C
    import folddisco_amd as fd
    from folddisco_amd._lib import MatchRec, u64p
    from folddisco_amd.api import FdgpuError
    from folddisco_amd.query import MATCH_DTYPE
    ctx = fd.Context(0)
    rng = np.random.Generator(np.random.PCG64(2027))
    i32p = C.POINTER(C.c_int32)
    for world in (2, 3, 8):
        nq = 5
        nres_per = np.array([6, 8, 6, 10, 4], np.uint64)
        counts = np.zeros((world, nq + 1), np.uint64)
        rank_m, rank_r = [], []
        for r in range(world):
            ms, rs = [], []
            for t in range(nq):
                if t == 3 or (r == 1 and world > 2):        # a query nobody matched; a rank without any match
                    continue
                slots = np.sort(rng.choice(np.arange(r, 64, world), size=int(rng.integers(0, 6)), replace=False))     # a slot belongs to ONE rank
                for s_ in slots:
                    for comp in range(int(rng.integers(1, 4))):      # components of a slot, in this order
                        m = np.zeros(1, MATCH_DTYPE)
                        m["cand"] = s_; m["same"] = comp; m["idf"] = rng.random(); m["rmsd"] = rng.random()
                        m["rot"] = rng.random(9); m["tran"] = rng.random(3); m["metrics"] = rng.random(5)
                        ms.append(m); rs.append(rng.integers(-1, 300, size=int(nres_per[t])).astype(np.int32))
                        counts[r, t] += 1
            rank_m.append(np.concatenate(ms) if ms else np.zeros(0, MATCH_DTYPE))
            rank_r.append(np.concatenate(rs) if rs else np.zeros(0, np.int32))

        def call(cnt):
            pm = (C.POINTER(MatchRec) * world)(*[x.ctypes.data_as(C.POINTER(MatchRec)) if len(x) else None for x in rank_m])
            pr = (i32p * world)(*[x.ctypes.data_as(i32p) if len(x) else None for x in rank_r])
            om, omo, orr, oro = C.POINTER(MatchRec)(), u64p(), i32p(), u64p()
            cnt = np.ascontiguousarray(cnt)
            ctx.check(ctx.L.fdgpu_debug_merge_retrieved(ctx.h, world, nq, cnt.ctypes.data_as(u64p), pm, pr, nres_per.ctypes.data_as(u64p),
                                                        C.byref(om), C.byref(omo), C.byref(orr), C.byref(oro)))
            mo = np.ctypeslib.as_array(omo, (nq + 1,)).copy(); ro = np.ctypeslib.as_array(oro, (nq + 1,)).copy()
            m = np.frombuffer(C.string_at(om, int(mo[-1]) * MATCH_DTYPE.itemsize), MATCH_DTYPE).copy() if mo[-1] else np.zeros(0, MATCH_DTYPE)
            res = np.ctypeslib.as_array(orr, (max(int(ro[-1]), 1),)).copy()[:int(ro[-1])]
            ctx.L.fdgpu_matches_free(om, orr); ctx.L.fdgpu_free(omo); ctx.L.fdgpu_free(oro)
            return m, mo, res, ro
        m, mo, res, ro = call(counts)
        # the expected merge: per query, every rank's records in rank order, stably sorted by slot
        at_m, at_r = [0] * world, [0] * world
        wm, wr = [], []
        exp_mo, exp_ro = [0], [0]
        for t in range(nq):
            refs = []
            for r in range(world):
                k = int(counts[r, t])
                for z in range(k):
                    refs.append((int(rank_m[r][at_m[r] + z]["cand"]), len(refs), r, at_m[r] + z, at_r[r] + z * int(nres_per[t])))
                at_m[r] += k; at_r[r] += k * int(nres_per[t])
            refs.sort(key=lambda x: (x[0], x[1]))
            for _, _, r, mi, ri in refs:
                wm.append(rank_m[r][mi:mi + 1]); wr.append(rank_r[r][ri:ri + int(nres_per[t])])
            exp_mo.append(len(wm)); exp_ro.append(sum(len(x) for x in wr))
        assert mo.tolist() == exp_mo and ro.tolist() == exp_ro
        assert m.tobytes() == (np.concatenate(wm).tobytes() if wm else b"") and res.tobytes() == (np.concatenate(wr).tobytes() if wr else b"")
        bad = counts.copy(); bad[world - 1, nq] = 5       # the last rank failed its local step: every rank returns the error
        with pytest.raises(FdgpuError):
            call(bad)


def test_bench_n_gt_1_path_runs_with_two_gloo_ranks_on_one_gpu():
    """`python bench.py --gpus 2` — the command the driver's scaling lease runs — executed end to end here with two ranks on ONE GPU over gloo
    (FD_BENCH_BACKEND=gloo) at 8,000 structures: the shard build per rank, the sharded query legs with their exchange, the replica leg through
    the query lanes, and the line's N > 1 fields.  Not a performance number: it keeps the driver's run from being the first execution of this
    code (a typo in run_replicas / value_from would otherwise surface there)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FD_BENCH_BACKEND="gloo", PYTHONPATH=root)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--structures", "8000", "--steps", "1", "--warmup", "1", "--queries", "64",
                        "--no-cpu-baseline", "--no-cli-index"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["structures"] == 8000 and out["config"]["structures_per_gpu"] == 4000
    q = out["query"]
    assert "error" not in q, q
    assert q["value_from"] in ("replicas", "sharded") and q["sharded_value"] > 0
    assert "error" not in q["replicas"] and q["replicas"]["value"] > 0 and q["replicas"]["replicas"] == 2
    assert q["value"] == max(q["sharded_value"], q["replicas"]["value"])
    assert q["exchange"] is not None and q["batched_with_matching"]["matches"] > 0
    assert out["roofline"]["frac"] > 0 and out["export_inclusive"]["value"] > 0
    # the N > 1 line carries BOTH build figures: independent shard builds (the headline) and the same step + the exchange that makes them one index
    si = out["single_index_inclusive"]
    assert "error" not in si and si["value"] > 0 and si["ms_per_step"] > out["ms_per_step"] * 0.5 and si["total_value_bytes"] > 0 and si["exchange_bytes_sent_by_rank0_estimate"] > 0, si
    assert "INDEPENDENT shard builds" in out["config"]["parallelism"]
    disk = out["index_on_disk_inclusive"]
    assert "error" not in disk and (disk.get("skipped") or (disk["value"] > 0 and disk["files_complete"])), disk
    # the N > 1 watchdog: with a limit shorter than the legs behind the timed build, rank 0 prints the line with the completed measurement and every
    # rank leaves with status 0 — what keeps a rank stuck in a never-executed collective from taking the headline with it
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--structures", "8000", "--steps", "1", "--warmup", "0", "--queries", "64",
                        "--no-cpu-baseline", "--no-cli-index"], cwd=root, env=dict(env, FD_BENCH_WATCHDOG_S="0.05"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["watchdog"]["fired_after_s"] == 0.05 and out["n_gpus"] == 2 and out["value"] > 0 and out["ms_per_step"] > 0 and out["query"] is None
