"""BASELINE.json configs 3 and 5 at their own sizes, against the oracle.

* configs[2] "Human AFDB proteome index + batch query/*.txt": a human-scale synthetic database (20,500 AFDB-shaped structures)
  with the motifs of the reference's five shipped query files (query/*.txt: serine peptidase triad, zinc finger, aminopeptidase,
  enolase with substitutions, knottin) PLANTED into known structures (rigidly moved copies, half of them with coordinate
  noise), so every query has real hits.  Query map, prefilter records and the matches of the top candidates — one at a time
  and through the batched entry points — equal the oracle's; the index is checked through S1 hash lists of sampled structures
  and through the device merge of five sub-builds.
* configs[4] whole-structure query mode (no -q): a 300-residue query (90 k hashes) against the same 20,500 structures:
  prefilter records of every touched structure and the matches of the top 20 equal the oracle's — the large-query code paths
  (hashed lookup, two scans, packed candidate pairs) at a size where they are the ones that run.

The oracle answers from the EXPORT of the GPU index (oracle.BorrowedIndex): its own table build needs ~15 minutes at this size,
and the index bytes are pinned to it at the sizes of tests/test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HUMAN = 20500
QDIR = os.path.join(HERE, "golden", "query")
QUERY_FILES = ["serine_peptidase.txt", "zinc_finger.txt", "aminopeptidase.txt", "enolase.txt", "knottin.txt"]
N_PLANT = 24


def _rot(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _parse_query_file(name):
    line = open(os.path.join(QDIR, name)).read().rstrip("\n").split("\t")
    return os.path.join(QDIR, os.path.basename(line[0])), line[1]


@pytest.fixture(scope="module")
def human():
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from folddisco_amd import synth
    ctx = fd.Context(0)
    ps = synth.to_packed(synth.generate(HUMAN, seed=2050, device="cuda"))
    rng = np.random.Generator(np.random.PCG64(77))
    queries = []
    ins_pos, ins = [], dict(n_xyz=[], ca_xyz=[], cb_xyz=[], aa=[])
    extra = np.zeros(HUMAN, np.int64)
    free = rng.permutation(HUMAN)
    for k, name in enumerate(QUERY_FILES):
        path, qstr = _parse_query_file(name)
        q = st.read_compact_structure(path)
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        assert all(i is not None for i, _ in pairs), name
        idx = np.array([i for i, _ in pairs])
        planted = np.sort(free[k * N_PLANT:(k + 1) * N_PLANT])
        for t, sid in enumerate(planted):
            R, tr = _rot(rng), rng.normal(0, 30, 3) + 60.0
            noise = 0.0 if t % 2 == 0 else 0.12
            for key, src in (("n_xyz", q.n_xyz), ("ca_xyz", q.ca_xyz), ("cb_xyz", q.cb_xyz)):
                x = src[idx].astype(np.float64) @ R.T + tr + rng.normal(0, 1, (len(idx), 3)) * noise
                ins[key].append(np.round(x, 3).astype(np.float32))
            ins["aa"].append(q.aa[idx])
            ins_pos.append(np.full(len(idx), int(ps.res_off[sid + 1])))
            extra[sid] += len(idx)
        queries.append(dict(name=name, path=path, qstr=qstr, q=q, idx=idx.astype(np.uint32), subs=[s for _, s in pairs], planted=planted))
    pos = np.concatenate(ins_pos)
    order = np.argsort(pos, kind="stable")
    cat = {k: np.concatenate(v)[order] for k, v in ins.items()}
    new_off = ps.res_off.astype(np.int64).copy()
    new_off[1:] += np.cumsum(extra)
    ps = fd.PackedStructures(new_off.astype(np.uint64), np.insert(ps.n_xyz, pos[order], cat["n_xyz"], axis=0), np.insert(ps.ca_xyz, pos[order], cat["ca_xyz"], axis=0),
                             np.insert(ps.cb_xyz, pos[order], cat["cb_xyz"], axis=0), np.insert(ps.aa, pos[order], cat["aa"]))
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    v, h, o = ix.export()
    oix = oracle.BorrowedIndex(h, o, v)
    nres = np.diff(ps.res_off).astype(np.uint64)
    return dict(ctx=ctx, ps=ps, batch=batch, ix=ix, oix=oix, nres=nres, pen=fd.length_penalty(nres, 0.5), queries=queries, export=(v, h, o))


def _ostruct(ps, s):
    a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
    return oracle.structure_from_packed(ps.n_xyz[a:b], ps.ca_xyz[a:b], ps.cb_xyz[a:b], ps.aa[a:b])


def _check_matches(got, cand, ps, oq, om, rmsd_tol=1e-4):
    n = 0
    for slot, nid in enumerate(cand):
        R = oracle.retrieve(_ostruct(ps, int(nid)), oq, om)
        mine = [g for g in got if g["cand"] == slot]
        assert len(mine) == len(R["processed"]), (slot, nid)
        for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
            assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]], (slot, nid)
            assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]], (slot, nid)
            assert abs(g["rmsd"] - rp["rmsd"]) <= rmsd_tol and g["idf"] == pytest.approx(rp["idf"], rel=1e-5)
            n += 1
    return n


def test_index_at_human_scale(human):
    """S1 hash lists of sampled structures <-> decoded posting lists (both directions), and five sub-builds merged on the device
    == the single build, byte for byte"""
    import folddisco_amd as fd
    ctx, ps, ix = human["ctx"], human["ps"], human["ix"]
    v, h, o = human["export"]
    rng = np.random.Generator(np.random.PCG64(1))
    sample = np.sort(rng.choice(HUMAN, 120, replace=False))
    items = []
    for s in sample:
        a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
        items.append(dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b]))
    hs, off = fd.get_geometric_hash_as_u32(ctx, ctx.upload(fd.PackedStructures.concat(items)))
    # structure -> index: every (hash, id) of the sampled structures is in the hash's posting list
    pick = rng.choice(len(hs), 4000, replace=False)
    sid_of = sample[np.searchsorted(off, pick, side="right") - 1]
    lists = ix.get_entries(hs[pick])
    assert all(int(s) in set(l.tolist()) if len(l) < 64 else bool(np.isin(s, l)) for s, l in zip(sid_of, lists))
    # index -> structure: ids of the sampled structures inside those lists really hold the hash
    own = {int(s): set(hs[int(off[k]):int(off[k + 1])].tolist()) for k, s in enumerate(sample)}
    n_back = 0
    for hq, l in zip(hs[pick][:1500], lists[:1500]):
        assert np.all(l[1:] > l[:-1]) and l[-1] < HUMAN
        for s in np.intersect1d(l, sample):
            assert int(hq) in own[int(s)]
            n_back += 1
    assert n_back >= 1500
    assert ix.num_postings == len(v) - int(np.count_nonzero(v & 0x80))   # one terminator byte per posting
    # five sub-builds over consecutive id ranges, merged on the device
    cuts = [0, 4100, 8200, 12300, 16400, HUMAN]
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        sl = slice(int(ps.res_off[a]), int(ps.res_off[b]))
        chunk = fd.PackedStructures((ps.res_off[a:b + 1] - ps.res_off[a]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl])
        parts.append(fd.FolddiscoIndex.build(ctx, ctx.upload(chunk), first_id=a))
    mv, mh, mo = fd.FolddiscoIndexSet(parts).merge().export()
    assert np.array_equal(mh, h) and np.array_equal(mo, o) and np.array_equal(mv, v)


@pytest.mark.parametrize("qk", range(len(QUERY_FILES)))
def test_shipped_motif_queries_at_human_scale(human, qk):
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    ctx, ps, batch, ix, oix = human["ctx"], human["ps"], human["batch"], human["ix"], human["oix"]
    Q = human["queries"][qk]
    q = Q["q"]
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    qm = fq.make_query_map(ctx, qb, Q["idx"], Q["subs"], ix, float(HUMAN))
    oq = oracle.read_pdb(Q["path"])
    om = oracle.make_query_map(oq, Q["qstr"], oix, float(HUMAN))
    oa = om.arrays()
    assert np.array_equal(qm.hash, oa["hash"]) and np.array_equal(qm.qi, oa["qi"]) and np.array_equal(qm.qj, oa["qj"])
    assert np.array_equal(qm.idf.view(np.uint32), oa["idf"].view(np.uint32))
    recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, human["pen"], total_structures=HUMAN, as_array=True)
    want = oracle.count_query(om, oix, human["nres"])
    assert [(int(r["nid"]), int(r["total_match_count"]), int(r["node_count"]), int(r["edge_count"])) for r in recs] == \
           [(w["nid"], w["total_match_count"], w["node_count"], w["edge_count"]) for w in want]
    assert np.allclose(recs["idf"], [w["idf"] for w in want], rtol=1e-5, atol=0)
    # the exact copies carry the whole motif
    by = {int(r["nid"]): r for r in recs}
    for t, sid in enumerate(Q["planted"]):
        assert int(sid) in by
        if t % 2 == 0:
            assert int(by[int(sid)]["node_count"]) == len(Q["idx"])
    top = fdist.rank_hits(recs, 40)
    cand = top["nid"].astype(np.uint32)
    assert len(np.intersect1d(cand, Q["planted"])) >= N_PLANT // 2          # planted copies lead the ranking
    std = np.ones(int(ps.res_off[-1]), np.uint8)
    got = fq.retrieve(ctx, batch, std, cand, qm, qb)
    n = _check_matches(got, cand, ps, oq, om)
    full = [g for g in got if sum(1 for x in g["processed"] if x >= 0) == len(Q["idx"])]
    assert n >= N_PLANT // 2 and len(full) >= N_PLANT // 2 and min(g["rmsd"] for g in full) < 0.01


def test_batch_of_shipped_queries_equals_singles(human):
    """`query -q query/*.txt` as one batch (fdgpu_make_query_map_batch + fdgpu_count_query_batch_top + fdgpu_retrieve_batch): the same
    candidates and matches as one query at a time"""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    ctx, ps, batch, ix = human["ctx"], human["ps"], human["batch"], human["ix"]
    Qs = human["queries"]
    qall = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item() for Q in Qs]))
    qms = fq.make_query_maps(ctx, qall, [(k, Q["idx"], Q["subs"]) for k, Q in enumerate(Qs)], ix, float(HUMAN))
    recs = fd.count_query_batch(ctx, ix, [(m.hash, m.qi, m.qj) for m in qms], human["pen"], total_structures=HUMAN, top_n=1000)
    std = np.ones(int(ps.res_off[-1]), np.uint8)
    cands = []
    for k, (Q, m, r) in enumerate(zip(Qs, qms, recs)):
        qb = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item()]))
        single = fq.make_query_map(ctx, qb, Q["idx"], Q["subs"], ix, float(HUMAN))
        assert single.hash.tobytes() == m.hash.tobytes() and single.idf.tobytes() == m.idf.tobytes()
        full = fd.count_query(ctx, ix, m.hash, m.qi, m.qj, human["pen"], total_structures=HUMAN, as_array=True)
        assert r.tobytes() == fdist.rank_hits(full, 1000).tobytes()      # top 1000 of 20,500 ranked on the device
        cands.append(fdist.rank_hits(r, 25)["nid"].astype(np.uint32))
    got = fq.retrieve_batch(ctx, batch, std, cands, qms, qall, list(range(len(Qs))))
    for k, (Q, m) in enumerate(zip(Qs, qms)):
        qb = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item()]))
        single = fq.retrieve(ctx, batch, std, cands[k], fq.make_query_map(ctx, qb, Q["idx"], Q["subs"], ix, float(HUMAN)), qb)
        assert len(single) == len(got[k]) and len(single) >= N_PLANT // 2
        for a, b in zip(got[k], single):
            assert a["cand"] == b["cand"] and a["processed"] == b["processed"] and a["from_hash"] == b["from_hash"]
            assert a["rmsd"] == b["rmsd"] and a["idf"] == b["idf"]


def test_fused_query_batch_equals_the_three_calls(human, monkeypatch):
    """fdgpu_query_batch (maps -> ranked top_n -> retrieval of the first match_top candidates in one call, stages overlapped inside it) returns
    the arrays of fdgpu_make_query_map_batch + fdgpu_count_query_maps_top + fdgpu_retrieve_batch bit for bit: the shipped motifs with their
    substitution lists, a query with no hit at all, top_n beyond the device selection (host ranking path inside the fused call), match_top
    above top_n, the residue-name filter."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd.api import count_query_maps
    ctx, ps, batch, ix = human["ctx"], human["ps"], human["batch"], human["ix"]
    Qs = human["queries"]
    qall = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item() for Q in Qs]))
    ix.set_penalty(human["pen"])
    queries = [(k, Q["idx"], Q["subs"]) for k, Q in enumerate(Qs)] * 3 + [(0, np.array([0, 1], np.uint32), None)]
    std = np.ones(int(ps.res_off[-1]), np.uint8)
    fields = ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi", "primary_hash")
    n_matches = 0
    for top_n, match_top, use_std, pen in ((1000, 25, True, None), (40, 64, False, None), (5000, 10, False, human["pen"])):
        maps = fq.make_query_maps(ctx, qall, queries, ix, float(HUMAN))
        recs, off = count_query_maps(ctx, ix, maps, pen, total_structures=HUMAN, top_n=top_n, flat=True)
        cl = [recs["nid"][int(off[t]): int(off[t]) + min(match_top, int(off[t + 1] - off[t]))].astype(np.uint32) for t in range(len(queries))]
        ref = fq.retrieve_batch(ctx, batch, std if use_std else None, cl, maps, qall, [q[0] for q in queries], as_arrays=True)
        fmaps, (frecs, foff), got = fq.query_batch(ctx, ix, batch, qall, queries, float(HUMAN), top_n, match_top, penalty=pen,
                                                  resname_std=std if use_std else None)
        for a, b in zip(maps, fmaps):
            for f in fields:
                assert getattr(a, f).tobytes() == getattr(b, f).tobytes(), f
        assert foff.tobytes() == off.tobytes() and frecs.tobytes() == recs.tobytes()
        for a, b, what in zip(ref, got, ("matches", "match_off", "residues", "res_off")):
            assert a.tobytes() == b.tobytes(), (top_n, what)
        n_matches += len(got[0])
    assert n_matches > 3 * 5 * N_PLANT // 2


def test_pipelined_submit_wait_equals_the_blocking_call(human):
    """fdgpu_query_batch_submit / fdgpu_query_batch_wait (batches in flight on the context's query lanes: sibling contexts driven by library
    threads) return, per batch, the arrays of the blocking fdgpu_query_batch bit for bit — different batches in flight at once (different
    queries, sizes, top_n per job), waits out of submission order, a job that is dropped without taking its results, a failing job, and the
    caller's own context usable in between."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    ctx, ps, batch, ix = human["ctx"], human["ps"], human["batch"], human["ix"]
    Qs = human["queries"]
    qall = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item() for Q in Qs]))
    ix.set_penalty(human["pen"])
    base = [(k, Q["idx"], Q["subs"]) for k, Q in enumerate(Qs)]
    jobs_in = [(base * 2, 1000, 25), (base[:2], 40, 8), (base[::-1] + [(0, np.array([0, 1], np.uint32), None)], 1000, 32), (base[1:4] * 5, 300, 16),
               (base, 1000, 25)]
    fields = ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi", "primary_hash")

    def same(ref, got):
        (m0, (r0, o0), t0), (m1, (r1, o1), t1) = ref, got
        assert len(m0) == len(m1)
        for a, b in zip(m0, m1):
            for f in fields:
                assert getattr(a, f).tobytes() == getattr(b, f).tobytes(), f
        assert o0.tobytes() == o1.tobytes() and r0.tobytes() == r1.tobytes()
        for a, b in zip(t0, t1):
            assert a.tobytes() == b.tobytes()
    refs = [fq.query_batch(ctx, ix, batch, qall, q, float(HUMAN), tn, mt) for q, tn, mt in jobs_in]
    assert sum(len(r[2][0]) for r in refs) > 100
    assert ctx.L.fdgpu_query_lanes(ctx.h, 0) == 0
    for rnd in range(3):
        jobs = [fq.query_batch_submit(ctx, ix, batch, qall, q, float(HUMAN), tn, mt) for q, tn, mt in jobs_in]
        assert ctx.L.fdgpu_query_lanes(ctx.h, 0) == 6          # the default number of lanes, made by the first submit
        mid = fq.query_batch(ctx, ix, batch, qall, jobs_in[1][0], float(HUMAN), 40, 8)      # the caller's own context while its lanes are busy
        same(refs[1], mid)
        order = list(range(len(jobs)))[::-1] if rnd == 1 else list(range(len(jobs)))
        for k in order:
            same(refs[k], jobs[k].wait())
    # generator form, depth 2
    for k, got in enumerate(fq.query_batch_pipelined(ctx, ix, batch, qall, [j[0] for j in jobs_in[:3]], float(HUMAN), 1000, 25, depth=2)):
        same(fq.query_batch(ctx, ix, batch, qall, jobs_in[k][0], float(HUMAN), 1000, 25), got)
    # a job nobody waits for is collected when the object goes away; a failing job reports through wait
    j = fq.query_batch_submit(ctx, ix, batch, qall, base, float(HUMAN), 1000, 25)
    del j
    bad = fq.query_batch_submit(ctx, ix, batch, qall, [(len(Qs) + 7, np.array([0, 1, 2], np.uint32), None)], float(HUMAN), 1000, 25)      # no such query structure
    with pytest.raises(fd.FdgpuError):
        bad.wait()
    with pytest.raises(RuntimeError):
        bad.wait()
    assert ctx.L.fdgpu_query_lanes(ctx.h, 5) == 6          # lanes are only added: five asked for, six there
    assert ctx.L.fdgpu_query_lanes(ctx.h, 7) == 7
    same(refs[0], fq.query_batch_submit(ctx, ix, batch, qall, jobs_in[0][0], float(HUMAN), 1000, 25).wait())


def test_whole_structure_query_at_human_scale(human, monkeypatch):
    """configs[4] (no -q): a ~300-residue database structure as the query, against all 20,500 structures"""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    ctx, ps, batch, ix, oix = human["ctx"], human["ps"], human["batch"], human["ix"], human["oix"]
    nres = human["nres"].astype(np.int64)
    s = int(np.nonzero((nres >= 295) & (nres <= 305))[0][3])
    a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
    item = dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b])
    qb = ctx.upload(fd.PackedStructures.concat([item]))
    qm = fq.make_query_map(ctx, qb, np.arange(b - a, dtype=np.uint32), None, ix, float(HUMAN))
    oq = _ostruct(ps, s)
    om = oracle.make_query_map(oq, "", oix, float(HUMAN))
    oa = om.arrays()
    assert len(qm.hash) > 50000
    assert np.array_equal(qm.hash, oa["hash"]) and np.array_equal(qm.qi, oa["qi"]) and np.array_equal(qm.qj, oa["qj"])
    assert np.array_equal(qm.idf.view(np.uint32), oa["idf"].view(np.uint32))
    recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, human["pen"], total_structures=HUMAN, as_array=True)
    want = oracle.count_query(om, oix, human["nres"])
    assert len(recs) == len(want) and len(recs) > HUMAN // 2
    assert [(int(r["nid"]), int(r["total_match_count"]), int(r["node_count"]), int(r["edge_count"])) for r in recs] == \
           [(w["nid"], w["total_match_count"], w["node_count"], w["edge_count"]) for w in want]
    assert np.allclose(recs["idf"], [w["idf"] for w in want], rtol=1e-5, atol=0)
    cand = fdist.rank_hits(recs, 20)["nid"].astype(np.uint32)
    assert int(cand[0]) == s
    # the same prefilter through the fused entry point (rows sliced at node boundaries, selection and ranking on the device)
    top = fd.count_query_maps(ctx, ix, [qm], human["pen"], total_structures=HUMAN, top_n=1000)[0]
    assert top.tobytes() == fdist.rank_hits(recs, 1000).tobytes()
    # ... which scores a query of this size per (tile of structures, slice of the rows) in LDS (k_qt_score<BIG>: sums per slice, reduction,
    # survivors' row bits by a second decode) — against the occupancy-row path (FDGPU_QTILE=0) and the ranked full list, other cuts included
    for N in (1, 20, 3000):
        monkeypatch.setenv("FDGPU_QTILE", "1")
        t1 = fd.count_query_maps(ctx, ix, [qm], human["pen"], total_structures=HUMAN, top_n=N)[0]
        monkeypatch.setenv("FDGPU_QTILE", "0")
        t0 = fd.count_query_maps(ctx, ix, [qm], human["pen"], total_structures=HUMAN, top_n=N)[0]
        assert t1.tobytes() == t0.tobytes() == fdist.rank_hits(recs, N).tobytes(), N
    monkeypatch.delenv("FDGPU_QTILE")
    got = fq.retrieve(ctx, batch, None, cand, qm, qb)
    # the first six candidates against oracle.retrieve (each is ~10 s of CPU for a 300-residue query against a 2,000-residue chain: all twenty
    # were 200 of the suite's 700 s); the other fourteen are retrieved on the GPU all the same and must each give a match
    n = _check_matches([g for g in got if g["cand"] < 6], cand[:6], ps, oq, om)
    assert n >= 6 and len({g["cand"] for g in got}) == 20
    assert max(sum(1 for x in g["processed"] if x >= 0) for g in got) == b - a     # the structure matches itself entirely


def test_tiled_scoring_equals_occupancy_rows(human, monkeypatch):
    """The tiled scoring of motif batches (k_qtile.hip: posting lists entered at the index's checkpoints, scores in LDS, records for the
    survivors by a second decode) against the occupancy-row path (FDGPU_QTILE=0) and against the ranked full list, byte for byte: two tiles
    of structures, queries of 1 to 1,000 rows that mix the longest lists of the index (entered mid-way, many 1 KB steps per wavefront) with
    short ones (decoded whole by every tile), more survivors than one round of row bits holds, a sub-range of the ids (first_id > 0,
    ids beyond the range skipped like count_query.rs:133 skips them) and the threshold bin's second level (ties by construction)."""
    import folddisco_amd as fd
    from folddisco_amd.dist import rank_hits
    ctx, ix, pen = human["ctx"], human["ix"], human["pen"]
    v, h, o = human["export"]
    rng = np.random.Generator(np.random.PCG64(2026))
    lens_b = np.diff(o.astype(np.int64))
    longest = np.argsort(lens_b)[-400:]
    queries = []
    for n in (1, 5, 44, 44, 200, 616, 1000):
        k_long = min(n // 2, 40)
        pick = np.concatenate([rng.choice(longest, size=k_long, replace=False), rng.choice(len(h), size=n - k_long, replace=False)])
        qh = np.unique(h[pick]).astype(np.uint32)
        queries.append((qh, rng.integers(0, 6, size=len(qh)).astype(np.uint32), rng.integers(0, 6, size=len(qh)).astype(np.uint32)))
    queries.insert(2, (np.zeros(0, np.uint32),) * 3)
    queries.append((np.array([0x3fffffff], np.uint32), np.zeros(1, np.uint32), np.ones(1, np.uint32)))      # absent hash only

    def run(index, penalty, total, top_n, tiled, tile="13"):
        monkeypatch.setenv("FDGPU_QTILE", "1" if tiled else "0")
        monkeypatch.setenv("FDGPU_QT_TILE", tile)
        return fd.count_query_batch(ctx, index, queries, penalty, total_structures=total, top_n=top_n)

    full = run(ix, pen, HUMAN, 0, False)
    assert max(len(f) for f in full) > 15000          # most of the tiles touched
    for N in (1, 20, 1000, 3000):
        a, a14, b = run(ix, pen, HUMAN, N, True), run(ix, pen, HUMAN, N, True, "14"), run(ix, pen, HUMAN, N, False)
        for f, x, x14, y in zip(full, a, a14, b):
            assert x.tobytes() == x14.tobytes() == y.tobytes() == rank_hits(f, N).tobytes(), N
    # query maps made against the index carry their lists' positions and lengths: pass B then reads the decoded stream pass A left
    # (k_qt_rows) instead of decoding the lists again — both forms, both tile sizes, against the ranked full list
    from folddisco_amd import query as fq
    Qs = human["queries"]
    qall = ctx.upload(fd.PackedStructures.concat([Q["q"].as_item() for Q in Qs]))
    qms = fq.make_query_maps(ctx, qall, [(k, Q["idx"], Q["subs"]) for k, Q in enumerate(Qs)], ix, float(HUMAN))
    monkeypatch.setenv("FDGPU_QTILE", "0")
    full_m = fd.count_query_maps(ctx, ix, qms, pen, total_structures=HUMAN, top_n=0)
    monkeypatch.setenv("FDGPU_QTILE", "1")
    monkeypatch.setenv("FDGPU_QT32", "0")
    for N in (5, 1000):
        for tile in ("13", "14"):
            for stream in ("1", "0"):
                monkeypatch.setenv("FDGPU_QT_TILE", tile); monkeypatch.setenv("FDGPU_QT_STREAM", stream)
                got = fd.count_query_maps(ctx, ix, qms, pen, total_structures=HUMAN, top_n=N)
                for f, x in zip(full_m, got):
                    assert x.tobytes() == rank_hits(f, N).tobytes(), (N, tile, stream)
    monkeypatch.delenv("FDGPU_QT_STREAM")
    # ... and the default since round 6: 32-bit sums over a planned slot stream (k_qscore32.hip; tiles of 2^14 or 2^15 structures), every cut
    # incl. the ones that take the whole tile's keys (more than the tile holds), ties around the cut, a shard's sub-range of the ids
    for mode in ("13", "14", "15"):
        monkeypatch.setenv("FDGPU_QT32", mode)
        for N in (1, 5, 1000, 3000):
            got = fd.count_query_maps(ctx, ix, qms, pen, total_structures=HUMAN, top_n=N)
            for f, x in zip(full_m, got):
                assert x.tobytes() == rank_hits(f, N).tobytes(), (N, mode)
    # a decoded stream too small for the batch (FDGPU_QT_STREAM_CAP: the bound the host sizes it by, cut down): the tiles that do not fit are not scored,
    # their queries raise the selection's overflow flag and the call is ranked by the compacting path — same records
    monkeypatch.setenv("FDGPU_QT32", "14")
    for cap_recs in ("64", "20000"):
        monkeypatch.setenv("FDGPU_QT_STREAM_CAP", cap_recs)
        got = fd.count_query_maps(ctx, ix, qms, pen, total_structures=HUMAN, top_n=1000)
        for f, x in zip(full_m, got):
            assert x.tobytes() == rank_hits(f, 1000).tobytes(), cap_recs
    monkeypatch.delenv("FDGPU_QT_STREAM_CAP")
    monkeypatch.setenv("FDGPU_QTILE", "0")
    ones = np.ones_like(pen)
    full_1 = fd.count_query_maps(ctx, ix, qms, ones, total_structures=HUMAN, top_n=0)
    sub_m = fd.FolddiscoIndex.load(ctx, h, o, v, 17000, first_id=3000)
    qms_s = fq.make_query_maps(ctx, qall, [(k, Q["idx"], Q["subs"]) for k, Q in enumerate(Qs)], sub_m, float(HUMAN))      # (a map remembers ITS index's lists)
    full_sm = fd.count_query_maps(ctx, sub_m, qms_s, pen[3000:20000].copy(), total_structures=HUMAN, top_n=0)
    assert all((f["nid"] >= 3000).all() and (f["nid"] < 20000).all() for f in full_sm if len(f))
    monkeypatch.setenv("FDGPU_QTILE", "1")
    for mode in ("13", "14", "15"):
        monkeypatch.setenv("FDGPU_QT32", mode)
        for N in (50, 1000):
            got = fd.count_query_maps(ctx, ix, qms, ones, total_structures=HUMAN, top_n=N)
            for f, x in zip(full_1, got):
                assert x.tobytes() == rank_hits(f, N).tobytes(), (N, mode, "ties")
            got = fd.count_query_maps(ctx, sub_m, qms_s, pen[3000:20000].copy(), total_structures=HUMAN, top_n=N)
            for f, x in zip(full_sm, got):
                assert x.tobytes() == rank_hits(f, N).tobytes(), (N, mode, "shard")
    monkeypatch.delenv("FDGPU_QT32")
    # ties: a penalty of 1 makes the key a function of the matched rows alone — thousands of equal keys around the cut
    one = np.ones_like(pen)
    full1 = run(ix, one, HUMAN, 0, False)
    for N in (50, 1000):
        a = run(ix, one, HUMAN, N, True)
        for f, x in zip(full1, a):
            assert x.tobytes() == rank_hits(f, N).tobytes(), N
    # a shard's view of the same bytes: ids [3000, 3000 + 17000) only
    sub = fd.FolddiscoIndex.load(ctx, h, o, v, 17000, first_id=3000)
    pen_s = pen[3000:20000].copy()
    full_s = run(sub, pen_s, HUMAN, 0, False)
    assert all((f["nid"] >= 3000).all() and (f["nid"] < 20000).all() for f in full_s if len(f))
    for N in (10, 1000):
        a = run(sub, pen_s, HUMAN, N, True)
        for f, x in zip(full_s, a):
            assert x.tobytes() == rank_hits(f, N).tobytes(), N
