"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI
(libfdgpu.so) and is compared with the CPU oracle on the same inputs.
Bit-exact: u32 hashes, posting bytes, offsets, counts, residue indices. RMSD: |d| <= 1e-4. idf: rel 1e-5."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from tests.helpers import (Q4CHA, SER, oracle_structs_to_packed, packed_to_oracle_structs, synthetic_packed)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import folddisco_amd as fd
    c = fd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ser(ctx):
    structs = [oracle.read_pdb(p) for p in SER]
    ps, std = oracle_structs_to_packed(structs)
    batch = ctx.upload(ps)
    return structs, ps, std, batch


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_device_libm_bit_exact(ctx):
    """gfx950 evaluation of csrc/fd_libm.h == host glibc, bit for bit, on 4M samples per function."""
    rng = np.random.Generator(np.random.PCG64(7))
    libm = C.CDLL("libm.so.6")
    n = 1 << 22
    def run(op, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        out = np.zeros_like(a)
        bp = None if b is None else np.ascontiguousarray(b, np.float32)
        ctx.check(ctx.L.fdgpu_debug_libm(ctx.h, op, a.ctypes.data_as(C.POINTER(C.c_float)),
                                         None if bp is None else bp.ctypes.data_as(C.POINTER(C.c_float)),
                                         out.ctypes.data_as(C.POINTER(C.c_float)), len(a)))
        return out
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, n - 8).astype(np.float32),
                          np.array([0, -0.0, np.pi, -np.pi, 1e-8, np.nan, 0.75, 0.7499999], np.float32)])
    for op, fn in ((0, np.sin), (1, np.cos)):
        cfn = getattr(libm, "sinf" if op == 0 else "cosf")
        cfn.restype = C.c_float; cfn.argtypes = [C.c_float]
        got = run(op, ang)
        idx = rng.integers(0, n, 200000)
        ref = np.array([cfn(float(x)) for x in ang[idx]], np.float32)
        ok = (_bits(got[idx]) == _bits(ref)) | (np.isnan(got[idx]) & np.isnan(ref))
        assert ok.all()
    x = np.concatenate([rng.uniform(-1.0000001, 1.0000001, n - 4).astype(np.float32), np.array([1, -1, 0, np.nan], np.float32)])
    libm.acosf.restype = C.c_float; libm.acosf.argtypes = [C.c_float]
    got = run(2, x)
    idx = rng.integers(0, n, 200000)
    ref = np.array([libm.acosf(float(v)) for v in x[idx]], np.float32)
    assert ((_bits(got[idx]) == _bits(ref)) | (np.isnan(got[idx]) & np.isnan(ref))).all()
    y = rng.uniform(-1, 1, n).astype(np.float32)
    z = rng.uniform(-1, 1, n).astype(np.float32)
    y[:4] = [0, -0.0, 1e-30, np.nan]
    libm.atan2f.restype = C.c_float; libm.atan2f.argtypes = [C.c_float, C.c_float]
    got = run(4, y, z)
    idx = np.concatenate([np.arange(8), rng.integers(0, n, 200000)])
    ref = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y[idx], z[idx])], np.float32)
    assert ((_bits(got[idx]) == _bits(ref)) | (np.isnan(got[idx]) & np.isnan(ref))).all()


def test_hash_raw_order_serine(ctx, ser):
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    h, off = fd.get_geometric_hash_as_u32(ctx, batch, sort_dedup=False)
    assert list(np.diff(off.astype(np.int64))) == [71946, 68430, 74664, 24148, 57974]
    for s, st in enumerate(structs):
        ref = oracle.hash_structure(st)
        assert np.array_equal(h[int(off[s]):int(off[s + 1])], ref), f"structure {s}"


def test_hash_sorted_unique_serine(ctx, ser):
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    h, off = fd.get_geometric_hash_as_u32(ctx, batch, sort_dedup=True)
    assert list(np.diff(off.astype(np.int64))) == [47512, 67129, 46795, 23853, 40385]
    for s, st in enumerate(structs):
        assert np.array_equal(h[int(off[s]):int(off[s + 1])], np.unique(oracle.hash_structure(st)))


def test_index_build_serine_byte_identical(ctx, ser, tmp_path):
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    ix = fd.FolddiscoIndex.build(ctx, batch)
    assert ix.num_hashes == 217612 and ix.value_len == 225674 and ix.num_postings == 225674
    v, h, o = ix.export()
    oix, nres, plddt = oracle.build_index(structs)
    assert np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
    # on-disk files byte-identical to the reference format written by the oracle
    ix.save(str(tmp_path / "gpu"))
    oix.save(str(tmp_path / "cpu"))
    for ext in ("", ".offset"):
        assert open(str(tmp_path / "gpu") + ext, "rb").read() == open(str(tmp_path / "cpu") + ext, "rb").read()


@pytest.mark.parametrize("n_struct,seed", [(1, 3), (37, 11), (160, 5)])
def test_index_build_synthetic(ctx, n_struct, seed):
    import folddisco_amd as fd
    ps = synthetic_packed(n_struct, seed)
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    v, h, o = ix.export()
    structs = packed_to_oracle_structs(ps)
    oix, _, _ = oracle.build_index(structs)
    assert np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())


def test_index_edge_cases(ctx):
    import folddisco_amd as fd
    # empty batch, empty structures, unknown residues, missing CB, multi-byte varints via first_id
    ps = fd.PackedStructures(np.zeros(1, np.uint64), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.uint8))
    ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps))
    v, h, o = ix.export()
    assert len(v) == 0 and len(h) == 0 and list(o) == [0]
    base = synthetic_packed(6, 21, lengths=np.array([40, 64, 65, 128, 1, 200]))
    aa = base.aa.copy(); aa[5] = 255; aa[70] = 255
    cbv = np.ones(len(aa), np.uint8); cbv[10] = 0; cbv[150] = 0
    # insert an empty structure in the middle (skipped/oversize structures keep their id, mod.rs:313-318)
    off = base.res_off.astype(np.int64)
    off2 = np.concatenate([off[:3], off[2:3], off[3:]]).astype(np.uint64)
    ps = fd.PackedStructures(off2, base.n_xyz, base.ca_xyz, base.cb_xyz, aa, cbv)
    first_id = 16380  # ids straddle the 1->2->3 byte varint boundaries (16384)
    ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=first_id)
    v, h, o = ix.export()
    structs = packed_to_oracle_structs(ps)
    lists = [np.unique(oracle.hash_structure(s)) for s in structs]
    L = oracle.lib()
    oix = L.fdo_index_new(30)
    for fn in (L.fdo_index_count_single_entry, L.fdo_index_add_single_entry):
        for s, lst in enumerate(lists):
            for x in lst:
                fn(oix, int(x), first_id + s)
        if fn is L.fdo_index_count_single_entry:
            L.fdo_index_allocate_entries(oix)
    L.fdo_index_finish(oix)
    oi = oracle.OIndex(oix)
    assert np.array_equal(h, oi.hashes()) and np.array_equal(o, oi.offsets()) and np.array_equal(v, oi.values())
    assert len(lists[2]) == 0 and len(lists[5]) == 0  # the inserted empty structure and the 1-residue one


def _query_arrays(m):
    a = m.arrays()
    return a["hash"], a["qi"].astype(np.uint32), a["qj"].astype(np.uint32)


def test_posting_lengths_and_count_query_serine(ctx, ser):
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    ix = fd.FolddiscoIndex.build(ctx, batch)
    oix, nres, plddt = oracle.build_index(structs)
    q = oracle.read_pdb(Q4CHA)
    m = oracle.make_query_map(q, "B57,B102,C195", oix, float(len(structs)))
    qh, qi, qj = _query_arrays(m)
    lens = ix.posting_lengths(qh)
    assert list(lens) == [len(oix.entries(int(x))) for x in qh]
    absent = ix.posting_lengths(np.array([0, 1, 2 ** 30 - 1], np.uint32))
    assert list(absent) == [0, 0, 0]
    pen = fd.length_penalty(nres, 0.5)
    got = fd.count_query(ctx, ix, qh, qi, qj, pen)
    ref = oracle.count_query(m, oix, nres)
    assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in got] == \
           [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in ref]
    for g, r in zip(got, ref):
        assert abs(g["idf"] - r["idf"]) <= 1e-5 * max(abs(r["idf"]), 1e-30)
    # README.md:237-241 directly
    assert {(r["nid"], "%.4f" % r["idf"]) for r in got} == {(4, "0.6138"), (3, "0.4869"), (1, "0.0617"), (2, "0.0584"), (0, "0.1856")}


def test_count_query_synthetic_long_postings(ctx):
    """lists with multi-byte deltas and > 64-byte blocks: whole-structure query of structure 0 against 300 structures."""
    import folddisco_amd as fd
    ps = synthetic_packed(300, 9, lengths=np.full(300, 60))
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch, first_id=0)
    structs = packed_to_oracle_structs(ps)
    oix, nres, _ = oracle.build_index(structs)
    m = oracle.make_query_map(structs[0], "", oix, 300.0)
    qh, qi, qj = _query_arrays(m)
    assert list(ix.posting_lengths(qh)) == [len(oix.entries(int(x))) for x in qh]
    got = fd.count_query(ctx, ix, qh, qi, qj, fd.length_penalty(nres))
    ref = oracle.count_query(m, oix, nres)
    assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in got] == \
           [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in ref]
    for g, r in zip(got, ref):
        assert abs(g["idf"] - r["idf"]) <= 1e-5 * max(abs(r["idf"]), 1e-30)


def test_match_pairs_and_kabsch_serine(ctx, ser):
    from folddisco_amd import match
    structs, ps, std, batch = ser
    oix, nres, _ = oracle.build_index(structs)
    q = oracle.read_pdb(Q4CHA)
    m = oracle.make_query_map(q, "B57,B102,C195", oix, float(len(structs)))
    a = m.arrays()
    for ca_cut in (1.0, 1.5):
        found, cands = match.match_pairs(ctx, batch, std, np.arange(5, dtype=np.uint32), a, ca_distance_cutoff=ca_cut)
        for nid, t in enumerate(structs):
            R = oracle.retrieve(t, q, m, ca_distance_cutoff=ca_cut)
            f = found[found[:, 0] == nid][:, 1:]
            c = cands[cands[:, 0] == nid][:, 1:]
            assert np.array_equal(f.astype(np.uint64), R["found"]), (nid, ca_cut)
            assert np.array_equal(c.astype(np.uint64), R["cand"]), (nid, ca_cut)
    # Kabsch on the reference's own triads (structure/kabsch.rs:563-616) + the oracle's results
    src = np.array([[6.994, 8.354, 42.405], [9.429, 7.479, 48.266], [5.547, 0.158, 42.050]], np.float32)
    t1 = np.array([[-13.958, -1.741, -4.223], [-12.833, 3.134, -7.780], [-5.720, -2.218, -3.368]], np.float32)
    t2 = np.array([[-4.924, 5.813, -9.485], [-0.499, 10.073, -8.059], [-0.792, 0.658, -4.430]], np.float32)
    xs = np.concatenate([t1, t2, src]); ys = np.concatenate([src, src, src])
    rmsd, rot, tran = match.kabsch_batch(ctx, xs, ys, np.array([0, 3, 6, 9], np.uint64))
    for k, (x, y) in enumerate(((t1, src), (t2, src), (src, src))):
        r, R, T = oracle.kabsch(x, y)
        assert abs(rmsd[k] - r) <= 1e-4 and np.allclose(rot[k], R, atol=1e-4) and np.allclose(tran[k], T, atol=1e-3)
    assert rmsd[0] < 0.2 and rmsd[1] < 0.2 and rmsd[2] < 1e-6


def test_index_build_medium_multitile(ctx):
    """~20M postings: many radix tiles, look-back chains across XCDs, multi-tile encode; byte-identical to the oracle."""
    import folddisco_amd as fd
    ps = synthetic_packed(600, 77)
    batch = ctx.upload(ps)
    structs = packed_to_oracle_structs(ps)
    h, off = oracle.hash_batch(structs)
    oix = oracle.build_index_from_lists_mt(h, off, 8)
    for first_id in (0, 1 << 20):
        ix = fd.FolddiscoIndex.build(ctx, batch, first_id=first_id)
        v, hh, o = ix.export()
        if first_id == 0:
            assert np.array_equal(hh, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
        else:
            assert np.array_equal(hh, oix.hashes())  # same hashes, longer varints
            assert ix.num_postings == len(h)
    # per-structure API agrees as well
    gh, goff = fd.get_geometric_hash_as_u32(ctx, batch, sort_dedup=True)
    assert np.array_equal(goff, off) and np.array_equal(gh, h)


@pytest.mark.gpu
def test_count_query_batch_equals_single(ctx):
    """fdgpu_count_query_batch == fdgpu_count_query per query (same records, same order), incl. an empty query and
    a query with only absent hashes."""
    import folddisco_amd as fd
    from folddisco_amd import synth
    ps = synth.to_packed(synth.generate(300, seed=77))
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch, first_id=5)
    hashes, _, _ = ix.export()[1], None, None
    rng = np.random.Generator(np.random.PCG64(5))
    nres = np.diff(ps.res_off).astype(np.uint64)
    pen = fd.length_penalty(nres, 0.5)
    queries = []
    for t in range(9):
        n = int(rng.integers(1, 40))
        qh = rng.choice(hashes, size=n, replace=False).astype(np.uint32)
        qi = rng.integers(0, 5, size=n).astype(np.uint32)
        qj = rng.integers(0, 5, size=n).astype(np.uint32)
        queries.append((qh, qi, qj))
    queries.insert(3, (np.zeros(0, np.uint32),) * 3)
    queries.insert(6, (np.array([0x3fffffff, 0x3ffffffe], np.uint32), np.array([0, 1], np.uint32), np.array([1, 0], np.uint32)))
    got = fd.count_query_batch(ctx, ix, queries, pen, total_structures=300)
    assert len(got) == len(queries)
    for (qh, qi, qj), g in zip(queries, got):
        want = fd.count_query(ctx, ix, qh, qi, qj, pen, total_structures=300, as_array=True)
        assert g.tobytes() == want.tobytes()
    assert len(got[3]) == 0 and len(got[6]) == 0
    # device-side selection of the top N (k_topn_* + k_topn_sort): exactly the first N of the ranked full list, in rank order
    from folddisco_amd.dist import rank_hits
    for N in (1, 3, 10, 50, 1000):
        sel = fd.count_query_batch(ctx, ix, queries, pen, total_structures=300, top_n=N)
        for full, short in zip(got, sel):
            assert len(short) <= len(full)
            assert short.tobytes() == rank_hits(full, N).tobytes()       # ranked and cut on the device (k_topn_sort) / in the library
    big = fd.count_query_batch(ctx, ix, queries, pen, total_structures=300, top_n=3)
    assert any(len(b) < len(f) for b, f in zip(big, got))          # the preselection did drop records


@pytest.mark.gpu
def test_speculative_torsions_equal_exact_path(ctx, monkeypatch):
    """The index build evaluates the torsion fields speculatively (rsq normalisations, sign-of-(|y| - t|x|) segment
    decisions with error margins) and re-evaluates undecided pairs exactly: the index must be byte-identical to the one
    built with FDGPU_EXACT=1, and the fallback must actually be exercised."""
    import folddisco_amd as fd
    from folddisco_amd import synth
    ps = synth.to_packed(synth.generate(1500, seed=99))
    batch = ctx.upload(ps)
    ctx.spec_fallbacks()
    monkeypatch.delenv("FDGPU_EXACT", raising=False)
    spec = fd.FolddiscoIndex.build(ctx, batch).export()
    n_fallback = ctx.spec_fallbacks()
    monkeypatch.setenv("FDGPU_EXACT", "1")
    exact = fd.FolddiscoIndex.build(ctx, batch).export()
    assert ctx.spec_fallbacks() == 0
    for a, b in zip(spec, exact):
        assert a.tobytes() == b.tobytes()
    assert n_fallback > 0          # ~2.4e-4 of ~2.4e7 pairs


@pytest.mark.gpu
def test_get_entries_decodes_posting_lists(ctx):
    """fdgpu_get_entries == the oracle's get_entries for present and absent hashes, short and long lists (lists longer than one
    64-byte decode block, multi-byte varints through first_id), and matches posting_lengths"""
    import folddisco_amd as fd
    ps = synthetic_packed(400, 21)
    batch = ctx.upload(ps)
    first_id = 70000                                # three-byte absolute ids, two-byte deltas never (deltas < 400)
    ix = fd.FolddiscoIndex.build(ctx, batch, first_id=first_id)
    v, h, o = ix.export()
    structs = packed_to_oracle_structs(ps)
    oix, _, _ = oracle.build_index(structs)
    lens = np.diff(o.astype(np.int64))
    longest = h[np.argsort(lens)[-5:]]
    rng = np.random.Generator(np.random.PCG64(3))
    qh = np.concatenate([longest, rng.choice(h, 200, replace=False), np.array([0x3fffffff, 1], np.uint32)]).astype(np.uint32)
    got = ix.get_entries(qh)
    want_len = ix.posting_lengths(qh)
    assert max(len(g) for g in got) > 64
    for hh, g, n in zip(qh, got, want_len):
        w = oix.entries(int(hh))
        assert len(g) == int(n) == len(w)
        assert np.array_equal(g.astype(np.int64) - first_id, w.astype(np.int64))


@pytest.mark.gpu
def test_both_sort_element_forms_give_the_same_index(ctx, monkeypatch):
    """6-byte sort elements (hash << 2 | id bits 17:16 with a u16 payload; shards of <= 2^18 structures) and the 8-byte form
    (u32 hash, u32 id; FDGPU_IDS32=1 or larger shards) build byte-identical indices, both equal to the oracle's"""
    import folddisco_amd as fd
    ps = synthetic_packed(120, 8)
    batch = ctx.upload(ps)
    monkeypatch.delenv("FDGPU_IDS32", raising=False)
    a = fd.FolddiscoIndex.build(ctx, batch, first_id=3).export()
    monkeypatch.setenv("FDGPU_IDS32", "1")
    b = fd.FolddiscoIndex.build(ctx, batch, first_id=3).export()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    oix, _, _ = oracle.build_index(packed_to_oracle_structs(ps))
    assert np.array_equal(a[1], oix.hashes()) and np.array_equal(a[2], oix.offsets())


@pytest.mark.gpu
def test_index_set_equals_single_index(ctx):
    """A shard built as several fdgpu_index_build calls (FolddiscoIndexSet: one resident sub-index per chunk of structures, the
    way a shard of more than 2^32 residue pairs is held) answers like the single index: posting lengths, decoded lists, scoring
    records byte for byte, and its merged export is the single index's."""
    import folddisco_amd as fd
    from folddisco_amd import synth
    d = synth.generate(180, seed=31)
    ps = synth.to_packed(d)
    off = ps.res_off.astype(np.int64)
    cuts = [0, 50, 51, 130, 180]                    # uneven chunks, one of a single structure
    chunks = []
    for a, b in zip(cuts, cuts[1:]):
        sl = slice(off[a], off[b])
        chunks.append(fd.PackedStructures((ps.res_off[a:b + 1] - ps.res_off[a]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl]))
    single = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=7)
    iset = fd.FolddiscoIndexSet.build(ctx, [ctx.upload(c) for c in chunks], first_id=7)
    assert iset.n_structures == 180 and iset.num_postings == single.num_postings
    v, h, o = single.export()
    mv, mh, mo = iset.export_merged()
    assert np.array_equal(mv, v) and np.array_equal(mh, h) and np.array_equal(mo, o)
    rng = np.random.Generator(np.random.PCG64(9))
    qh = np.concatenate([rng.choice(h, 60, replace=False), np.array([0x3ffffffe], np.uint32)]).astype(np.uint32)
    assert np.array_equal(iset.posting_lengths(qh), single.posting_lengths(qh))
    for a, b in zip(iset.get_entries(qh), single.get_entries(qh)):
        assert np.array_equal(a, b)
    qi = rng.integers(0, 4, len(qh)).astype(np.uint32)
    qj = rng.integers(0, 4, len(qh)).astype(np.uint32)
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    want = fd.count_query(ctx, single, qh, qi, qj, pen, as_array=True)
    got = fd.count_query_set(ctx, iset, qh, qi, qj, pen)
    assert got.tobytes() == want.tobytes() and len(got) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("first_id,cuts", [(7, [0, 50, 51, 130, 180]), (70000, [0, 1, 90, 90, 180]), (2100000, [0, 180]), (16380, [0, 3, 6, 60, 61, 62, 100, 180])])
def test_device_merge_equals_single_index(ctx, first_id, cuts):
    """fdgpu_index_merge: the parts' posting lists concatenated per hash on the device (first varint of every continuation
    re-based to a delta) == the index one build over all the structures produces, byte for byte — with hashes absent from
    some parts, an empty part, a single-part merge, and first ids whose absolute / delta varints differ in length (1, 2, 3 and
    4 bytes)."""
    import folddisco_amd as fd
    from folddisco_amd import synth
    ps = synth.to_packed(synth.generate(180, seed=31))
    off = ps.res_off.astype(np.int64)
    chunks = []
    for a, b in zip(cuts, cuts[1:]):
        sl = slice(off[a], off[b])
        chunks.append(fd.PackedStructures((ps.res_off[a:b + 1] - ps.res_off[a]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl]))
    single = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=first_id)
    iset = fd.FolddiscoIndexSet.build(ctx, [ctx.upload(c) for c in chunks], first_id=first_id)
    merged = iset.merge()
    v, h, o = single.export()
    mv, mh, mo = merged.export()
    assert merged.num_hashes == single.num_hashes and merged.value_len == single.value_len and merged.num_postings == single.num_postings
    assert np.array_equal(mh, h) and np.array_equal(mo, o) and np.array_equal(mv, v)
    if len(cuts) > 2:
        hv, hh, ho = iset.export_merged()           # the host-side merge agrees too
        assert np.array_equal(hv, v) and np.array_equal(hh, h) and np.array_equal(ho, o)
    # the merged index answers queries like the single one
    rng = np.random.Generator(np.random.PCG64(5))
    qh = rng.choice(h, 40, replace=False).astype(np.uint32)
    qi = rng.integers(0, 4, len(qh)).astype(np.uint32)
    qj = rng.integers(0, 4, len(qh)).astype(np.uint32)
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    assert fd.count_query(ctx, merged, qh, qi, qj, pen, as_array=True).tobytes() == fd.count_query(ctx, single, qh, qi, qj, pen, as_array=True).tobytes()


@pytest.mark.gpu
def test_device_merge_long_lists_and_wide_hashes(ctx):
    """posting lists of thousands of ids (one repeated structure: every hash in every structure, lists longer than the 64-byte
    decode block, parts of uneven size) and the 2^32 hash space (a part with an overflowed hash, test_degenerate_geometry)"""
    import folddisco_amd as fd
    from folddisco_amd import synth
    one = synth.to_packed(synth.generate(1, seed=77, lengths=np.array([60])))
    n = 1500
    item = dict(n_xyz=one.n_xyz, ca_xyz=one.ca_xyz, cb_xyz=one.cb_xyz, aa=one.aa)
    far = dict(n_xyz=one.n_xyz.copy(), ca_xyz=one.ca_xyz.copy(), cb_xyz=one.cb_xyz.copy(), aa=one.aa)
    far["cb_xyz"][7] = np.float32(3.0e38)               # CB-CB distance overflows to inf -> hash beyond 30 bits
    items = [item] * 700 + [far] + [item] * (n - 701)
    ps = fd.PackedStructures.concat(items)
    single = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=100)
    cuts = [0, 100, 699, 702, 1200, n]
    parts, fid = [], 100
    for a, b in zip(cuts, cuts[1:]):
        parts.append(fd.FolddiscoIndex.build(ctx, ctx.upload(fd.PackedStructures.concat(items[a:b])), first_id=fid))
        fid += b - a
    merged = fd.FolddiscoIndexSet(parts).merge()
    v, h, o = single.export()
    mv, mh, mo = merged.export()
    assert h.max() >= (1 << 30) and np.diff(o).max() > 1400
    assert np.array_equal(mh, h) and np.array_equal(mo, o) and np.array_equal(mv, v)


@pytest.mark.gpu
def test_degenerate_geometry_matches_oracle(ctx):
    """Bit-exactness where the arithmetic degenerates: CB on top of CA (zero-length vectors -> NaN angles), duplicated residues
    (distance 0), collinear N-CA-CB (zero cross products), residues on a perfect line / lattice (torsion operands exactly 0 ->
    the atan2 special cases), huge and non-finite coordinates.  The raw hash lists (S1, generic chain) and the index (frames,
    tables, speculative path with its exact fallback) must both equal the oracle."""
    import folddisco_amd as fd
    rng = np.random.Generator(np.random.PCG64(123))
    items = []

    def add(n_xyz, ca_xyz, cb_xyz, aa=None):
        n = len(ca_xyz)
        items.append(dict(n_xyz=np.asarray(n_xyz, np.float32), ca_xyz=np.asarray(ca_xyz, np.float32), cb_xyz=np.asarray(cb_xyz, np.float32),
                          aa=np.asarray(aa if aa is not None else rng.integers(0, 20, n), np.uint8)))
    base = rng.normal(0, 6, (24, 3)).astype(np.float32)
    off = lambda s: (base + rng.normal(0, s, base.shape)).astype(np.float32)
    add(off(1.0), base, base.copy())                                   # CB == CA
    add(base + 1.0, base, base + np.float32(2.0))                      # N, CA, CB collinear
    dup = np.repeat(base[:8], 3, axis=0)
    add(dup + rng.normal(0, 1, dup.shape), dup, dup + rng.normal(0, 1, dup.shape))     # duplicated CA positions (distance 0)
    line = np.stack([np.arange(20) * 3.8, np.zeros(20), np.zeros(20)], axis=1).astype(np.float32)
    add(line + [0, 1.46, 0], line, line + [0, 0, 1.53])                # perfect line, parallel frames: torsion operands exactly 0
    g = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3)), -1).reshape(-1, 3).astype(np.float32) * 4.0
    add(g + [1, 0, 0], g, g + [0, 1, 0])                               # cubic lattice
    big = off(1.0) * np.float32(1e7)
    add(big + 1.0, big, big - 1.0)                                     # huge coordinates (all pairs beyond the cutoff or absorbed)
    weird = base.copy(); wn = off(1.0); wb = off(1.0)
    weird[3, 0] = np.nan; wn[5, 1] = np.inf; wb[7, 2] = -np.inf; weird[9] = 0.0; wn[9] = 0.0; wb[9] = 0.0
    add(wn, weird, wb)                                                 # NaN / inf coordinates, an all-zero residue
    tiny = (base * np.float32(1e-20)).astype(np.float32)
    add(tiny + np.float32(1e-21), tiny, tiny - np.float32(1e-21))      # denormal-scale differences
    ps = fd.PackedStructures.concat(items)
    batch = ctx.upload(ps)
    structs = packed_to_oracle_structs(ps)
    raw, roff = fd.get_geometric_hash_as_u32(ctx, batch, sort_dedup=False)
    for s, st in enumerate(structs):
        want = oracle.hash_structure(st)
        assert np.array_equal(raw[int(roff[s]):int(roff[s + 1])], want), f"structure {s}"
    # index of everything: an infinite distance saturates the quantiser and the unmasked OR sets hash bits 30-31 — beyond the
    # reference's 2^30-entry table (it would panic there), so this part is checked against the S1 lists: the build must keep
    # the full u32 (it falls back from 6-byte to 8-byte sort elements for such a shard)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    v, h, o = ix.export()
    assert raw.max() >= (1 << 30)
    assert np.array_equal(h, np.unique(raw))
    per_hash = [np.unique(np.repeat(np.arange(len(structs)), np.diff(roff).astype(np.int64))[raw == hh]) for hh in (h[0], h[len(h) // 2], h[-1])]
    for hh, want_ids in zip((h[0], h[len(h) // 2], h[-1]), per_hash):
        assert np.array_equal(ix.get_entries(np.array([hh], np.uint32))[0], want_ids)
    # index of the finite structures against the oracle's index, byte for byte
    keep = [k for k in range(len(items)) if k != 6]
    ps2 = fd.PackedStructures.concat([items[k] for k in keep])
    v2, h2, o2 = fd.FolddiscoIndex.build(ctx, ctx.upload(ps2)).export()
    oix, _, _ = oracle.build_index(packed_to_oracle_structs(ps2))
    assert np.array_equal(h2, oix.hashes()) and np.array_equal(o2, oix.offsets()) and np.array_equal(v2, oix.values())
    assert len(h2) > 500


@pytest.mark.gpu
def test_random_unknown_residues_and_missing_cb(ctx):
    """randomly sprinkled unknown residue types (aa = 255) and residues without a CB (cb_valid = 0) — pairs with either are
    rejected by get_single_feature (controller/feature.rs:11-24) — plus structures of 0, 1 and 2 residues: hash lists and index
    equal the oracle's"""
    import folddisco_amd as fd
    from folddisco_amd import synth
    rng = np.random.Generator(np.random.PCG64(77))
    ps = synth.to_packed(synth.generate(70, seed=13, lengths=np.concatenate([[0, 1, 2, 2, 3], rng.integers(40, 180, 65)])))
    aa = ps.aa.copy()
    aa[rng.uniform(size=len(aa)) < 0.08] = 255
    cbv = (rng.uniform(size=len(aa)) >= 0.1).astype(np.uint8)
    ps = fd.PackedStructures(ps.res_off, ps.n_xyz, ps.ca_xyz, ps.cb_xyz, aa, cbv)
    batch = ctx.upload(ps)
    structs = packed_to_oracle_structs(ps)
    h, off = fd.get_geometric_hash_as_u32(ctx, batch, sort_dedup=False)
    for s, st in enumerate(structs):
        assert np.array_equal(h[int(off[s]):int(off[s + 1])], oracle.hash_structure(st)), s
    v, hh, oo = fd.FolddiscoIndex.build(ctx, batch).export()
    oix, _, _ = oracle.build_index(structs)
    assert np.array_equal(hh, oix.hashes()) and np.array_equal(oo, oix.offsets()) and np.array_equal(v, oix.values())


@pytest.mark.gpu
def test_lms_qcp_batch_matches_oracle(ctx):
    """--partial-fit superposition (src/structure/lms_qcp.rs): the wavefront-per-problem kernel takes the same discrete decisions
    as the CPU restatement (seed triple out of the 500 xorshift trials, joining order, stop) and the same numbers: core index
    lists identical, rms / rotation / translation bit-identical (f64 sums and f32 residuals in the reference's order)."""
    from folddisco_amd import match
    from tests.test_oracle_golden import _lms_problem
    rng = np.random.default_rng(5)
    probs = []
    for n, n_out, noise in ((3, 0, 0.0), (4, 1, 0.0), (5, 0, 0.3), (8, 2, 0.05), (16, 5, 0.1), (17, 0, 0.5), (32, 9, 0.0), (63, 20, 0.2),
                            (64, 3, 0.2), (65, 30, 0.02), (130, 40, 0.3), (400, 150, 0.1)):
        x, y, _ = _lms_problem(rng, n, n_out, noise)
        probs.append((x, y))
    line = np.outer(np.arange(9, dtype=np.float32), np.array([1.0, 2.0, -1.0], np.float32))
    probs.append((line, line + np.float32(1.0)))                     # collinear moving points: every trial is rejected, seed = 0,1,2
    pts = (rng.normal(size=(10, 3)) * 3).astype(np.float32)
    probs.append((pts, pts.copy()))                                  # identical sets
    dup = pts.copy(); dup[1:6] = dup[0]
    probs.append((dup, pts))                                         # repeated points: collinearity retries in the random stream
    xs = np.concatenate([p[0] for p in probs]); ys = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).astype(np.uint64)
    rmsd, rot, tran, cores = match.lms_qcp_batch(ctx, xs, ys, off)
    for k, (x, y) in enumerate(probs):
        r, R, T, core = oracle.lms_qcp(x, y)
        assert np.array_equal(cores[k].astype(np.uint64), core), (k, len(x), cores[k], core)
        if np.isnan(r):
            assert np.isnan(rmsd[k])
            continue
        assert _bits(rmsd[k]) == _bits(r), (k, len(x), rmsd[k], r)
        assert np.array_equal(_bits(rot[k]), _bits(R)) and np.array_equal(_bits(tran[k]), _bits(T)), (k, len(x))
    with pytest.raises(Exception):
        match.lms_qcp_batch(ctx, xs[:2], ys[:2], np.array([0, 2], np.uint64))     # the reference asserts >= 3 pairs


@pytest.mark.gpu
@pytest.mark.parametrize("htype,nbd,nba", [(0, 0, 0), (1, 0, 0), (7, 0, 0), (8, 0, 0), (0, 16, 8), (1, 12, 5), (7, 6, 20), (8, 40, 12), (3, 16, 0)])
def test_other_encodings_hash_index_query(ctx, ser, htype, nbd, nba):
    """The encodings that share the PDBTrRosetta descriptor (HashType 0 PDBMotif, 1 PDBMotifSinCos, 7 FolddiscoAngle, 8 FolddiscoDist,
    §8f rank 3): raw and sorted hash lists, the index bytes, posting lengths and count_query records equal the CPU restatement;
    (3, 16, 0) checks the reference's "either bin count 0 -> both default" rule on the default encoding."""
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    with oracle.hash_type(htype):
        h, off = fd.get_geometric_hash_as_u32(ctx, batch, nbin_dist=nbd, nbin_angle=nba, sort_dedup=False, hash_type=htype)
        for s, st in enumerate(structs):
            ref = oracle.hash_structure(st, nbin_dist=nbd, nbin_angle=nba)
            assert np.array_equal(h[int(off[s]):int(off[s + 1])], ref), f"structure {s}"
        ix = fd.FolddiscoIndex.build(ctx, batch, nbin_dist=nbd, nbin_angle=nba, hash_type=htype)
        oix, nres, plddt = oracle.build_index(structs, nbin_dist=nbd, nbin_angle=nba)
        v, hh, o = ix.export()
        assert np.array_equal(hh, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
        q = oracle.read_pdb(Q4CHA)
        m = oracle.make_query_map(q, "B57,B102,C195", oix, float(len(structs)), nbin_dist=nbd, nbin_angle=nba)
        qh, qi, qj = _query_arrays(m)
        assert list(ix.posting_lengths(qh)) == [len(oix.entries(int(x))) for x in qh]
        got = fd.count_query(ctx, ix, qh, qi, qj, fd.length_penalty(nres, 0.5))
        ref = oracle.count_query(m, oix, nres)
        assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in got] == \
               [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in ref]
        for g, r in zip(got, ref):
            assert abs(g["idf"] - r["idf"]) <= 1e-5 * max(abs(r["idf"]), 1e-30)


@pytest.mark.gpu
def test_unknown_encodings_are_refused(ctx, ser):
    import folddisco_amd as fd
    structs, ps, std, batch = ser
    for htype in (9, 12):
        with pytest.raises(Exception):
            fd.get_geometric_hash_as_u32(ctx, batch, hash_type=htype)
        with pytest.raises(Exception):
            fd.FolddiscoIndex.build(ctx, batch, hash_type=htype)


@pytest.mark.gpu
def test_kabsch_large_problems_wave_path(ctx):
    """Whole-structure matches superpose hundreds of points: problems of >= 128 points take the wavefront-per-problem kernel
    (tree-order f64 sums).  RMSD within 1e-4 of the restatement, rotation / translation within 1e-4 / 1e-3, next to small problems
    in the same batch."""
    from folddisco_amd import match
    rng = np.random.default_rng(3)
    probs = []
    for n in (3, 127, 128, 129, 600, 2000, 16):
        y = (rng.normal(size=(n, 3)) * 12).astype(np.float32)
        a = rng.uniform(0, np.pi)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        x = ((y - 3.0) @ R.T + rng.normal(size=(n, 3)) * 0.4).astype(np.float32)
        probs.append((x, y))
    xs = np.concatenate([p[0] for p in probs]); ys = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).astype(np.uint64)
    rmsd, rot, tran = match.kabsch_batch(ctx, xs, ys, off)
    for k, (x, y) in enumerate(probs):
        r, R, T = oracle.kabsch(x, y)
        assert abs(rmsd[k] - r) <= 1e-4, (k, len(x), rmsd[k], r)
        assert np.allclose(rot[k], R, atol=1e-4) and np.allclose(tran[k], T, atol=1e-3), (k, len(x))


@pytest.mark.gpu
def test_small_superpositions_four_to_a_wavefront_equal_the_wavefront_form(ctx, monkeypatch):
    """Batches of small problems (<= 16 points on average) run four to a wavefront (k_superpose4 / k_metrics4): same bits as a wavefront per
    problem (FDGPU_SP_PACK=0) — including groups of four that hold a larger problem (run one after the other), empty problems, degenerate inputs
    and a problem count that is not a multiple of four — and within the oracle's tolerance."""
    from folddisco_amd import match
    rng = np.random.default_rng(17)
    sizes = [int(v) for v in rng.integers(0, 17, size=397)]
    sizes[5] = 40; sizes[6] = 0; sizes[100] = 17; sizes[101] = 16; sizes[396] = 64        # a few groups of four take the sequential path
    probs = []
    for k, n in enumerate(sizes):
        y = (rng.normal(size=(n, 3)) * 9).astype(np.float32)
        if k % 29 == 3 and n:
            x = np.repeat(y[:1], n, axis=0)                                               # every moving point the same
        elif k % 31 == 4 and n > 2:
            x = (y[0] + np.outer(np.arange(n), [1.0, 0.5, -0.25])).astype(np.float32)     # collinear
        else:
            x = (y + rng.normal(size=(n, 3)) * rng.choice([0.1, 1.0, 5.0])).astype(np.float32)
        probs.append((x, y))
    xs = np.concatenate([p[0] for p in probs]); ys = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    assert off[-1] <= 16 * len(sizes)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FDGPU_SP_PACK", mode)
        rmsd, rot, tran = match.kabsch_batch(ctx, xs, ys, off)
        met = match.metrics_batch(ctx, ys, xs, off, rot, tran)
        res[mode] = (rmsd, rot, tran, met)
    monkeypatch.delenv("FDGPU_SP_PACK")
    for a, b in zip(res["1"], res["0"]):
        assert a.tobytes() == b.tobytes()
    rmsd, rot, tran, met = res["1"]
    for k in range(0, len(probs), 7):
        x, y = probs[k]
        if len(x) == 0:
            continue
        r, R, T = oracle.kabsch(x, y)
        assert (abs(rmsd[k] - r) <= 1e-4 * max(1.0, abs(r))) or (rmsd[k] == r), (k, rmsd[k], r)
        want = oracle.metrics(y, x, rot[k].reshape(9), tran[k])
        assert np.allclose(met[k], want, rtol=1e-6, atol=1e-6), (k, len(x), met[k], want)


@pytest.mark.gpu
def test_device_merge_of_loaded_indices(ctx):
    """parts that came through fdgpu_index_load carry no per-list last ids: the merge computes them with one decode pass"""
    import folddisco_amd as fd
    from folddisco_amd import synth
    ps = synth.to_packed(synth.generate(90, seed=8))
    off = ps.res_off.astype(np.int64)
    single = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=300)
    parts, loaded = [], []
    for a, b in ((0, 40), (40, 41), (41, 90)):
        sl = slice(off[a], off[b])
        chunk = fd.PackedStructures((ps.res_off[a:b + 1] - ps.res_off[a]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl])
        p = fd.FolddiscoIndex.build(ctx, ctx.upload(chunk), first_id=300 + a)
        v, h, o = p.export()
        loaded.append(fd.FolddiscoIndex.load(ctx, h, o, v, b - a, first_id=300 + a))
    merged = fd.FolddiscoIndexSet(loaded).merge()
    v, h, o = single.export()
    mv, mh, mo = merged.export()
    assert np.array_equal(mh, h) and np.array_equal(mo, o) and np.array_equal(mv, v)


@pytest.mark.gpu
def test_context_is_safe_under_concurrent_callers(ctx):
    """the reference calls its seams from many rayon workers (controller/mod.rs:291, query_pdb.rs:348,415).  Four host
    threads hammer ONE context through the C ABI (ctypes releases the GIL) with index builds, posting lookups and scoring: every
    result equals the serial one (the context serialises its entry points; one stream, one scratch set)."""
    import threading
    import folddisco_amd as fd
    from folddisco_amd import synth
    sets = [synth.to_packed(synth.generate(60 + 7 * k, seed=40 + k)) for k in range(4)]
    batches = [ctx.upload(ps) for ps in sets]
    want = []
    for ps, b in zip(sets, batches):
        ix = fd.FolddiscoIndex.build(ctx, b, first_id=5)
        v, h, o = ix.export()
        qh = h[:: max(1, len(h) // 50)][:50].astype(np.uint32)
        pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
        recs = fd.count_query(ctx, ix, qh, np.arange(len(qh), dtype=np.uint32) % 3, np.arange(len(qh), dtype=np.uint32) % 5, pen, as_array=True)
        want.append((v.tobytes(), h.tobytes(), o.tobytes(), ix.posting_lengths(qh).tobytes(), recs.tobytes(), qh, pen))
    errors = []

    def worker(k):
        try:
            for _ in range(6):
                ix = fd.FolddiscoIndex.build(ctx, batches[k], first_id=5)
                v, h, o = ix.export()
                wv, wh, wo, wl, wr, qh, pen = want[k]
                assert v.tobytes() == wv and h.tobytes() == wh and o.tobytes() == wo
                assert ix.posting_lengths(qh).tobytes() == wl
                recs = fd.count_query(ctx, ix, qh, np.arange(len(qh), dtype=np.uint32) % 3, np.arange(len(qh), dtype=np.uint32) % 5, pen, as_array=True)
                assert recs.tobytes() == wr
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.gpu
def test_superposition_degenerate_inputs_and_device_metrics(ctx):
    """k_superpose on the inputs where Kabsch's eigen construction degenerates — one point, two points, collinear points, every moving
    point equal, identical sets, a mirror image (negative determinant), coplanar points — gives the restatement's rmsd / rotation /
    translation (the reference reports the identity and a zero translation where the frame cannot be built); and k_metrics
    (TM-score, GDT-TS, GDT-HA, Chamfer, Hausdorff on the device) equals the restatement for 1 ... 700 points."""
    from folddisco_amd import match
    rng = np.random.default_rng(11)
    base = (rng.normal(size=(9, 3)) * 5).astype(np.float32)
    line = np.outer(np.arange(6, dtype=np.float32), np.array([1.0, 2.0, -0.5], np.float32)).astype(np.float32)
    plane = np.concatenate([rng.normal(size=(7, 2)) * 4, np.zeros((7, 1))], axis=1).astype(np.float32)
    probs = [
        (base[:1] + 1.0, base[:1]),                                          # one point
        (base[:2] * 1.1, base[:2]),                                          # two points
        (line + 0.5, line[::-1].copy()),                                     # collinear
        (np.repeat(base[:1], 5, axis=0), base[:5]),                          # every moving point equal
        (base.copy(), base.copy()),                                          # identical
        (base * np.array([1, 1, -1], np.float32), base),                     # mirror image
        (plane @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32), plane),   # coplanar
        (np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32)),        # all zero
        ((base + rng.normal(size=base.shape) * 0.3).astype(np.float32), base),
    ]
    xs = np.concatenate([p[0] for p in probs]); ys = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).astype(np.uint64)
    rmsd, rot, tran = match.kabsch_batch(ctx, xs, ys, off)
    for k, (x, y) in enumerate(probs):
        r, R, T = oracle.kabsch(x, y)
        assert (abs(rmsd[k] - r) <= 1e-4 * max(1.0, abs(r))) or (rmsd[k] == r), (k, rmsd[k], r)
        assert np.allclose(rot[k], R, atol=1e-4) and np.allclose(tran[k], T, atol=1e-3), (k, rot[k], R, tran[k], T)
    # metrics: sizes around the d0 switch (21 / 22 points) and beyond one wavefront
    mp = []
    for n in (1, 2, 6, 21, 22, 64, 65, 300, 700):
        y = (rng.normal(size=(n, 3)) * 9).astype(np.float32)
        x = (y + rng.normal(size=(n, 3)) * rng.choice([0.2, 1.5, 6.0])).astype(np.float32)
        mp.append((x, y))
    xs = np.concatenate([p[0] for p in mp]); ys = np.concatenate([p[1] for p in mp])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in mp])]).astype(np.uint64)
    rmsd, rot, tran = match.kabsch_batch(ctx, xs, ys, off)
    got = match.metrics_batch(ctx, ys, xs, off, rot, tran)
    for k, (x, y) in enumerate(mp):
        want = oracle.metrics(y, x, rot[k].reshape(9), tran[k])
        assert np.allclose(got[k], want, rtol=1e-6, atol=1e-6), (k, len(x), got[k], want)


@pytest.mark.gpu
def test_count_query_against_recount_packed_and_wide(ctx):
    """count_query's records against a numpy recount from the decoded posting lists, bit for bit (the idf sum is exact in 2^-22 fixed
    point on both sides): in the packed accumulator form (idf < 32) and in the wide form that an idf >= 32 selects (here:
    total_structures = 2^45), through the single and the batched entry point, with and without the device-side top-N."""
    import folddisco_amd as fd
    from folddisco_amd import synth
    from folddisco_amd.dist import rank_hits
    S = 300
    ps = synth.to_packed(synth.generate(S, seed=31))
    ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=0)
    hashes = ix.export()[1]
    rng = np.random.Generator(np.random.PCG64(8))
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    queries = []
    for n in (1, 7, 40, 300):
        qh = rng.choice(hashes, size=n, replace=False).astype(np.uint32)
        queries.append((qh, rng.integers(0, 6, size=n).astype(np.uint32), rng.integers(0, 6, size=n).astype(np.uint32)))
    for total in (S, 2 ** 45):
        want = []
        for qh, qi, qj in queries:
            lens = ix.posting_lengths(qh)
            idf = fd.idf_of_lengths(lens, total).astype(np.float32)
            fix = np.floor(idf.astype(np.float64) * 4194304.0 + 0.5).astype(np.int64)
            assert (idf >= 32).all() == (total != S)
            cnt, acc = np.zeros(S, np.int64), np.zeros(S, np.int64)
            nodes, edges = [set() for _ in range(S)], [set() for _ in range(S)]
            for k, ids in enumerate(ix.get_entries(qh)):
                for s in ids.astype(np.int64):
                    cnt[s] += 1; acc[s] += fix[k]; nodes[s].add(int(qi[k])); edges[s].add((int(qi[k]), int(qj[k])))
            rec = np.zeros(int((cnt > 0).sum()), fd.api.REC_DTYPE)
            t = np.nonzero(cnt)[0]
            rec["nid"], rec["total_match_count"] = t, cnt[t]
            rec["node_count"], rec["edge_count"] = [len(nodes[s]) for s in t], [len(edges[s]) for s in t]
            rec["idf"] = (acc[t].astype(np.float64) / 4194304.0).astype(np.float32) * pen[t]
            want.append(rec)
        for (qh, qi, qj), w in zip(queries, want):
            got = fd.count_query(ctx, ix, qh, qi, qj, pen, total_structures=total, as_array=True)
            assert got.tobytes() == w.tobytes(), (total, len(qh))
        gb = fd.count_query_batch(ctx, ix, queries, pen, total_structures=total)
        gt = fd.count_query_batch(ctx, ix, queries, pen, total_structures=total, top_n=10)
        for g, g10, w in zip(gb, gt, want):
            assert g.tobytes() == w.tobytes() and g10.tobytes() == rank_hits(w, 10).tobytes()


def test_msd_build_equals_structure_major_build(ctx):
    """The default index build of the default encoding (keys bucketed by the top six hash bits at emit time, residues visited in
    amino-acid order, three segmented 8-bit sort passes) against the structure-major stream + four passes (FDGPU_MSD=0) and against the
    buckets without the amino-acid order (FDGPU_MSD_PERM=0): byte-identical indices, for shapes that stress the bucket bookkeeping — one
    residue type only, unknown residues and missing CB mixed in, a structure longer than one tile row, empty structures, a shard that
    starts at a late id."""
    import os
    import folddisco_amd as fd
    from folddisco_amd import synth
    rng = np.random.Generator(np.random.PCG64(31))
    d = synth.generate(700, seed=31, lengths=np.concatenate([rng.integers(40, 400, 690), [0, 0, 1, 2, 1500, 2600, 64, 65, 63, 128]]))
    ps = synth.to_packed(d)
    aa = ps.aa.copy()
    off = ps.res_off.astype(np.int64)
    aa[off[10]:off[11]] = 17                                    # one residue type only (Trp): two buckets
    aa[off[20]:off[21]] = np.where(rng.random(off[21] - off[20]) < 0.3, 255, aa[off[20]:off[21]])      # unknown residues
    cbv = np.ones(len(aa), np.uint8)
    cbv[off[30]:off[31]] = rng.random(off[31] - off[30]) > 0.2  # missing CB
    ps = fd.PackedStructures(ps.res_off, ps.n_xyz, ps.ca_xyz, ps.cb_xyz, aa, cbv)
    ps_odd = fd.PackedStructures(ps.res_off, ps.n_xyz, ps.ca_xyz, ps.cb_xyz, np.where(np.arange(len(aa)) % 97 == 5, 25, aa).astype(np.uint8), cbv)
    outs = {}
    try:
        for tag, env in (("msd", {}), ("plain", {"FDGPU_MSD": "0"}), ("noperm", {"FDGPU_MSD_PERM": "0"})):
            for k in ("FDGPU_MSD", "FDGPU_MSD_PERM"):
                os.environ.pop(k, None)
            os.environ.update(env)
            ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=70000)
            outs[tag] = ix.export()
            # residue types outside map_aa_to_u8's 0..19 marked hashable: forty buckets cannot hold them — the build falls back to the 8-byte
            # structure-major path and equals what the plain build gives for the same input
            outs[tag + "_odd"] = fd.FolddiscoIndex.build(ctx, ctx.upload(ps_odd), first_id=70000).export()
    finally:
        for k in ("FDGPU_MSD", "FDGPU_MSD_PERM"):
            os.environ.pop(k, None)
    for tag in ("plain", "noperm"):
        for a, b in zip(outs["msd"], outs[tag]):
            assert np.array_equal(a, b), tag
        for a, b in zip(outs["msd_odd"], outs[tag + "_odd"]):
            assert np.array_equal(a, b), tag
    assert not np.array_equal(outs["msd"][1], outs["msd_odd"][1])
    assert len(outs["msd"][1]) > 10 ** 6


def test_release_workspaces_keeps_indices_and_batches_valid():
    """fdgpu_release_workspaces hands the sort buffers, scratch and cached index blocks back to the device: indices, batches and query maps made
    before stay valid, and the next calls allocate what they need again — same index bytes, same query results."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import synth
    c2 = fd.Context(0)
    ps = synth.to_packed(synth.generate(300, seed=91))
    b = c2.upload(ps)
    ix = fd.FolddiscoIndex.build(c2, b, first_id=5)
    before = [a.copy() for a in ix.export()]
    x, y = int(ps.res_off[7]), int(ps.res_off[8])
    qb = c2.upload(fd.PackedStructures.concat([dict(n_xyz=ps.n_xyz[x:y], ca_xyz=ps.ca_xyz[x:y], cb_xyz=ps.cb_xyz[x:y], aa=ps.aa[x:y])]))
    qm = fq.make_query_map(c2, qb, np.array([1, 4, 9, 12], np.uint32), None, ix, 300.0)
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    r0 = fd.count_query(c2, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=300, as_array=True)
    junk = fd.FolddiscoIndex.build(c2, b)          # a destroyed index leaves its blocks in the context's pool
    del junk
    c2.release_workspaces()
    c2.release_workspaces()                        # idempotent
    after = ix.export()
    for u, v in zip(before, after):
        assert np.array_equal(u, v)
    r1 = fd.count_query(c2, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=300, as_array=True)
    assert r0.tobytes() == r1.tobytes() and len(r0) > 0
    ix2 = fd.FolddiscoIndex.build(c2, b, first_id=5)
    for u, v in zip(before, ix2.export()):
        assert np.array_equal(u, v)
    got = fq.retrieve(c2, b, None, r1["nid"][:4].astype(np.uint32) - 5, qm, qb)
    assert any(g["cand"] >= 0 for g in got)
    c2.close()
