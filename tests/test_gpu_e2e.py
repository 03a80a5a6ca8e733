"""End-to-end on the GPU through the product path only (host ingest -> C ABI -> kernels), checked against the
reference's README rows (G2) and against the oracle for the intermediate products."""
import os

import numpy as np
import pytest

import oracle
from tests.helpers import Q1G2F, Q4CHA, SER, packed_to_oracle_structs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import folddisco_amd as fd
    from folddisco_amd import structure as st
    ctx = fd.Context(0)
    structs = [st.read_compact_structure(p) for p in SER]
    ps = fd.PackedStructures.concat([s.as_item() for s in structs])
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    nres = np.array([s.n for s in structs], np.uint64)
    plddt = np.array([s.avg_plddt() for s in structs], np.float32)
    tids = ["data/serine_peptidases/" + p.split("/")[-1] for p in SER]
    yield ctx, structs, batch, ix, nres, plddt, tids
    ctx.close()


def test_readme_rows_through_product_path(env):
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    q = st.read_compact_structure(Q4CHA)
    rows, matches = fq.query_pdb(ctx, ix, batch, structs, tids, nres, plddt, q, "B57,B102,C195")
    got = [fq.format_match_row(m) for m in matches]
    # README.md:218-223, in the reference's default order (idf desc, rmsd asc)
    assert got == [
        "data/serine_peptidases/4cha.pdb\t3\t8.7616\t0.0000\tB57,B102,C195\tB57,B102,C195",
        "data/serine_peptidases/4cha.pdb\t3\t8.7616\t0.0874\tF57,F102,G195\tB57,B102,C195",
        "data/serine_peptidases/1pq5.pdb\t3\t4.1178\t0.2609\tA56,A99,A195\tB57,B102,C195",
        "data/serine_peptidases/1ju3.pdb\t2\t1.4739\t0.7792\t_,A223,A234\tB57,B102,C195",
        "data/serine_peptidases/1l7a.pdb\t2\t1.4739\t0.7883\t_,A146,A127\tB57,B102,C195",
        "data/serine_peptidases/1l7a.pdb\t2\t1.4739\t0.8078\t_,B146,B127\tB57,B102,C195",
    ]
    # README.md:237-241 per-structure columns
    st_rows = {r["tid"].split("/")[-1]: ("%.4f" % r["idf"], r["total_match_count"], r["node_count"], r["edge_count"],
                                         r["max_matching_node_count"], "%.4f" % r["min_rmsd_with_max_match"], r["nres"], "%.4f" % r["plddt"], r["db_key"])
               for r in rows}
    assert st_rows["4cha.pdb"] == ("0.6138", 8, 3, 6, 3, "0.0000", 477, "13.5404", 4)
    assert st_rows["1pq5.pdb"] == ("0.4869", 4, 3, 4, 3, "0.2609", 224, "5.1340", 3)
    assert st_rows["1ju3.pdb"] == ("0.0617", 2, 2, 2, 2, "0.7792", 570, "19.4881", 1)
    assert st_rows["1l7a.pdb"] == ("0.0584", 2, 2, 2, 2, "0.7883", 636, "11.7037", 2)
    assert st_rows["1azw.pdb"][:4] == ("0.1856", 2, 2, 2)
    # stale README row reappears with --ca-distance 1.5
    _, m15 = fq.query_pdb(ctx, ix, batch, structs, tids, nres, plddt, q, "B57,B102,C195", ca_distance=1.5)
    assert "data/serine_peptidases/1azw.pdb\t2\t4.6439\t0.9234\tA179,_,B176\tB57,B102,C195" in [fq.format_match_row(m) for m in m15]


@pytest.mark.parametrize("qpath,qstr", [(Q4CHA, "B57,B102,C195"), (Q1G2F, "F207,F212,F225,F229"), (Q1G2F, "F207:C,F212,F225:HX,F229"),
                                        (Q4CHA, "B57-60,C195:ST")])
def test_make_query_map_matches_oracle(env, qpath, qstr):
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    ostructs = [oracle.read_pdb(p) for p in SER]
    oix, _, _ = oracle.build_index(ostructs)
    oq = oracle.read_pdb(qpath)
    om = oracle.make_query_map(oq, qstr, oix, 5.0).arrays()
    q = st.read_compact_structure(qpath)
    res = fq.parse_query_string(qstr, q.chains[0])
    idx = [q.get_index(c, r) for c, r, _ in res]
    subs = [s for _, _, s in res]
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    m = fq.make_query_map(ctx, qb, idx, subs, ix, 5.0)
    assert np.array_equal(m.hash, om["hash"]) and np.array_equal(m.qi, om["qi"]) and np.array_equal(m.qj, om["qj"])
    assert np.array_equal(m.is_primary, om["is_primary"])
    assert np.array_equal(m.idf.view(np.uint32), om["idf"].view(np.uint32))   # glibc log2f on both sides
    assert np.array_equal(m.aad_aa1, om["aad_aa1"]) and np.array_equal(m.aad_aa2, om["aad_aa2"])
    assert np.array_equal(m.aad_dist.view(np.uint32), om["aad_dist"].view(np.uint32)) and np.array_equal(m.aad_qi, om["aad_qi"])


def test_retrieve_matches_oracle(env):
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    ostructs = [oracle.read_pdb(p) for p in SER]
    oix, _, _ = oracle.build_index(ostructs)
    for qpath, qstr in ((Q4CHA, "B57,B102,C195"), (Q4CHA, "B57,B102,C195,B58,B59,C999")):
        oq = oracle.read_pdb(qpath)
        om = oracle.make_query_map(oq, qstr, oix, 5.0)
        q = st.read_compact_structure(qpath)
        res = fq.parse_query_string(qstr, q.chains[0])
        qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        pairs = [(i, s) for i, s in pairs if i is not None]   # unresolved residues are dropped (query.rs:238-246)
        m = fq.make_query_map(ctx, qb, [i for i, _ in pairs], [s for _, s in pairs], ix, 5.0)
        std = np.concatenate([s.resname_std() for s in structs])
        for ca_cut in (1.0, 1.5, 3.0):
            got = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut)
            for nid in range(5):
                R = oracle.retrieve(ostructs[nid], oq, om, ca_distance_cutoff=ca_cut)
                mine = [g for g in got if g["cand"] == nid]
                assert len(mine) == len(R["processed"]), (qstr, ca_cut, nid)
                for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                    assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                    assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                    assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and abs(g["rmsd_from_hash"] - rh["rmsd"]) <= 1e-4
                    assert g["idf"] == pytest.approx(rp["idf"], rel=1e-6)
                    # similarity metrics of the reported superposition (structure/metrics.rs) against the oracle's restatement
                    # fed with the oracle's own rotation / translation: tolerance 1e-4 like the RMSD
                    qpos = [i for (i, _), x in zip(pairs, rp["residues"]) if x is not None]
                    tpos = [x[2] for x in rp["residues"] if x is not None]
                    oa, ta = oq.arrays(), ostructs[nid].arrays()
                    qpts = np.stack([p for i in qpos for p in (oa["ca_xyz"][i], oa["cb_xyz"][i])])
                    tpts = np.stack([p for i in tpos for p in (ta["ca_xyz"][i], ta["cb_xyz"][i])])
                    want = oracle.metrics(qpts, tpts, rp["rot"], rp["tran"])
                    assert np.allclose(g["metrics"], want, atol=1e-4), (g["metrics"], want)


def test_partial_fit_retrieval_matches_oracle(env):
    """--partial-fit (retrieve.rs:733-746, 787-812): mappings of more than 3 residues are superposed by the LMS-QCP fit and
    report the rms over its core; smaller ones keep Kabsch; the mappings themselves do not change."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    oq = oracle.read_pdb(Q4CHA)
    oa = oq.arrays()
    ostructs = [oracle.read_pdb(p) for p in SER]
    q = st.read_compact_structure(Q4CHA)
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    std = np.concatenate([s.resname_std() for s in structs])
    n_lms = 0
    for qstr in ("B57,B102,C195,B58,B59,C999", "B57,B102,C195,C194,C196,B56,C214"):
        res = fq.parse_query_string(qstr, q.chains[0])
        idx = [q.get_index(c, r) for c, r, _ in res]
        keep = [k for k, i in enumerate(idx) if i is not None]
        qidx = [idx[k] for k in keep]
        m = fq.make_query_map(ctx, qb, qidx, [res[k][2] for k in keep], ix, 5.0)
        plain = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=3.0)
        part = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=3.0, partial_fit=True)
        assert len(plain) == len(part)
        for a, b in zip(plain, part):
            assert a["cand"] == b["cand"] and a["processed"] == b["processed"] and a["from_hash"] == b["from_hash"]
            tpos = [x for x in b["processed"] if x >= 0]
            qpos = [i for i, x in zip(qidx, b["processed"]) if x >= 0]
            if len(tpos) <= 3:
                assert b["rmsd"] == a["rmsd"] and np.array_equal(a["rot"], b["rot"])
                continue
            n_lms += 1
            ta = ostructs[b["cand"]].arrays()
            qpts = np.stack([p for i in qpos for p in (oa["ca_xyz"][i], oa["cb_xyz"][i])])
            tpts = np.stack([p for i in tpos for p in (ta["ca_xyz"][i], ta["cb_xyz"][i])])
            rms, rot, tran, core = oracle.lms_qcp(tpts, qpts)
            assert abs(b["rmsd"] - rms) <= 1e-4, (qstr, b["cand"], b["rmsd"], rms)
            assert np.allclose(b["rot"], rot, atol=1e-5) and np.allclose(b["tran"], tran, atol=1e-3)
            assert np.allclose(b["metrics"], oracle.metrics(qpts, tpts, rot, tran), atol=1e-4)
    assert n_lms > 0


@pytest.mark.parametrize("htype", [0, 1, 7, 8])
def test_other_encodings_query_map_and_retrieval(env, htype, tmp_path):
    """§8f rank 3, the encodings over the PDBTrRosetta descriptor: make_query_map (threshold expansion over the encoding's own
    dist / angle feature indices, substitutions), retrieval (pair scan hashes, symmetry flags) and the CLI round trip through
    `.type` equal the CPU restatement."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from folddisco_amd.__main__ import main as cli
    ctx, structs, batch, _ix, nres, plddt, tids = env
    ostructs = [oracle.read_pdb(p) for p in SER]
    ix = fd.FolddiscoIndex.build(ctx, batch, hash_type=htype)
    std = np.concatenate([s.resname_std() for s in structs])
    with oracle.hash_type(htype):
        oix, _, _ = oracle.build_index(ostructs)
        for qpath, qstr in ((Q4CHA, "B57,B102,C195"), (Q1G2F, "F207:C,F212,F225:HX,F229"), (Q4CHA, "B57,B102,C195,B58,B59")):
            oq = oracle.read_pdb(qpath)
            om_ = oracle.make_query_map(oq, qstr, oix, 5.0, dist_thr=(0.5, 1.0), angle_thr=(5.0, 10.0))
            om = om_.arrays()
            q = st.read_compact_structure(qpath)
            res = fq.parse_query_string(qstr, q.chains[0])
            idx = [q.get_index(c, r) for c, r, _ in res]
            qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
            m = fq.make_query_map(ctx, qb, idx, [x for _, _, x in res], ix, 5.0, dist_thr=(0.5, 1.0), angle_thr=(5.0, 10.0), hash_type=htype)
            assert np.array_equal(m.hash, om["hash"]) and np.array_equal(m.qi, om["qi"]) and np.array_equal(m.qj, om["qj"])
            assert np.array_equal(m.is_primary, om["is_primary"]) and np.array_equal(m.idf.view(np.uint32), om["idf"].view(np.uint32))
            for ca_cut in (1.0, 3.0):
                got = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut, hash_type=htype)
                for nid in range(5):
                    R = oracle.retrieve(ostructs[nid], oq, om_, ca_distance_cutoff=ca_cut)
                    mine = [g for g in got if g["cand"] == nid]
                    assert len(mine) == len(R["processed"]), (htype, qstr, ca_cut, nid)
                    for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                        assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                        assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                        assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and g["idf"] == pytest.approx(rp["idf"], rel=1e-6)
    # CLI: index -y <name> writes the type into PREFIX.type, query reads it back
    from folddisco_amd._lib import HASH_TYPE_NAMES
    prefix = str(tmp_path / "ix")
    cli(["index", "-p", os.path.dirname(SER[0]), "-i", prefix, "-y", HASH_TYPE_NAMES[htype]])
    assert f'hash_type = "{HASH_TYPE_NAMES[htype]}"' in open(prefix + ".type").read()
    v, hh, o = ix.export()
    from folddisco_amd import indexio
    dv, dh, do = indexio.read_index_files(prefix)
    assert np.array_equal(dv, v) and np.array_equal(dh, hh) and np.array_equal(do, o)
    out = str(tmp_path / "out.tsv")
    cli(["query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", prefix, "-o", out])
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    q = st.read_compact_structure(Q4CHA)
    _, want = fq.query_pdb(ctx, ix, batch, structs, [os.path.join(os.path.dirname(SER[0]), os.path.basename(p)) for p in SER], nres, plddt, q,
                           "B57,B102,C195", hash_type=htype, sort_by="")      # the CLI default: no --sort-by (idf descending, rmsd ascending)
    assert [r[1:] for r in rows] == [fq.format_match_row(m).split("\t")[1:] for m in want] and len(rows) > 0


def test_sharded_build_merge_and_disk_roundtrip(env, tmp_path):
    """index build shards by structure: two sub-indices built with id offsets merge into the byte-identical single index;
    the files written to disk load back (product loader and oracle loader) and answer the query identically."""
    import folddisco_amd as fd
    from folddisco_amd import indexio
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    whole = ix.export()
    parts = []
    for lo, hi in ((0, 2), (2, 5)):
        ps = fd.PackedStructures.concat([s.as_item() for s in structs[lo:hi]])
        sub = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=lo)
        parts.append(sub.export())
    v, h, o = indexio.merge_subindices(parts)
    assert np.array_equal(v, whole[0]) and np.array_equal(h, whole[1]) and np.array_equal(o, whole[2])
    pre = str(tmp_path / "serine_folddisco")
    indexio.write_index_files(pre, v, h, o)
    indexio.save_lookup(pre + ".lookup", tids, nres, plddt)
    indexio.save_type(pre + ".type", len(tids))
    ix.save(str(tmp_path / "direct"))
    for ext in ("", ".offset"):
        assert open(pre + ext, "rb").read() == open(str(tmp_path / "direct") + ext, "rb").read()
    # load back
    oix = oracle.load_index(pre)
    assert oix.H == 217612
    v2, h2, o2 = indexio.read_index_files(pre)
    t2, n2, p2, _ = indexio.load_lookup(pre + ".lookup")
    ix2 = fd.FolddiscoIndex.load(ctx, h2, o2, v2, len(t2))
    q = st.read_compact_structure(Q4CHA)
    rows1, m1 = fq.query_pdb(ctx, ix, batch, structs, tids, nres, plddt, q, "B57,B102,C195")
    rows2, m2 = fq.query_pdb(ctx, ix2, batch, structs, t2, n2, p2, q, "B57,B102,C195")
    assert [fq.format_match_row(m) for m in m1] == [fq.format_match_row(m) for m in m2]
    assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"], r["idf"]) for r in rows1] == \
           [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"], r["idf"]) for r in rows2]


def test_cli_index_and_query_reproduce_readme(tmp_path):
    """`index -p data/serine_peptidases -i …` then `query -p 4CHA.pdb -q B57,B102,C195 -i …` (README.md:200-241)"""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "data" / "serine_peptidases"
    d.mkdir(parents=True)
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    pre = str(tmp_path / "index" / "serine_folddisco")
    os.makedirs(os.path.dirname(pre))
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre, "-t", "4"], cwd=tmp_path, env=env)
    for ext in ("", ".offset", ".lookup", ".type"):
        assert os.path.exists(pre + ext)
    assert os.path.getsize(pre) == 225674 and os.path.getsize(pre + ".offset") == 2611360
    # chunked build (two structures per GPU call, sub-indices merged): byte-identical files
    pre2 = str(tmp_path / "index" / "serine_chunked")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre2, "--chunk", "2"], cwd=tmp_path, env=env)
    for ext in ("", ".offset", ".lookup", ".type"):
        assert open(pre + ext, "rb").read() == open(pre2 + ext, "rb").read(), ext
    # one structure per build call, the five resident sub-indices merged on the device in rounds of two: byte-identical again
    pre3 = str(tmp_path / "index" / "serine_rounds")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre3, "--chunk", "1", "-v"], cwd=tmp_path,
                          env=dict(env, FD_MERGE_GROUP="2"))
    for ext in ("", ".offset", ".lookup"):
        assert open(pre + ext, "rb").read() == open(pre3 + ext, "rb").read(), ext
    # the reference's --mmap-on-disk only chooses where ITS posting array lives while it is filled (indextable.rs:215-226,247): same files; accepted here
    pre4 = str(tmp_path / "index" / "serine_mmap")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre4, "--mmap-on-disk"], cwd=tmp_path, env=env)
    for ext in ("", ".offset", ".lookup", ".type"):
        assert open(pre + ext, "rb").read() == open(pre4 + ext, "rb").read(), ext
    out = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--header"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "tid\tnode_count\tidf\trmsd\tmatching_residues\tquery_residues"
    assert out[1:] == [
        "data/serine_peptidases/4cha.pdb\t3\t8.7616\t0.0000\tB57,B102,C195\tB57,B102,C195",
        "data/serine_peptidases/4cha.pdb\t3\t8.7616\t0.0874\tF57,F102,G195\tB57,B102,C195",
        "data/serine_peptidases/1pq5.pdb\t3\t4.1178\t0.2609\tA56,A99,A195\tB57,B102,C195",
        "data/serine_peptidases/1ju3.pdb\t2\t1.4739\t0.7792\t_,A223,A234\tB57,B102,C195",
        "data/serine_peptidases/1l7a.pdb\t2\t1.4739\t0.7883\t_,A146,A127\tB57,B102,C195",
        "data/serine_peptidases/1l7a.pdb\t2\t1.4739\t0.8078\t_,B146,B127\tB57,B102,C195",
    ]
    ps = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--per-structure"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    # README.md:237-240 (the 1azw row's match columns are stale)
    assert ps[0] == "data/serine_peptidases/4cha.pdb\t0.6138\t8\t3\t6\t3\t0.0000\t477\t13.5404\tB57,B102,C195:0.0000;F57,F102,G195:0.0874\t4\tB57,B102,C195"
    assert ps[1] == "data/serine_peptidases/1pq5.pdb\t0.4869\t4\t3\t4\t3\t0.2609\t224\t5.1340\tA56,A99,A195:0.2609\t3\tB57,B102,C195"
    assert "data/serine_peptidases/1ju3.pdb\t0.0617\t2\t2\t2\t2\t0.7792\t570\t19.4881\t_,A223,A234:0.7792\t1\tB57,B102,C195" in ps
    assert "data/serine_peptidases/1l7a.pdb\t0.0584\t2\t2\t2\t2\t0.7883\t636\t11.7037\t_,A146,A127:0.7883;_,B146,B127:0.8078\t2\tB57,B102,C195" in ps
    # no --sort-by = StructureSortStrategy::default (idf descending, cli/main.rs:84): 1azw (idf 0.1856, no match) comes third —
    # "node_count,rmsd" would put it last
    assert ps[2].startswith("data/serine_peptidases/1azw.pdb\t0.1856\t")
    # the per-structure table prints the NORMALISED query string (ranges expanded, default chain filled in; query_pdb.rs:359-365)
    pr = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57-57,B102,C195", "-i", pre, "--per-structure"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    assert pr == ps


@pytest.mark.gpu
def test_cli_filters_sort_and_sampling(tmp_path):
    """filtering options, --sort-by and --sampling-* of the query subcommand (query_pdb.rs:60-84, controller/filter.rs,
    controller/sort.rs, count_query.rs:222-253) on the README example"""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "data" / "serine_peptidases"
    d.mkdir(parents=True)
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    pre = str(tmp_path / "serine_folddisco")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre], cwd=tmp_path, env=env)

    def q(*extra):
        return subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, *extra],
                              cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    base = q()
    assert len(base) == 6
    # --partial-fit: matches of at most 3 residues keep the Kabsch superposition (retrieve.rs:735-740)
    assert q("--partial-fit") == base
    # --skip-ca-match: the from-hash mapping of every match (result.rs:54-69): same rows here except where the rescue added a residue
    sk = q("--skip-ca-match", "--ca-distance", "1.5")
    pr = q("--ca-distance", "1.5")
    assert len(sk) == len(pr) and sorted(r.split("\t")[0] for r in sk) == sorted(r.split("\t")[0] for r in pr)
    assert all(r.split("\t")[4].count("_") >= p_.split("\t")[4].count("_") for r, p_ in zip(sorted(sk), sorted(pr)))
    # --web: per-match rows with the superposition columns (query_pdb.rs:481-493), whatever --per-structure says
    web = q("--web", "--per-structure", "--header")
    assert web[0].split("\t")[:6] == ["tid", "node_count", "idf", "rmsd", "matching_residues", "u_matrix"] and len(web) == 7
    assert [r.split("\t")[:5] for r in web[1:]] == [b.split("\t")[:5] for b in base]
    bad = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--per-structure", "--per-match"],
                         cwd=tmp_path, env=env, capture_output=True, text=True)
    assert bad.returncode != 0
    # index --id: the lookup ids (controller/mode.rs:69-126)
    pre_id = str(tmp_path / "serine_ids")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre_id, "--id", "pdb"], cwd=tmp_path, env=env)
    assert [l.split("\t")[1] for l in open(pre_id + ".lookup")] == sorted(os.path.basename(p)[:-4] for p in SER)
    # MatchFilter: connected node count / ratio, rmsd, idf score
    assert q("--connected-node", "3") == base[:3]
    assert q("--connected-node-ratio", "0.9") == base[:3]
    assert q("--rmsd", "0.1") == base[:2]
    # --score is both the per-structure prefilter idf cutoff (0.6138 / 0.4869 pass, 0.0617 / 0.0584 / 0.1856 do not) and the per-match one
    assert q("--score", "0.4") == base[:3] and q("--score", "4.0") == []
    # StructureFilter before matching: covered nodes (prefilter node_count: 3 only for 4cha and 1pq5), after matching: max node
    assert q("--covered-node", "3") == base[:3]
    assert q("--max-node", "3") == base[:3]
    assert q("--num-residue", "300") == [r for r in base if "1pq5" in r]
    # --sort-by: idf ascending puts the 2-node matches (idf 1.4739) first, rmsd ascending inside ties
    by_idf = q("--sort-by", "idf:asc,rmsd")
    assert sorted(by_idf) == sorted(base) and [r.split("\t")[2] for r in by_idf] == ["1.4739"] * 3 + ["4.1178"] + ["8.7616"] * 2
    # sampling: all hashes kept == no sampling; keeping fewer hashes can only lower the prefilter counts
    assert q("--sampling-ratio", "1.0") == base
    full = q("--per-structure", "--skip-match")
    half = q("--per-structure", "--skip-match", "--sampling-ratio", "0.5")
    tot = lambda rows: {r.split("\t")[0]: int(r.split("\t")[2]) for r in rows}
    assert all(tot(half).get(k, 0) <= v for k, v in tot(full).items()) and sum(tot(half).values()) < sum(tot(full).values())


@pytest.mark.gpu
def test_sample_query_keeps_shortest_posting_lists():
    import folddisco_amd as fd
    from folddisco_amd import synth
    from folddisco_amd.query import sample_query_hashes
    ctx = fd.Context(0)
    ps = synth.to_packed(synth.generate(200, seed=5))
    ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps))
    _, hashes, _ = ix.export()
    rng = np.random.Generator(np.random.PCG64(1))
    qh = rng.choice(hashes, size=37, replace=False).astype(np.uint32)
    lens = ix.posting_lengths(qh)
    order = np.argsort(lens, kind="stable")
    assert list(sample_query_hashes(ix, qh)) == list(range(37))
    assert list(sample_query_hashes(ix, qh, sampling_ratio=0.5, sampling_count=3)) == list(range(37))    # both given -> all
    assert list(sample_query_hashes(ix, qh, sampling_count=5)) == list(order[:5])
    assert list(sample_query_hashes(ix, qh, sampling_ratio=0.3)) == list(order[: int(np.ceil(np.float32(0.3) * np.float32(37)))])


@pytest.mark.gpu
def test_cli_format_output_metrics(tmp_path):
    """--format-output with the similarity-metric columns and a metric filter (query_pdb.rs:79-99, result.rs:280-298)"""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "data" / "serine_peptidases"
    d.mkdir(parents=True)
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    pre = str(tmp_path / "serine_folddisco")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre], cwd=tmp_path, env=env)
    out = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--header",
                          "--format-output", "tid,rmsd,tm_score,gdt_ts,gdt_ha,chamfer_distance,hausdorff_distance"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].split("\t") == ["tid", "rmsd", "tm_score", "gdt_ts", "gdt_ha", "chamfer_distance", "hausdorff_distance"]
    rows = [r.split("\t") for r in out[1:]]
    assert len(rows) == 6 and rows[0][1:] == ["0.0000", "1.0000", "1.0000", "1.0000", "0.0000", "0.0000"]    # the query itself
    for r in rows[1:]:
        rmsd, tm, ts, ha, ch, hd = map(float, r[1:])
        assert 0.0 < tm < 1.0 and 0.0 <= ha <= ts <= 1.0 and 0.0 < ch <= hd and ch <= rmsd * 1.0001 + 1e-4
    sup = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--superpose"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    f0 = sup[0].split("\t")                          # tid node_count idf rmsd matching_residues u_matrix t_vector matching_coordinates db_key query
    assert len(f0) == 10 and f0[4] == "B57,B102,C195" and f0[8] == "4"
    U = np.array(f0[5].split(","), float).reshape(3, 3)
    assert np.allclose(U, np.eye(3), atol=1e-3) and np.allclose(np.array(f0[6].split(","), float), 0.0, atol=1e-2)   # the query on itself
    assert len(f0[7].split(",")) == 9               # three C-alpha positions
    keep = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", pre, "--tm-score", "0.9"],
                          cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(keep) == sum(float(r[2]) >= 0.9 for r in rows) and len(keep) >= 1


@pytest.mark.gpu
def test_query_map_batch_equals_single(env):
    """fdgpu_make_query_map_batch == fdgpu_make_query_map per query (same hashes in the same insertion order, same node / edge
    labels, idf, observed-distance lists), with substitutions and an unresolvable residue in the mix"""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    q1, q2 = st.read_compact_structure(Q4CHA), st.read_compact_structure(Q1G2F)
    qall = ctx.upload(fd.PackedStructures.concat([q1.as_item(), q2.as_item()]))
    specs = [(0, q1, "B57,B102,C195"), (1, q2, "F207,F212,F225,F229"), (0, q1, "B57:HKR,B102,C195:ST"), (1, q2, "F207:C,F212,F225:HX,F229")]
    reqs, singles = [], []
    for sidx, q, qstr in specs:
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        pairs = [(i, s) for i, s in pairs if i is not None]
        reqs.append((sidx, [i for i, _ in pairs], [s for _, s in pairs]))
        qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
        singles.append(fq.make_query_map(ctx, qb, [i for i, _ in pairs], [s for _, s in pairs], ix, 5.0))
    got = fq.make_query_maps(ctx, qall, reqs, ix, 5.0)
    assert len(got) == len(singles)
    for g, w in zip(got, singles):
        for f in ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi"):
            assert getattr(g, f).tobytes() == getattr(w, f).tobytes(), f
    assert all(len(g.hash) > 0 for g in got) and len(got[2].hash) > len(got[0].hash)      # substitutions add hashes


@pytest.mark.gpu
def test_cli_batch_query_file(tmp_path):
    """`query -q <file.txt>`: one query per line `pdb<TAB>residues<TAB>[output]` (the reference's query/*.txt, README.md:243-258)"""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "data" / "serine_peptidases"
    d.mkdir(parents=True)
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    (tmp_path / "query").mkdir()
    shutil.copy(Q4CHA, tmp_path / "query" / "4CHA.pdb")
    shutil.copy(Q1G2F, tmp_path / "query" / "1G2F.pdb")
    here = os.path.dirname(os.path.abspath(__file__))
    lines = [open(os.path.join(here, "golden", "query", f)).read().rstrip("\n") for f in ("serine_peptidase.txt", "zinc_finger_with_output.txt")]
    (tmp_path / "batch.txt").write_text("\n".join(lines) + "\n")
    root = os.path.dirname(here)
    env = dict(os.environ, PYTHONPATH=root)
    pre = str(tmp_path / "serine_folddisco")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", pre], cwd=tmp_path, env=env)
    out = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-q", "batch.txt", "-i", pre], cwd=tmp_path, env=env,
                         capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == 6 and out[0] == "data/serine_peptidases/4cha.pdb\t3\t8.7616\t0.0000\tB57,B102,C195\tB57,B102,C195"
    zf = (tmp_path / "zinc_finger.folddisco.out.tsv").read_text().splitlines()     # no zinc finger among the serine peptidases
    single = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", "query/1G2F.pdb", "-q", "F207,F212,F225,F229", "-i", pre],
                            cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    assert zf == single


@pytest.mark.gpu
def test_sharded_query_equals_single_index():
    """SURVEY §8e through the product path: two ranks (gloo, one GPU), index and coordinates sharded by structure id, idf from
    all-reduced posting lengths, all-gather of candidate records and of the matches found on the owning rank — records
    byte-identical and matches identical to the single-index query (tests/shard_query_worker.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29733", os.path.join(root, "tests", "shard_query_worker.py")],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert "SHARDED_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("QUERY") == 3 and "DIFFERENT" not in out.stdout


@pytest.mark.gpu
def test_cli_two_ranks_sharded_index_and_query(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 -m folddisco_amd index|query` (gloo, both ranks on one GPU): the index build is
    sharded by structure, rank 0 merges the shards into the reference's single index (byte-identical to the one-rank build);
    the query runs against the shards (all-reduced posting lengths, all-gathered candidates and matches) and prints what the
    one-rank query prints."""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "data" / "serine_peptidases"
    d.mkdir(parents=True)
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, FD_BENCH_BACKEND="gloo")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/serine_peptidases", "-i", one], cwd=tmp_path, env=env)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29744",
              "-m", "folddisco_amd"]
    subprocess.run(launch + ["index", "-p", "data/serine_peptidases", "-i", two], cwd=tmp_path, env=env, check=True, capture_output=True, timeout=600)
    for ext in ("", ".offset", ".lookup", ".type"):
        assert open(one + ext, "rb").read() == open(two + ext, "rb").read(), ext
    assert os.path.exists(two + ".shard0of2.offset") and os.path.exists(two + ".shard1of2.offset")
    for extra in ([], ["--per-structure"], ["--rmsd", "0.5", "--sort-by", "idf"]):
        q = ["query", "-p", Q4CHA, "-q", "B57,B102,C195"] + extra
        want = subprocess.run([sys.executable, "-m", "folddisco_amd"] + q + ["-i", one], cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout
        got = subprocess.run(launch + q + ["-i", two], cwd=tmp_path, env=env, capture_output=True, text=True, check=True, timeout=600).stdout
        rows = [l for l in got.splitlines() if l.startswith("data/")]
        assert rows == want.splitlines() and len(rows) >= 2, (extra, got[-2000:])


@pytest.mark.gpu
def test_retrieve_batch_equals_single(env):
    """fdgpu_retrieve_batch (one pair scan, one gather, one Kabsch launch for all queries) == fdgpu_retrieve per query: same
    matches in the same order, bit-identical floats; queries with different candidate lists, one with no candidates"""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    q1, q2 = st.read_compact_structure(Q4CHA), st.read_compact_structure(Q1G2F)
    qall = ctx.upload(fd.PackedStructures.concat([q1.as_item(), q2.as_item()]))
    std = np.concatenate([s.resname_std() for s in structs])
    specs = [(0, q1, "B57,B102,C195", [0, 1, 2, 3, 4]), (1, q2, "F207,F212,F225,F229", [4, 2]), (0, q1, "B57:HKR,B102,C195:ST", [3, 4]),
             (0, q1, "B57,B102", [])]
    reqs, singles = [], []
    for sidx, q, qstr, cand in specs:
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        reqs.append((sidx, [i for i, _ in pairs], [s for _, s in pairs]))
        qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
        qm = fq.make_query_map(ctx, qb, [i for i, _ in pairs], [s for _, s in pairs], ix, 5.0)
        singles.append(fq.retrieve(ctx, batch, std, np.array(cand, np.uint32), qm, qb))
    qms = fq.make_query_maps(ctx, qall, reqs, ix, 5.0)
    got = fq.retrieve_batch(ctx, batch, std, [np.array(sp[3], np.uint32) for sp in specs], qms, qall, [sp[0] for sp in specs])
    assert [len(g) for g in got] == [len(w) for w in singles] and len(got[0]) == 6 and got[3] == []
    marr, moff, rarr, roff = fq.retrieve_batch(ctx, batch, std, [np.array(sp[3], np.uint32) for sp in specs], qms, qall, [sp[0] for sp in specs], as_arrays=True)
    assert moff.tolist() == np.concatenate([[0], np.cumsum([len(g) for g in got])]).tolist() and len(marr) == moff[-1]
    assert [float(x) for x in marr["rmsd"][:6]] == [g["rmsd"] for g in got[0]] and rarr[:3].tolist() == got[0][0]["from_hash"]
    for g, w in zip(got, singles):
        for a, b in zip(g, w):
            assert a["cand"] == b["cand"] and a["processed"] == b["processed"] and a["from_hash"] == b["from_hash"] and a["same"] == b["same"]
            for f in ("idf", "rmsd", "rmsd_from_hash"):
                assert np.float32(a[f]).tobytes() == np.float32(b[f]).tobytes(), f
            assert a["rot"].tobytes() == b["rot"].tobytes() and a["tran"].tobytes() == b["tran"].tobytes() and a["metrics"].tobytes() == b["metrics"].tobytes()


@pytest.mark.gpu
def test_synthetic_database_full_query_matches_oracle():
    """Planted motif queries against a synthetic database (80 AFDB-shaped structures): query map, prefilter records (counts exact,
    idf rel 1e-5), and retrieval of the top candidates (matched residues exact, subgraph idf rel 1e-6, RMSD 1e-4) against the
    oracle — the README goldens only exercise five real structures."""
    import torch
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    from folddisco_amd import querybench, synth
    ctx = fd.Context(0)
    S = 80
    d = synth.generate(S, seed=2024)
    ps = synth.to_packed(d)
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    structs = packed_to_oracle_structs(ps)
    oix, onres, _ = oracle.build_index(structs)
    nres = np.diff(ps.res_off).astype(np.uint64)
    pen = fd.length_penalty(nres, 0.5)
    n_checked = 0
    for s, idx, item in querybench._pick_queries(d, S, 6, seed=99):
        qb = ctx.upload(fd.PackedStructures.concat([item]))
        qm = fq.make_query_map(ctx, qb, idx, None, ix, float(S))
        oq = structs[s]
        om = oracle.make_query_map(oq, ",".join(f"A{int(i) + 1}" for i in idx), oix, float(S))
        oa = om.arrays()
        assert np.array_equal(qm.hash, oa["hash"]) and np.array_equal(qm.qi, oa["qi"]) and np.array_equal(qm.qj, oa["qj"])
        recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
        want = oracle.count_query(om, oix, onres)
        assert [int(r["nid"]) for r in recs] == [w["nid"] for w in want]
        for r, w in zip(recs, want):
            assert (int(r["total_match_count"]), int(r["node_count"]), int(r["edge_count"])) == (w["total_match_count"], w["node_count"], w["edge_count"])
            assert float(r["idf"]) == pytest.approx(w["idf"], rel=1e-5)
        assert s in recs["nid"]
        cand = fdist.rank_hits(recs, 10)["nid"].astype(np.uint32)
        got = fq.retrieve(ctx, batch, None, cand, qm, qb)
        for slot, nid in enumerate(cand):
            R = oracle.retrieve(structs[int(nid)], oq, om)
            mine = [g for g in got if g["cand"] == slot]
            assert len(mine) == len(R["processed"]), (s, nid)
            for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and g["idf"] == pytest.approx(rp["idf"], rel=1e-6)
                n_checked += 1
    assert n_checked >= 6      # at least every query's own structure matches itself


def test_whole_structure_query_matches_oracle():
    """BASELINE config 5, whole-structure query mode (no -q): every residue of the query structure is a query residue
    (query.rs:226-233), the query map has far more than 200 hashes so retrieval scans all N^2 residue pairs without the amino-acid
    prefilter (retrieve.rs:146-153, 569-571).  Query map, prefilter records and the matches of the top candidates equal the oracle."""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from folddisco_amd import query as fq
    from tests.helpers import synthetic_packed
    ctx = fd.Context(0)
    S = 24
    ps = synthetic_packed(S, 31, lengths=np.full(S, 48))
    batch = ctx.upload(ps)
    ix = fd.FolddiscoIndex.build(ctx, batch)
    structs = packed_to_oracle_structs(ps)
    oix, onres, _ = oracle.build_index(structs)
    pen = fd.length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5)
    for s in (0, 7):
        a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
        item = dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b])
        qb = ctx.upload(fd.PackedStructures.concat([item]))
        qm = fq.make_query_map(ctx, qb, np.arange(b - a, dtype=np.uint32), None, ix, float(S))
        om = oracle.make_query_map(structs[s], "", oix, float(S))
        oa = om.arrays()
        assert len(qm.hash) > 200
        assert np.array_equal(qm.hash, oa["hash"]) and np.array_equal(qm.qi, oa["qi"]) and np.array_equal(qm.qj, oa["qj"])
        assert np.array_equal(qm.idf.view(np.uint32), oa["idf"].view(np.uint32))
        recs = fd.count_query(ctx, ix, qm.hash, qm.qi, qm.qj, pen, total_structures=S, as_array=True)
        want = oracle.count_query(om, oix, onres)
        assert [(int(r["nid"]), int(r["total_match_count"]), int(r["node_count"]), int(r["edge_count"])) for r in recs] == \
               [(w["nid"], w["total_match_count"], w["node_count"], w["edge_count"]) for w in want]
        cand = fdist.rank_hits(recs, 4)["nid"].astype(np.uint32)
        assert int(cand[0]) == s                                  # the structure itself ranks first
        got = fq.retrieve(ctx, batch, None, cand, qm, qb)
        n = 0
        for slot, nid in enumerate(cand):
            R = oracle.retrieve(structs[int(nid)], structs[s], om)
            mine = [g for g in got if g["cand"] == slot]
            assert len(mine) == len(R["processed"]), (s, nid)
            for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and g["idf"] == pytest.approx(rp["idf"], rel=1e-5)
                n += 1
        assert n >= 1
    ctx.close()


@pytest.mark.parametrize("htype", [3, 0, 1, 2, 4, 5, 6, 7, 8])
def test_query_map_device_expansion_equals_host_expansion(env, monkeypatch, htype):
    """Large queries without substitutions expand, hash and dedupe their candidates on the device (k_qm_expand_hash + sort + first-of-run,
    fd_query_map.hip); FDGPU_QM_DEVICE=1 takes that path for any size, 0 the host loop: every array of the query map is identical — every
    encoding (their own threshold fields), several threshold lists (the f32 restore drift of expand_and_insert accumulates over them), motif and
    whole-structure queries, with and without an index."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from tests.helpers import synthetic_packed
    ctx = env[0]
    S = 12
    ps = synthetic_packed(S, 177 + htype, lengths=np.concatenate([np.full(S - 2, 70), [150, 9]]))
    sb = ctx.upload(ps)
    six = fd.FolddiscoIndex.build(ctx, sb, hash_type=htype)
    fields = ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi", "primary_hash")
    n_checked = 0
    for s, idx in ((3, None), (10, None), (11, None), (5, [1, 7, 8, 30, 31]), (10, [0, 149])):
        a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
        qb = ctx.upload(fd.PackedStructures.concat([dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b])]))
        qi = np.arange(b - a, dtype=np.uint32) if idx is None else np.array(idx, np.uint32)
        for dthr, athr in (((0.5,), (5.0,)), ((0.5, 1.0, 0.25), (5.0, 10.0)), ((), (7.5,))):
            for index in (six, None):
                got = {}
                for mode in ("0", "1", "2", None):      # host, device incl. the sort-based dedupe, device hashes only, default (chain for small queries)
                    if mode is None:
                        monkeypatch.delenv("FDGPU_QM_DEVICE", raising=False)
                    else:
                        monkeypatch.setenv("FDGPU_QM_DEVICE", mode)
                    got[mode] = fq.make_query_map(ctx, qb, qi, None, index, float(S), dist_thr=dthr, angle_thr=athr, hash_type=htype)
                monkeypatch.delenv("FDGPU_QM_DEVICE", raising=False)
                for f in fields:
                    for mode in ("1", "2", None):
                        x, y = getattr(got["0"], f), getattr(got[mode], f)
                        assert x.dtype == y.dtype and x.tobytes() == y.tobytes(), (htype, s, f, mode)
                n_checked += len(got["0"].hash)
    assert n_checked > 10000


def test_query_map_batch_device_chain_equals_host_forms(env, monkeypatch):
    """A batch of motif queries expands, hashes, dedupes (k_qm_dedupe: first insertion wins in an LDS table) and looks its posting lengths up in
    one chain of kernels; FDGPU_QM_DEVICE=2 keeps the dedupe and the length pass on the host, 0 the expansion too.  The maps — and what
    scoring makes of the lengths / list positions they remember — are identical; a query beyond the dedupe kernel's table in the batch
    sends the whole batch down the host dedupe."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd.api import count_query_maps, length_penalty
    from tests.helpers import synthetic_packed
    ctx = env[0]
    S = 40
    rng = np.random.default_rng(5)
    ps = synthetic_packed(S, 911, lengths=np.concatenate([rng.integers(40, 120, S - 1), [150]]))
    sb = ctx.upload(ps)
    six = fd.FolddiscoIndex.build(ctx, sb)
    six.set_penalty(length_penalty(np.diff(ps.res_off).astype(np.uint64), 0.5))
    fields = ("hash", "qi", "qj", "is_primary", "idf", "indices", "aad_aa1", "aad_aa2", "aad_dist", "aad_qi", "primary_hash")
    small = []
    for s in range(0, 24):
        n = int(ps.res_off[s + 1] - ps.res_off[s])
        small.append((s, np.sort(rng.choice(n, size=int(rng.integers(2, 9)), replace=False)).astype(np.uint32)))
    small.append((3, np.array([5, 5, 9], np.uint32)))            # a repeated residue: pairs (5, 5) are skipped, the rest twice
    small.append((7, np.array([1000, 3], np.uint32)))            # an index outside the structure
    big = small + [(S - 1, np.arange(150, dtype=np.uint32))]     # 22 k pairs x 11 candidates: beyond the dedupe kernel's table
    n_checked = 0
    for queries in (small, big):
        for index in (six, None):
            got = {}
            for mode in ("0", "2", None):
                if mode is None:
                    monkeypatch.delenv("FDGPU_QM_DEVICE", raising=False)
                else:
                    monkeypatch.setenv("FDGPU_QM_DEVICE", mode)
                maps = fq.make_query_maps(ctx, sb, queries, index, float(S))
                recs = count_query_maps(ctx, six, maps, None, total_structures=S, top_n=10) if index is not None else None
                got[mode] = (maps, recs)
            monkeypatch.delenv("FDGPU_QM_DEVICE", raising=False)
            for mode in ("2", None):
                for t in range(len(queries)):
                    for f in fields:
                        x, y = getattr(got["0"][0][t], f), getattr(got[mode][0][t], f)
                        assert x.dtype == y.dtype and x.tobytes() == y.tobytes(), (mode, t, f)
                    if index is not None:
                        assert got["0"][1][t].tobytes() == got[mode][1][t].tobytes(), (mode, t)
                    n_checked += len(got["0"][0][t].hash)
    assert n_checked > 20000


def test_two_pass_retrieval_equals_single_pass(env, monkeypatch):
    """Large queries scan the candidates twice (found triples, then candidate pairs restricted to the partner residues some
    component mapped — the only ones the rescue counts): same matches, rescued residues included, as the single scan."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from tests.helpers import synthetic_packed
    ctx, structs, batch, ix, nres, plddt, tids = env
    std = np.concatenate([s.resname_std() for s in structs])

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x["cand"] == y["cand"] and x["processed"] == y["processed"] and x["from_hash"] == y["from_hash"] and x["same"] == y["same"]
            assert x["rmsd"] == y["rmsd"] and x["rmsd_from_hash"] == y["rmsd_from_hash"] and x["idf"] == y["idf"]

    n_rescued = 0
    q = st.read_compact_structure(Q4CHA)
    qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
    for qstr, ca_cut in (("B57,B102,C195", 1.0), ("B57,B102,C195,B58,B59,C999", 3.0), ("B57,B102,C195,C194,C196,B56,C214", 3.0), ("B40-70", 2.0)):
        res = fq.parse_query_string(qstr, q.chains[0])
        idx = [i for i in (q.get_index(c, r) for c, r, _ in res) if i is not None]
        m = fq.make_query_map(ctx, qb, idx, None, ix, 5.0)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("FDGPU_TWO_PASS", mode)
            out[mode] = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut)
            monkeypatch.setenv("FDGPU_PACK_MIN", "0")      # candidate pairs packed and sorted on the device (the large-query form)
            out[mode + "p"] = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut)
            monkeypatch.delenv("FDGPU_PACK_MIN")
        # the second scan counts the rescue votes on the device by default; FDGPU_DEVICE_VOTES=0: candidate pairs copied back and counted on the host
        monkeypatch.setenv("FDGPU_DEVICE_VOTES", "0")
        out["1h"] = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut)
        monkeypatch.delenv("FDGPU_DEVICE_VOTES")
        monkeypatch.setenv("FDGPU_FOUND_SORT", "device")      # the first scan's found triples ordered by two radix sorts on the device (default from 32,768 triples)
        out["1d"] = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut)
        monkeypatch.delenv("FDGPU_FOUND_SORT")
        same(out["0"], out["1"]); same(out["0"], out["0p"]); same(out["0"], out["1p"]); same(out["0"], out["1h"]); same(out["0"], out["1d"])
        n_rescued += sum(1 for g in out["1"] if not g["same"])
    assert n_rescued > 0          # the rescue path ran
    # whole-structure queries on a synthetic shard
    S = 16
    ps = synthetic_packed(S, 77, lengths=np.full(S, 64))
    sb = ctx.upload(ps)
    six = fd.FolddiscoIndex.build(ctx, sb)
    for s in (3, 9):
        a, b = int(ps.res_off[s]), int(ps.res_off[s + 1])
        qb2 = ctx.upload(fd.PackedStructures.concat([dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b])]))
        m = fq.make_query_map(ctx, qb2, np.arange(b - a, dtype=np.uint32), None, six, float(S))
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("FDGPU_TWO_PASS", mode)
            out[mode] = fq.retrieve(ctx, sb, None, np.arange(S, dtype=np.uint32), m, qb2, ca_distance_cutoff=1.5)
            monkeypatch.setenv("FDGPU_PACK_MIN", "0")
            out[mode + "p"] = fq.retrieve(ctx, sb, None, np.arange(S, dtype=np.uint32), m, qb2, ca_distance_cutoff=1.5)
            monkeypatch.delenv("FDGPU_PACK_MIN")
        monkeypatch.setenv("FDGPU_DEVICE_VOTES", "0")
        out["1h"] = fq.retrieve(ctx, sb, None, np.arange(S, dtype=np.uint32), m, qb2, ca_distance_cutoff=1.5)
        monkeypatch.delenv("FDGPU_DEVICE_VOTES")
        monkeypatch.setenv("FDGPU_FOUND_SORT", "device")
        out["1d"] = fq.retrieve(ctx, sb, None, np.arange(S, dtype=np.uint32), m, qb2, ca_distance_cutoff=1.5)
        monkeypatch.delenv("FDGPU_FOUND_SORT")
        same(out["0"], out["1"]); same(out["0"], out["0p"]); same(out["0"], out["1p"]); same(out["0"], out["1h"]); same(out["0"], out["1d"])
        assert any(g["cand"] == s for g in out["1"])
    monkeypatch.delenv("FDGPU_TWO_PASS")


@pytest.mark.parametrize("htype,bins", [(3, [(16, 4), (8, 3)]), (3, [(12, 4), (16, 4), (6, 2)]), (7, [(8, 32), (4, 12)])])
def test_multiple_bins_index_query_and_retrieval(env, htype, bins, tmp_path):
    """--multiple-bins (build_index.rs:45, feature.rs:211-215, query.rs:59-70, retrieve.rs:124-131): every residue pair is hashed
    once per (dist, angle) bin pair into one index, queries insert every expansion under every bin pair, retrieval reports a
    found triple per matching bin pair.  Index bytes, query map, count records and matches equal the CPU restatement."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, _ix, nres, plddt, tids = env
    ostructs = [oracle.read_pdb(p) for p in SER]
    std = np.concatenate([s.resname_std() for s in structs])
    ix = fd.FolddiscoIndex.build(ctx, batch, hash_type=htype, multiple_bins=bins)
    with oracle.hash_type(htype), oracle.multiple_bins(bins):
        oix, onres, _ = oracle.build_index(ostructs)
        v, hh, o = ix.export()
        assert np.array_equal(hh, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
        pen = fd.length_penalty(onres, 0.5)
        for qpath, qstr in ((Q4CHA, "B57,B102,C195"), (Q1G2F, "F207:C,F212,F225:HX,F229"), (Q4CHA, "B57,B102,C195,B58,B59")):
            oq = oracle.read_pdb(qpath)
            om_ = oracle.make_query_map(oq, qstr, oix, 5.0)
            om = om_.arrays()
            q = st.read_compact_structure(qpath)
            res = fq.parse_query_string(qstr, q.chains[0])
            idx = [q.get_index(c, r) for c, r, _ in res]
            qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
            m = fq.make_query_map(ctx, qb, idx, [x for _, _, x in res], ix, 5.0, hash_type=htype, multiple_bins=bins)
            assert np.array_equal(m.hash, om["hash"]) and np.array_equal(m.qi, om["qi"]) and np.array_equal(m.qj, om["qj"])
            assert np.array_equal(m.is_primary, om["is_primary"]) and np.array_equal(m.idf.view(np.uint32), om["idf"].view(np.uint32))
            got = fd.count_query(ctx, ix, m.hash, m.qi, m.qj, pen)
            ref = oracle.count_query(om_, oix, onres)
            assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in got] == \
                   [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in ref]
            for ca_cut in (1.0, 3.0):
                ms = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut, hash_type=htype, multiple_bins=bins)
                for nid in range(5):
                    R = oracle.retrieve(ostructs[nid], oq, om_, ca_distance_cutoff=ca_cut)
                    mine = [g for g in ms if g["cand"] == nid]
                    assert len(mine) == len(R["processed"]), (qstr, ca_cut, nid)
                    for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                        assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                        assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                        assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and g["idf"] == pytest.approx(rp["idf"], rel=1e-6)
    # CLI round trip through PREFIX.type
    from folddisco_amd.__main__ import main as cli
    from folddisco_amd._lib import HASH_TYPE_NAMES
    from folddisco_amd import indexio
    prefix = str(tmp_path / "ix")
    cli(["index", "-p", os.path.dirname(SER[0]), "-i", prefix, "-y", HASH_TYPE_NAMES[htype], "--multiple-bins", ",".join(f"{d}-{a}" for d, a in bins)])
    assert indexio.load_type(prefix + ".type")["multiple_bin"] == bins
    dv, dh, do = indexio.read_index_files(prefix)
    assert np.array_equal(dv, v) and np.array_equal(dh, hh) and np.array_equal(do, o)
    out = str(tmp_path / "out.tsv")
    cli(["query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", prefix, "-o", out])
    q = st.read_compact_structure(Q4CHA)
    _, want = fq.query_pdb(ctx, ix, batch, structs, [os.path.join(os.path.dirname(SER[0]), os.path.basename(p)) for p in SER], nres, plddt, q,
                           "B57,B102,C195", hash_type=htype, multiple_bins=bins, sort_by="")
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert [r[1:] for r in rows] == [fq.format_match_row(m).split("\t")[1:] for m in want] and len(rows) > 0
    with pytest.raises(Exception):
        fd.FolddiscoIndex.build(ctx, batch, multiple_bins=[(16, 0)])


@pytest.mark.parametrize("htype,nbd,nba", [(2, 0, 0), (4, 0, 0), (5, 0, 0), (6, 0, 0), (2, 6, 4), (4, 12, 6), (5, 10, 5), (6, 12, 3)])
def test_own_descriptor_encodings(env, htype, nbd, nba):
    """§8f rank 3, the encodings with their own pair descriptors — 2 TrRosetta (CB-distance cutoff, five angles), 4 PointPairFeature
    (orientation-dependent acceptance), 5 TertiaryInteraction and 6 Hybrid (chain-interior residues, neighbouring CA atoms): raw
    hash lists in the reference's order, index bytes, query map, count records and matches equal the CPU restatement.  For 5 / 6
    the retrieval prefilter (which panics in the reference for <= 200 query hashes) scans every pair on both sides."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, _ix, nres, plddt, tids = env
    ostructs = [oracle.read_pdb(p) for p in SER]
    std = np.concatenate([s.resname_std() for s in structs])
    with oracle.hash_type(htype):
        h, off = fd.get_geometric_hash_as_u32(ctx, batch, nbin_dist=nbd, nbin_angle=nba, sort_dedup=False, hash_type=htype)
        for s, ost in enumerate(ostructs):
            assert np.array_equal(h[int(off[s]):int(off[s + 1])], oracle.hash_structure(ost, nbin_dist=nbd, nbin_angle=nba)), f"structure {s}"
        ix = fd.FolddiscoIndex.build(ctx, batch, nbin_dist=nbd, nbin_angle=nba, hash_type=htype)
        oix, onres, _ = oracle.build_index(ostructs, nbin_dist=nbd, nbin_angle=nba)
        v, hh, o = ix.export()
        assert np.array_equal(hh, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
        pen = fd.length_penalty(onres, 0.5)
        for qpath, qstr in ((Q4CHA, "B57,B102,C195"), (Q1G2F, "F207:C,F212,F225:HX,F229"), (Q4CHA, "B57,B102,C195,B58,B59")):
            oq = oracle.read_pdb(qpath)
            om_ = oracle.make_query_map(oq, qstr, oix, 5.0, dist_thr=(0.5, 1.0), angle_thr=(5.0, 10.0), nbin_dist=nbd, nbin_angle=nba)
            om = om_.arrays()
            q = st.read_compact_structure(qpath)
            res = fq.parse_query_string(qstr, q.chains[0])
            idx = [q.get_index(c, r) for c, r, _ in res]
            qb = ctx.upload(fd.PackedStructures.concat([q.as_item()]))
            m = fq.make_query_map(ctx, qb, idx, [x for _, _, x in res], ix, 5.0, dist_thr=(0.5, 1.0), angle_thr=(5.0, 10.0), nbin_dist=nbd,
                                  nbin_angle=nba, hash_type=htype)
            assert np.array_equal(m.hash, om["hash"]) and np.array_equal(m.qi, om["qi"]) and np.array_equal(m.qj, om["qj"])
            assert np.array_equal(m.is_primary, om["is_primary"]) and np.array_equal(m.idf.view(np.uint32), om["idf"].view(np.uint32))
            assert np.array_equal(m.aad_aa1, om["aad_aa1"]) and np.array_equal(m.aad_dist.view(np.uint32), om["aad_dist"].view(np.uint32))
            got = fd.count_query(ctx, ix, m.hash, m.qi, m.qj, pen)
            ref = oracle.count_query(om_, oix, onres)
            assert [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in got] == \
                   [(r["nid"], r["total_match_count"], r["node_count"], r["edge_count"]) for r in ref]
            for ca_cut in (1.0, 3.0):
                ms = fq.retrieve(ctx, batch, std, np.arange(5, dtype=np.uint32), m, qb, ca_distance_cutoff=ca_cut, nbin_dist=nbd, nbin_angle=nba,
                                 hash_type=htype)
                for nid in range(5):
                    R = oracle.retrieve(ostructs[nid], oq, om_, ca_distance_cutoff=ca_cut, nbin_dist=nbd, nbin_angle=nba)
                    mine = [g for g in ms if g["cand"] == nid]
                    assert len(mine) == len(R["processed"]), (htype, qstr, ca_cut, nid)
                    for g, rp, rh in zip(mine, R["processed"], R["from_hash"]):
                        assert g["processed"] == [-1 if x is None else x[2] for x in rp["residues"]]
                        assert g["from_hash"] == [-1 if x is None else x[2] for x in rh["residues"]]
                        assert abs(g["rmsd"] - rp["rmsd"]) <= 1e-4 and g["idf"] == pytest.approx(rp["idf"], rel=1e-6)


@pytest.mark.parametrize("alias,name", [("trrosetta", "TrRosetta"), ("ppf", "PointPairFeature"), ("3di", "TertiaryInteraction"), ("hybrid", "Hybrid")])
def test_cli_own_descriptor_encodings(env, alias, name, tmp_path):
    """`index -y <alias>` / `query` round trip for the encodings with their own descriptors: PREFIX.type carries the HashType name
    (HashType::get_with_str aliases, geometry/core.rs:42-56), the query reads it back and prints what the API path returns."""
    from folddisco_amd import indexio
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from folddisco_amd.__main__ import main as cli
    from folddisco_amd._lib import hash_type_index
    import folddisco_amd as fd
    ctx, structs, batch, _ix, nres, plddt, tids = env
    prefix = str(tmp_path / "ix")
    cli(["index", "-p", os.path.dirname(SER[0]), "-i", prefix, "-y", alias])
    assert indexio.load_type(prefix + ".type")["hash_type"] == name
    htype = hash_type_index(name)
    ix = fd.FolddiscoIndex.build(ctx, batch, hash_type=htype)
    v, hh, o = ix.export()
    dv, dh, do = indexio.read_index_files(prefix)
    assert np.array_equal(dv, v) and np.array_equal(dh, hh) and np.array_equal(do, o)
    out = str(tmp_path / "out.tsv")
    cli(["query", "-p", Q4CHA, "-q", "B57,B102,C195", "-i", prefix, "-o", out, "--ca-distance", "1.5"])
    q = st.read_compact_structure(Q4CHA)
    _, want = fq.query_pdb(ctx, ix, batch, structs, [os.path.join(os.path.dirname(SER[0]), os.path.basename(p)) for p in SER], nres, plddt, q,
                           "B57,B102,C195", hash_type=htype, sort_by="", ca_distance=1.5)
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert [r[1:] for r in rows] == [fq.format_match_row(m).split("\t")[1:] for m in want]
    assert any("B57,B102,C195" == r[4] for r in rows)          # 4cha matches itself under every encoding


@pytest.mark.gpu
def test_analyze_enrichment_against_restatement(tmp_path):
    """`analyze -i PREFIX -p DIR` (enrichment branch, src/controller/summary.rs:262-628): the structure set's encodings against the index
    as background.  Checked against an independent restatement built from the ORACLE's per-pair hashes: structures per encoding,
    hypergeometric p-values (log-space pmf with the reference's Stirling log-factorial, summed over the whole tail), the enriched set in
    p-value order, the residue pairs carrying every enriched encoding, the supported positions and the per-structure query strings."""
    import math
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    d = tmp_path / "db"
    d.mkdir()
    for p in SER:
        shutil.copy(p, d / os.path.basename(p))
    qd = tmp_path / "set"
    qd.mkdir()
    sel = [p for p in SER if os.path.basename(p) in ("1pq5.pdb", "4cha.pdb")]
    for p in sel:
        shutil.copy(p, qd / os.path.basename(p))
    pre = str(tmp_path / "ix")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", str(d), "-i", pre], cwd=tmp_path, env=env)
    out = str(tmp_path / "enr")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "analyze", "-i", pre, "-p", str(qd), "-o", out, "--p-value", "0.3",
                           "--min-support", "2", "--max-pos", "6", "-t", "4"], cwd=tmp_path, env=env)
    # ---- restatement
    from folddisco_amd import indexio
    _, bgh, off = indexio.read_index_files(pre)
    bgc = dict(zip(bgh.tolist(), np.diff(off.astype(np.int64)).tolist()))
    paths = sorted(str(qd / os.path.basename(p)) for p in sel)
    ostructs = [oracle.read_pdb(p) for p in paths]
    per_struct, pos_of = [], {}
    for s, st_ in enumerate(ostructs):
        a = st_.arrays()
        lab = [f"{chr(int(c))}{int(r)}" for c, r in zip(a["chain"], a["serial"])]
        hs = set()
        n = st_.n
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                r = oracle.pair_hash(st_, i, j)
                if r is None:
                    continue
                hs.add(int(r[0]))
                pos_of.setdefault(int(r[0]), []).append((s, lab[i], lab[j]))
        per_struct.append(hs)
    qcount = {}
    for hs in per_struct:
        for h in hs:
            qcount[h] = qcount.get(h, 0) + 1
    tq, tb = sum(qcount.values()), sum(bgc.values())

    def lf(n):
        if n <= 1:
            return 0.0
        if n < 20:
            return sum(math.log(i) for i in range(2, n + 1))
        return n * math.log(n) - n + 0.5 * math.log(2.0 * math.pi * n)

    def lb(n, k):
        if k > n:
            return -math.inf
        if k == 0 or k == n:
            return 0.0
        return lf(n) - lf(k) - lf(n - k)
    want = []
    for h in sorted(qcount):
        x, K, N = qcount[h], bgc.get(h, 0) + qcount[h], tb + tq
        pv = 0.0
        for i in range(x, min(tq, K) + 1):
            t = lb(K, i) + lb(N - K, tq - i) - lb(N, tq)
            pv += math.exp(t) if t > -745 else 0.0
        pv = min(pv, 1.0)
        if pv < 0.3:
            want.append((h, pv))
    want.sort(key=lambda t: t[1])
    rows = [l.rstrip("\n").split("\t") for l in open(out + "_enriched_hashes.tsv")][1:]
    assert len(rows) == len(want) and len(rows) > 10
    got_p = {int(r[0]): float(r[1]) for r in rows}
    for h, pv in want:
        assert got_p[h] == pytest.approx(pv, rel=1e-3), h          # four printed decimals
    assert [float(r[1]) for r in rows] == sorted(float(r[1]) for r in rows)
    for r in rows[:50]:
        h = int(r[0])
        assert r[3].split(",") == [f"{paths[s]}-{a}-{b}" for s, a, b in pos_of[h]], h
    # positions: count of enriched encodings per (structure, residue) above --min-support, query strings of at most --max-pos residues
    cnt = {}
    for h, _ in want:
        for s, a, b in pos_of[h]:
            cnt[(s, a)] = cnt.get((s, a), 0) + 1
            cnt[(s, b)] = cnt.get((s, b), 0) + 1
    prow = [l.rstrip("\n").split("\t") for l in open(out + "_enriched_positions.tsv")][1:]
    assert {(r[0], r[1]): int(r[2]) for r in prow} == {(paths[s], a): c for (s, a), c in cnt.items() if c > 2}
    qrow = [l.rstrip("\n").split("\t") for l in open(out + "_query_summary.tsv")][1:]
    assert len(qrow) >= 1 and all(1 <= len(r[1].split(",")) <= 6 for r in qrow)
    for r in qrow:
        s = paths.index(r[0])
        assert all(cnt[(s, pos)] > 2 for pos in r[1].split(","))


@pytest.mark.gpu
def test_cli_index_and_query_from_foldcomp_database(tmp_path):
    """`index -p <Foldcomp DB>` (cli/workflows/build_index.rs:109-123): entries in key order, names from DB.lookup, db_key column,
    .type naming the database; the index equals the oracle's over the same decoded structures (decoder pinned against the reference's
    own in tests/test_foldcomp.py); the query reads the hit coordinates back from the database by db_key and accepts DB:NAME as the
    query structure (controller/io.rs:303-333)."""
    import shutil
    import subprocess
    import sys
    from folddisco_amd import indexio
    from folddisco_amd import structure as st
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "foldcomp")
    d = tmp_path / "data"
    d.mkdir()
    for ext in ("", ".index", ".lookup", ".dbtype"):
        shutil.copy(os.path.join(g, "example_db" + ext), d / ("scop_foldcomp" + ext))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/scop_foldcomp", "-t", "4"], cwd=tmp_path, env=env)
    pre = str(d / "scop_folddisco")                       # X_foldcomp -> X_folddisco (controller/io.rs:460-470)
    fc = st.FoldcompDb(str(d / "scop_foldcomp"))
    tids, nres, plddt, keys = indexio.load_lookup(pre + ".lookup")
    assert tids == fc.names and keys.tolist() == fc.keys.tolist()
    cfg = indexio.load_type(pre + ".type")
    assert cfg["input_format"] == "FCZDB" and cfg["foldcomp_db"] == "data/scop_foldcomp" and cfg["chunk_size"] == 24
    structs, ok = st.read_compact_structures(fc.keys, threads=4, foldcomp=fc)
    assert ok.all() and nres.tolist() == [s.n for s in structs]
    import folddisco_amd as fd
    ostructs = packed_to_oracle_structs(fd.PackedStructures.concat([s.as_item() for s in structs]))
    oix, onres, oplddt = oracle.build_index(ostructs)
    v, h, o = indexio.read_index_files(pre)
    assert np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
    assert np.array_equal(plddt, np.array([np.float32(indexio.format_f32_display(s.avg_plddt())) for s in structs], np.float32))
    # chunked build over the database: byte-identical files
    pre2 = str(tmp_path / "chunked")
    subprocess.check_call([sys.executable, "-m", "folddisco_amd", "index", "-p", "data/scop_foldcomp", "-i", pre2, "--chunk", "7"], cwd=tmp_path, env=env)
    for ext in ("", ".offset", ".lookup", ".type"):
        assert open(pre + ext, "rb").read() == open(pre2 + ext, "rb").read(), ext
    # query: an entry of the database against the index built from it — itself first, all residues, RMSD 0, db_key printed
    s0 = structs[3]
    pick = [0, 5, 9]
    qstr = ",".join(f"{chr(s0.chain[i])}{int(s0.serial[i])}" for i in pick)
    out = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", f"data/scop_foldcomp:{fc.names[3]}", "-q", qstr, "-i", pre,
                          "--format-output", "tid,db_key,node_count,rmsd,matching_residues"], cwd=tmp_path, env=env, capture_output=True, text=True,
                         check=True).stdout.splitlines()
    assert out[0] == f"{fc.names[3]}\t{int(fc.keys[3])}\t3\t0.0000\t{qstr}"
    ps = subprocess.run([sys.executable, "-m", "folddisco_amd", "query", "-p", f"data/scop_foldcomp:{fc.names[3]}", "-q", qstr, "-i", pre, "--per-structure"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, check=True).stdout.splitlines()
    f = ps[0].split("\t")
    assert f[0] == fc.names[3] and f[10] == str(int(fc.keys[3])) and f[9].startswith(qstr + ":0.0000")


@pytest.mark.gpu
def test_device_retrieval_glue_equals_host_glue(env, monkeypatch):
    """fdgpu_retrieve_batch's device path (k_retrieve.hip: components / votes / assignment / rescue one wavefront per candidate,
    superposition on the device) against the host C++ path it replaces (FDGPU_HOST_GLUE=1, itself pinned to the oracle by
    test_retrieve_matches_oracle): identical match tables, bit for bit, on the serine set with substitution and range queries and on
    a synthetic database with planted motifs; the stage timings prove which path ran."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    from folddisco_amd import synth
    from folddisco_amd.api import count_query_batch, length_penalty
    ctx, structs, batch, ix, nres, plddt, tids = env
    q1, q2 = st.read_compact_structure(Q4CHA), st.read_compact_structure(Q1G2F)
    qall = ctx.upload(fd.PackedStructures.concat([q1.as_item(), q2.as_item()]))
    std = np.concatenate([s.resname_std() for s in structs])
    specs = [(0, q1, "B57,B102,C195", [0, 1, 2, 3, 4]), (1, q2, "F207,F212,F225,F229", [4, 2, 0]), (0, q1, "B57:HKR,B102,C195:ST", [3, 4, 1]),
             (0, q1, "B57,B102,C195,B58,B59,C999", [0, 1, 2, 3, 4]), (0, q1, "B57-60,B102,C195", [0, 3])]
    reqs = []
    for sidx, q, qstr, cand in specs:
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        pairs = [(i, s) for i, s in pairs if i is not None]
        reqs.append((sidx, [i for i, _ in pairs], [s for _, s in pairs]))
    qms = fq.make_query_maps(ctx, qall, reqs, ix, 5.0)

    def run(db, stdn, cl, qms_, qb, qs, ca):
        ctx.enable_timing(True)
        out = fq.retrieve_batch(ctx, db, stdn, cl, qms_, qb, qs, ca_distance_cutoff=ca, as_arrays=True)
        ctx.synchronize()
        names = [n for n, _, _ in ctx.last_timings()]
        ctx.enable_timing(False)
        return out, names

    def both(db, stdn, cl, qms_, qb, qs, ca=1.0):
        monkeypatch.delenv("FDGPU_HOST_GLUE", raising=False)
        dev, n_dev = run(db, stdn, cl, qms_, qb, qs, ca)
        monkeypatch.setenv("FDGPU_HOST_GLUE", "1")
        host, n_host = run(db, stdn, cl, qms_, qb, qs, ca)
        monkeypatch.delenv("FDGPU_HOST_GLUE", raising=False)
        assert "retrieve_slots" in n_dev and "retrieve_slots" not in n_host
        for a, b, what in zip(dev, host, ("matches", "match_off", "residues", "res_off")):
            assert a.tobytes() == b.tobytes(), what
        # the records' order made on the device (slot bases from record counts) against the host's sort of the record headers,
        # and the slots launched heaviest first against slot order
        # ... and the split glue (set-up per slot + a wavefront per component, the default) against k_rs_slots alone
        for env_name, val in (("FDGPU_RS_HOST_ORDER", "1"), ("FDGPU_RS_ORDER", "0"), ("FDGPU_MP_ITEMS", "0"), ("FDGPU_RS_SPLIT", "0")):      # (MP_ITEMS: work items built on the host)
            monkeypatch.setenv(env_name, val)
            alt, _ = run(db, stdn, cl, qms_, qb, qs, ca)
            monkeypatch.delenv(env_name)
            for a, b, what in zip(dev, alt, ("matches", "match_off", "residues", "res_off")):
                assert a.tobytes() == b.tobytes(), (env_name, what)
        return dev

    for ca in (1.0, 1.5, 3.0):
        m = both(batch, std, [np.array(sp[3], np.uint32) for sp in specs], qms, qall, [sp[0] for sp in specs], ca)[0]
        assert len(m) >= 6
    # a candidate beyond the kernel's limits (here: the node limit lowered to 2) raises the overflow flag and the call falls back to
    # the host path as a whole: same tables again
    monkeypatch.setenv("FDGPU_RS_NODE_CAP", "2")
    over, names = run(batch, std, [np.array(sp[3], np.uint32) for sp in specs], qms, qall, [sp[0] for sp in specs], 1.0)
    monkeypatch.delenv("FDGPU_RS_NODE_CAP")
    ref_tables, _ = run(batch, std, [np.array(sp[3], np.uint32) for sp in specs], qms, qall, [sp[0] for sp in specs], 1.0)
    assert names == ["match_pairs"]       # the host path's own scan restarted the stage list: the device attempt was abandoned
    for a, b in zip(over, ref_tables):
        assert a.tobytes() == b.tobytes()
    # planted motifs in a synthetic database: 24 queries x their top 24 candidates
    S = 1500
    d = synth.generate(S, seed=91)
    ps = synth.to_packed(d)
    sb = ctx.upload(ps)
    six = fd.FolddiscoIndex.build(ctx, sb)
    rng = np.random.Generator(np.random.PCG64(5))
    off = ps.res_off.astype(np.int64)
    queries = []
    while len(queries) < 24:
        s = int(rng.integers(0, S))
        a, b = int(off[s]), int(off[s + 1])
        if b - a < 30:
            continue
        c = int(rng.integers(0, b - a))
        near = np.nonzero(np.linalg.norm(ps.ca_xyz[a:b] - ps.ca_xyz[a + c], axis=1) < 10.0)[0]
        if len(near) < 5:
            continue
        idx = sorted(rng.choice(near, 5, replace=False).tolist())
        item = dict(n_xyz=ps.n_xyz[a:b], ca_xyz=ps.ca_xyz[a:b], cb_xyz=ps.cb_xyz[a:b], aa=ps.aa[a:b], cb_ok=None)
        queries.append((s, idx, item))
    qb2 = ctx.upload(fd.PackedStructures.concat([it for _, _, it in queries]))
    qms2 = fq.make_query_maps(ctx, qb2, [(k, queries[k][1]) for k in range(len(queries))], six, float(S))
    pen = length_penalty(np.diff(ps.res_off.astype(np.int64)).astype(np.uint64), 0.5)
    recs = count_query_batch(ctx, six, [(qm.hash, qm.qi, qm.qj) for qm in qms2], pen, total_structures=S, top_n=24)
    from folddisco_amd import dist as fdist
    cl = [fdist.rank_hits(r, 24)["nid"].astype(np.uint32) for r in recs]
    m2 = both(sb, None, cl, qms2, qb2, list(range(len(queries))))[0]
    assert len(m2) >= len(queries)          # every query finds at least itself


@pytest.mark.gpu
def test_count_query_maps_equals_count_query_batch(env):
    """fdgpu_count_query_maps_top (query maps handed back to the library as they are, posting lengths + idf + scoring in one call,
    penalty resident on the device) == the posting_lengths / idf_of_lengths / count_query_batch route, byte for byte; with and without
    the top-N selection; a query with no hash in the index gives no records."""
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    q1, q2 = st.read_compact_structure(Q4CHA), st.read_compact_structure(Q1G2F)
    qall = ctx.upload(fd.PackedStructures.concat([q1.as_item(), q2.as_item()]))
    reqs = []
    for sidx, q, qstr in [(0, q1, "B57,B102,C195"), (1, q2, "F207,F212,F225,F229"), (0, q1, "B57:HKR,B102,C195:ST"), (0, q1, "B57-62")]:
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        reqs.append((sidx, [i for i, _ in pairs], [s for _, s in pairs]))
    qms = fq.make_query_maps(ctx, qall, reqs, ix, 5.0)
    pen = fd.length_penalty(nres, 0.5)
    for top_n in (0, 3):
        want = fd.count_query_batch(ctx, ix, [(m.hash, m.qi, m.qj) for m in qms], pen, total_structures=5, top_n=top_n)
        got = fd.count_query_maps(ctx, ix, qms, pen, total_structures=5, top_n=top_n)
        ix.set_penalty(pen)
        res_pen = fd.count_query_maps(ctx, ix, qms, None, total_structures=5, top_n=top_n)
        ix.set_penalty(None)
        assert len(got) == len(want) == 4 and sum(len(w) for w in want) > 0
        for g, w, r in zip(got, want, res_pen):
            assert g.tobytes() == w.tobytes() and r.tobytes() == w.tobytes()
    with pytest.raises(Exception):
        fd.count_query_maps(ctx, ix, qms, None, total_structures=5)          # no penalty given and none resident


@pytest.mark.gpu
def test_two_contexts_share_one_resident_index(env):
    """Two host threads, one context (stream + workspaces) each, query the SAME resident index and coordinate batch at the same
    time (the bench's multi-thread leg; how rayon workers would call the ABI): every result equals the serial one."""
    import threading
    import folddisco_amd as fd
    from folddisco_amd import query as fq
    from folddisco_amd import structure as st
    ctx, structs, batch, ix, nres, plddt, tids = env
    q1, q2 = st.read_compact_structure(Q4CHA), st.read_compact_structure(Q1G2F)
    qall = ctx.upload(fd.PackedStructures.concat([q1.as_item(), q2.as_item()]))
    std = np.concatenate([s.resname_std() for s in structs])
    reqs = []
    for sidx, q, qstr in [(0, q1, "B57,B102,C195"), (1, q2, "F207,F212,F225,F229"), (0, q1, "B57:HKR,B102,C195:ST")]:
        res = fq.parse_query_string(qstr, q.chains[0])
        pairs = [(q.get_index(c, r), s) for c, r, s in res]
        reqs.append((sidx, [i for i, _ in pairs], [s for _, s in pairs]))
    ix.set_penalty(fd.length_penalty(nres, 0.5))

    def once(cx):
        qms = fq.make_query_maps(cx, qall, reqs, ix, 5.0)
        recs = fd.count_query_maps(cx, ix, qms, None, total_structures=5, top_n=5)
        cl = [r["nid"].astype(np.uint32) for r in recs]
        tabs = fq.retrieve_batch(cx, batch, std, cl, qms, qall, [r[0] for r in reqs], as_arrays=True)
        return b"".join(r.tobytes() for r in recs) + b"".join(t.tobytes() for t in tabs)

    want = once(ctx)
    ctx2 = fd.Context(0)
    out, errs = {}, []

    def worker(name, cx):
        try:
            out[name] = [once(cx) for _ in range(25)]
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=("a", ctx)), threading.Thread(target=worker, args=("b", ctx2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ix.set_penalty(None)
    ctx2.close()
    assert not errs, errs
    assert all(x == want for x in out["a"]) and all(x == want for x in out["b"])


def test_cli_indexes_structures_between_dash_n_and_65535_like_the_reference(tmp_path):
    """`index -n 50000` (the default): the reference never applies -n — its skip test reads the hard-wired 65535 (src/controller/mod.rs:40,124,313;
    set_max_residue mod.rs:189 has no caller) and -n only lands in PREFIX.type (src/cli/main.rs:42, build_index.rs:222).  Structures whose raw
    residue count is 50,002 and 65,535 must therefore have postings, one of 65,536 must keep its id with nres 0 / plddt 0 and no postings.
    Expected index: the oracle's pair feature + hash on the pairs of the sparse layout (residue k pairs with k ^ 1 and with nothing else:
    checked against the oracle's full O(R^2) enumeration on the small structure), posting lists built by the oracle's table builder."""
    import subprocess
    import sys
    from folddisco_amd import indexio
    from tests.helpers import write_sparse_pdb
    d = tmp_path / "big"
    d.mkdir()
    sizes = [600, 50001, 65534, 65535]                       # raw residue counts 601, 50002, 65535, 65536
    for k, n in enumerate(sizes):
        write_sparse_pdb(str(d / f"s{k}.pdb"), n, seed=10 + k)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env_ = dict(os.environ, PYTHONPATH=root)
    pre = str(tmp_path / "big_folddisco")
    r = subprocess.run([sys.executable, "-m", "folddisco_amd", "index", "-p", str(d), "-i", pre, "-t", "4", "--residue", "50000"], cwd=tmp_path, env=env_,
                       capture_output=True, text=True, check=True)
    assert r.stderr.count("has too many residues. Skipping") == 1 and "s3.pdb has too many residues" in r.stderr
    # expected lists
    lists, nres_want, plddt_want = [], [], []
    for k, n in enumerate(sizes):
        s = oracle.read_pdb(str(d / f"s{k}.pdb"))
        raw = s.ptr.contents.num_residues_raw
        assert raw == n + 1 and s.n == n
        if raw > 65535:
            lists.append(np.zeros(0, np.uint32))
            nres_want.append(0)
            plddt_want.append(np.float32(0.0))
            continue
        hs = []
        for i in range(n):
            ph = oracle.pair_hash(s, i, i ^ 1) if (i ^ 1) < n else None
            if ph is not None:
                hs.append(ph[0])
        h = np.unique(np.array(hs, np.uint32))
        if k == 0:
            assert np.array_equal(h, np.unique(oracle.hash_structure(s)))      # the layout's claim: no pair but (k, k ^ 1) inside the cutoff
        lists.append(h)
        nres_want.append(n)
        plddt_want.append(np.float32(s.avg_plddt()))
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    oix = oracle.build_index_from_lists(np.concatenate(lists), off)
    v, h, o = indexio.read_index_files(pre)
    assert np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets()) and np.array_equal(v, oix.values())
    assert len(lists[1]) > 100 and len(lists[2]) > 100
    ids = set()
    for hh in oix.hashes()[:: max(1, oix.H // 64)]:
        ids |= set(int(x) for x in oix.entries(int(hh)))
    assert ids <= {0, 1, 2} and {1, 2} <= ids                 # ids 1 and 2 (50,002 and 65,535 raw residues) are in the posting lists, id 3 never
    look = [l.rstrip("\n").split("\t") for l in open(pre + ".lookup")]
    assert [int(l[2]) for l in look] == nres_want
    assert [np.float32(float(l[3])).tobytes() for l in look] == [np.float32(x).tobytes() for x in plddt_want]
    assert indexio.load_type(pre + ".type")["max_residue"] == 50000


def test_one_index_from_n_ranks_by_hash_ranges(tmp_path):
    """SURVEY §8e row 2, Option A (csrc/fd_shard_index.hip): sub-indices over consecutive id ranges (what N ranks hold after a build sharded by structure) ->
    hash bounds of equal posting bytes -> every part sliced at the bounds -> the pieces of a range concatenated per hash on the device (fdgpu_index_merge) ->
    every range written into its regions of PREFIX / PREFIX.offset (fdgpu_index_save_part): byte-identical to the files of ONE build, for 2, 3 and 8 hand-made
    ranks (ragged id ranges, one rank without structures), with degenerate bounds (empty ranges), and — world of one — through the RCCL entry point
    fdgpu_comm_single_index with its ncclSend / ncclRecv to itself."""
    import folddisco_amd as fd
    from folddisco_amd import dist as fdist
    from tests.helpers import synthetic_packed
    ctx = fd.Context(0)
    ps = synthetic_packed(700, seed=91)
    whole = fd.FolddiscoIndex.build(ctx, ctx.upload(ps))
    whole.save(str(tmp_path / "one"))
    want = [open(tmp_path / ("one" + ext), "rb").read() for ext in ("", ".offset")]
    H, V = whole.num_hashes, whole.value_len
    assert len(want[0]) == V and len(want[1]) == 8 + 4 * H + 8 * (H + 1)

    def sub(a, b):
        sl = slice(int(ps.res_off[a]), int(ps.res_off[b]))
        chunk = fd.PackedStructures((ps.res_off[a:b + 1] - ps.res_off[a]).astype(np.uint64), ps.n_xyz[sl], ps.ca_xyz[sl], ps.cb_xyz[sl], ps.aa[sl])
        return fd.FolddiscoIndex.build(ctx, ctx.upload(chunk), first_id=a)

    def run(cuts, edges_override=None, tag="x"):
        parts = [sub(a, b) for a, b in zip(cuts, cuts[1:])]
        W = len(parts)
        b0 = parts[0].range_bounds(W) if edges_override is None else np.array(edges_override, np.uint32)
        assert len(b0) == W - 1 and np.all(np.diff(b0.astype(np.int64)) >= 0)
        edge = [0] + [int(x) for x in b0] + [1 << 32]
        pre = str(tmp_path / tag)
        for ext in ("", ".offset"):
            if os.path.exists(pre + ext):
                os.remove(pre + ext)
        ranges = []
        for j in range(W):
            pieces = [p.slice(edge[j], edge[j + 1]) for p in parts]
            ranges.append(fd.FolddiscoIndexSet(pieces).merge() if W > 1 else pieces[0])
        assert sum(r.num_hashes for r in ranges) == H and sum(r.value_len for r in ranges) == V
        assert sum(r.num_postings for r in ranges) == whole.num_postings
        hb = vb = 0
        order = list(range(W))[::-1]            # ranks write concurrently in any order: here the last range first
        pos = []
        for r in ranges:
            pos.append((hb, vb)); hb += r.num_hashes; vb += r.value_len
        for j in order:
            ranges[j].save_part(pre, pos[j][0], pos[j][1], H, V, write_header=(j == 0), is_last=(j == W - 1))
        got = [open(pre + ext, "rb").read() for ext in ("", ".offset")]
        assert got[0] == want[0] and got[1] == want[1], (cuts, edges_override)
        return [r.value_len for r in ranges]
    sizes = run([0, 350, 700], tag="w2")
    assert min(sizes) > 0.3 * V / 2                                           # ranges of about equal posting bytes
    run([0, 100, 460, 700], tag="w3")
    sizes8 = run([0, 50, 50, 200, 290, 400, 555, 640, 700], tag="w8")          # a rank without structures
    assert max(sizes8) < 3 * V / 8
    run([0, 350, 700], edges_override=[0], tag="d0")                           # first range empty
    run([0, 233, 466, 700], edges_override=[0xffffffff, 0xffffffff], tag="d1")  # everything in the first range
    mid = int(whole.range_bounds(2)[0])
    run([0, 233, 466, 700], edges_override=[mid, mid], tag="d2")               # an empty range in the middle
    # the RCCL entry point with a world of one: bounds, slices, send / recv (to itself), merge, sizes — every line the N-rank call runs
    comm = fdist.Comm(ctx)
    a0, g0 = comm.stats()
    rng, hb, vb, ht, vt = comm.single_index(whole)
    assert (hb, vb, ht, vt) == (0, 0, H, V) and comm.stats()[1] == g0 + 3
    rng.save_part(str(tmp_path / "rccl"), hb, vb, ht, vt, write_header=True, is_last=True)
    assert [open(tmp_path / ("rccl" + ext), "rb").read() for ext in ("", ".offset")] == want
    comm.close()
    ctx.close()
