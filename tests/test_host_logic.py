"""CPU-only tests (no GPU): the C ABI library loads and exports every declared symbol, host-side logic
(ingest, query grammar, ranking/sharding) agrees with the oracle, and the device arithmetic headers — compiled for
the host — are bit-identical to glibc / the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from tests.helpers import Q1G2F, Q4CHA, SER

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from folddisco_amd import _lib
    from folddisco_amd import build as fb
    fb.build()
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "fdgpu.h")).read()
    public = set(re.findall(r"\b(fdgpu_[a-z0-9_]+)\s*\(", header))
    debug = set(re.findall(r"\b(fdgpu_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "fdgpu_debug.h")).read()))
    declared = public | debug
    assert len(public) >= 30 and debug and all(n.startswith("fdgpu_debug_") for n in debug) and not any(n.startswith("fdgpu_debug_") for n in public)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared <= bound, f"unbound: {declared - bound}"
    assert b"gfx950" in L.fdgpu_version()
    # the reference-side binding a maintainer would add (INTEGRATION.md's extern "C" block) names every public entry point
    rust = set(re.findall(r"\bfn (fdgpu_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert public <= rust, f"not in INTEGRATION.md: {sorted(public - rust)}"
    assert not (rust - public), f"INTEGRATION.md binds what include/fdgpu.h does not declare: {sorted(rust - public)}"


def test_no_gpu_means_loud_failure_not_fallback():
    import folddisco_amd as fd
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fd.FdgpuError):
        fd.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "folddisco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "fd_oracle.h" not in src and "libfdoracle" not in src, f


def test_ingest_matches_oracle_parser():
    from folddisco_amd import structure as st
    for p in SER + [Q1G2F]:
        a = st.read_compact_structure(p)
        o = oracle.read_pdb(p).arrays()
        for k in ("n_xyz", "ca_xyz", "cb_xyz", "cb_ok", "aa", "chain", "serial", "bfac"):
            assert np.array_equal(getattr(a, k), o[k]), (p, k)
        assert np.float32(a.avg_plddt()).view(np.uint32) == np.float32(oracle.read_pdb(p).avg_plddt()).view(np.uint32)
    s = st.read_compact_structure(SER[4])
    assert s.n == 477 and "%.4f" % s.avg_plddt() == "13.5404"   # README.md:237


def test_query_grammar_matches_oracle():
    from folddisco_amd import query as fq
    for q, dc in [("A250,A232,A269", ord("A")), ("250,232,269", ord("C")), ("A1-3,B5", ord("C")), ("164:H,195,247:ND,297:Xp", ord("A")),
                  ("B57 , B102:a,C195", ord("Q")), ("11:X", ord("1")), ("", ord("A"))]:
        _, ref = oracle.parse_query_string(q, dc)
        got = fq.parse_query_string(q, dc)
        assert [(c, r) for c, r, _ in got] == [(c, r) for c, r, _ in ref]
        assert [s for _, _, s in got] == [s for _, _, s in ref]
    with pytest.raises(ValueError):
        fq.parse_query_string("A12,,B3")
    with pytest.raises(ValueError):
        oracle.parse_query_string("A12,,B3")


def test_shard_ranges_and_ranking():
    from folddisco_amd import dist as fdist
    for S, N in [(542000, 8), (7, 3), (5, 8), (0, 2)]:
        rs = [fdist.shard_range(r, N, S) for r in range(N)]
        assert rs[0][0] == 0 and rs[-1][1] == S and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    recs = np.zeros(5, fdist.REC_DTYPE)
    recs["nid"] = [4, 3, 1, 2, 0]
    recs["idf"] = [0.5, 0.9, 0.5, 0.1, 0.5]
    assert list(fdist.rank_hits(recs)["nid"]) == [3, 0, 1, 4, 2]       # idf desc, ties by nid asc
    assert list(fdist.rank_hits(recs, 2)["nid"]) == [3, 0]


def _gcc(src, out, extra=()):
    subprocess.check_call(["gcc" if src.endswith(".c") else "g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", src, "-o", out, *extra, "-lm"],
                          cwd=ROOT)


def test_device_libm_header_equals_glibc_sampled(tmp_path):
    """folddisco_amd/csrc/fd_libm.h compiled for the host == glibc, on a 1/4096 stride of all float bit patterns and 2M atan2f
    pairs (the exhaustive run — stride 1, 4e9 pairs, 0 mismatches on glibc 2.35 — is `tools/check_libm 1 4000000000`)."""
    exe = str(tmp_path / "check_libm")
    _gcc("tools/check_libm.c", exe)
    out = subprocess.run([exe, "4096", "2000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sinf 0 cosf 0 acosf 0 atanf 0 atan2f 0" in out.stdout


def test_device_geometry_header_equals_oracle(tmp_path):
    """fd_geom.h (direct, shared-subexpression and table forms) compiled for the host == oracle on every residue pair of the
    five serine_peptidases structures and on degenerate random clouds."""
    oracle.build()
    exe = str(tmp_path / "check_geom")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "tools/check_geom.cpp", "-Loracle", "-lfdoracle",
                           f"-Wl,-rpath,{ROOT}/oracle", "-o", exe], cwd=ROOT)
    out = subprocess.run([exe] + SER, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "total mismatches: 0" in out.stdout


def test_bin_tables_are_consistent_with_glibc():
    """spot-check folddisco_amd/csrc/fd_bin_tables.h (generated exhaustively by tools/gen_bin_tables.c) against glibc at and
    around every breakpoint."""
    hdr = open(os.path.join(ROOT, "folddisco_amd", "csrc", "fd_bin_tables.h")).read()
    thr = [int(x, 16) for x in re.search(r"fd_theta_thr_bits\[FD_THETA_NSEG\] = \{([^}]*)\}", hdr).group(1).replace("u", "").split(",")]
    keys = [int(x) for x in re.search(r"fd_theta_key\[FD_THETA_NSEG\] = \{([^}]*)\}", hdr).group(1).split(",")]
    libm = C.CDLL("libm.so.6")
    for f in ("acosf", "sinf", "cosf"):
        getattr(libm, f).restype = C.c_float
        getattr(libm, f).argtypes = [C.c_float]

    def q4(v):
        t = np.float32(np.float32(np.float32(v) + np.float32(1.0)) * np.float32(1.5)) + np.float32(0.5)
        return 0 if not (t == t) or t <= 0 else int(t)

    def key_ref(c):
        a = libm.acosf(float(c))
        return (q4(np.float32(libm.sinf(a))) << 2) | q4(np.float32(libm.cosf(a)))

    def key_tab(c):
        n = sum(1 for t in thr[1:] if c >= np.uint32(t).view(np.float32))
        return keys[n] if -1.0 <= c <= 1.0 else 0

    for t in thr:
        for d in (-2, -1, 0, 1, 2):
            b = np.uint32(t + d if not (t & 0x80000000) else t - d)
            c = b.view(np.float32)
            if -1.0 <= c <= 1.0:
                assert key_tab(c) == key_ref(c), hex(int(b))
    rng = np.random.Generator(np.random.PCG64(1))
    for c in rng.uniform(-1, 1, 20000).astype(np.float32):
        assert key_tab(c) == key_ref(c)


def test_merge_subindices_equals_single_build_oracle():
    """per-shard sub-indices (oracle-built here, so the test runs without a GPU) merge into the byte-identical single index"""
    from folddisco_amd import indexio
    structs = [oracle.read_pdb(p) for p in SER]
    lists = [np.unique(oracle.hash_structure(s)) for s in structs]
    L = oracle.lib()

    def build(ids):
        ix = L.fdo_index_new(30)
        for fn in (L.fdo_index_count_single_entry, L.fdo_index_add_single_entry):
            for i in ids:
                for x in lists[i % 5]:
                    fn(ix, int(x), i)
            if fn is L.fdo_index_count_single_entry:
                L.fdo_index_allocate_entries(ix)
        L.fdo_index_finish(ix)
        o = oracle.OIndex(ix)
        return o.values(), o.hashes(), o.offsets()
    ids = list(range(0, 5)) + list(range(16380, 16390))          # multi-byte deltas across the shard boundary
    whole = build(ids)
    for cut in ([5], [2, 9], [1, 2, 3, 4, 14]):
        bounds = [0] + cut + [len(ids)]
        parts = [build(ids[a:b]) for a, b in zip(bounds, bounds[1:])]
        v, h, o = indexio.merge_subindices(parts)
        assert np.array_equal(v, whole[0]) and np.array_equal(h, whole[1]) and np.array_equal(o, whole[2]), cut
    with pytest.raises(RuntimeError):
        indexio.merge_subindices([build(ids[5:]), build(ids[:5])])


def test_index_files_roundtrip_and_text_files(tmp_path):
    from folddisco_amd import indexio
    structs = [oracle.read_pdb(p) for p in SER]
    oix, nres, plddt = oracle.build_index(structs)
    pre = str(tmp_path / "o")
    oix.save(pre)
    v, h, o = indexio.read_index_files(pre)
    assert np.array_equal(v, oix.values()) and np.array_equal(h, oix.hashes()) and np.array_equal(o, oix.offsets())
    indexio.write_index_files(str(tmp_path / "w"), v, h, o)
    for ext in ("", ".offset"):
        assert open(pre + ext, "rb").read() == open(str(tmp_path / "w") + ext, "rb").read()
    tids = ["data/serine_peptidases/" + os.path.basename(p) for p in SER]
    indexio.save_lookup(pre + ".lookup", tids, nres, plddt)
    # same bytes as the oracle's writer (Rust `{}` float formatting restated on both sides)
    arr = (C.c_char_p * 5)(*[t.encode() for t in tids])
    oracle.lib().fdo_save_lookup((pre + ".lookup2").encode(), arr, nres.ctypes.data_as(oracle.u64p), plddt.ctypes.data_as(oracle.f32p), None, 5)
    assert open(pre + ".lookup").read() == open(pre + ".lookup2").read()
    t2, n2, p2, k2 = indexio.load_lookup(pre + ".lookup")
    assert t2 == tids and np.array_equal(n2, nres) and np.array_equal(p2.view(np.uint32), plddt.view(np.uint32)) and list(k2) == [0, 1, 2, 3, 4]
    indexio.save_type(pre + ".type", 5)
    cfg = indexio.load_type(pre + ".type")
    assert cfg == {"chunk_size": 5, "grid_width": 20.0, "hash_type": "PDBTrRosetta", "input_format": "PDB", "max_residue": 50000,
                   "num_bin_angle": 0, "num_bin_dist": 0}
    for x, want in [(50.0, "50"), (0.0, "0"), (13.540419, "13.540419"), (0.1, "0.1"), (1e-7, "0.0000001"), (float("nan"), "NaN"), (1.5e10, "15000000000")]:
        assert indexio.format_f32_display(x) == want


def test_sort_by_grammar():
    """--sort-by (src/controller/sort.rs:160-204, 400-440; unit tests :512-548, :671-698)"""
    from folddisco_amd.query import parse_sort_by, sort_rows
    assert parse_sort_by("node_count,rmsd", False) == [("node_count", -1), ("rmsd", 1)]
    assert parse_sort_by("", False) == [("idf", -1), ("rmsd", 1)]                     # MatchSortStrategy::default
    assert parse_sort_by("  ", True) == [("idf", -1), ("min_rmsd_with_max_match", 1)]  # StructureSortStrategy::default
    assert parse_sort_by("NODE_COUNT:ASC, score", False) == [("node_count", 1), ("idf", -1)]
    assert parse_sort_by("rmsd,max-node:a", True) == [("min_rmsd_with_max_match", 1), ("max_matching_node_count", 1)]
    assert parse_sort_by("tm_score,chamfer", False) == [("tm_score", -1), ("chamfer_distance", 1)]
    for bad in ("lddt", "rmsd:up", "a:b:c"):
        with pytest.raises(ValueError):
            parse_sort_by(bad, False)
    rows = [dict(node_count=2, rmsd=0.5, idf=1.0, k=0), dict(node_count=3, rmsd=0.9, idf=0.5, k=1), dict(node_count=3, rmsd=0.1, idf=0.2, k=2),
            dict(node_count=3, rmsd=0.1, idf=0.9, k=3)]
    assert [r["k"] for r in sort_rows(list(rows), parse_sort_by("node_count,rmsd", False))] == [2, 3, 1, 0]   # stable on ties
    assert [r["k"] for r in sort_rows(list(rows), parse_sort_by("idf", False))] == [0, 3, 1, 2]


def test_native_ingest_matches_python_and_oracle(tmp_path):
    """csrc/fd_ingest.cpp (fdgpu_parse_structures; host only, no GPU needed) against the Python restatement and the oracle's C
    parser on the reference's PDB fixtures (incl. a gzipped .ent), and mmCIF against PDB for an AlphaFold model, whose two
    files hold the same ATOM records (data/io_test/cif/)."""
    import glob
    import gzip
    import oracle
    from folddisco_amd import structure
    here = os.path.dirname(os.path.abspath(__file__))
    pdbs = sorted(glob.glob(os.path.join(here, "golden", "serine_peptidases", "*.pdb"))) + sorted(glob.glob(os.path.join(here, "golden", "query", "*.pdb")))
    io = os.path.join(here, "golden", "io")
    gz = [os.path.join(io, "1b72a-.ent.gz"), os.path.join(io, "2wnb.pdb.gz"), os.path.join(io, "AF-A0A4S3KKF6-F1-model_v4.pdb.gz")]
    nat, ok = structure.read_compact_structures(pdbs + gz, threads=3)
    assert ok.tolist() == [1] * len(nat)

    def same(a, b):
        return (a.n == b.n and a.n_xyz.tobytes() == b.n_xyz.tobytes() and a.ca_xyz.tobytes() == b.ca_xyz.tobytes() and a.cb_xyz.tobytes() == b.cb_xyz.tobytes()
                and a.cb_ok.tobytes() == b.cb_ok.tobytes() and a.aa.tobytes() == b.aa.tobytes() and list(a.resname) == list(b.resname)
                and a.chain.tobytes() == b.chain.tobytes() and a.serial.tobytes() == b.serial.tobytes() and a.bfac.tobytes() == b.bfac.tobytes()
                and a.num_residues_raw == b.num_residues_raw)
    for p, a in zip(pdbs + gz, nat):
        assert same(a, structure.read_compact_structure(p)), p
        assert a.chains[:1] == structure.read_compact_structure(p).chains[:1]
    for p, a in zip(pdbs, nat):                      # the oracle reads plain files
        os_ = oracle.read_pdb(p)
        o = os_.arrays()
        assert a.n == os_.n and a.ca_xyz.tobytes() == o["ca_xyz"].tobytes() and a.cb_xyz.tobytes() == o["cb_xyz"].tobytes()
        assert a.n_xyz.tobytes() == o["n_xyz"].tobytes() and a.aa.tobytes() == o["aa"].tobytes()
        assert np.float32(a.avg_plddt()).tobytes() == np.float32(os_.avg_plddt()).tobytes()
    # mmCIF: same model, same arrays; HETATM rows are not dropped by the reference's reader (cif.rs:279-287)
    cif, ok2 = structure.read_compact_structures([os.path.join(io, "AF-A0A4S3KKF6-F1-model_v4.cif.gz"), os.path.join(io, "2wnb.cif.gz")], threads=2)
    assert ok2.tolist() == [1, 1] and same(cif[0], nat[-1]) and cif[0].n == 738
    assert cif[1].n == 273 and nat[-2].n == 270 and cif[1].num_residues_raw == 642
    # plain-text mmCIF, a missing file, max_residue
    plain = tmp_path / "af.cif"
    plain.write_bytes(gzip.open(os.path.join(io, "AF-A0A4S3KKF6-F1-model_v4.cif.gz")).read())
    r, okf = structure.read_compact_structures([str(plain), str(tmp_path / "nope.pdb"), pdbs[0]], threads=2, max_residue=700)
    assert okf.tolist() == [1, 0, 1] and r[0].n == 0 and r[0].num_residues_raw == 738 and r[1].n == 0 and r[2].n == nat[0].n


def test_hash_type_names_and_aliases():
    """HashType::get_with_str / to_string (src/geometry/core.rs:42-75)"""
    from folddisco_amd._lib import HASH_TYPE_NAMES, hash_type_index
    for k, v in HASH_TYPE_NAMES.items():
        assert hash_type_index(v) == k and hash_type_index(str(k)) == k
    for alias, want in (("default", 3), ("folddisco", 3), ("pdbtr", 3), ("pdb", 1), ("pyscomotif", 0), ("orig_pdb", 0), ("tr", 2), ("ppf", 4),
                        ("3di", 5), ("tertiary", 5), ("hybrid", 6), ("angle", 7), ("folddisco_angle", 7), ("dist", 8), ("distance", 8),
                        ("folddisco_dist", 8)):
        assert hash_type_index(alias) == want
    import pytest
    with pytest.raises(ValueError):
        hash_type_index("nonsense")


def test_analyze_summary_files(tmp_path):
    """`analyze -i PREFIX` (src/cli/workflows/analyze.rs summary branch, src/controller/summary.rs:121-260, 490-541) on the
    serine_peptidases index written by the oracle: totals, density, the count = posting BYTES quirk, top-N decode through
    reverse_hash, the amino-acid pair table and the power-of-two count distribution."""
    import oracle
    from folddisco_amd import analyze, indexio
    from folddisco_amd.__main__ import main as cli
    from tests.helpers import SER
    structs = [oracle.read_pdb(p) for p in SER]
    oix, nres, plddt = oracle.build_index(structs)
    prefix = str(tmp_path / "ser")
    oix.save(prefix)
    indexio.save_type(prefix + ".type", len(SER))
    cli(["analyze", "-i", prefix, "--top", "5"])
    stats = dict(l.rstrip("\n").split("\t") for l in open(prefix + "_summary_stats.tsv"))
    possible = 16 * 16 * (4 * 4) ** 3 * 400                      # dist_bins * angle_bins * aa_bins (feature.rs:293-345)
    assert stats == {"metric": "value", "total": "217612", "possible": str(possible), "empty": str(possible - 217612), "nonempty": "217612",
                     "density": "%.4f" % (217612 / possible * 100)}
    hashes, offsets = oix.hashes(), oix.offsets()
    counts = np.diff(offsets.astype(np.int64))
    top = [l.rstrip("\n").split("\t") for l in open(prefix + "_summary_top5.tsv")]
    assert top[0] == ["rank", "hash", "count", "aa1", "aa2", "ca_dist", "cb_dist", "ca_cb_angle", "phi1", "phi2"] and len(top) == 6
    order = np.argsort(-counts, kind="stable")
    for r in range(5):
        h = int(hashes[order[r]])
        assert top[r + 1][:3] == [str(r + 1), str(h), str(int(counts[order[r]]))]
        want = np.zeros(7, np.float32)
        oracle.lib().fdo_reverse_hash_pdbtr(h, want.ctypes.data_as(oracle.f32p))      # the oracle's restatement of pdb_tr.rs:95-136
        assert top[r + 1][3:5] == [analyze.AA3[int(want[0])], analyze.AA3[int(want[1])]]
        assert top[r + 1][5:] == ["%.4f" % x for x in want[2:]]
    aa = [l.rstrip("\n").split(",") for l in open(prefix + "_summary_aa_pairs.csv")]
    assert aa[0] == ["aa1_aa2"] + analyze.AA3 and [r[0] for r in aa[1:]] == analyze.AA3
    table = np.array([[int(x) for x in r[1:]] for r in aa[1:]])
    assert table.sum() == int(offsets[-1])                       # every posting byte is counted once
    a1, a2 = (hashes >> 25) & 31, (hashes >> 20) & 31
    assert table[3, 8] == int(counts[(a1 == 3) & (a2 == 8)].sum())
    dist = [tuple(int(x) for x in l.split("\t")) for l in list(open(prefix + "_summary_count_distribution.tsv"))[1:]]
    assert [b for b, _ in dist][:3] == [1, 2, 4] and dist[-1][0] == int(counts.max())
    assert sum(c for _, c in dist) == len(hashes) and dist[0][1] == int((counts <= 1).sum()) and dist[1][1] == int((counts == 2).sum())
    # the other built encodings decode through their own reverse_hash (bit fields round-trip)
    for t, feat in ((0, [13, 19, 14.0, 15.9, 116.0]), (7, [13, 19, 14.0, 15.9, 2.0, 1.4, -1.7]), (8, [13, 19, 14.0, 15.9, 2.0, 1.4, -1.7])):
        with oracle.hash_type(t):
            h = oracle.hash_any(feat)
        dd, da = analyze._BINS[t][:2]
        v = analyze.reverse_hash(t, [h], dd, da)[0]
        assert (int(v[0]), int(v[1])) == (13, 19) and abs(v[2] - 14.0) <= 18.0 / (dd - 1) / 2 + 1e-4 and abs(v[3] - 15.9) <= 18.0 / (dd - 1) / 2 + 1e-4
    with oracle.hash_type(4):                                    # PointPairFeature: 1 distance, 3 sin/cos angle pairs (8 / 3 bins)
        h = oracle.hash_any([13, 19, 7.5, 0.7, 1.7, 2.9])
    v = analyze.reverse_hash(4, [h], 8, 3)[0]
    assert (int(v[0]), int(v[1])) == (13, 19) and abs(v[2] - 7.5) <= 18.0 / 7 / 2 + 1e-4 and analyze.total_bins(4, 0, 0) == 8 * 729 * 400
    with pytest.raises(SystemExit):
        cli(["analyze", "-i", prefix, "-p", "somewhere"])


def test_index_id_types():
    """--id of the index subcommand: IdType::get_with_str + parse_path_by_id_type (src/controller/mode.rs:18-30, 69-126)"""
    from folddisco_amd.indexio import parse_path_by_id_type as f
    assert f("data/x/pdb1abc.ent", "pdb") == "1abc" and f("d/1abc.pdb", "PDB") == "1abc"
    assert f("d/AF-P12345-F1-model_v4.pdb", "afdb") == "AF-P12345-F1-model_v4" and f("d/AF-P12345-F1-model_v4.pdb", "uniprot") == "P12345"
    assert f("d/zzz.pdb", "afdb") == "zzz" and f("d/zzz.pdb", "uniprot") == "zzz"
    assert f("d/a.pdb.gz", "filename") == "a.pdb" and f("d/a.pdb.gz", "basename") == "a.pdb.gz"
    assert f("d/a.pdb", "relpath") == "d/a.pdb" and f("d/a.pdb", "default") == "d/a.pdb" and f("d/a.pdb", "whatever") == "d/a.pdb"
    assert f("tests/helpers.py", "abspath") == os.path.realpath("tests/helpers.py")


def test_oracle_query_bench_driver_equals_python_driven_oracle():
    """bench.py's query cpu_baseline driver (oracle/fdo_bench.c: fdo_query_bench over a borrowed index) does the work the
    Python-driven oracle does query by query: same candidate counts, same number of matches, for 1 and 4 threads."""
    from folddisco_amd import querybench, synth
    from tests.helpers import packed_to_oracle_structs
    S = 40
    d = synth.generate(S, seed=77)
    ps = synth.to_packed(d)
    structs = packed_to_oracle_structs(ps)
    oix, onres, _ = oracle.build_index(structs)
    qs = querybench._pick_queries(d, S, 5, seed=3)
    want_hits = want_matches = 0
    for s, idx, _ in qs:
        om = oracle.make_query_map(structs[s], ",".join(f"A{int(i) + 1}" for i in idx), oix, float(S))
        recs = oracle.count_query(om, oix, onres)
        order = sorted(range(len(recs)), key=lambda k: -recs[k]["idf"])      # stable
        want_hits += min(len(recs), 1000)
        for k in order[:8]:
            want_matches += len(oracle.retrieve(structs[recs[k]["nid"]], structs[s], om)["processed"])
    for nt in (1, 4):
        r = oracle.query_bench(oix.hashes(), oix.offsets(), oix.values(), onres, ps.res_off, ps.n_xyz, ps.ca_xyz, ps.cb_xyz, ps.aa,
                               [(s, idx) for s, idx, _ in qs], top_n=1000, match_top=8, n_threads=nt)
        assert (r["hits"], r["matches"]) == (want_hits, want_matches) and want_matches >= 5


def test_multi_model_files_plain_versus_gzip(tmp_path):
    """The reference's plain-file reader stops behind the first MODEL (pdb.rs:46-58); its gzip reader has no MODEL handling and keeps
    the ATOM records of every model (pdb.rs:79-124) — a multi-model .pdb.gz therefore yields different residues than the same file
    uncompressed.  Both ingest paths (fd_ingest.cpp and structure.py) follow it."""
    import gzip
    from folddisco_amd import structure as st

    def atom(serial, name, res, chain, rser, x, y, z):
        return "ATOM  %5d %-4s %3s %1s%4d    %8.3f%8.3f%8.3f  1.00 20.00           C  " % (serial, name, res, chain, rser, x, y, z)
    lines, k = [], 1
    for model in (1, 2):
        lines.append("MODEL     %4d" % model)
        for r in range(1, 6):
            for name, dx in ((" N  ", 0.0), (" CA ", 1.4), (" C  ", 2.5), (" CB ", 1.6)):
                lines.append(atom(k, name, "ALA" if model == 1 else "SER", "A", r + 10 * (model - 1), 3.8 * r + dx, 0.3 * model, 0.1 * r))
                k += 1
        lines.append("ENDMDL")
    txt = "\n".join(lines) + "\n"
    plain, gz = tmp_path / "m.pdb", tmp_path / "m.pdb.gz"
    plain.write_text(txt)
    with gzip.open(gz, "wt") as fh:
        fh.write(txt)
    a, b = st.read_compact_structure(str(plain)), st.read_compact_structure(str(gz))
    assert a.n < b.n and set(a.aa.tolist()) == {0} and set(b.aa.tolist()) == {0, 15}
    ps = st.read_packed([str(plain), str(gz)], threads=2)[0]
    off = ps.res_off.astype(int)
    assert off[1] - off[0] == a.n and off[2] - off[1] == b.n
    assert np.array_equal(ps.ca_xyz[off[1]:off[2]], b.ca_xyz) and np.array_equal(ps.aa[off[1]:off[2]], b.aa)


def test_bench_input_writers_round_trip_through_the_ingest(tmp_path):
    """bench.py's drop-in `index` leg writes synthetic structures as gzipped PDB files and a Foldcomp database by repeating the reference's
    fixture entries: both must come back through the native ingest as what was written (residue counts, types, N / CA coordinates at
    PDB precision; CB of glycines is the reference's virtual one, so it is not compared)."""
    from folddisco_amd import structure, synth
    d = synth.generate(12, seed=5)
    n = synth.write_pdb_gz(d, str(tmp_path / "pdb"), workers=2)
    assert n == 12
    paths = sorted(str(p) for p in (tmp_path / "pdb").iterdir())
    ps, nres, plddt, raw, ok = structure.read_packed(paths, threads=2)
    off = d["res_off"].numpy()
    assert ok.all() and np.array_equal(nres, np.diff(off))
    assert np.array_equal(ps.aa, d["aa"].numpy())
    assert np.abs(ps.ca_xyz - d["ca_xyz"].numpy()).max() < 1e-3 and np.abs(ps.n_xyz - d["n_xyz"].numpy()).max() < 1e-3
    keep = d["aa"].numpy() != 7
    keep[off[1:] - 1] = False           # the last residue of a file takes the builder's own path (structure/core.rs:116-155 quirks): not the writer's concern
    assert np.abs(ps.cb_xyz[keep] - d["cb_xyz"].numpy()[keep]).max() < 1e-3
    src = os.path.join(ROOT, "tests", "golden", "foldcomp", "example_db")
    db = str(tmp_path / "rep_foldcomp")
    assert synth.replicate_foldcomp_db(src, db, 60) == 60
    fc = structure.FoldcompDb(db)
    assert len(fc.keys) == 60 and fc.names[0].endswith("_000000") and fc.names[59].endswith("_000059")
    ps2, nres2, *_ = structure.read_packed(fc.keys, threads=2, foldcomp=fc)
    ref = structure.FoldcompDb(src)
    ps0, nres0, *_ = structure.read_packed(ref.keys, threads=2, foldcomp=ref)
    assert np.array_equal(nres2, np.tile(nres0, 3)[:60]) and np.array_equal(ps2.ca_xyz[: len(ps0.ca_xyz)], ps0.ca_xyz)


def test_read_packed_views_outlive_the_call_and_equal_the_per_structure_arrays(tmp_path):
    """read_packed hands out VIEWS of the library's packed arrays (no copy: the coordinates are ~40 bytes per residue) and the library copies
    the per-structure parts into them on several threads: the views must stay valid after the call's other results are dropped and garbage
    is collected, be released with the last view, and hold what read_compact_structures returns structure by structure — for one thread,
    for more threads than structures, and for enough structures that every copying thread gets several blocks of 64."""
    import gc
    from folddisco_amd import structure
    src = os.path.join(ROOT, "tests", "golden", "foldcomp", "example_db")
    db = str(tmp_path / "rep_foldcomp")
    from folddisco_amd import synth
    synth.replicate_foldcomp_db(src, db, 700)
    fc = structure.FoldcompDb(db)
    keys = np.asarray(fc.keys, np.uint64)
    cs, okc = structure.read_compact_structures(keys[:40], threads=3, foldcomp=fc)
    for n_keys, threads in ((40, 1), (3, 8), (700, 4)):
        ps, nres, plddt, raw, ok = structure.read_packed(keys[:n_keys], threads=threads, foldcomp=fc)
        ca = ps.ca_xyz
        assert not ca.flags.owndata and ca.base is not None      # a view, kept alive through its base
        del ps, nres, plddt, raw, ok
        gc.collect()
        junk = [np.full(1 << 20, 7, np.uint8) for _ in range(8)]   # freed memory would be handed out again here
        off = 0
        for k in range(min(n_keys, 40)):
            m = cs[k].n
            assert np.array_equal(ca[off:off + m], cs[k].ca_xyz), (n_keys, threads, k)
            off += m
        del junk
    ps, nres, *_ = structure.read_packed(keys, threads=4, foldcomp=fc)
    P = len(structure.FoldcompDb(src).keys)                    # the database repeats the fixture's entries with this period
    per = np.array([c.n for c in cs[:P]], np.uint64)
    assert np.array_equal(nres, np.tile(per, 700 // P + 1)[:700])
    ro = ps.res_off.astype(np.int64)
    last = 699 % P
    assert np.array_equal(ps.n_xyz[ro[699]:ro[700]], ps.n_xyz[ro[last]:ro[last + 1]]) and np.array_equal(ps.cb_xyz[ro[699]:ro[700]], ps.cb_xyz[ro[last]:ro[last + 1]])
    assert np.array_equal(ps.aa[:ro[P]], ps.aa[ro[P]:ro[2 * P]])


def test_fixed_format_number_fields_parse_like_strtof(tmp_path):
    """The ingest's fast path for "%8.3f" / "%6.2f" fields (mantissa below 2^24, one IEEE division by a power of ten) returns the bits the
    correctly rounded decimal -> f32 conversion returns (Rust's str::parse::<f32>, glibc strtof, numpy): random coordinates over the whole
    range of the format incl. negative zero, no fraction, leading '+', and fields that leave the fast path (8 digits, exponents)."""
    from folddisco_amd.structure import read_compact_structures
    rng = np.random.Generator(np.random.PCG64(404))
    fields = ["%8.3f" % v for v in rng.uniform(-999.999, 9999.999, 3000)]
    fields += ["  -0.000", "   0.000", "9999.999", "-999.999", "    12.5", "     100", "  +1.250", "16777.21", "99999.99", "1.25e+01", " 1677721", "16777217"]
    lines, want = [], []
    for k in range(0, len(fields) - 2, 3):
        x, y, z = fields[k:k + 3]
        r = k // 3 + 1
        b = "%6.2f" % rng.uniform(0, 100)
        for name in (" N  ", " CA ", " C  ", " CB "):
            lines.append("ATOM  %5d %s ALA A%4d    %s%s%s  1.00%s           C" % (r % 100000, name, r % 10000, x.rjust(8)[:8], y.rjust(8)[:8], z.rjust(8)[:8], b))
        want.append([np.float32(x), np.float32(y), np.float32(z)])
    p = tmp_path / "fields.pdb"
    p.write_text("\n".join(lines) + "\nEND\n")
    cs = read_compact_structures([str(p)], threads=1)[0][0]
    got = np.asarray(cs.ca_xyz, np.float32)
    w = np.asarray(want, np.float32)[:len(got)]
    assert len(got) >= len(want) - 1
    assert np.array_equal(got.view(np.uint32), w.view(np.uint32))


def test_own_gzip_decoder_equals_zlib(tmp_path, monkeypatch):
    """The ingest's gzip decoder (csrc/fd_inflate.cpp, written from RFC 1951 / 1952) returns exactly what zlib returns: stored, fixed and dynamic
    blocks at every level and strategy, sync flushes, several members, optional header fields, trailing bytes; damaged input is DECLINED (the
    ingest then goes through zlib), never decoded to something else.  And the ingest reads the same structures through either path."""
    import ctypes as C
    import gzip
    import struct
    import zlib
    from folddisco_amd import _lib
    from folddisco_amd.structure import read_compact_structures
    L = _lib.load()

    def gunzip(b):
        buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
        out, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        if L.fdgpu_debug_gunzip(buf, len(b), C.byref(out), C.byref(n)):
            return None
        r = bytes(C.string_at(out, n.value))
        L.fdgpu_free(out)
        return r

    rng = np.random.default_rng(1)
    pdb = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "query", "4CHA.pdb"), "rb").read()
    line = b"ATOM      1  N   MET A   1      12.345  23.456  34.567  1.00 50.00           N  \n"
    texts = [b"", b"a", b"hello hello hello hello", bytes(rng.integers(0, 256, 100000, dtype=np.uint8)), bytes(rng.integers(0, 4, 300000, dtype=np.uint8)),
             b"A" * 100000, line * 5000, pdb, pdb * 5, bytes(rng.integers(65, 70, 50, dtype=np.uint8)) * 3000]
    n_ok = 0
    for t in texts:
        for lvl in (0, 1, 4, 6, 9):
            assert gunzip(gzip.compress(t, compresslevel=lvl)) == t, (len(t), lvl)
            n_ok += 1
        for strat in (zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED):
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
            assert gunzip(co.compress(t) + co.flush()) == t, (len(t), strat)
            n_ok += 1
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        z = co.compress(t[:len(t) // 2]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(t[len(t) // 2:]) + co.flush()
        assert gunzip(z + gzip.compress(b"second member " + t[:1000])) == t + b"second member " + t[:1000]
        assert gunzip(z + b"trailing bytes") == t
    assert n_ok == 90
    for n in list(range(0, 300)) + [1023, 4096 + 7, 65536 + 63]:       # every tail length of the carry-less-multiplication CRC (64-byte folds, 16-byte folds, table tail)
        t = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert gunzip(gzip.compress(t, compresslevel=1)) == t, n
    z = gzip.compress(b"payload payload payload")
    hdr = bytearray(z[:10])
    hdr[3] = 4 | 8 | 16
    assert gunzip(bytes(hdr) + struct.pack("<H", 5) + b"extra" + b"name.pdb\0" + b"a comment\0" + z[10:]) == b"payload payload payload"
    t = line * 5000
    z = bytearray(gzip.compress(t))
    declined = 0
    for k in range(0, len(z), max(1, len(z) // 300)):
        zz = bytearray(z)
        zz[k] ^= 0x55
        r = gunzip(bytes(zz))
        declined += r is None
        assert r is None or r == t, k
    assert declined > 250
    assert gunzip(bytes(z[:len(z) // 2])) is None and gunzip(b"\x1f\x8b" + b"\0" * 30) is None and gunzip(b"not gzip at all, plain text....") is None
    # the ingest: same structure through the own decoder and through zlib, and a damaged file fails the same way through both
    p = tmp_path / "q.pdb.gz"
    p.write_bytes(gzip.compress(pdb))
    bad = tmp_path / "bad.pdb.gz"
    bad.write_bytes(gzip.compress(pdb)[:2000])
    a = read_compact_structures([str(p), str(bad)], threads=1)
    monkeypatch.setenv("FDGPU_ZLIB", "1")
    b = read_compact_structures([str(p), str(bad)], threads=1)
    monkeypatch.delenv("FDGPU_ZLIB")
    assert list(a[1]) == list(b[1])
    assert np.array_equal(np.asarray(a[0][0].ca_xyz), np.asarray(b[0][0].ca_xyz)) and len(a[0][0].ca_xyz) > 200


def test_cli_skip_threshold_is_the_references_hard_wired_65535(tmp_path):
    """The reference skips a structure when Structure.num_residues > 65535 — Folddisco::new sets max_residue = DEFAULT_MAX_RESIDUE
    (src/controller/mod.rs:40,124), set_max_residue (mod.rs:189) has no caller, the test is mod.rs:313 — and its -n/--residue flag
    (src/cli/main.rs:42, default 50000) only lands in PREFIX.type (build_index.rs:222).  The CLI's ingest step must do the same: raw counts
    of 50,002 and 65,535 are indexed, 65,536 is skipped (id kept, nres 0, plddt 0), whatever -n says; the oracle agrees."""
    import argparse
    import inspect
    import oracle
    from folddisco_amd import __main__ as cli, indexio, structure
    from tests.helpers import write_sparse_pdb
    sizes = [30, 50001, 65534, 65535]                       # raw residue counts 31, 50002, 65535, 65536
    paths = []
    for k, n in enumerate(sizes):
        paths.append(str(tmp_path / f"s{k}.pdb"))
        write_sparse_pdb(paths[-1], n, seed=k)
    for dash_n in (50000, 10, 70000):                       # -n never changes what is indexed
        a = argparse.Namespace(fc=None, fc_keys=None, threads=2, max_residue=dash_n)
        ps, nres, plddt, raw, ok = cli._ingest(a, structure, paths, 0)
        assert ok.tolist() == [1, 1, 1, 1]
        assert raw.tolist() == [31, 50002, 65535, 65536]
        assert nres.tolist() == [30, 50001, 65534, 0]
        assert np.diff(ps.res_off.astype(np.int64)).tolist() == [30, 50001, 65534, 0]
        assert plddt[3] == 0.0 and plddt[1] > 0.0
    assert cli.REF_SKIP_MAX_RESIDUE == 65535
    # the oracle: same counts from its own parser, same threshold by default (strict >)
    assert inspect.signature(oracle.build_index).parameters["max_residue"].default == 65535
    for p, n, r, pl in zip(paths, nres.tolist(), raw.tolist(), plddt):
        s = oracle.read_pdb(p)
        assert s.ptr.contents.num_residues_raw == r and (s.n == n or r > 65535)
        if n:
            assert np.float32(s.avg_plddt()).tobytes() == np.float32(pl).tobytes()
    small = [oracle.read_pdb(paths[0])] * 2                                   # raw count 31
    for thr, want in ((31, [30, 30]), (30, [0, 0])):
        _, onres, opl = oracle.build_index(small, max_residue=thr)
        assert onres.tolist() == want and (opl[0] == 0.0) == (want[0] == 0)
    # -n reaches PREFIX.type and nothing else (cli/config.rs:66-97)
    indexio.save_type(str(tmp_path / "x.type"), 4, max_residue=10)
    assert "max_residue = 10\n" in open(tmp_path / "x.type").read()


def test_lookup_writer_prints_f32_like_rust_display(tmp_path):
    """PREFIX.lookup (src/index/lookup.rs:35-56): the library's writer (fdgpu_write_lookup, what the CLI uses) against the Python restatement of
    Rust's `{}` for f32 — shortest round-trip digits, never exponent form, integral values without a fraction, NaN / inf — on fixed cases and
    50,000 random bit patterns; the file round-trips through load_lookup."""
    import ctypes as C
    from folddisco_amd import _lib, indexio
    fixed = [(50.0, "50"), (0.0, "0"), (100.0, "100"), (0.1, "0.1"), (1e-7, "0.0000001"), (3.4028235e38, "340282350000000000000000000000000000000"), (1e10, "10000000000"),
             (13.540356, "13.540356"), (16777216.0, "16777216"), (0.5, "0.5"), (1.5e-5, "0.000015"), (-2.5, "-2.5"), (-0.0, "-0")]
    v = np.array([x for x, _ in fixed], np.float32)
    out = C.create_string_buffer(64 * len(v))
    assert _lib.load().fdgpu_format_f32_display(v.ctypes.data_as(_lib.f32p), len(v), out) == 0
    got = [out.raw[64 * k:64 * k + 64].split(b"\0")[0].decode() for k in range(len(v))]
    assert got == [w for _, w in fixed] and got == [indexio.format_f32_display(x) for x in v]
    rng = np.random.default_rng(5)
    with np.errstate(all="ignore"):
        vals = np.concatenate([rng.integers(0, 2 ** 32, 50000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                               np.array([np.nan, np.inf, -np.inf, -0.0, 1e-45, 1.17549435e-38], np.float32), rng.uniform(0, 100, 5000).astype(np.float32)])
        tids = ["d/s%d.pdb" % i for i in range(len(vals))]
        nres = rng.integers(0, 70000, len(vals)).astype(np.uint64)
        keys = rng.integers(0, 2 ** 40, len(vals)).astype(np.uint64)
        indexio.save_lookup(str(tmp_path / "a.lookup"), tids, nres, vals, db_keys=keys)
        indexio.save_lookup_py(str(tmp_path / "b.lookup"), tids, nres, vals, db_keys=keys)
    assert open(tmp_path / "a.lookup", "rb").read() == open(tmp_path / "b.lookup", "rb").read()
    indexio.save_lookup(str(tmp_path / "c.lookup"), tids[:100], nres[:100], vals[-100:])
    t2, n2, p2, k2 = indexio.load_lookup(str(tmp_path / "c.lookup"))
    assert t2 == tids[:100] and np.array_equal(n2, nres[:100]) and p2.tobytes() == vals[-100:].tobytes() and np.array_equal(k2, np.arange(100))


def test_squared_distance_table_equals_sqrt_and_quantiser():
    """folddisco_amd/csrc/fd_dist_table.h (tools/gen_dist_table.c: every float squared distance pushed through sqrtf + the quantiser of
    src/utils/convert.rs:32-36 with the default 16 distance bins, src/geometry/pdb_tr.rs:22-44): counting the breakpoints at or below x equals the
    chain, at and around every breakpoint and on 2 M random squared distances; and the device's correction bin = b + (x >= T[b + 1]) - (x < T[b])
    lands on it from any first guess within one bin."""
    hdr = open(os.path.join(ROOT, "folddisco_amd", "csrc", "fd_dist_table.h")).read()
    T = np.array([int(x, 16) for x in re.search(r"fd_dist_thr_bits\[FD_DIST_NTHR\] = \{([^}]*)\}", hdr).group(1).replace("u", "").split(",")], np.uint32)
    thr = T.view(np.float32)
    assert len(T) == 33 and T[0] == 0 and np.all(np.diff(T.astype(np.int64)) > 0)
    disc = np.float32(1.0) / (np.float32(18.0) / np.float32(15.0))

    def chain(x):
        with np.errstate(invalid="ignore"):
            v = (np.sqrt(x.astype(np.float32)) - np.float32(2.0)) * disc + np.float32(0.5)
        return np.where(v > 0, np.floor(np.maximum(v, np.float32(0))), 0).astype(np.int64)

    def table(x):
        return np.searchsorted(thr[1:], x, side="right").astype(np.int64)
    edge = np.concatenate([(T[1:, None].astype(np.int64) + np.arange(-3, 4)[None, :]).ravel().astype(np.uint32).view(np.float32), np.array([0.0, 1e-30, 6.75, 400.0, 424.36], np.float32)])
    assert np.array_equal(chain(edge), table(edge))
    rng = np.random.Generator(np.random.PCG64(3))
    x = np.concatenate([rng.uniform(0, 1600, 1500000), rng.uniform(0, 30, 500000) ** 2]).astype(np.float32)
    x = x[x < thr[32]]
    want = chain(x)
    assert np.array_equal(want, table(x)) and want.max() == 31
    for dg in (-1, 0, 1):                                     # the device's correction from a guess one bin off
        b = np.clip(want + dg, 0, 31)
        got = b + (x >= thr[b + 1]).astype(np.int64) - (x < thr[b]).astype(np.int64)
        ok = np.abs(b - want) <= 1
        assert np.array_equal(got[ok], want[ok])


def test_residue_name_table_matches_oracle(tmp_path):
    """The ingest's residue-name table (a hash table since round 5; src/utils/convert.rs:53-81) against the oracle's map for every listed name — canonical
    and modified — plus names it must not know (lower case, shifted, nucleotides, water); resname_std is set for the canonical names only."""
    import oracle
    from folddisco_amd.structure import read_compact_structures
    groups = ["ALA ABA ORN DAL AIB ALC MDO MAA DAB", "ARG DAR CIR AGM", "ASN DSG MEN SNN", "ASP 0TD DAS IAS PHD BFD ASX",
              "CYS CSO CSD CME OCS CAS CSX CSS YCM DCY SMC SCH SCY CAF SNC SEC", "GLN DGN CRQ MEQ", "GLU PCA DGL CGU FGA B3E GLX", "GLY CR2 SAR GHP GL3",
              "HIS HIC DHI NEP CR8 MHS", "ILE DIL", "LEU DLE NLE MLE MK8", "LYS KCX LLP MLY M3L ALY MLZ DLY KPI PYL", "MET MSE FME NRQ CXM SME MHO MED",
              "PHE DPN PHI MEA PHL", "PRO HYP DPR", "SER CSH SEP DSN SAC GYS DHA OAS", "THR TPO CRO DTH BMT CRF", "TRP DTR TRQ TOX 0AF", "TYR PTR TYS TPQ DTY OMY",
              "VAL DVA MVA FVA"]
    names = [n for g in groups for n in g.split()] + ["UNK", "XXX", "HOH", "A  ", " DA", "ala", "GLy", "AL ", "  A"]
    lines, ser = [], 1
    for r, nm in enumerate(names, 1):
        for an, xyz in ((" N  ", (0.0, 0.0, 0.0)), (" CA ", (1.458, 0.0, 0.0)), (" C  ", (2.0, 1.4, 0.0)), (" CB ", (2.0, -0.7, 1.2))):
            lines.append("ATOM  %5d %s %s A%4d    %8.3f%8.3f%8.3f  1.00 50.00           C" % (ser, an, nm, r, xyz[0] + r * 3.8, xyz[1], xyz[2]))
            ser += 1
    p = tmp_path / "names.pdb"
    p.write_text("\n".join(lines) + "\nEND\n")
    want = oracle.read_pdb(str(p)).arrays()
    cs = read_compact_structures([str(p)], threads=1)[0][0]
    assert len(cs.aa) == len(names)
    assert np.array_equal(np.asarray(cs.aa), np.asarray(want["aa"]))
    canonical = {g.split()[0] for g in groups}
    assert [bool(x) for x in cs.resname_std()] == [n in canonical for n in names]
    assert [int(a) for a in cs.aa[:9]] == [0] * 9 and int(cs.aa[len(names) - 9]) == 255


def test_host_glue_components_and_symmetry_flags_without_a_gpu():
    """The host pieces of the retrieval of > 64-node graphs (csrc/fd_host_query.hip), called without a GPU: the components of the found-pair graph
    (graph.rs:16-50: nodes by first appearance, strongly + weakly connected sets of at least node_count nodes, sorted, without duplicates) against
    scipy's labels on random directed graphs of up to 3,000 nodes; the default encoding's symmetry flag (a 16-entry table of the torsion angles,
    geometry/pdb_tr.rs:158-162) against the oracle over every residue-type pair x every combination of the four angle-bin fields."""
    import ctypes as C
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    from folddisco_amd import _lib
    L = _lib.load()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(23)

    def want(ei, ej, node_count):
        node = {}
        for a, b in zip(ei.tolist(), ej.tolist()):
            node.setdefault(a, len(node)); node.setdefault(b, len(node))
        res = np.array(list(node.keys()), np.uint32)
        n = len(node)
        if n == 0:
            return []
        u = np.array([node[a] for a in ei.tolist()]); v = np.array([node[b] for b in ej.tolist()])
        g = csr_matrix((np.ones(len(u)), (u, v)), shape=(n, n))
        sets = set()
        for conn in ("strong", "weak"):
            _, lab = connected_components(g, directed=True, connection=conn)
            for l in np.unique(lab):
                members = tuple(np.nonzero(lab == l)[0].tolist())
                if len(members) >= node_count:
                    sets.add(members)
        return [res[list(m)].tolist() for m in sorted(sets)]

    def got(ei, ej, node_count):
        r, o, nc = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.c_uint64()
        rc = L.fdgpu_debug_host_components(ei.ctypes.data_as(C.POINTER(C.c_uint32)), ej.ctypes.data_as(C.POINTER(C.c_uint32)), len(ei), node_count, C.byref(r), C.byref(o),
                                           C.byref(nc))
        assert rc == 0
        off = np.ctypeslib.as_array(o, shape=(nc.value + 1,)).copy()
        res = np.ctypeslib.as_array(r, shape=(max(int(off[-1]), 1),)).copy()
        libc.free(C.cast(r, C.c_void_p)); libc.free(C.cast(o, C.c_void_p))
        return [res[off[k]:off[k + 1]].tolist() for k in range(nc.value)]

    cases = [(0, 0, 2), (5, 4, 1), (12, 30, 2), (70, 90, 3), (70, 400, 4), (300, 350, 2), (300, 25000, 4), (3000, 4000, 2), (3000, 12000, 4)]
    for n_nodes, n_edges, node_count in cases:
        ids = rng.permutation(70000)[:max(n_nodes, 1)].astype(np.uint32)
        ei = ids[rng.integers(0, max(n_nodes, 1), n_edges)].astype(np.uint32) if n_edges else np.zeros(0, np.uint32)
        ej = ids[rng.integers(0, max(n_nodes, 1), n_edges)].astype(np.uint32) if n_edges else np.zeros(0, np.uint32)
        assert got(ei, ej, node_count) == want(ei, ej, node_count), (n_nodes, n_edges, node_count)
    # chains and rings: strongly connected rings inside a weakly connected whole
    ei = np.array([1, 2, 3, 3, 10, 11, 12, 20], np.uint32); ej = np.array([2, 3, 1, 10, 11, 12, 10, 21], np.uint32)
    assert got(ei, ej, 2) == want(ei, ej, 2) and got(ei, ej, 3) == want(ei, ej, 3) and got(ei, ej, 7) == want(ei, ej, 7)

    # symmetry flags of the default encoding (hash type 3): every (aa1, aa2) x every value of the low eight bits, plus random hashes
    aa1, aa2, low = np.meshgrid(np.arange(32, dtype=np.uint32), np.arange(32, dtype=np.uint32), np.arange(256, dtype=np.uint32), indexing="ij")
    mid = rng.integers(0, 1 << 12, size=aa1.size, dtype=np.uint32)
    h = np.concatenate([(aa1.ravel() << 25) | (aa2.ravel() << 20) | (mid << 8) | low.ravel(), rng.integers(0, 1 << 30, size=20000, dtype=np.uint32)]).astype(np.uint32)
    out = np.zeros(len(h), np.uint8)
    assert L.fdgpu_debug_hash_is_symmetric(3, h.ctypes.data_as(C.POINTER(C.c_uint32)), len(h), out.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    OL = oracle.lib()
    exp = np.fromiter((OL.fdo_hash_is_symmetric(int(x)) for x in h[::7]), np.uint8)
    assert np.array_equal(out[::7], exp) and 0 < int(out.sum()) < len(out)
