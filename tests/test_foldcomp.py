"""Foldcomp input: folddisco_amd/csrc/fd_fcz.cpp against the reference's own decoder.

The expected atom records (tests/golden/foldcomp/ref_atoms.npz) were produced by oracle/_ref/libfoldcomp_ref.so — the vendored
decoder under /root/reference/lib/foldcomp, compiled from its sources where they lie — over the data files the reference's own
tests read (src/structure/io/fcz.rs:300-393: data/foldcomp/7m0y.fcz, data/foldcomp/example_db); tools/make_foldcomp_golden.py is the
generator.  Where the reference tree exists the same comparison also runs live.  Bit-exact: names, residue names, chain,
residue serial, temperature factor for every atom; coordinates for every atom this path reads (N, CA, C, O, CB, OXT).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from folddisco_amd import _lib, structure

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "foldcomp")
DB = os.path.join(G, "example_db")
PLACED = [b" N  ", b" CA ", b" C  ", b" O  ", b" CB ", b" OXT"]


def product_decode(buf: bytes):
    L = _lib.load()
    p, n = C.POINTER(_lib.FoldcompAtom)(), C.c_uint64()
    rc = L.fdgpu_foldcomp_decode(buf, len(buf), C.byref(p), C.byref(n))
    if rc != 0:
        return rc
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 32,))[:n.value * 32].copy()
    L.fdgpu_free(p)
    return a.view(oracle.FCZ_ATOM_DTYPE)


def entries():
    out = [("fcz_7m0y", open(os.path.join(G, "7m0y.fcz"), "rb").read())]
    db = open(DB, "rb").read()
    for line in open(DB + ".index"):
        k, s, l = (int(t) for t in line.split())
        out.append((f"db_{k}", db[s:s + l]))
    return out


def assert_same_atoms(got, want, tag):
    assert len(got) == len(want), tag
    for f in ("name", "res", "chain", "rser"):
        assert np.array_equal(got[f], want[f]), (tag, f)
    assert np.array_equal(got["b"].view(np.uint32), want["b"].view(np.uint32)), (tag, "b")
    placed = np.isin(want["name"], PLACED)
    assert placed.sum() >= 4 * len(np.unique(want["rser"])) - 4
    for f in "xyz":
        assert np.array_equal(got[f][placed].view(np.uint32), want[f][placed].view(np.uint32)), (tag, f)


def test_decoder_matches_reference_golden_vectors():
    gold = np.load(os.path.join(G, "ref_atoms.npz"))
    ents = entries()
    assert len(ents) == 25 and {k for k, _ in ents} == set(gold.files)
    for tag, buf in ents:
        got = product_decode(buf)
        assert not isinstance(got, int), (tag, got)
        assert_same_atoms(got, gold[tag].view(oracle.FCZ_ATOM_DTYPE), tag)
    # the example database holds an UNK residue (backbone only, no side-chain torsions consumed): db_1
    assert b"UNK" in set(gold["db_1"].view(oracle.FCZ_ATOM_DTYPE)["res"].tolist())


@pytest.mark.skipif(not oracle.foldcomp_ref_available(), reason="oracle/_ref/libfoldcomp_ref.so needs the reference tree")
def test_decoder_matches_reference_build_live():
    for tag, buf in entries():
        assert_same_atoms(product_decode(buf), oracle.foldcomp_ref_decode(buf), tag)


def test_malformed_entries_are_rejected():
    buf = entries()[0][1]
    assert product_decode(b"") == -1 or product_decode(b"") < 0
    assert isinstance(product_decode(b"XXXX" + buf[4:]), int)
    for cut in (3, 40, 76, 200, len(buf) - 1):
        assert isinstance(product_decode(buf[:cut]), int), cut
    # residue codes the format's decoder has no geometry for (ASX = 20) are refused, not guessed
    hdr_n_anchor = buf[4 + 8]
    first_res = 4 + 72 + 4 * hdr_n_anchor + int.from_bytes(buf[4 + 20:4 + 24], "little") + 36 * hdr_n_anchor + 1 + 12
    bad = bytearray(buf)
    bad[first_res + 8] = (20 << 3) | (bad[first_res + 8] & 7)
    assert isinstance(product_decode(bytes(bad)), int)
    # hostile anchor indices (3 * index wraps a 32-bit int; non-monotone; out of range) cost the entry, not the process
    a0 = 4 + 72
    for i0, i1 in ((0x2AAAAAAB, 272), (5, 3), (-1, 10), (0, 10 ** 6)):
        bad = bytearray(buf)
        bad[a0:a0 + 4] = int(i0 & 0xffffffff).to_bytes(4, "little")
        bad[a0 + 4:a0 + 8] = int(i1 & 0xffffffff).to_bytes(4, "little")
        assert isinstance(product_decode(bytes(bad)), int), (i0, i1)
    # counts in the header larger than the entry: refused before anything is allocated with them
    for field_off in (4 + 0, 4 + 8, 4 + 12, 4 + 20):      # n_residue, n_anchor, n_side_torsion (u16 / u32 fields), len_title
        bad = bytearray(buf)
        bad[field_off:field_off + 2] = b"\xff\xff"
        assert isinstance(product_decode(bytes(bad)), int), field_off


def gold_atoms_as_tuples(rec):
    return [(np.float32(a["x"]), np.float32(a["y"]), np.float32(a["z"]), a["name"].decode(), a["res"].decode(), int(a["rser"]), int(a["chain"]),
             np.float32(a["b"])) for a in rec]


def test_database_listing_and_ingest():
    fc = structure.FoldcompDb(DB)
    lk = dict((int(l.split("\t")[0]), l.split("\t")[1].strip()) for l in open(DB + ".lookup"))
    keys = sorted(int(l.split("\t")[0]) for l in open(DB + ".index"))
    assert fc.keys.tolist() == keys and fc.names == [lk[k] for k in keys] and len(fc) == 24
    assert structure.is_foldcomp_db(DB) and not structure.is_foldcomp_db(G) and not structure.is_foldcomp_db(os.path.join(G, "7m0y.fcz"))
    gold = np.load(os.path.join(G, "ref_atoms.npz"))
    structs, ok = structure.read_compact_structures(fc.keys, threads=4, foldcomp=fc)
    assert ok.all() and len(structs) == 24
    for k, s in zip(keys, structs):
        want = structure.build_compact(gold_atoms_as_tuples(gold[f"db_{k}"].view(oracle.FCZ_ATOM_DTYPE)))
        assert s.n == want.n and s.num_residues_raw == want.num_residues_raw
        for f in ("n_xyz", "ca_xyz", "cb_xyz"):
            assert np.array_equal(getattr(s, f).view(np.uint32), getattr(want, f).view(np.uint32)), (k, f)
        assert np.array_equal(s.aa, want.aa) and np.array_equal(s.chain, want.chain) and np.array_equal(s.serial, want.serial)
        assert np.array_equal(s.bfac.view(np.uint32), want.bfac.view(np.uint32)) and list(s.resname) == list(want.resname)
    # a subset in caller order, one key the database does not hold: that slot is flagged unreadable and empty
    sub = [keys[5], 10 ** 9, keys[0]]
    ps, nres, plddt, raw, okf = structure.read_packed(sub, threads=2, foldcomp=fc)
    assert okf.tolist() == [1, 0, 1] and nres.tolist() == [structs[5].n, 0, structs[0].n]
    assert np.array_equal(ps.ca_xyz[:structs[5].n].view(np.uint32), structs[5].ca_xyz.view(np.uint32))
    assert np.float32(plddt[2]) == structs[0].avg_plddt()
    # --max-residue: the id stays, nothing is hashed (controller/mod.rs:313-318)
    ps2, nres2, _, raw2, ok2 = structure.read_packed(fc.keys[:3], threads=1, max_residue=100, foldcomp=fc)
    assert nres2.tolist() == [0, 0, 0] and (raw2 > 100).all() and ok2.all()


def test_type_file_and_lookup_carry_the_database(tmp_path):
    from folddisco_amd import indexio
    p = str(tmp_path / "x.type")
    indexio.save_type(p, 24, input_format="FCZDB", foldcomp_db="data/foldcomp/example_db", multiple_bins=[(16, 4), (8, 3)])
    txt = open(p).read().splitlines()
    assert txt[0] == "chunk_size = 24" and txt[1] == 'foldcomp_db = "data/foldcomp/example_db"' and txt[2] == "grid_width = 20.0"
    cfg = indexio.load_type(p)
    assert cfg["input_format"] == "FCZDB" and cfg["foldcomp_db"] == "data/foldcomp/example_db" and cfg["multiple_bin"] == [(16, 4), (8, 3)]
    lp = str(tmp_path / "x.lookup")
    indexio.save_lookup(lp, ["a", "b"], [10, 20], np.array([50.0, 60.5], np.float32), db_keys=[100, 110])
    assert open(lp).read() == "0\ta\t10\t50\t100\n1\tb\t20\t60.5\t110\n"
    assert indexio.load_lookup(lp)[3].tolist() == [100, 110]
