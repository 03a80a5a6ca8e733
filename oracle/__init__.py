"""oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/libfdoracle.so, the CPU restatement of the reference hot
path (see oracle/fd_oracle.h for the file:line citations and the parity-pinning
status).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; folddisco_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfdoracle.so")
_REF_FOLDCOMP = os.path.join(_HERE, "_ref", "libfoldcomp_ref.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    if os.path.isdir("/root/reference/lib/foldcomp") and not os.path.exists(_REF_FOLDCOMP):
        subprocess.call(["make", "-C", _HERE, "-s", "ref"])      # the reference's own Foldcomp decoder (oracle/_ref), where the tree exists
    return _LIB_PATH


class Structure(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("num_residues_raw", C.c_int32), ("num_atoms", C.c_int32),
        ("n_xyz", f32p), ("ca_xyz", f32p), ("cb_xyz", f32p), ("cb_ok", u8p), ("resname", u8p),
        ("aa", u8p), ("chain", u8p), ("serial", u64p), ("bfac", f32p),
        ("num_chains", C.c_int32), ("chains", C.c_uint8 * 256),
    ]


class QueryMap(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("hash", u32p), ("qi", u64p), ("qj", u64p), ("is_primary", u8p), ("idf", f32p),
        ("n_indices", C.c_uint64), ("indices", u64p),
        ("n_aad", C.c_uint64), ("aad_aa1", u8p), ("aad_aa2", u8p), ("aad_dist", f32p), ("aad_qi", u64p),
    ]


class QuerySpec(C.Structure):
    _fields_ = [("n", C.c_uint64), ("chain", u8p), ("serial", u64p), ("subs", C.POINTER(u8p)), ("n_subs", u64p)]


class CountResult(C.Structure):
    _fields_ = [("nid", C.c_uint64), ("total_match_count", C.c_uint64), ("node_count", C.c_uint64),
                ("edge_count", C.c_uint64), ("idf", C.c_float)]


class Match(C.Structure):
    _fields_ = [("n", C.c_uint64), ("has", u8p), ("chain", u8p), ("serial", u64p), ("tindex", i64p),
                ("rmsd", C.c_float), ("idf", C.c_float), ("rot", C.c_float * 9), ("tran", C.c_float * 3)]


class Retrieval(C.Structure):
    _fields_ = [("n_matches", C.c_uint64), ("from_hash", C.POINTER(Match)), ("processed", C.POINTER(Match)),
                ("max_matching_node_count", C.c_uint64), ("min_rmsd_with_max_match", C.c_float),
                ("n_found", C.c_uint64), ("found_i", u64p), ("found_j", u64p), ("found_hash", u32p),
                ("n_cand", C.c_uint64), ("cand_qi", u64p), ("cand_i", u64p), ("cand_j", u64p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    SP = C.POINTER(Structure)
    L.fdo_read_pdb.restype = SP
    L.fdo_read_pdb.argtypes = [C.c_char_p]
    L.fdo_structure_from_atoms.restype = SP
    L.fdo_structure_from_atoms.argtypes = [C.c_int32, f32p, u8p, u8p, u64p, u8p, f32p]
    L.fdo_structure_from_packed.restype = SP
    L.fdo_structure_from_packed.argtypes = [C.c_int32, f32p, f32p, f32p, u8p, u8p, f32p]
    L.fdo_structure_free.argtypes = [SP]
    L.fdo_avg_plddt.restype = C.c_float
    L.fdo_avg_plddt.argtypes = [SP]
    L.fdo_get_index.restype = C.c_int64
    L.fdo_get_index.argtypes = [SP, C.c_uint8, C.c_uint64]
    L.fdo_map_aa_to_u8.restype = C.c_uint8
    L.fdo_map_aa_to_u8.argtypes = [C.c_char_p]
    L.fdo_map_u8_to_aa.restype = C.c_char_p
    L.fdo_map_u8_to_aa.argtypes = [C.c_uint8]
    L.fdo_pair_feature.restype = C.c_int
    L.fdo_pair_feature.argtypes = [SP, C.c_int64, C.c_int64, C.c_float, f32p]
    L.fdo_discretize.restype = C.c_uint32
    L.fdo_discretize.argtypes = [C.c_float] * 4
    L.fdo_hash_pdbtr.restype = C.c_uint32
    L.fdo_hash_pdbtr.argtypes = [f32p, C.c_uint64, C.c_uint64]
    L.fdo_reverse_hash_pdbtr.argtypes = [C.c_uint32, f32p]
    L.fdo_hash_is_symmetric.restype = C.c_int
    L.fdo_hash_is_symmetric.argtypes = [C.c_uint32]
    L.fdo_hash_structure.argtypes = [SP, C.c_uint64, C.c_uint64, C.c_float, C.POINTER(u32p), u64p]
    L.fdo_sort_dedup_u32.restype = C.c_uint64
    L.fdo_sort_dedup_u32.argtypes = [u32p, C.c_uint64]
    L.fdo_free.argtypes = [C.c_void_p]
    VP = C.c_void_p
    L.fdo_index_new.restype = VP
    L.fdo_index_new.argtypes = [C.c_uint32]
    L.fdo_index_count_single_entry.argtypes = [VP, C.c_uint32, C.c_uint64]
    L.fdo_index_allocate_entries.argtypes = [VP]
    L.fdo_index_add_single_entry.argtypes = [VP, C.c_uint32, C.c_uint64]
    L.fdo_index_finish.argtypes = [VP]
    L.fdo_index_save.restype = C.c_int
    L.fdo_index_save.argtypes = [VP, C.c_char_p]
    L.fdo_index_load.restype = VP
    L.fdo_index_load.argtypes = [C.c_char_p]
    L.fdo_index_free.argtypes = [VP]
    L.fdo_index_num_hashes.restype = C.c_uint64
    L.fdo_index_num_hashes.argtypes = [VP]
    L.fdo_index_hashes.restype = u32p
    L.fdo_index_hashes.argtypes = [VP]
    L.fdo_index_offsets.restype = u64p
    L.fdo_index_offsets.argtypes = [VP]
    L.fdo_index_values.restype = u8p
    L.fdo_index_values.argtypes = [VP]
    L.fdo_index_value_len.restype = C.c_uint64
    L.fdo_index_value_len.argtypes = [VP]
    L.fdo_index_get_entries.restype = C.c_uint64
    L.fdo_index_get_entries.argtypes = [VP, C.c_uint32, C.POINTER(u64p)]
    L.fdo_split_by_seven_bits.restype = C.c_uint64
    L.fdo_split_by_seven_bits.argtypes = [C.c_uint64, u8p]
    L.fdo_build_index.restype = VP
    L.fdo_build_index.argtypes = [C.POINTER(SP), C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_uint64, u64p, f32p]
    L.fdo_build_index_from_lists.restype = VP
    L.fdo_build_index_from_lists.argtypes = [u32p, u64p, C.c_uint64]
    L.fdo_build_index_from_lists_mt.restype = VP
    L.fdo_build_index_from_lists_mt.argtypes = [u32p, u64p, C.c_uint64, C.c_int]
    L.fdo_hash_batch.restype = C.c_int
    L.fdo_hash_batch.argtypes = [C.POINTER(SP), C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.POINTER(u32p), C.POINTER(u64p)]
    L.fdo_save_lookup.restype = C.c_int
    L.fdo_save_lookup.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u64p, f32p, u64p, C.c_uint64]
    L.fdo_save_type.restype = C.c_int
    L.fdo_save_type.argtypes = [C.c_char_p, C.c_uint64, C.c_float, C.c_uint64, C.c_uint64, C.c_uint64]
    L.fdo_format_f32_display.restype = C.c_int
    L.fdo_format_f32_display.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
    L.fdo_parse_query_string.restype = C.POINTER(QuerySpec)
    L.fdo_parse_query_string.argtypes = [C.c_char_p, C.c_uint8]
    L.fdo_query_spec_free.argtypes = [C.POINTER(QuerySpec)]
    L.fdo_make_query_map.restype = C.POINTER(QueryMap)
    L.fdo_make_query_map.argtypes = [SP, C.POINTER(QuerySpec), C.c_uint64, C.c_uint64, f32p, C.c_uint64, f32p,
                                     C.c_uint64, C.c_float, C.c_int, VP, C.c_float]
    L.fdo_query_map_free.argtypes = [C.POINTER(QueryMap)]
    L.fdo_count_query.restype = C.c_uint64
    L.fdo_count_query.argtypes = [C.POINTER(QueryMap), VP, u64p, C.c_uint64, C.c_float, C.c_float,
                                  C.POINTER(C.POINTER(CountResult))]
    L.fdo_index_borrow.restype = VP
    L.fdo_index_borrow.argtypes = [u32p, u64p, C.c_uint64, u8p, C.c_uint64]
    L.fdo_index_free_borrowed.argtypes = [VP]
    L.fdo_query_bench.restype = C.c_double
    L.fdo_query_bench.argtypes = [VP, u64p, C.c_uint64, u64p, f32p, f32p, f32p, u8p, C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_uint64,
                                  C.c_int, u64p, u64p, C.POINTER(C.c_double)]
    L.fdo_retrieve.restype = C.POINTER(Retrieval)
    L.fdo_retrieve.argtypes = [SP, SP, C.POINTER(QueryMap), C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float]
    L.fdo_retrieval_free.argtypes = [C.POINTER(Retrieval)]
    L.fdo_kabsch.restype = C.c_float
    L.fdo_kabsch.argtypes = [f32p, f32p, C.c_uint64, C.c_int, f32p, f32p]
    L.fdo_set_hash_type.restype = C.c_int
    L.fdo_set_hash_type.argtypes = [C.c_uint32]
    L.fdo_get_hash_type.restype = C.c_uint32
    L.fdo_hash_any.restype = C.c_uint32
    L.fdo_hash_any.argtypes = [f32p, C.c_uint64, C.c_uint64]
    L.fdo_set_multiple_bins.restype = C.c_int
    L.fdo_set_multiple_bins.argtypes = [C.c_uint64, u64p]
    L.fdo_lms_qcp.restype = C.c_float
    L.fdo_lms_qcp.argtypes = [f32p, f32p, C.c_uint64, f32p, f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fdo_metrics.restype = None
    L.fdo_metrics.argtypes = [f32p, f32p, C.c_uint64, f32p, f32p, f32p]
    _lib = L
    return L


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


# ---------------------------------------------------------------- high-level helpers
class OStructure:
    """Owning wrapper of an fdo_structure."""

    def __init__(self, ptr):
        if not ptr:
            raise IOError("oracle: could not read/build structure")
        self.ptr = ptr

    def __del__(self):
        try:
            lib().fdo_structure_free(self.ptr)
        except Exception:
            pass

    @property
    def n(self):
        return self.ptr.contents.n

    def arrays(self):
        s = self.ptr.contents
        n = s.n
        return dict(
            n_xyz=_np(s.n_xyz, 3 * n, np.float32).reshape(n, 3),
            ca_xyz=_np(s.ca_xyz, 3 * n, np.float32).reshape(n, 3),
            cb_xyz=_np(s.cb_xyz, 3 * n, np.float32).reshape(n, 3),
            cb_ok=_np(s.cb_ok, n, np.uint8), aa=_np(s.aa, n, np.uint8), chain=_np(s.chain, n, np.uint8),
            serial=_np(s.serial, n, np.uint64), bfac=_np(s.bfac, n, np.float32),
            resname=_np(s.resname, 3 * n, np.uint8).reshape(n, 3),
        )

    def avg_plddt(self):
        return float(lib().fdo_avg_plddt(self.ptr))

    def first_chain(self):
        s = self.ptr.contents
        return s.chains[0] if s.num_chains > 0 else ord("A")


def read_pdb(path: str) -> OStructure:
    return OStructure(lib().fdo_read_pdb(path.encode()))


def structure_from_packed(n_xyz, ca_xyz, cb_xyz, aa, cb_ok=None, bfac=None) -> OStructure:
    n = len(aa)
    a1, p1 = _f32(n_xyz)
    a2, p2 = _f32(ca_xyz)
    a3, p3 = _f32(cb_xyz)
    aa = np.ascontiguousarray(aa, dtype=np.uint8)
    ok = None if cb_ok is None else np.ascontiguousarray(cb_ok, dtype=np.uint8)
    bf = None if bfac is None else np.ascontiguousarray(bfac, dtype=np.float32)
    return OStructure(lib().fdo_structure_from_packed(
        n, p1, p2, p3, None if ok is None else ok.ctypes.data_as(u8p), aa.ctypes.data_as(u8p),
        None if bf is None else bf.ctypes.data_as(f32p)))


def hash_structure(s: OStructure, nbin_dist=0, nbin_angle=0, cutoff=20.0) -> np.ndarray:
    out = u32p()
    n = C.c_uint64()
    lib().fdo_hash_structure(s.ptr, nbin_dist, nbin_angle, cutoff, C.byref(out), C.byref(n))
    r = _np(out, n.value, np.uint32)
    lib().fdo_free(out)
    return r


def pair_hash(s: OStructure, i: int, j: int, cutoff=20.0):
    feat = (C.c_float * 9)()
    if not lib().fdo_pair_feature(s.ptr, i, j, cutoff, feat):
        return None
    return int(lib().fdo_hash_pdbtr(feat, 16, 4)), np.array(list(feat)[:7], dtype=np.float32)


class OIndex:
    def __init__(self, ptr):
        if not ptr:
            raise IOError("oracle: index null")
        self.ptr = ptr

    def __del__(self):
        try:
            lib().fdo_index_free(self.ptr)
        except Exception:
            pass

    @property
    def H(self):
        return int(lib().fdo_index_num_hashes(self.ptr))

    def hashes(self):
        return _np(lib().fdo_index_hashes(self.ptr), self.H, np.uint32)

    def offsets(self):
        return _np(lib().fdo_index_offsets(self.ptr), self.H + 1, np.uint64)

    def values(self):
        return _np(lib().fdo_index_values(self.ptr), int(lib().fdo_index_value_len(self.ptr)), np.uint8)

    def entries(self, h: int) -> np.ndarray:
        out = u64p()
        n = lib().fdo_index_get_entries(self.ptr, h, C.byref(out))
        r = _np(out, n, np.uint64)
        if n:
            lib().fdo_free(out)
        return r

    def save(self, prefix: str):
        if lib().fdo_index_save(self.ptr, prefix.encode()) != 0:
            raise IOError("oracle: index save failed")


def build_index(structs, nbin_dist=0, nbin_angle=0, cutoff=20.0, max_residue=65535):
    S = len(structs)
    arr = (C.POINTER(Structure) * S)(*[s.ptr for s in structs])
    nres = np.zeros(S, dtype=np.uint64)
    plddt = np.zeros(S, dtype=np.float32)
    ix = lib().fdo_build_index(arr, S, nbin_dist, nbin_angle, cutoff, max_residue,
                               nres.ctypes.data_as(u64p), plddt.ctypes.data_as(f32p))
    return OIndex(ix), nres, plddt


def build_index_from_lists(hashes: np.ndarray, off: np.ndarray) -> OIndex:
    hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    return OIndex(lib().fdo_build_index_from_lists(hashes.ctypes.data_as(u32p), off.ctypes.data_as(u64p), len(off) - 1))


def build_index_from_lists_mt(hashes: np.ndarray, off: np.ndarray, n_threads: int) -> OIndex:
    hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    return OIndex(lib().fdo_build_index_from_lists_mt(hashes.ctypes.data_as(u32p), off.ctypes.data_as(u64p), len(off) - 1, n_threads))


def hash_batch(structs, nbin_dist=0, nbin_angle=0, cutoff=20.0, n_threads=0):
    """sorted-unique per-structure hash lists as CSR (OpenMP over structures; n_threads > 0 sets the thread count of the region)."""
    S = len(structs)
    if n_threads:
        lib().fdo_set_threads(int(n_threads))
    arr = (C.POINTER(Structure) * S)(*[s.ptr for s in structs])
    oh, oo = u32p(), u64p()
    lib().fdo_hash_batch(arr, S, nbin_dist, nbin_angle, cutoff, C.byref(oh), C.byref(oo))
    off = _np(oo, S + 1, np.uint64)
    h = _np(oh, int(off[-1]), np.uint32)
    lib().fdo_free(oh)
    lib().fdo_free(oo)
    return h, off


def load_index(prefix: str) -> OIndex:
    return OIndex(lib().fdo_index_load(prefix.encode()))


class OQueryMap:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            lib().fdo_query_map_free(self.ptr)
        except Exception:
            pass

    def arrays(self):
        m = self.ptr.contents
        n = m.n
        return dict(hash=_np(m.hash, n, np.uint32), qi=_np(m.qi, n, np.uint64), qj=_np(m.qj, n, np.uint64),
                    is_primary=_np(m.is_primary, n, np.uint8), idf=_np(m.idf, n, np.float32),
                    indices=_np(m.indices, m.n_indices, np.uint64),
                    aad_aa1=_np(m.aad_aa1, m.n_aad, np.uint8), aad_aa2=_np(m.aad_aa2, m.n_aad, np.uint8),
                    aad_dist=_np(m.aad_dist, m.n_aad, np.float32), aad_qi=_np(m.aad_qi, m.n_aad, np.uint64))


def parse_query_string(q: str, default_chain: int = ord("A")):
    sp = lib().fdo_parse_query_string(q.encode(), default_chain)
    if not sp:
        raise ValueError("oracle: malformed query string (reference would panic)")
    s = sp.contents
    out = []
    for k in range(s.n):
        subs = None
        if s.subs[k]:
            subs = [s.subs[k][z] for z in range(s.n_subs[k])]
        out.append((s.chain[k], int(s.serial[k]), subs))
    return sp, out


def make_query_map(qs: OStructure, query: str, index: OIndex | None, total_structures: float,
                   dist_thr=(0.5,), angle_thr=(5.0,), nbin_dist=0, nbin_angle=0, cutoff=20.0, serial_query=False):
    sp, _ = parse_query_string(query, qs.first_chain())
    d, dp = _f32(list(dist_thr))
    a, ap = _f32(list(angle_thr))
    m = lib().fdo_make_query_map(qs.ptr, sp, nbin_dist, nbin_angle, dp, len(d), ap, len(a), cutoff,
                                 int(serial_query), index.ptr if index is not None else None, total_structures)
    lib().fdo_query_spec_free(sp)
    return OQueryMap(m)


def count_query(m: OQueryMap, index: OIndex, nres: np.ndarray, freq_filter=-1.0, length_penalty=0.5):
    nres = np.ascontiguousarray(nres, dtype=np.uint64)
    out = C.POINTER(CountResult)()
    n = lib().fdo_count_query(m.ptr, index.ptr, nres.ctypes.data_as(u64p), len(nres), freq_filter, length_penalty,
                              C.byref(out))
    res = [dict(nid=int(out[k].nid), total_match_count=int(out[k].total_match_count),
                node_count=int(out[k].node_count), edge_count=int(out[k].edge_count), idf=float(out[k].idf))
           for k in range(n)]
    lib().fdo_free(out)
    return res


def _match_to_py(mt):
    res = []
    for k in range(mt.n):
        res.append((chr(mt.chain[k]), int(mt.serial[k]), int(mt.tindex[k])) if mt.has[k] else None)
    return dict(residues=res, rmsd=float(mt.rmsd), idf=float(mt.idf), rot=np.array(list(mt.rot), dtype=np.float32),
                tran=np.array(list(mt.tran), dtype=np.float32))


def retrieve(target: OStructure, query: OStructure, m: OQueryMap, node_count=2, nbin_dist=0, nbin_angle=0,
             cutoff=20.0, ca_distance_cutoff=1.0):
    r = lib().fdo_retrieve(target.ptr, query.ptr, m.ptr, node_count, nbin_dist, nbin_angle, cutoff, ca_distance_cutoff)
    R = r.contents
    out = dict(
        from_hash=[_match_to_py(R.from_hash[c]) for c in range(R.n_matches)],
        processed=[_match_to_py(R.processed[c]) for c in range(R.n_matches)],
        max_matching_node_count=int(R.max_matching_node_count),
        min_rmsd_with_max_match=float(R.min_rmsd_with_max_match),
        found=np.stack([_np(R.found_i, R.n_found, np.uint64), _np(R.found_j, R.n_found, np.uint64),
                        _np(R.found_hash, R.n_found, np.uint32).astype(np.uint64)], axis=1) if R.n_found else np.zeros((0, 3), np.uint64),
        cand=np.stack([_np(R.cand_qi, R.n_cand, np.uint64), _np(R.cand_i, R.n_cand, np.uint64),
                       _np(R.cand_j, R.n_cand, np.uint64)], axis=1) if R.n_cand else np.zeros((0, 3), np.uint64),
    )
    lib().fdo_retrieval_free(r)
    return out


def kabsch(x: np.ndarray, y: np.ndarray, mode=2):
    x, xp = _f32(x)
    y, yp = _f32(y)
    rot = (C.c_float * 9)()
    tran = (C.c_float * 3)()
    r = lib().fdo_kabsch(xp, yp, len(x.reshape(-1, 3)), mode, rot, tran)
    return float(r), np.array(list(rot), dtype=np.float32).reshape(3, 3), np.array(list(tran), dtype=np.float32)


class hash_type:
    """with oracle.hash_type(7): ...  — every oracle function inside encodes with that HashType (0 PDBMotif, 1 PDBMotifSinCos,
    3 PDBTrRosetta, 7 FolddiscoAngle, 8 FolddiscoDist); restored on exit"""

    def __init__(self, t: int):
        self.t = int(t)

    def __enter__(self):
        self.prev = int(lib().fdo_get_hash_type())
        if lib().fdo_set_hash_type(self.t) != 0:
            raise ValueError(f"oracle: hash type {self.t} is not restated")
        return self

    def __exit__(self, *exc):
        lib().fdo_set_hash_type(self.prev)
        return False


class multiple_bins:
    """with oracle.multiple_bins([(16, 4), (8, 3)]): ...  — the --multiple-bins list for every oracle function inside"""

    def __init__(self, pairs):
        self.pairs = [(int(d), int(a)) for d, a in pairs]

    def __enter__(self):
        arr = np.array(self.pairs, np.uint64).reshape(-1)
        if lib().fdo_set_multiple_bins(len(self.pairs), arr.ctypes.data_as(u64p)) != 0:
            raise ValueError("oracle: at most 8 bin pairs")
        return self

    def __exit__(self, *exc):
        lib().fdo_set_multiple_bins(0, None)
        return False


def hash_any(feature, nbin_dist=0, nbin_angle=0) -> int:
    f = np.zeros(9, np.float32)
    f[: len(feature)] = np.asarray(feature, np.float32)
    return int(lib().fdo_hash_any(f.ctypes.data_as(f32p), nbin_dist, nbin_angle))


def lms_qcp(x: np.ndarray, y: np.ndarray):
    """partial (least-median-of-squares) superposition of x onto y (src/structure/lms_qcp.rs, default parameters):
    (rms over the core, rot, tran, core indices in insertion order)"""
    x, xp = _f32(x)
    y, yp = _f32(y)
    n = len(x.reshape(-1, 3))
    rot = np.zeros(9, np.float32)
    tran = np.zeros(3, np.float32)
    core = np.zeros(n, np.uint64)
    nc = C.c_uint64(0)
    r = lib().fdo_lms_qcp(xp, yp, n, rot.ctypes.data_as(f32p), tran.ctypes.data_as(f32p), core.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(nc))
    return float(r), rot.reshape(3, 3), tran, core[:nc.value].copy()


def metrics(ref: np.ndarray, mov: np.ndarray, rot: np.ndarray, tran: np.ndarray) -> np.ndarray:
    """{tm_score, gdt_ts, gdt_ha, chamfer, hausdorff} of rot * mov + tran against ref (src/structure/metrics.rs)"""
    r_, rp = _f32(ref)
    m_, mp = _f32(mov)
    ro, rop = _f32(rot)
    t_, tp = _f32(tran)
    out = np.zeros(5, np.float32)
    lib().fdo_metrics(rp, mp, len(r_.reshape(-1, 3)), rop, tp, out.ctypes.data_as(f32p))
    return out


def query_bench(hashes, offsets, values, nres, res_off, n_xyz, ca_xyz, cb_xyz, aa, queries, top_n=1000, match_top=32, n_threads=1):
    """bench.py's query cpu_baseline (fdo_query_bench): queries = [(structure index, residue indices)] against the index given as
    arrays (borrowed, not copied).  -> dict(wall_s, hits, matches, stage thread-seconds)"""
    hashes = np.ascontiguousarray(hashes, np.uint32); offsets = np.ascontiguousarray(offsets, np.uint64); values = np.ascontiguousarray(values, np.uint8)
    nres = np.ascontiguousarray(nres, np.uint64); res_off = np.ascontiguousarray(res_off, np.uint64)
    n_xyz, ca_xyz, cb_xyz = (np.ascontiguousarray(a, np.float32) for a in (n_xyz, ca_xyz, cb_xyz))
    aa = np.ascontiguousarray(aa, np.uint8)
    qs = np.ascontiguousarray([q[0] for q in queries], np.uint64)
    q_off = np.concatenate([[0], np.cumsum([len(q[1]) for q in queries])]).astype(np.uint64)
    q_res = np.ascontiguousarray(np.concatenate([np.asarray(q[1], np.uint64) for q in queries]), np.uint64)
    L = lib()
    ix = L.fdo_index_borrow(hashes.ctypes.data_as(u32p), offsets.ctypes.data_as(u64p), len(hashes), values.ctypes.data_as(u8p), len(values))
    hits, matches = C.c_uint64(), C.c_uint64()
    st = (C.c_double * 3)()
    try:
        wall = L.fdo_query_bench(ix, nres.ctypes.data_as(u64p), len(nres), res_off.ctypes.data_as(u64p), n_xyz.ctypes.data_as(f32p),
                                 ca_xyz.ctypes.data_as(f32p), cb_xyz.ctypes.data_as(f32p), aa.ctypes.data_as(u8p), len(queries),
                                 qs.ctypes.data_as(u64p), q_off.ctypes.data_as(u64p), q_res.ctypes.data_as(u64p), top_n, match_top, n_threads,
                                 C.byref(hits), C.byref(matches), st)
    finally:
        L.fdo_index_free_borrowed(ix)
    return dict(wall_s=float(wall), hits=int(hits.value), matches=int(matches.value),
                stage_thread_s=dict(make_query_map=st[0], count_query=st[1], retrieval=st[2]))


class BorrowedIndex(OIndex):
    """OIndex over arrays that stay the caller's (e.g. the export of a GPU-built index, which the small-size parity tests pin byte
    for byte to build_index): lets the oracle answer queries at database sizes its own table build would need minutes for."""

    def __init__(self, hashes, offsets, values):
        self._keep = (np.ascontiguousarray(hashes, np.uint32), np.ascontiguousarray(offsets, np.uint64), np.ascontiguousarray(values, np.uint8))
        h, o, v = self._keep
        L = lib()
        super().__init__(L.fdo_index_borrow(h.ctypes.data_as(u32p), o.ctypes.data_as(u64p), len(h), v.ctypes.data_as(u8p), len(v)))

    def __del__(self):
        try:
            lib().fdo_index_free_borrowed(self.ptr)
        except Exception:
            pass


class _AtomT(C.Structure):      # lib/foldcomp/foldcompffi.h atom_t
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("atom", C.c_char * 4), ("atomIdx", C.c_uint64), ("chain", C.c_char),
                ("aa", C.c_char * 3), ("resIdx", C.c_uint64), ("bfactor", C.c_float)]


FCZ_ATOM_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("b", np.float32), ("name", "S4"), ("res", "S3"), ("chain", np.uint8),
                           ("rser", np.uint64)])


def foldcomp_ref_available() -> bool:
    return os.path.exists(_REF_FOLDCOMP)


def foldcomp_ref_decode(entry: bytes) -> np.ndarray:
    """the REFERENCE's Foldcomp decoder (oracle/_ref/libfoldcomp_ref.so = lib/foldcomp built as it lies): foldcomp_create /
    foldcomp_process / foldcomp_free / foldcomp_destroy exactly as src/structure/io/fcz.rs:82-93 calls them -> FCZ_ATOM_DTYPE records"""
    L = C.CDLL(_REF_FOLDCOMP)
    L.foldcomp_create.restype = C.c_void_p
    L.foldcomp_process.restype = C.POINTER(_AtomT)
    L.foldcomp_process.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.foldcomp_free.argtypes = [C.POINTER(_AtomT)]
    L.foldcomp_destroy.argtypes = [C.c_void_p]
    inst = L.foldcomp_create()
    n = C.c_size_t()
    p = L.foldcomp_process(inst, entry, len(entry), C.byref(n))
    out = np.zeros(n.value, FCZ_ATOM_DTYPE)
    for k in range(n.value):
        a = p[k]
        out[k] = (a.x, a.y, a.z, a.bfactor, bytes(a.atom), bytes(a.aa), a.chain[0], a.resIdx)
    L.foldcomp_free(p)
    L.foldcomp_destroy(inst)
    return out
