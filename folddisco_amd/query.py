"""Query-side host mirror: parse_query_string (src/controller/query.rs:331-384), make_query_map, retrieval and the
query_pdb workflow (src/cli/workflows/query_pdb.rs:348-519) over the C ABI.  Numerics (features, hashes, postings,
pair scan, Kabsch) run in libfdgpu.so; this module only marshals and formats."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import HashParams, MatchRec, QueryMap, f32p, owned_view, u8p, u32p, u64p
from .api import Batch, Context, FolddiscoIndex, PackedStructures, count_query, length_penalty
from .structure import CompactStructure

_ONE = "ARNDCQEGHILKMFPSTWYV"
_GROUPS = {"B": [2, 3], "Z": [5, 6], "X": list(range(20)), "x": list(range(20)), "J": [9, 10], "U": [4], "O": [11],
           "p": [1, 8, 11], "n": [3, 6], "h": [2, 5, 15, 16, 18], "b": [0, 4, 7, 9, 10, 12, 13, 14, 19], "a": [8, 13, 17, 18]}


def _one_letter(c: str):
    if c in _ONE:
        return [_ONE.index(c)]
    return _GROUPS.get(c, [255])


def parse_query_string(q: str, default_chain: int = ord("A")):
    """-> list of (chain u8, serial, substitutions or None); raises ValueError where the reference panics."""
    out = []
    if not q:
        return out
    if not chr(default_chain).isalpha() or default_chain > 127:
        default_chain = ord("A")
    for seg in q.replace(" ", "").split(","):
        chain = default_chain
        if seg and seg[0].isascii() and seg[0].isalpha():
            chain, seg = ord(seg[0]), seg[1:]
        subs = None
        if ":" in seg:
            seg, s = seg.split(":", 1)
            subs = [v for ch in s if ch.isascii() and ch.isalpha() for v in _one_letter(ch)]
        def num(t):
            t2 = t[1:] if t.startswith("+") else t
            if not t2.isdigit():
                raise ValueError(f"Invalid residue '{t}'")
            return int(t2)
        if "-" in seg:
            a, b = seg.split("-", 1)
            for r in range(num(a), num(b) + 1):
                out.append((chain, r, None if subs is None else list(subs)))
        else:
            out.append((chain, num(seg), subs))
    return out


def res_chain_to_string(res):
    return ",".join(f"{chr(c)}{r}" for c, r, _ in res)


_QM_FIELDS = {   # attribute -> (pointer field, length field, dtype) of fd_query_map
    "hash": ("hash", "n", np.uint32), "qi": ("qi", "n", np.uint32), "qj": ("qj", "n", np.uint32), "is_primary": ("is_primary", "n", np.uint8),
    "idf": ("idf", "n", np.float32), "indices": ("indices", "n_indices", np.uint32), "aad_aa1": ("aad_aa1", "n_aad", np.uint8),
    "aad_aa2": ("aad_aa2", "n_aad", np.uint8), "aad_dist": ("aad_dist", "n_aad", np.float32), "aad_qi": ("aad_qi", "n_aad", np.uint32),
    "primary_hash": ("primary_hash", "n", np.uint32),
}


class QueryMapResult:
    """fd_query_map handle + numpy copies of its arrays, made on first use (a batch of queries that goes straight from
    fdgpu_make_query_map_batch to fdgpu_count_query_maps_top / fdgpu_retrieve_batch never needs them on the Python side)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.handle, self._cache = ctx, handle, {}

    def __getattr__(self, name):
        f = _QM_FIELDS.get(name)
        if f is None:
            raise AttributeError(name)
        c = self.__dict__["_cache"]
        if name not in c:
            m = self.handle.contents
            c[name] = _arr(getattr(m, f[0]), getattr(m, f[1]), f[2])
        return c[name]

    def set_idf(self, idf: np.ndarray):
        """overwrite the per-entry idf (also inside the C struct fdgpu_retrieve reads): a caller that shards the index computes it
        from GLOBAL posting lengths of primary_hash (dist.global_posting_lengths)"""
        idf = np.ascontiguousarray(idf, np.float32)
        assert len(idf) == int(self.handle.contents.n)
        C.memmove(self.handle.contents.idf, idf.ctypes.data, idf.nbytes)
        self._cache["idf"] = idf.copy()

    def __del__(self):
        try:
            if self.handle is not None and self.ctx is not None and self.ctx.h:
                self.ctx.L.fdgpu_query_map_free(self.handle)
        except Exception:
            pass


def _arr(ptr, n, dt):
    n = int(n)
    if n == 0:
        return np.zeros(0, dt)
    return np.frombuffer((ptr._type_ * n).from_address(C.addressof(ptr.contents)), dtype=dt).copy()


def make_query_map(ctx: Context, qbatch: Batch, q_indices, subs=None, index: FolddiscoIndex | None = None,
                   total_structures: float = 0.0, dist_thr=(0.5,), angle_thr=(5.0,), nbin_dist=0, nbin_angle=0,
                   dist_cutoff=20.0, hash_type=3, multiple_bins=None) -> QueryMapResult:
    qi = np.ascontiguousarray(q_indices, dtype=np.uint32)
    n = len(qi)
    sub_ptrs = (u8p * max(n, 1))()
    n_subs = np.zeros(max(n, 1), np.uint32)
    keep = []
    if subs is not None:
        for k, s in enumerate(subs):
            if s is not None:
                a = np.ascontiguousarray(s if len(s) else [0], dtype=np.uint8)
                keep.append(a)
                sub_ptrs[k] = a.ctypes.data_as(u8p)
                n_subs[k] = len(s)
    d = np.ascontiguousarray(dist_thr, np.float32)
    a = np.ascontiguousarray(angle_thr, np.float32)
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    out = C.POINTER(QueryMap)()
    ctx.check(ctx.L.fdgpu_make_query_map(ctx.h, qbatch.h, qi.ctypes.data_as(u32p), n, sub_ptrs, n_subs.ctypes.data_as(u32p),
                                         d.ctypes.data_as(f32p), len(d), a.ctypes.data_as(f32p), len(a), C.byref(p),
                                         index.h if index is not None else None, total_structures, C.byref(out)))
    return _wrap_query_map(ctx, out)


def _wrap_query_map(ctx, out) -> QueryMapResult:
    return QueryMapResult(ctx, out)


def make_query_maps(ctx: Context, qbatch: Batch, queries, index: FolddiscoIndex | None = None, total_structures: float = 0.0,
                    dist_thr=(0.5,), angle_thr=(5.0,), nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, hash_type=3, multiple_bins=None):
    """Many query maps with three launches in total (fdgpu_make_query_map_batch).  queries: list of (structure index in
    qbatch, residue indices[, substitution lists]).  -> list of QueryMapResult."""
    nq = len(queries)
    q_struct = np.ascontiguousarray([q[0] for q in queries], np.uint32)
    idx = [np.ascontiguousarray(q[1], np.uint32) for q in queries]
    q_off = np.concatenate([[0], np.cumsum([len(x) for x in idx])]).astype(np.uint64)
    q_index = np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros(0, np.uint32))
    ntot = len(q_index)
    sub_ptrs = (u8p * max(ntot, 1))()
    n_subs = np.zeros(max(ntot, 1), np.uint32)
    keep = []
    for t, q in enumerate(queries):
        if len(q) > 2 and q[2] is not None:
            for k, sl in enumerate(q[2]):
                if sl is not None:
                    a = np.ascontiguousarray(sl if len(sl) else [0], dtype=np.uint8)
                    keep.append(a)
                    sub_ptrs[int(q_off[t]) + k] = a.ctypes.data_as(u8p)
                    n_subs[int(q_off[t]) + k] = len(sl)
    d = np.ascontiguousarray(dist_thr, np.float32)
    a = np.ascontiguousarray(angle_thr, np.float32)
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    outs = (C.POINTER(QueryMap) * max(nq, 1))()
    ctx.check(ctx.L.fdgpu_make_query_map_batch(ctx.h, qbatch.h, nq, q_struct.ctypes.data_as(u32p), q_off.ctypes.data_as(u64p),
                                               q_index.ctypes.data_as(u32p), sub_ptrs, n_subs.ctypes.data_as(u32p), d.ctypes.data_as(f32p), len(d),
                                               a.ctypes.data_as(f32p), len(a), C.byref(p), index.h if index is not None else None,
                                               total_structures, outs))
    return [_wrap_query_map(ctx, outs[t]) for t in range(nq)]


def retrieve(ctx: Context, db: Batch, resname_std, cand, qm: QueryMapResult, qbatch: Batch, ca_distance_cutoff=1.0,
             node_count=2, nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, partial_fit=False, hash_type=3, multiple_bins=None):
    """-> list of dicts per match: cand slot, idf, rmsd, from_hash / processed target residue indices (-1 = none)."""
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    std = None if resname_std is None else np.ascontiguousarray(resname_std, np.uint8)
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    mp = C.POINTER(MatchRec)()
    rp = C.POINTER(C.c_int32)()
    nm = C.c_uint64()
    ctx.check(ctx.L.fdgpu_retrieve(ctx.h, db.h, None if std is None else std.ctypes.data_as(u8p), cand.ctypes.data_as(u32p), len(cand),
                                   qm.handle, qbatch.h, C.byref(p), ca_distance_cutoff, node_count, int(bool(partial_fit)), C.byref(mp), C.byref(nm), C.byref(rp)))
    nq = len(qm.indices)
    n = int(nm.value)
    out = _match_dicts(mp, rp, n, nq)
    ctx.L.fdgpu_matches_free(mp, rp)
    return out


def _match_dicts(mp, rp, n, nq, first=0, res_base=0):
    """n match records from mp[first ..) with their 2 * nq residue ints each from rp[res_base ..) as the dicts retrieve() returns — through
    numpy views of the two arrays (element-wise ctypes reads of a whole-structure query's 10^5 residue ints cost milliseconds)"""
    if n == 0:
        return []
    assert C.sizeof(MatchRec) == MATCH_DTYPE.itemsize
    recs = np.frombuffer((C.c_uint8 * ((first + n) * MATCH_DTYPE.itemsize)).from_address(C.addressof(mp.contents)), dtype=MATCH_DTYPE)[first:].copy()
    res = (np.frombuffer((C.c_int32 * (res_base + 2 * nq * n)).from_address(C.addressof(rp.contents)), dtype=np.int32)[res_base:].reshape(n, 2, nq).tolist()
           if nq else [[[], []] for _ in range(n)])
    out = []
    for k in range(n):
        r = recs[k]
        out.append(dict(cand=int(r["cand"]), idf=float(r["idf"]), rmsd=float(r["rmsd"]), rmsd_from_hash=float(r["rmsd_from_hash"]), same=bool(r["same"]),
                        from_hash=res[k][0], processed=res[k][1],
                        rot=r["rot"].reshape(3, 3).copy(), tran=r["tran"].copy(), metrics=r["metrics"].copy(),
                        rot_from_hash=r["rot_from_hash"].reshape(3, 3).copy(), tran_from_hash=r["tran_from_hash"].copy(),
                        metrics_from_hash=r["metrics_from_hash"].copy()))
    return out


_MATCH_KEYS = {"tm_score": ("tm_score", -1), "tm-score": ("tm_score", -1), "tmscore": ("tm_score", -1), "tm": ("tm_score", -1),
               "gdt_ts": ("gdt_ts", -1), "gdt-ts": ("gdt_ts", -1), "gdtts": ("gdt_ts", -1), "gdt": ("gdt_ts", -1), "gdt_ha": ("gdt_ha", -1), "gdt-ha": ("gdt_ha", -1),
               "gdtha": ("gdt_ha", -1), "chamfer_distance": ("chamfer_distance", 1), "chamfer-distance": ("chamfer_distance", 1),
               "chamfer": ("chamfer_distance", 1), "hausdorff_distance": ("hausdorff_distance", 1),
               "hausdorff-distance": ("hausdorff_distance", 1), "hausdorff": ("hausdorff_distance", 1),
               "node_count": ("node_count", -1), "node-count": ("node_count", -1), "nodes": ("node_count", -1), "node": ("node_count", -1),
               "n": ("node_count", -1), "idf": ("idf", -1), "score": ("idf", -1), "rmsd": ("rmsd", 1)}
_STRUCT_KEYS = {"max_node_count": ("max_matching_node_count", -1), "max-node-count": ("max_matching_node_count", -1),
                "max_node": ("max_matching_node_count", -1), "max-node": ("max_matching_node_count", -1),
                "max_nodes": ("max_matching_node_count", -1), "max-nodes": ("max_matching_node_count", -1),
                "node_count": ("node_count", -1), "node-count": ("node_count", -1), "nodes": ("node_count", -1), "node": ("node_count", -1),
                "n": ("node_count", -1), "idf": ("idf", -1), "score": ("idf", -1), "min_rmsd": ("min_rmsd_with_max_match", 1),
                "min-rmsd": ("min_rmsd_with_max_match", 1), "rmsd": ("min_rmsd_with_max_match", 1),
                "total_match_count": ("total_match_count", -1), "total-match-count": ("total_match_count", -1),
                "total_match": ("total_match_count", -1), "total-match": ("total_match_count", -1), "matches": ("total_match_count", -1),
                "match": ("total_match_count", -1), "edge_count": ("edge_count", -1), "edge-count": ("edge_count", -1),
                "edges": ("edge_count", -1), "edge": ("edge_count", -1), "e": ("edge_count", -1), "nres": ("nres", -1),
                "num_residues": ("nres", -1), "num-residues": ("nres", -1), "length": ("nres", -1), "residues": ("nres", -1),
                "residue": ("nres", -1), "l": ("nres", -1), "plddt": ("plddt", -1)}


def parse_sort_by(s: str, per_structure: bool):
    """--sort-by grammar of src/controller/sort.rs:160-204 / 400-440: comma-separated keys, optional :asc / :desc, default
    order per key; empty -> default strategy (idf desc, rmsd asc).  Returns [(dict key, sign)] (sign -1 = descending)."""
    table = _STRUCT_KEYS if per_structure else _MATCH_KEYS
    out = []
    for part in (s or "").split(","):
        part = part.strip()
        if not part:
            continue
        comps = part.split(":")
        if len(comps) > 2:
            raise ValueError(f"Invalid format: '{part}'. Use 'key:order' or just 'key'")
        k = comps[0].strip().lower()
        if k not in table:
            raise ValueError(f"Unknown sort key: '{comps[0]}'")
        field, sign = table[k]
        if len(comps) == 2:
            o = comps[1].strip().lower()
            if o in ("asc", "ascending", "a"):
                sign = 1
            elif o in ("desc", "descending", "d"):
                sign = -1
            else:
                raise ValueError(f"Unknown sort order: '{comps[1]}'")
        out.append((field, sign))
    if not out:
        out = [("idf", -1), ("min_rmsd_with_max_match" if per_structure else "rmsd", 1)]
    return out


def sort_rows(rows, strategy):
    """stable multi-key sort (Vec::sort_by with the strategy's compare)"""
    for field, sign in reversed(strategy):
        rows.sort(key=lambda r: sign * r[field])
    return rows


def sample_query_hashes(index: FolddiscoIndex, q_hash, sampling_ratio=None, sampling_count=None):
    """sample_query (src/controller/count_query.rs:222-253): keep the hashes with the shortest posting lists (stable sort by
    length); ratio -> ceil(ratio * n) evaluated in f32; both or neither given -> all.  Returns the kept positions."""
    n = len(q_hash)
    if (sampling_ratio is None) == (sampling_count is None):
        return np.arange(n)
    lens = index.posting_lengths(np.ascontiguousarray(q_hash, np.uint32))
    order = np.argsort(lens, kind="stable")
    if sampling_ratio is not None:
        keep = int(np.ceil(np.float32(sampling_ratio) * np.float32(n)))
    else:
        keep = int(sampling_count)
    return order[: max(0, min(keep, n))]


MATCH_DTYPE = np.dtype([("cand", np.uint32), ("same", np.uint32), ("idf", np.float32), ("rmsd", np.float32), ("rmsd_from_hash", np.float32),
                        ("rot", np.float32, (9,)), ("tran", np.float32, (3,)), ("metrics", np.float32, (5,)),
                        ("rot_from_hash", np.float32, (9,)), ("tran_from_hash", np.float32, (3,)), ("metrics_from_hash", np.float32, (5,))])


def retrieve_batch(ctx: Context, db: Batch, resname_std, cands, qms, qbatch: Batch, q_structs, ca_distance_cutoff=1.0, node_count=2,
                   nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, as_arrays=False, partial_fit=False, hash_type=3, multiple_bins=None):
    """retrieve() for many queries with one pair scan / gather / Kabsch launch in total (fdgpu_retrieve_batch).  cands[t]:
    candidate structure indices of query t (a list of arrays, or one [T, n] array), qms[t] its QueryMapResult, q_structs[t] its structure in qbatch.
    -> list (per query) of lists of match dicts like retrieve(); with as_arrays=True the raw tables instead:
    (matches MATCH_DTYPE[], match_off u64[T+1], residues int32[], res_off u64[T+1]) — 2 * len(qms[t].indices) residue indices
    per match (from-hash mapping, then processed mapping)."""
    T = len(qms)
    if isinstance(cands, np.ndarray) and cands.ndim == 2:       # [T, n] candidates per query as one array (no per-query list handling)
        cand = np.ascontiguousarray(cands, dtype=np.uint32).reshape(-1)
        cand_off = (np.arange(T + 1, dtype=np.uint64) * np.uint64(cands.shape[1]))
    else:
        cl = [np.ascontiguousarray(c, dtype=np.uint32) for c in cands]
        cand_off = np.concatenate([[0], np.cumsum([len(c) for c in cl])]).astype(np.uint64)
        cand = np.ascontiguousarray(np.concatenate(cl) if cl else np.zeros(0, np.uint32))
    std = None if resname_std is None else np.ascontiguousarray(resname_std, np.uint8)
    qs = np.ascontiguousarray(q_structs, np.uint32)
    handles = (C.POINTER(QueryMap) * max(T, 1))(*[q.handle for q in qms])
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    mp, rp = C.POINTER(MatchRec)(), C.POINTER(C.c_int32)()
    mo, ro = u64p(), u64p()
    ctx.check(ctx.L.fdgpu_retrieve_batch(ctx.h, db.h, None if std is None else std.ctypes.data_as(u8p), T, cand.ctypes.data_as(u32p),
                                         cand_off.ctypes.data_as(u64p), handles, qbatch.h, qs.ctypes.data_as(u32p), C.byref(p), ca_distance_cutoff,
                                         node_count, int(bool(partial_fit)), C.byref(mp), C.byref(mo), C.byref(rp), C.byref(ro)))
    if as_arrays:
        moff = np.ctypeslib.as_array(mo, shape=(T + 1,)).copy()
        roff = np.ctypeslib.as_array(ro, shape=(T + 1,)).copy()
        nm, nr = int(moff[-1]), int(roff[-1])
        assert C.sizeof(MatchRec) == MATCH_DTYPE.itemsize
        marr = owned_view(ctx.L, mp, nm * MATCH_DTYPE.itemsize, MATCH_DTYPE)       # no copies: the blocks go back to the library with the arrays
        rarr = owned_view(ctx.L, rp, nr * 4, np.int32)
        ctx.L.fdgpu_free(mo)
        ctx.L.fdgpu_free(ro)
        return marr, moff, rarr, roff
    out = []
    for t in range(T):
        out.append(_match_dicts(mp, rp, int(mo[t + 1]) - int(mo[t]), len(qms[t].indices), int(mo[t]), int(ro[t])))
    ctx.L.fdgpu_matches_free(mp, rp)
    ctx.L.fdgpu_free(mo)
    ctx.L.fdgpu_free(ro)
    return out


def _query_batch_args(queries, penalty, resname_std, dist_thr, angle_thr, nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins):
    """the argument block fdgpu_query_batch and fdgpu_query_batch_submit share -> (ctypes arguments between qb and the outputs, objects to keep alive)"""
    nq = len(queries)
    q_struct = np.ascontiguousarray([q[0] for q in queries], np.uint32)
    idx = [np.ascontiguousarray(q[1], np.uint32) for q in queries]
    q_off = np.concatenate([[0], np.cumsum([len(x) for x in idx])]).astype(np.uint64)
    q_index = np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros(0, np.uint32))
    ntot = len(q_index)
    sub_ptrs = (u8p * max(ntot, 1))()
    n_subs = np.zeros(max(ntot, 1), np.uint32)
    keep = []
    for t, q in enumerate(queries):
        if len(q) > 2 and q[2] is not None:
            for k, sl in enumerate(q[2]):
                if sl is not None:
                    a = np.ascontiguousarray(sl if len(sl) else [0], dtype=np.uint8)
                    keep.append(a)
                    sub_ptrs[int(q_off[t]) + k] = a.ctypes.data_as(u8p)
                    n_subs[int(q_off[t]) + k] = len(sl)
    d = np.ascontiguousarray(dist_thr, np.float32)
    a = np.ascontiguousarray(angle_thr, np.float32)
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    pen = None if penalty is None else np.ascontiguousarray(penalty, dtype=np.float32)
    std = None if resname_std is None else np.ascontiguousarray(resname_std, np.uint8)
    head = (None if std is None else std.ctypes.data_as(u8p),)
    mid = (nq, q_struct.ctypes.data_as(u32p), q_off.ctypes.data_as(u64p), q_index.ctypes.data_as(u32p), sub_ptrs, n_subs.ctypes.data_as(u32p), d.ctypes.data_as(f32p), len(d),
           a.ctypes.data_as(f32p), len(a), C.byref(p))
    return head, mid, (None if pen is None else pen.ctypes.data_as(f32p)), (q_struct, q_off, q_index, sub_ptrs, n_subs, keep, d, a, p, pen, std)


def _query_batch_results(ctx, nq, outs, rp_, ro_, mp, mo, resp, reso):
    from .api import REC_DTYPE
    maps = [_wrap_query_map(ctx, outs[t]) for t in range(nq)]
    rec_off = np.ctypeslib.as_array(ro_, shape=(nq + 1,)).copy()
    moff = np.ctypeslib.as_array(mo, shape=(nq + 1,)).copy()
    roff = np.ctypeslib.as_array(reso, shape=(nq + 1,)).copy()
    recs = owned_view(ctx.L, rp_, int(rec_off[-1]) * 20, REC_DTYPE)
    marr = owned_view(ctx.L, mp, int(moff[-1]) * MATCH_DTYPE.itemsize, MATCH_DTYPE)
    rarr = owned_view(ctx.L, resp, int(roff[-1]) * 4, np.int32)
    for x in (ro_, mo, reso):
        ctx.L.fdgpu_free(x)
    return maps, (recs, rec_off), (marr, moff, rarr, roff)


def query_batch(ctx: Context, index: FolddiscoIndex, db: Batch, qbatch: Batch, queries, total_structures: float, top_n: int, match_top: int,
                penalty=None, resname_std=None, dist_thr=(0.5,), angle_thr=(5.0,), ca_distance_cutoff=1.0, node_count=2, nbin_dist=0, nbin_angle=0,
                dist_cutoff=20.0, hash_type=3, multiple_bins=None):
    """make_query_maps + count_query_maps(top_n) + retrieve_batch over the first match_top records of every ranking in ONE library call
    (fdgpu_query_batch: the stages overlap inside it).  queries as make_query_maps.  -> (maps, (records REC_DTYPE[], rec_off), (matches
    MATCH_DTYPE[], match_off, residues int32[], res_off)) — the arrays the three calls return."""
    nq = len(queries)
    head, mid, pen, keep = _query_batch_args(queries, penalty, resname_std, dist_thr, angle_thr, nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    outs = (C.POINTER(QueryMap) * max(nq, 1))()
    rp_, ro_ = C.POINTER(_lib.CountRec)(), u64p()
    mp, resp = C.POINTER(MatchRec)(), C.POINTER(C.c_int32)()
    mo, reso = u64p(), u64p()
    ctx.check(ctx.L.fdgpu_query_batch(ctx.h, index.h, db.h, *head, qbatch.h, *mid, float(total_structures), pen,
                                      int(top_n), int(match_top), ca_distance_cutoff, node_count, outs, C.byref(rp_), C.byref(ro_), C.byref(mp), C.byref(mo),
                                      C.byref(resp), C.byref(reso)))
    del keep
    return _query_batch_results(ctx, nq, outs, rp_, ro_, mp, mo, resp, reso)


class QueryJob:
    """a batch handed to fdgpu_query_batch_submit; wait() returns what query_batch returns (once)"""

    def __init__(self, ctx, handle, nq, keep):
        self.ctx, self.h, self.nq, self._keep = ctx, handle, nq, keep

    def wait(self):
        if self.h is None:
            raise RuntimeError("QueryJob.wait() called twice")
        nq = self.nq
        outs = (C.POINTER(QueryMap) * max(nq, 1))()
        rp_, ro_ = C.POINTER(_lib.CountRec)(), u64p()
        mp, resp = C.POINTER(MatchRec)(), C.POINTER(C.c_int32)()
        mo, reso = u64p(), u64p()
        h, self.h = self.h, None
        rc = self.ctx.L.fdgpu_query_batch_wait(self.ctx.h, h, outs, C.byref(rp_), C.byref(ro_), C.byref(mp), C.byref(mo), C.byref(resp), C.byref(reso))
        self._keep = None
        self.ctx.check(rc)
        return _query_batch_results(self.ctx, nq, outs, rp_, ro_, mp, mo, resp, reso)

    def __del__(self):      # a job nobody waited for: collect and drop its results (the library requires exactly one wait per job)
        try:
            if self.h is not None and self.ctx.h:
                h, self.h = self.h, None
                self.ctx.L.fdgpu_query_batch_wait(self.ctx.h, h, None, None, None, None, None, None, None)
        except Exception:
            pass


def query_batch_submit(ctx: Context, index: FolddiscoIndex, db: Batch, qbatch: Batch, queries, total_structures: float, top_n: int, match_top: int,
                       penalty=None, resname_std=None, dist_thr=(0.5,), angle_thr=(5.0,), ca_distance_cutoff=1.0, node_count=2, nbin_dist=0, nbin_angle=0,
                       dist_cutoff=20.0, hash_type=3, multiple_bins=None) -> QueryJob:
    """query_batch without the wait: the batch goes to one of the context's query lanes (fdgpu_query_batch_submit) -> QueryJob.  One host
    thread keeps several batches in flight: submit k, k + 1, k + 2; wait k; submit k + 3; ..."""
    nq = len(queries)
    head, mid, pen, keep = _query_batch_args(queries, penalty, resname_std, dist_thr, angle_thr, nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    job = C.c_void_p()
    ctx.check(ctx.L.fdgpu_query_batch_submit(ctx.h, index.h, db.h, *head, qbatch.h, *mid, float(total_structures), pen, int(top_n), int(match_top),
                                             ca_distance_cutoff, node_count, C.byref(job)))
    return QueryJob(ctx, job, nq, (keep[-2], keep[-1], index, db, qbatch))      # borrowed until the wait: penalty, resname_std, index, batches


def query_batch_pipelined(ctx: Context, index: FolddiscoIndex, db: Batch, qbatch: Batch, batches, total_structures: float, top_n: int, match_top: int, depth: int = 3,
                          **kw):
    """generator over the results of `batches` (an iterable of query lists) with `depth` batches in flight, in submission order"""
    from collections import deque
    pend = deque()
    for b in batches:
        pend.append(query_batch_submit(ctx, index, db, qbatch, b, total_structures, top_n, match_top, **kw))
        if len(pend) >= depth:
            yield pend.popleft().wait()
    while pend:
        yield pend.popleft().wait()


def query_pdb(ctx: Context, index: FolddiscoIndex, db: Batch, db_structs: list[CompactStructure], tids: list[str], nres, plddt,
              query: CompactStructure, query_string: str, dist_thr=(0.5,), angle_thr=(5.0,), ca_distance=1.0, top_n=None,
              length_penalty_power=0.5, skip_match=False, serial_query=False, freq_filter=None, dist_cutoff=20.0, nbin_dist=0, nbin_angle=0,
              sampling_ratio=None, sampling_count=None, filters=None, sort_by="", shard=None, partial_fit=False, hash_type=3, multiple_bins=None,
              skip_ca_match=False, match_top_n="same", db_keys=None):
    """The per-query body of query_pdb (src/cli/workflows/query_pdb.rs:348-519).  shard = dict(lo=first structure id, device=torch
    device or None): `index`, `db` and `db_structs` then cover only structures [lo, lo + len(db_structs)) of the database that
    tids / nres / plddt describe: idf comes from all-reduced posting lengths, the touched-structure records and the
    matches found on the owning rank are all-gathered, every rank returns the same result as the single-index call.
    `filters`: the reference's filtering options
    (total_match, covered_node, covered_node_ratio, max_node, max_node_ratio, score, connected_node, connected_node_ratio,
    num_residue, plddt, rmsd; 0 / absent = off), applied as StructureFilter before / after matching and MatchFilter
    (controller/filter.rs:76-131, 194-235).  Returns (structure rows, match rows) as lists of dicts."""
    F = dict(total_match=0, covered_node=0, covered_node_ratio=0.0, max_node=0, max_node_ratio=0.0, score=0.0, connected_node=0,
             connected_node_ratio=0.0, num_residue=0, plddt=0.0, rmsd=0.0, tm_score=0.0, gdt_ts=0.0, gdt_ha=0.0, chamfer=0.0, hausdorff=0.0)
    F.update(filters or {})
    S = len(tids)
    qres = parse_query_string(query_string, query.chains[0] if query.chains else ord("A"))
    if qres:
        idx, subs = [], []
        for c, r, s in qres:
            k = r if serial_query else query.get_index(c, r)
            if k is not None:
                idx.append(k); subs.append(s)
    else:   # whole-structure query: every residue (query.rs:226-233); --serial-index takes the residue number itself as the index (:236-238)
        idx = [int(query.serial[k]) if serial_query else query.get_index(int(query.chain[k]), int(query.serial[k])) for k in range(query.n)]
        idx = [k for k in idx if k is not None]
        subs = [None] * len(idx)
    qbatch = ctx.upload(PackedStructures.concat([query.as_item()]))
    pen = length_penalty(nres, length_penalty_power)
    if shard is None:
        lo, n_local = 0, S
        qm = make_query_map(ctx, qbatch, idx, subs, index, float(S), dist_thr, angle_thr, nbin_dist=nbin_dist, nbin_angle=nbin_angle,
                            dist_cutoff=dist_cutoff, hash_type=hash_type, multiple_bins=multiple_bins)
        keep = sample_query_hashes(index, qm.hash, sampling_ratio, sampling_count)
        rows = count_query(ctx, index, qm.hash[keep], qm.qi[keep], qm.qj[keep], pen, total_structures=S, freq_filter=freq_filter)
    else:
        from . import dist as fdist
        from .api import idf_of_lengths
        # absolute: the shard index was loaded over the whole id range (fdgpu_index_load of a shard file) and takes the full
        # penalty vector; otherwise it was built in place with first_id = lo and covers n_local structures
        lo, dev = int(shard["lo"]), shard.get("device")
        n_local = int(shard.get("n_local", len(db_structs) if db_structs is not None else index.n_structures))
        qm = make_query_map(ctx, qbatch, idx, subs, None, float(S), dist_thr, angle_thr, nbin_dist=nbin_dist, nbin_angle=nbin_angle,
                            dist_cutoff=dist_cutoff, hash_type=hash_type, multiple_bins=multiple_bins)
        pl = fdist.global_posting_lengths(index, qm.primary_hash, dev)
        qm.set_idf(np.where(pl > 0, idf_of_lengths(np.maximum(pl, 1), S), 0.0).astype(np.float32))
        lens = fdist.global_posting_lengths(index, qm.hash, dev)
        order = np.argsort(lens, kind="stable")                    # sample_query on the global lengths
        if (sampling_ratio is None) == (sampling_count is None):
            keep = np.arange(len(lens))
        else:
            k = int(np.ceil(np.float32(sampling_ratio) * np.float32(len(lens)))) if sampling_ratio is not None else int(sampling_count)
            keep = order[: max(0, min(k, len(lens)))]
        local = count_query(ctx, index, qm.hash[keep], qm.qi[keep], qm.qj[keep], pen if shard.get("absolute") else pen[lo:lo + n_local], total_structures=S,
                            freq_filter=freq_filter, as_array=True, lengths=lens[keep])
        allrec = fdist.allgather_hits(local, dev if dev is not None else "cpu")
        allrec = allrec[np.argsort(allrec["nid"], kind="stable")]  # the single-index call starts from ascending nid
        rows = [dict(nid=int(r["nid"]), total_match_count=int(r["total_match_count"]), node_count=int(r["node_count"]),
                     edge_count=int(r["edge_count"]), idf=float(r["idf"])) for r in allrec]
    qnorm = res_chain_to_string(qres) if qres else query_string     # both output modes print the normalised form (query_pdb.rs:359-365)
    # residue_count (query_pdb.rs:354-358): the PARSED residues, resolved in the structure or not; all residues for an empty query
    n_expected = np.float32(len(qres) if qres else query.n)
    for r in rows:
        r.update(tid=tids[r["nid"]], nres=int(nres[r["nid"]]), plddt=float(plddt[r["nid"]]), db_key=r["nid"] if db_keys is None else int(db_keys[r["nid"]]), matches=[],
                 max_matching_node_count=0, min_rmsd_with_max_match=0.0, query_residues=qnorm)

    def before(r):   # StructureFilter::filter_before_matching
        ok = True
        if F["total_match"] > 0: ok = ok and r["total_match_count"] >= F["total_match"]
        if F["covered_node"] > 0: ok = ok and r["node_count"] >= F["covered_node"]
        if F["covered_node_ratio"] > 0.0: ok = ok and np.float32(r["node_count"]) / n_expected >= np.float32(F["covered_node_ratio"])
        if F["score"] > 0.0: ok = ok and np.float32(r["idf"]) >= np.float32(F["score"])
        if F["num_residue"] > 0: ok = ok and r["nres"] <= F["num_residue"]
        if F["plddt"] > 0.0: ok = ok and np.float32(r["plddt"]) >= np.float32(F["plddt"])
        return ok
    rows = [r for r in rows if before(r)]
    rows.sort(key=lambda r: -r["idf"])  # par_sort_by idf desc, stable (query_pdb.rs:404)
    if top_n is not None:
        rows = rows[:top_n]
    match_rows = []
    if not skip_match and rows:
        std = np.concatenate([s.resname_std() for s in db_structs]) if db_structs else np.zeros(0, np.uint8)
        owned = [k for k, r in enumerate(rows) if lo <= r["nid"] < lo + n_local]       # candidates this rank holds coordinates of
        cand = np.array([rows[k]["nid"] - lo for k in owned], np.uint32)
        ms = retrieve(ctx, db, std, cand, qm, qbatch, ca_distance, nbin_dist=nbin_dist, nbin_angle=nbin_angle, dist_cutoff=dist_cutoff,
                      partial_fit=partial_fit, hash_type=hash_type, multiple_bins=multiple_bins) if len(cand) else []
        for m in ms:
            m["cand"] = owned[m["cand"]]                                                 # -> position in `rows`
            t = db_structs[rows[m["cand"]]["nid"] - lo]
            m["labels"] = ["_" if x < 0 else f"{chr(int(t.chain[x]))}{int(t.serial[x])}" for x in m["processed"]]
            m["ca"] = np.array([t.ca_xyz[x] for x in m["processed"] if x >= 0], np.float32).reshape(-1, 3)   # matching C-alpha coordinates
            m["labels_h"] = ["_" if x < 0 else f"{chr(int(t.chain[x]))}{int(t.serial[x])}" for x in m["from_hash"]]
            m["ca_h"] = np.array([t.ca_xyz[x] for x in m["from_hash"] if x >= 0], np.float32).reshape(-1, 3)
        if shard is not None:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
                box = [None] * tdist.get_world_size()
                tdist.all_gather_object(box, ms)
                ms = sorted((m for part in box for m in part), key=lambda m: m["cand"])   # stable: components keep their order
        for m in ms:
            r = rows[m["cand"]]
            lab = lambda lst: m["labels"]
            r.setdefault("match_strs", []).append(",".join(lab(m["processed"])) + ":%.4f" % m["rmsd"])
            # --skip-ca-match: the per-match table shows the from-hash mapping, i.e. before the C-alpha distance rescue
            # (StructureResult::into_match_query_results, result.rs:54-69); the per-structure columns keep the processed one
            sk = "_from_hash" if skip_ca_match else ""
            mres, mlab, mca = (m["from_hash"], m["labels_h"], m["ca_h"]) if skip_ca_match else (m["processed"], m["labels"], m["ca"])
            mm = m["metrics" + sk]
            m2 = dict(tid=r["tid"], nid=r["nid"], node_count=sum(x >= 0 for x in mres), idf=m["idf"], rmsd=m["rmsd" + sk],
                      matching_residues=",".join(mlab), query_residues=res_chain_to_string(qres) if qres else query_string,
                      tm_score=float(mm[0]), gdt_ts=float(mm[1]), gdt_ha=float(mm[2]),
                      chamfer_distance=float(mm[3]), hausdorff_distance=float(mm[4]), db_key=r["db_key"],
                      u_matrix=m["rot" + sk], t_vector=m["tran" + sk], matching_coordinates=mca)
            r["matches"].append(m2)
            cnt = sum(x >= 0 for x in m["processed"])
            if cnt > r["max_matching_node_count"]:
                r["max_matching_node_count"], r["min_rmsd_with_max_match"] = cnt, m["rmsd"]
            elif cnt == r["max_matching_node_count"] and m["rmsd"] < r["min_rmsd_with_max_match"]:
                r["min_rmsd_with_max_match"] = m["rmsd"]
            match_rows.append(m2)
        def after(r):    # StructureFilter::filter_after_matching
            ok = True
            if F["max_node"] > 0: ok = ok and r["max_matching_node_count"] >= F["max_node"]
            if F["max_node_ratio"] > 0.0: ok = ok and np.float32(r["max_matching_node_count"]) / n_expected >= np.float32(F["max_node_ratio"])
            if F["rmsd"] > 0.0: ok = ok and np.float32(r["min_rmsd_with_max_match"]) <= np.float32(F["rmsd"])
            return ok
        rows = [r for r in rows if after(r)]
        kept = {r["nid"] for r in rows}

        def mfilter(m):  # MatchFilter::filter
            ok = m["nid"] in kept
            if F["connected_node"] > 0: ok = ok and m["node_count"] >= F["connected_node"]
            if F["connected_node_ratio"] > 0.0: ok = ok and np.float32(m["node_count"]) / n_expected >= np.float32(F["connected_node_ratio"])
            if F["score"] > 0.0: ok = ok and np.float32(m["idf"]) >= np.float32(F["score"])
            if F["rmsd"] > 0.0: ok = ok and np.float32(m["rmsd"]) <= np.float32(F["rmsd"])
            if F["tm_score"] > 0.0: ok = ok and np.float32(m["tm_score"]) >= np.float32(F["tm_score"])
            if F["gdt_ts"] > 0.0: ok = ok and np.float32(m["gdt_ts"]) >= np.float32(F["gdt_ts"])
            if F["gdt_ha"] > 0.0: ok = ok and np.float32(m["gdt_ha"]) >= np.float32(F["gdt_ha"])
            if F["chamfer"] > 0.0: ok = ok and np.float32(m["chamfer_distance"]) <= np.float32(F["chamfer"])
            if F["hausdorff"] > 0.0: ok = ok and np.float32(m["hausdorff_distance"]) <= np.float32(F["hausdorff"])
            return ok
        match_rows = [m for m in match_rows if mfilter(m)]
        sort_rows(match_rows, parse_sort_by(sort_by, False))
        mt = top_n if match_top_n == "same" else match_top_n     # --web prints at most 1000 matches whatever --top says (query_pdb.rs:489)
        if mt is not None:
            match_rows = match_rows[:mt]
    return rows, match_rows


def format_match_row(m) -> str:
    """default per-match columns (result.rs:331-339), floats {:.4}"""
    return "\t".join([m["tid"], str(m["node_count"]), "%.4f" % m["idf"], "%.4f" % m["rmsd"], m["matching_residues"], m["query_residues"]])


def format_structure_row(r, query_residues: str) -> str:
    """default per-structure columns (result.rs:301-314), floats {:.4}"""
    ms = ";".join(r.get("match_strs", [])) or "NA"
    return "\t".join([r["tid"], "%.4f" % r["idf"], str(r["total_match_count"]), str(r["node_count"]), str(r["edge_count"]),
                      str(r["max_matching_node_count"]), "%.4f" % r["min_rmsd_with_max_match"], str(r["nres"]), "%.4f" % r["plddt"], ms,
                      str(r["db_key"]), query_residues])


# --format-output column names of the per-match table (src/controller/result.rs:280-298); floats {:.4}
MATCH_COLUMNS = {
    "tid": lambda m: m["tid"], "nid": lambda m: str(m["nid"]), "db_key": lambda m: str(m["db_key"]),
    "node_count": lambda m: str(m["node_count"]), "idf": lambda m: "%.4f" % m["idf"], "rmsd": lambda m: "%.4f" % m["rmsd"],
    "matching_residues": lambda m: m["matching_residues"], "query_residues": lambda m: m["query_residues"],
    "tm_score": lambda m: "%.4f" % m["tm_score"], "gdt_ts": lambda m: "%.4f" % m["gdt_ts"], "gdt_ha": lambda m: "%.4f" % m["gdt_ha"],
    "chamfer_distance": lambda m: "%.4f" % m["chamfer_distance"], "hausdorff_distance": lambda m: "%.4f" % m["hausdorff_distance"],
    "u_matrix": lambda m: ",".join("%.4f" % x for x in np.asarray(m["u_matrix"]).reshape(-1)),
    "t_vector": lambda m: ",".join("%.4f" % x for x in np.asarray(m["t_vector"]).reshape(-1)),
    # target C-alpha coordinates of the matched residues, in query-residue order (the reference lists them in the order its
    # rescue pass pushed them, retrieve.rs:769-771 — the same set)
    "matching_coordinates": lambda m: ",".join("%.4f" % x for x in np.asarray(m["matching_coordinates"]).reshape(-1)),
}
# --superpose / --web column set (src/controller/result.rs:341-353)
MATCH_SUPERPOSE_COLUMNS = ["tid", "node_count", "idf", "rmsd", "matching_residues", "u_matrix", "t_vector", "matching_coordinates", "db_key",
                           "query_residues"]


def format_match_columns(m, columns) -> str:
    return "\t".join(MATCH_COLUMNS[c](m) for c in columns)
