"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Names follow the reference: `Folddisco` index builder (src/controller/mod.rs:49-71),
`FolddiscoIndex` (src/index/indextable.rs:8-18), `count_query` (src/controller/count_query.rs:82),
`get_geometric_hash_as_u32_from_structure` (src/controller/feature.rs:198).  Everything numeric
happens in libfdgpu.so on the GPU; this module only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import BatchDesc, CountRec, HashParams, f32p, hash_type_index, u8p, u32p, u64p


class FdgpuError(RuntimeError):
    pass


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


@dataclass
class PackedStructures:
    """CompactStructure batch flattened (fd_batch_desc). res_off[s]..res_off[s+1] = residues of structure s."""
    res_off: np.ndarray   # u64 [S+1]
    n_xyz: np.ndarray     # f32 [R,3]
    ca_xyz: np.ndarray
    cb_xyz: np.ndarray
    aa: np.ndarray        # u8 [R]
    cb_valid: np.ndarray | None = None

    def __post_init__(self):
        self.res_off = np.ascontiguousarray(self.res_off, dtype=np.uint64)
        self.n_xyz = np.ascontiguousarray(self.n_xyz, dtype=np.float32).reshape(-1, 3)
        self.ca_xyz = np.ascontiguousarray(self.ca_xyz, dtype=np.float32).reshape(-1, 3)
        self.cb_xyz = np.ascontiguousarray(self.cb_xyz, dtype=np.float32).reshape(-1, 3)
        self.aa = np.ascontiguousarray(self.aa, dtype=np.uint8)
        if self.cb_valid is not None:
            self.cb_valid = np.ascontiguousarray(self.cb_valid, dtype=np.uint8)
        R = int(self.res_off[-1])
        if not (len(self.n_xyz) == len(self.ca_xyz) == len(self.cb_xyz) == len(self.aa) == R):
            raise ValueError("PackedStructures: array lengths do not match res_off[-1]")

    @property
    def n_struct(self) -> int:
        return len(self.res_off) - 1

    @property
    def nres(self) -> np.ndarray:
        return np.diff(self.res_off.astype(np.int64)).astype(np.uint64)

    @staticmethod
    def concat(items) -> "PackedStructures":
        items = list(items)
        lens = [len(it["aa"]) for it in items]
        off = np.zeros(len(items) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        cat = lambda k, w: (np.concatenate([np.asarray(it[k]).reshape(-1, w) for it in items]) if items else np.zeros((0, w)))
        cbv = None
        if any(it.get("cb_ok") is not None for it in items):
            cbv = np.concatenate([np.asarray(it.get("cb_ok") if it.get("cb_ok") is not None else np.ones(len(it["aa"]), np.uint8)) for it in items])
        return PackedStructures(off, cat("n_xyz", 3), cat("ca_xyz", 3), cat("cb_xyz", 3),
                                np.concatenate([np.asarray(it["aa"], dtype=np.uint8) for it in items]) if items else np.zeros(0, np.uint8), cbv)


class Context:
    """fdgpu_ctx: one per GPU."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.fdgpu_create(device, C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.fdgpu_last_error(h).decode() if h else "fdgpu_create failed"
            if h:
                self.L.fdgpu_destroy(h)
            self.h = None
            raise FdgpuError(f"fdgpu_create({device}) failed ({rc}): {msg}")
        if stream is not None:
            self.check(self.L.fdgpu_set_stream(self.h, C.c_void_p(stream)))

    def check(self, rc: int):
        if rc != 0:
            raise FdgpuError(f"libfdgpu error {rc}: {self.L.fdgpu_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.fdgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def spec_fallbacks(self) -> int:
        """pairs re-evaluated by the exact routine since the last call (speculative torsion path of the index build)"""
        out = np.zeros(1, np.uint64)
        self.check(self.L.fdgpu_spec_fallbacks(self.h, _ptr(out, u64p)))
        return int(out[0])

    def synchronize(self):
        self.check(self.L.fdgpu_synchronize(self.h))

    def release_workspaces(self):
        """fdgpu_release_workspaces: sort buffers, scratch and cached index blocks back to the device (indices stay valid)"""
        self.check(self.L.fdgpu_release_workspaces(self.h))

    def enable_timing(self, on: bool = True):
        self.check(self.L.fdgpu_enable_timing(self.h, int(on)))

    def last_timings(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        by = (C.c_uint64 * cap)()
        n = self.L.fdgpu_last_timings(self.h, names, ms, by, cap)
        return [(names[i].decode(), float(ms[i]), int(by[i])) for i in range(n)]

    # ---- batches
    def upload(self, ps: PackedStructures) -> "Batch":
        d = BatchDesc(ps.n_struct, _ptr(ps.res_off, u64p), _ptr(ps.n_xyz, f32p), _ptr(ps.ca_xyz, f32p), _ptr(ps.cb_xyz, f32p),
                      _ptr(ps.aa, u8p), _ptr(ps.cb_valid, u8p) if ps.cb_valid is not None else None)
        h = C.c_void_p()
        self.check(self.L.fdgpu_batch_upload(self.h, C.byref(d), C.byref(h)))
        return Batch(self, h, ps.n_struct)

    def wrap_device(self, n_struct: int, total_residues: int, res_off_ptr: int, n_ptr: int, ca_ptr: int, cb_ptr: int, aa_ptr: int,
                    cb_valid_ptr: int | None = None, keepalive=None) -> "Batch":
        d = BatchDesc(n_struct, C.cast(res_off_ptr, u64p), C.cast(n_ptr, f32p), C.cast(ca_ptr, f32p), C.cast(cb_ptr, f32p),
                      C.cast(aa_ptr, u8p), C.cast(cb_valid_ptr, u8p) if cb_valid_ptr else None)
        h = C.c_void_p()
        self.check(self.L.fdgpu_batch_wrap_device(self.h, C.byref(d), total_residues, C.byref(h)))
        b = Batch(self, h, n_struct)
        b._keepalive = keepalive
        return b


class Batch:
    def __init__(self, ctx: Context, h, n_struct: int):
        self.ctx, self.h, self.n_struct = ctx, h, n_struct

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.L.fdgpu_batch_destroy(self.h)
        except Exception:
            pass


def _params(nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, hash_type=3, multiple_bins=None) -> HashParams:
    return HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type_index(hash_type), multiple_bins)


def get_geometric_hash_as_u32(ctx: Context, batch: Batch, nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, sort_dedup=True, hash_type=3):
    """S1 for every structure of the batch -> (hashes u32[...], off u64[S+1])."""
    p = _params(nbin_dist, nbin_angle, dist_cutoff, hash_type)
    hp, op = u32p(), u64p()
    ctx.check(ctx.L.fdgpu_hash_batch(ctx.h, batch.h, C.byref(p), int(sort_dedup), C.byref(hp), C.byref(op)))
    off = np.ctypeslib.as_array(op, shape=(batch.n_struct + 1,)).copy()
    n = int(off[-1])
    h = np.ctypeslib.as_array(hp, shape=(max(n, 1),))[:n].copy()
    ctx.L.fdgpu_free(hp)
    ctx.L.fdgpu_free(op)
    return h, off


class FolddiscoIndex:
    """Inverted index resident in HBM (fdgpu_index)."""

    def __init__(self, ctx: Context, h, n_structures: int, first_id: int = 0):
        self.ctx, self.h, self.n_structures, self.first_id = ctx, h, n_structures, first_id

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.L.fdgpu_index_destroy(self.h)
        except Exception:
            pass

    @staticmethod
    def build(ctx: Context, batch: Batch, nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, first_id=0, hash_type=3, multiple_bins=None) -> "FolddiscoIndex":
        p = _params(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
        h = C.c_void_p()
        ctx.check(ctx.L.fdgpu_index_build(ctx.h, batch.h, C.byref(p), first_id, C.byref(h)))
        return FolddiscoIndex(ctx, h, batch.n_struct, first_id)

    @staticmethod
    def load(ctx: Context, hashes: np.ndarray, offsets: np.ndarray, value: np.ndarray, n_structures: int, first_id: int = 0) -> "FolddiscoIndex":
        hashes = np.ascontiguousarray(hashes, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        value = np.ascontiguousarray(value, dtype=np.uint8)
        h = C.c_void_p()
        ctx.check(ctx.L.fdgpu_index_load(ctx.h, _ptr(hashes, u32p), _ptr(offsets, u64p), len(hashes), _ptr(value, u8p), len(value),
                                         n_structures, C.byref(h)))
        if first_id:
            ctx.check(ctx.L.fdgpu_index_set_first_id(h, first_id))
        return FolddiscoIndex(ctx, h, n_structures, first_id)

    @property
    def num_hashes(self) -> int:
        return int(self.ctx.L.fdgpu_index_num_hashes(self.h))

    @property
    def value_len(self) -> int:
        return int(self.ctx.L.fdgpu_index_value_len(self.h))

    @property
    def num_postings(self) -> int:
        return int(self.ctx.L.fdgpu_index_num_postings(self.h))

    def export(self):
        """-> (value u8[], hashes u32[H], offsets u64[H+1]) in the on-disk layout."""
        vp, hp, op = u8p(), u32p(), u64p()
        vl, H = C.c_uint64(), C.c_uint64()
        L = self.ctx.L
        self.ctx.check(L.fdgpu_index_export(self.ctx.h, self.h, C.byref(vp), C.byref(vl), C.byref(hp), C.byref(op), C.byref(H)))
        v = np.ctypeslib.as_array(vp, shape=(max(vl.value, 1),))[: vl.value].copy()
        h = np.ctypeslib.as_array(hp, shape=(max(H.value, 1),))[: H.value].copy()
        o = np.ctypeslib.as_array(op, shape=(H.value + 1,)).copy()
        for p in (vp, hp, op):
            L.fdgpu_free(p)
        return v, h, o

    def save(self, prefix: str):
        self.ctx.check(self.ctx.L.fdgpu_index_save(self.ctx.h, self.h, prefix.encode()))

    # ---- one on-disk index from N ranks (fd_shard_index.hip; SURVEY §8e row 2, Option A)
    def range_bounds(self, n_ranges: int) -> np.ndarray:
        """n_ranges - 1 ascending hash values cutting this index into ranges of about equal posting bytes"""
        b = np.zeros(max(n_ranges - 1, 1), np.uint32)
        self.ctx.check(self.ctx.L.fdgpu_index_range_bounds(self.ctx.h, self.h, n_ranges, _ptr(b, u32p)))
        return b[: n_ranges - 1]

    def slice(self, hash_lo: int, hash_hi: int) -> "FolddiscoIndex":
        """the lists with hash_lo <= hash < hash_hi as a resident index of their own (same id range)"""
        h = C.c_void_p()
        self.ctx.check(self.ctx.L.fdgpu_index_slice(self.ctx.h, self.h, int(hash_lo), int(hash_hi), C.byref(h)))
        return FolddiscoIndex(self.ctx, h, self.n_structures, self.first_id)

    def save_part(self, prefix: str, hashes_before: int, value_before: int, total_hashes: int, total_value: int, write_header: bool, is_last: bool):
        """this index = one hash range of the database's single index: its regions of PREFIX / PREFIX.offset"""
        self.ctx.check(self.ctx.L.fdgpu_index_save_part(self.ctx.h, self.h, prefix.encode(), int(hashes_before), int(value_before), int(total_hashes), int(total_value),
                                                        int(bool(write_header)), int(bool(is_last))))

    def get_entries(self, q_hash) -> list:
        """decoded posting lists (ascending structure ids) of the given hashes (get_entries, index/indextable.rs:83-86)"""
        q = np.ascontiguousarray(q_hash, dtype=np.uint32)
        ip, op = u32p(), u64p()
        self.ctx.check(self.ctx.L.fdgpu_get_entries(self.ctx.h, self.h, _ptr(q, u32p), len(q), C.byref(ip), C.byref(op)))
        off = np.ctypeslib.as_array(op, shape=(len(q) + 1,)).copy()
        n = int(off[-1])
        ids = np.ctypeslib.as_array(ip, shape=(max(n, 1),))[:n].copy()
        self.ctx.L.fdgpu_free(ip)
        self.ctx.L.fdgpu_free(op)
        return [ids[int(off[k]): int(off[k + 1])] for k in range(len(q))]

    def posting_lengths(self, q_hash: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(q_hash, dtype=np.uint32)
        out = np.zeros(len(q), dtype=np.uint64)
        self.ctx.check(self.ctx.L.fdgpu_posting_lengths(self.ctx.h, self.h, _ptr(q, u32p), len(q), _ptr(out, u64p)))
        return out

    def set_penalty(self, penalty):
        """keep the length penalty of the index's structures on the device (fdgpu_index_set_penalty): count queries may then
        pass penalty=None instead of uploading n_structures floats per call; None drops it"""
        pen = None if penalty is None else np.ascontiguousarray(penalty, dtype=np.float32)
        assert pen is None or len(pen) == self.n_structures
        self.ctx.check(self.ctx.L.fdgpu_index_set_penalty(self.ctx.h, self.h, None if pen is None else _ptr(pen, f32p)))

    def posting_bytes(self, q_hash: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(q_hash, dtype=np.uint32)
        out = np.zeros(len(q), dtype=np.uint64)
        self.ctx.check(self.ctx.L.fdgpu_posting_bytes(self.ctx.h, self.h, _ptr(q, u32p), len(q), _ptr(out, u64p)))
        return out

    def export_view(self):
        """export() without the extra host copy: numpy views of the malloc'd buffers (freed with the arrays)"""
        vp, hp, op = u8p(), u32p(), u64p()
        vl, H = C.c_uint64(), C.c_uint64()
        L = self.ctx.L
        self.ctx.check(L.fdgpu_index_export(self.ctx.h, self.h, C.byref(vp), C.byref(vl), C.byref(hp), C.byref(op), C.byref(H)))

        class _Owner:
            def __init__(self, ptrs):
                self.ptrs = ptrs

            def __del__(self):
                for p in self.ptrs:
                    L.fdgpu_free(p)
        own = _Owner((vp, hp, op))

        def view(ptr, n, dt):
            a = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n]
            a = a.view(dt) if a.dtype != dt else a
            return _Keep(a, own)
        return view(vp, vl.value, np.uint8), view(hp, H.value, np.uint32), view(op, H.value + 1, np.uint64)


class _Keep(np.ndarray):
    """ndarray view that keeps the owner of its buffer alive"""
    def __new__(cls, arr, owner):
        obj = np.asarray(arr).view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class FolddiscoIndexSet:
    """Several resident sub-indices over consecutive, disjoint structure-id ranges, queried as one index: a shard of more than
    2^32 residue pairs is built as a sequence of fdgpu_index_build calls (one per chunk of structures) and never merged on the
    device.  Posting lengths add up over the parts and every part scores its own structures, so the concatenated records
    equal those of the merged index."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.ctx = self.parts[0].ctx if self.parts else None
        self.first_id = self.parts[0].first_id if self.parts else 0
        self.n_structures = sum(p.n_structures for p in self.parts)
        for a, b in zip(self.parts, self.parts[1:]):
            assert b.first_id == a.first_id + a.n_structures, "parts must cover consecutive id ranges"

    @staticmethod
    def build(ctx: Context, batches, first_id=0, **kw) -> "FolddiscoIndexSet":
        parts = []
        for b in batches:
            parts.append(FolddiscoIndex.build(ctx, b, first_id=first_id, **kw))
            first_id += b.n_struct
        return FolddiscoIndexSet(parts)

    def posting_lengths(self, q_hash) -> np.ndarray:
        q = np.ascontiguousarray(q_hash, dtype=np.uint32)
        tot = np.zeros(len(q), np.uint64)
        for p in self.parts:
            tot += p.posting_lengths(q)
        return tot

    def get_entries(self, q_hash) -> list:
        lists = [p.get_entries(q_hash) for p in self.parts]
        return [np.concatenate([l[k] for l in lists]) for k in range(len(np.atleast_1d(q_hash)))]

    @property
    def num_postings(self) -> int:
        return sum(p.num_postings for p in self.parts)

    def merge(self) -> "FolddiscoIndex":
        """ONE resident index over all the parts (fdgpu_index_merge: per-hash concatenation on the device)"""
        ctx = self.ctx
        arr = (C.c_void_p * len(self.parts))(*[p.h for p in self.parts])
        h = C.c_void_p()
        ctx.check(ctx.L.fdgpu_index_merge(ctx.h, arr, len(self.parts), C.byref(h)))
        return FolddiscoIndex(ctx, h, self.n_structures, self.first_id)

    def export_merged(self):
        """-> (value, hashes, offsets) of the single merged index (host-side per-hash concatenation)"""
        from . import indexio
        return indexio.merge_subindices([p.export() for p in self.parts])


def count_query_set(ctx: Context, iset: FolddiscoIndexSet, q_hash, q_node, q_edge_j, penalty: np.ndarray, total_structures: int | None = None,
                    freq_filter: float | None = None, lengths: np.ndarray | None = None) -> np.ndarray:
    """count_query over a FolddiscoIndexSet -> REC_DTYPE array in ascending nid.  penalty covers the set's structures."""
    S = iset.n_structures if total_structures is None else total_structures
    lens = iset.posting_lengths(q_hash) if lengths is None else np.asarray(lengths, np.uint64)
    out, base = [], 0
    for p in iset.parts:
        out.append(count_query(ctx, p, q_hash, q_node, q_edge_j, penalty[base:base + p.n_structures], total_structures=S,
                               freq_filter=freq_filter, as_array=True, lengths=lens))
        base += p.n_structures
    return np.concatenate(out) if out else np.zeros(0, REC_DTYPE)


def length_penalty(nres: np.ndarray, lp: float = 0.5) -> np.ndarray:
    """(nres as f32).powf(-lp) per structure (count_query.rs:200), evaluated with the C library's powf."""
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    nres = np.asarray(nres)
    uniq, inv = np.unique(nres, return_inverse=True)          # one libm call per distinct length
    lut = np.array([libm.powf(float(np.float32(n)), -lp) for n in uniq], dtype=np.float32)
    return lut[inv] if len(nres) else np.zeros(0, np.float32)


def idf_of_lengths(lengths: np.ndarray, total_structures: int) -> np.ndarray:
    """log2(S / len) in f32 with the C library's log2f (query.rs:26, count_query.rs:130)."""
    libm = C.CDLL("libm.so.6")
    libm.log2f.restype = C.c_float
    libm.log2f.argtypes = [C.c_float]
    S = np.float32(total_structures)
    lengths = np.asarray(lengths)
    uniq, inv = np.unique(lengths, return_inverse=True)       # one libm call per distinct length
    lut = np.array([libm.log2f(float(S / np.float32(n))) if n > 0 else np.inf for n in uniq], dtype=np.float32)
    return lut[inv] if len(lengths) else np.zeros(0, np.float32)


REC_DTYPE = np.dtype([("nid", np.uint32), ("total_match_count", np.uint32), ("node_count", np.uint32),
                      ("edge_count", np.uint32), ("idf", np.float32)])


def count_query(ctx: Context, index: FolddiscoIndex, q_hash, q_node, q_edge_j, penalty: np.ndarray, total_structures: int | None = None,
                freq_filter: float | None = None, as_array: bool = False, lengths: np.ndarray | None = None):
    """count_query (src/controller/count_query.rs:82-220) -> fd_count_rec rows in ascending nid
    (list of dicts, or a numpy structured array with as_array=True)."""
    q_hash = np.ascontiguousarray(q_hash, dtype=np.uint32)
    q_node = np.ascontiguousarray(q_node, dtype=np.uint32)
    q_edge_j = np.ascontiguousarray(q_edge_j, dtype=np.uint32)
    S = index.n_structures if total_structures is None else total_structures
    # lengths: posting lengths over the WHOLE database when `index` is one shard of it (dist.global_posting_lengths)
    lens = index.posting_lengths(q_hash) if lengths is None else np.asarray(lengths, np.uint64)
    keep = np.ones(len(q_hash), dtype=bool)
    if freq_filter is not None:
        keep &= ~((lens.astype(np.float32) / np.float32(S)) > np.float32(freq_filter))
    keep &= lens > 0
    idf = idf_of_lengths(lens, S)
    qh, qn, qe, qi = (np.ascontiguousarray(a[keep]) for a in (q_hash, q_node, q_edge_j, idf.astype(np.float32)))
    pen = np.ascontiguousarray(penalty, dtype=np.float32)
    out = C.POINTER(CountRec)()
    n = C.c_uint64()
    ctx.check(ctx.L.fdgpu_count_query(ctx.h, index.h, _ptr(qh, u32p), _ptr(qn, u32p), _ptr(qe, u32p), _ptr(qi, f32p), len(qh),
                                      _ptr(pen, f32p), C.byref(out), C.byref(n)))
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 20,))[: n.value * 20].copy().view(REC_DTYPE)
    ctx.L.fdgpu_free(out)
    if as_array:
        return arr
    return [dict(nid=int(r["nid"]), total_match_count=int(r["total_match_count"]), node_count=int(r["node_count"]),
                 edge_count=int(r["edge_count"]), idf=float(r["idf"])) for r in arr]


def count_query_maps(ctx: Context, index: FolddiscoIndex, qms, penalty=None, total_structures: int | None = None, top_n: int = 0, flat: bool = False):
    """count_query_batch for the QueryMapResults of make_query_maps, handed to the library as they are (fdgpu_count_query_maps_top:
    posting lengths, idf = log2f(S / len) per hash and the scoring in one call; penalty=None uses FolddiscoIndex.set_penalty's
    resident copy).  -> list of REC_DTYPE arrays like count_query_batch; flat=True: (all records, offsets[T + 1]) — the library's two
    output arrays as they are (per-query numpy slicing of a 128-query batch costs more host time than the library call's own host side)."""
    S = index.n_structures if total_structures is None else total_structures
    T = len(qms)
    handles = (C.c_void_p * max(T, 1))(*[C.cast(q.handle, C.c_void_p) for q in qms])
    pen = None if penalty is None else np.ascontiguousarray(penalty, dtype=np.float32)
    out = C.POINTER(CountRec)()
    ooff = u64p()
    ctx.check(ctx.L.fdgpu_count_query_maps_top(ctx.h, index.h, T, handles, None if pen is None else _ptr(pen, f32p), float(S), int(top_n),
                                               C.byref(out), C.byref(ooff)))
    off = np.frombuffer((C.c_uint64 * (T + 1)).from_address(C.addressof(ooff.contents)), dtype=np.uint64).copy()
    n = int(off[-1])
    arr = _lib.owned_view(ctx.L, out, n * 20, REC_DTYPE)         # the library's (pooled, page-locked) block itself, handed back when the views die
    ctx.L.fdgpu_free(ooff)
    if flat:
        return arr, off
    offs = off.tolist()
    return [arr[offs[t]: offs[t + 1]] for t in range(T)]


def count_query_batch(ctx: Context, index: FolddiscoIndex, queries, penalty: np.ndarray, total_structures: int | None = None,
                      top_n: int = 0, lengths_fn=None):
    """queries: list of (q_hash, q_node, q_edge_j) arrays.  One posting-length launch + one scoring pass for the whole
    batch.  Returns a list of REC_DTYPE arrays, one per query: every touched structure in ascending nid, or with top_n > 0
    the top_n records ranked like the candidate selection (idf descending, ties by ascending nid — selected and sorted on the
    device, what dist.rank_hits(recs, top_n) gives on the full list)."""
    S = index.n_structures if total_structures is None else total_structures
    qh = np.ascontiguousarray(np.concatenate([np.asarray(q[0], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
    qn = np.ascontiguousarray(np.concatenate([np.asarray(q[1], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
    qe = np.ascontiguousarray(np.concatenate([np.asarray(q[2], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
    lens_all = index.posting_lengths(qh)
    if lengths_fn is not None:          # sharded index: local lengths -> lengths over the whole database (one all-reduce)
        lens_all = lengths_fn(lens_all)
    idf = idf_of_lengths(lens_all, S).astype(np.float32)
    keep = lens_all > 0
    qid = np.repeat(np.arange(len(queries)), [len(q[0]) for q in queries])
    kept_per_q = np.bincount(qid[keep], minlength=len(queries))
    q_off = np.concatenate([[0], np.cumsum(kept_per_q)]).astype(np.uint64)
    qh, qn, qe, idf = (np.ascontiguousarray(a[keep]) for a in (qh, qn, qe, idf))
    pen = np.ascontiguousarray(penalty, dtype=np.float32)
    out = C.POINTER(CountRec)()
    ooff = u64p()
    ctx.check(ctx.L.fdgpu_count_query_batch_top(ctx.h, index.h, len(queries), _ptr(q_off, u64p), _ptr(qh, u32p), _ptr(qn, u32p), _ptr(qe, u32p),
                                                _ptr(idf, f32p), _ptr(pen, f32p), int(top_n), C.byref(out), C.byref(ooff)))
    off = np.ctypeslib.as_array(ooff, shape=(len(queries) + 1,)).copy()
    n = int(off[-1])
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n, 1) * 20,))[: n * 20].copy().view(REC_DTYPE)
    ctx.L.fdgpu_free(out)
    ctx.L.fdgpu_free(ooff)
    return [arr[int(off[t]): int(off[t + 1])] for t in range(len(queries))]
