"""S4 wrappers: candidate pair scan (retrieve_with_prefilter, src/controller/retrieve.rs:52-156) and
batched Kabsch (src/structure/kabsch.rs:157-554) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CandRec, HashParams, MatchQuery, PairRec, f32p, u8p, u32p, u64p
from .api import Batch, Context

PREFILTER_AA_SKIPPING_SIZE = 200  # retrieve.rs:24


def match_pairs(ctx: Context, db: Batch, resname_std: np.ndarray | None, cand: np.ndarray, qmap: dict,
                ca_distance_cutoff: float = 1.0, nbin_dist: int = 0, nbin_angle: int = 0, dist_cutoff: float = 20.0, hash_type: int = 3, multiple_bins=None):
    """qmap: dict with 'hash', 'aad_aa1', 'aad_aa2', 'aad_dist', 'aad_qi' (make_query_map outputs).
    Returns (found u32[n,4] = cand,i,j,hash ; cands u32[m,4] = cand,qi,i,j) in the reference's scan order."""
    hashes = np.unique(np.asarray(qmap["hash"], dtype=np.uint32))
    a1 = np.ascontiguousarray(qmap["aad_aa1"], dtype=np.uint8)
    a2 = np.ascontiguousarray(qmap["aad_aa2"], dtype=np.uint8)
    ad = np.ascontiguousarray(qmap["aad_dist"], dtype=np.float32)
    aq = np.ascontiguousarray(qmap["aad_qi"], dtype=np.uint32)
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    q = MatchQuery(hashes.ctypes.data_as(u32p), len(hashes), a1.ctypes.data_as(u8p), a2.ctypes.data_as(u8p), ad.ctypes.data_as(f32p),
                   aq.ctypes.data_as(u32p), len(ad), ca_distance_cutoff, int(len(hashes) <= PREFILTER_AA_SKIPPING_SIZE))
    p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
    std = None if resname_std is None else np.ascontiguousarray(resname_std, dtype=np.uint8)
    fp, cp = C.POINTER(PairRec)(), C.POINTER(CandRec)()
    nf, nc = C.c_uint64(), C.c_uint64()
    ctx.check(ctx.L.fdgpu_match_pairs(ctx.h, db.h, None if std is None else std.ctypes.data_as(u8p), cand.ctypes.data_as(u32p), len(cand),
                                      C.byref(q), C.byref(p), C.byref(fp), C.byref(nf), C.byref(cp), C.byref(nc)))
    found = np.ctypeslib.as_array(C.cast(fp, u32p), shape=(max(nf.value, 1) * 4,))[: nf.value * 4].reshape(-1, 4).copy()
    cands = np.ctypeslib.as_array(C.cast(cp, u32p), shape=(max(nc.value, 1) * 4,))[: nc.value * 4].reshape(-1, 4).copy()
    ctx.L.fdgpu_free(fp)
    ctx.L.fdgpu_free(cp)
    return found, cands


def kabsch_batch(ctx: Context, x: np.ndarray, y: np.ndarray, off: np.ndarray):
    """problem k: moving points x[off[k]:off[k+1]] (target) onto fixed y[...] (query). -> rmsd[n], rot[n,3,3], tran[n,3]"""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    rmsd = np.zeros(n, np.float32)
    rot = np.zeros((n, 3, 3), np.float32)
    tran = np.zeros((n, 3), np.float32)
    ctx.check(ctx.L.fdgpu_kabsch_batch(ctx.h, x.ctypes.data_as(f32p), y.ctypes.data_as(f32p), off.ctypes.data_as(u64p), n,
                                       rmsd.ctypes.data_as(f32p), rot.ctypes.data_as(f32p), tran.ctypes.data_as(f32p)))
    return rmsd, rot, tran


def metrics_batch(ctx: Context, ref: np.ndarray, mov: np.ndarray, off: np.ndarray, rot: np.ndarray, tran: np.ndarray) -> np.ndarray:
    """similarity metrics of n superpositions on the device (fdgpu_metrics_batch): problem k compares the fixed points
    ref[off[k]:off[k+1]] with rot[k] @ mov[...] + tran[k]. -> [n, 5] = tm_score, gdt_ts, gdt_ha, chamfer, hausdorff"""
    ref = np.ascontiguousarray(ref, dtype=np.float32).reshape(-1, 3)
    mov = np.ascontiguousarray(mov, dtype=np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    rot = np.ascontiguousarray(rot, dtype=np.float32).reshape(-1, 9)
    tran = np.ascontiguousarray(tran, dtype=np.float32).reshape(-1, 3)
    n = len(off) - 1
    out = np.zeros((n, 5), np.float32)
    ctx.check(ctx.L.fdgpu_metrics_batch(ctx.h, ref.ctypes.data_as(f32p), mov.ctypes.data_as(f32p), off.ctypes.data_as(u64p), n,
                                        rot.ctypes.data_as(f32p), tran.ctypes.data_as(f32p), out.ctypes.data_as(f32p)))
    return out


def lms_qcp_batch(ctx: Context, x: np.ndarray, y: np.ndarray, off: np.ndarray):
    """--partial-fit superposition (src/structure/lms_qcp.rs, default parameters) of x[off[k]:off[k+1]] onto y[...], >= 3 pairs each.
    -> rms over the core[n], rot[n,3,3], tran[n,3], list of core index arrays (joining order)"""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    rmsd = np.zeros(n, np.float32)
    rot = np.zeros((n, 3, 3), np.float32)
    tran = np.zeros((n, 3), np.float32)
    clen = np.zeros(max(n, 1), np.uint32)
    core = np.zeros(max(len(x), 1), np.uint32)
    ctx.check(ctx.L.fdgpu_lms_qcp_batch(ctx.h, x.ctypes.data_as(f32p), y.ctypes.data_as(f32p), off.ctypes.data_as(u64p), n,
                                        rmsd.ctypes.data_as(f32p), rot.ctypes.data_as(f32p), tran.ctypes.data_as(f32p),
                                        clen.ctypes.data_as(u32p), core.ctypes.data_as(u32p)))
    return rmsd, rot, tran, [core[int(off[k]): int(off[k]) + int(clen[k])].copy() for k in range(n)]
