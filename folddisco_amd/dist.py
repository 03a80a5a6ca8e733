"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

Index build shards by structure: rank r owns the contiguous id range [r*S/N, (r+1)*S/N) and builds a complete
sub-index for it with no communication (ids ascend inside every posting list because the range is contiguous).
A query is scored by every rank against its own shard — all per-structure counters are complete locally — and
the only exchange is an all-gather of the candidate records (20 B each), after which every rank holds the global
ranking.  The same code runs over gloo on CPUs (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .api import REC_DTYPE  # fd_count_rec layout (20 bytes)


def shard_range(rank: int, world: int, n_structures: int):
    """contiguous, balanced id ranges; the union over ranks is [0, n_structures)"""
    base, rem = divmod(n_structures, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_hits(recs: np.ndarray, top_n: int | None = None) -> np.ndarray:
    """idf descending, ties by ascending nid (= the reference's stable sort over nid-ordered input,
    src/cli/workflows/query_pdb.rs:404-411)"""
    if len(recs) == 0:
        return recs
    # one u64 key per record: order-preserving image of the f32 idf (inverted for descending) above the nid
    b = (recs["idf"] + np.float32(0.0)).view(np.uint32)
    ordered = np.where(b & np.uint32(0x80000000), ~b, b | np.uint32(0x80000000))
    key = ((~ordered).astype(np.uint64) << np.uint64(32)) | recs["nid"].astype(np.uint64)
    if top_n is not None and top_n < len(recs):
        part = np.argpartition(key, top_n)[:top_n]
        order = part[np.argsort(key[part], kind="stable")]
    else:
        order = np.argsort(key, kind="stable")
    return recs[order]


def records_from_rows(rows) -> np.ndarray:
    a = np.zeros(len(rows), dtype=REC_DTYPE)
    for k, r in enumerate(rows):
        a[k] = (r["nid"], r["total_match_count"], r["node_count"], r["edge_count"], r["idf"])
    return a


def allgather_hits(local: np.ndarray, device: torch.device, top_n: int | None = None) -> np.ndarray:
    """all-gather of per-rank candidate records (variable length): sizes first, then one padded all_gather of the
    raw 20-byte records.  Returns the global ranking (identical on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rank_hits(local, top_n)
    world = dist.get_world_size()
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")     # gloo (CPU tests, single-GPU smoke runs): host tensors
    if top_n is not None:
        local = rank_hits(local, top_n)  # a rank never contributes more than top_n rows
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = np.zeros(mx, dtype=REC_DTYPE)
    buf[: len(local)] = local
    t = torch.from_numpy(buf.view(np.uint8).copy()).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = [o.cpu().numpy().view(REC_DTYPE)[:s] for o, s in zip(out, sizes)]
    return rank_hits(np.concatenate(parts), top_n)


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _dev(device):
    return device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")


def allreduce_sum(x: int, device: torch.device | None = None) -> int:
    if not _active():
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def allgather_array(local: np.ndarray, device: torch.device | None = None) -> np.ndarray:
    """all-gather of a variable-length array of fixed-size records (any dtype, e.g. query.MATCH_DTYPE): sizes first, then one padded
    all_gather of the raw bytes.  Returns the concatenation in rank order (identical on every rank)."""
    if not _active():
        return local
    world, dv = dist.get_world_size(), _dev(device)
    n = torch.tensor([len(local)], dtype=torch.int64, device=dv)
    sizes = torch.zeros(world, dtype=torch.int64, device=dv)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.cpu().tolist()
    isz = local.dtype.itemsize
    mx = max(max(sizes), 1)
    buf = np.zeros(mx * isz, np.uint8)
    buf[: len(local) * isz] = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    out = torch.empty(world * mx * isz, dtype=torch.uint8, device=dv)
    dist.all_gather_into_tensor(out, torch.from_numpy(buf).to(dv))
    o = out.cpu().numpy().reshape(world, mx * isz)
    return np.concatenate([o[r, : sizes[r] * isz].view(local.dtype) for r in range(world)])


def allgather_hits_many(locals_: list, device: torch.device | None = None, top_n: int | None = None, ranked: bool = False) -> list:
    """allgather_hits for a batch of queries with TWO collectives in total (all the sizes, then one padded payload): locals_[t] =
    this rank's candidate records of query t.  Returns the global ranking of every query (identical on every rank).
    ranked=True: the local lists are already ranked and cut to top_n (count_query_batch with top_n > 0 ranks on the device)."""
    if top_n is not None and not ranked:
        locals_ = [rank_hits(r, top_n) for r in locals_]
    if not _active():
        return [r if top_n is not None else rank_hits(r, None) for r in locals_]
    world, dv, T = dist.get_world_size(), _dev(device), len(locals_)
    n = torch.tensor([len(r) for r in locals_], dtype=torch.int64, device=dv)
    sizes = torch.zeros(world * T, dtype=torch.int64, device=dv)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.cpu().numpy().reshape(world, T)
    tot = sizes.sum(axis=1)
    mx = max(int(tot.max()), 1)
    isz = REC_DTYPE.itemsize
    buf = np.zeros(mx * isz, np.uint8)
    if T and int(tot[dist.get_rank()]):
        cat = np.concatenate(locals_)
        buf[: len(cat) * isz] = cat.view(np.uint8).reshape(-1)
    out = torch.empty(world * mx * isz, dtype=torch.uint8, device=dv)
    dist.all_gather_into_tensor(out, torch.from_numpy(buf).to(dv))
    o = out.cpu().numpy().reshape(world, mx * isz)
    starts = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(sizes, axis=1)], axis=1)
    res = []
    for t in range(T):
        parts = [o[r, starts[r, t] * isz: starts[r, t + 1] * isz].view(REC_DTYPE) for r in range(world)]
        res.append(rank_hits(np.concatenate(parts), top_n))
    return res


def reduce_lengths(lens: np.ndarray, device: torch.device | None = None) -> np.ndarray:
    """posting lengths of one shard -> posting lengths over all shards (all-reduce SUM; identity without a process group).
    idf = log2(S / len) must see the whole database, or the sharded hit list differs from the single-index one."""
    lens = np.asarray(lens, np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return lens
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.from_numpy(lens.astype(np.int64)).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def global_posting_lengths(index, hashes, device: torch.device | None = None) -> np.ndarray:
    return reduce_lengths(index.posting_lengths(np.ascontiguousarray(hashes, np.uint32)), device)


def sharded_query(ctx, shard_index, lo: int, db_shard, qbatch, q_indices, subs, penalty_shard, total_structures: int, device=None,
                  top_n: int | None = None, resname_std_shard=None, ca_distance=1.0, dist_thr=(0.5,), angle_thr=(5.0,), retrieve_matches=True):
    """One motif query against an index sharded by structure id: this rank holds the postings and coordinates of
    structures [lo, lo + shard_index.n_structures).  Scoring is local with idf from GLOBAL posting lengths (one all-reduce),
    the candidate records are all-gathered and ranked (idf descending, nid ascending, top_n), every candidate is matched on
    the rank that owns it and the match lists are all-gathered.  Returns (records, matches) — identical on every rank and to
    the single-index query.  matches: dicts of query.retrieve() with the global structure id under "nid"."""
    from .api import count_query, idf_of_lengths
    from .query import make_query_map, retrieve
    S = int(total_structures)
    qm = make_query_map(ctx, qbatch, q_indices, subs, None, float(S), dist_thr, angle_thr)
    lens = global_posting_lengths(shard_index, qm.hash, device)
    pl = global_posting_lengths(shard_index, qm.primary_hash, device)
    qm.set_idf(np.where(pl > 0, idf_of_lengths(np.maximum(pl, 1), S), 0.0).astype(np.float32))
    local = count_query(ctx, shard_index, qm.hash, qm.qi, qm.qj, penalty_shard, total_structures=S, as_array=True, lengths=lens)
    recs = allgather_hits(local, device if device is not None else torch.device("cpu"), top_n=top_n)
    matches = []
    if retrieve_matches:
        n_local = shard_index.n_structures
        mine = [int(n) for n in recs["nid"] if lo <= int(n) < lo + n_local]
        got = []
        if mine:
            cand = np.array([n - lo for n in mine], np.uint32)
            for m in retrieve(ctx, db_shard, resname_std_shard, cand, qm, qbatch, ca_distance):
                m = dict(m, nid=mine[m["cand"]])
                got.append(m)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [None] * dist.get_world_size()
            dist.all_gather_object(box, got)
            got = [m for part in box for m in part]
        order = {int(n): k for k, n in enumerate(recs["nid"])}      # candidate order of the global ranking, components in order
        matches = sorted(got, key=lambda m: order[m["nid"]])
    return recs, matches


class Comm:
    """fdgpu_comm: the RCCL communicator of libfdgpu.so itself (csrc/fd_comm.hip) — the exchange steps of the sharded query behind the
    C ABI, for hosts that do not run torch.  The 128-byte unique id travels from rank 0 to the others by whatever the host has
    (here: torch.distributed's broadcast when a process group exists)."""

    def __init__(self, ctx, rank: int = 0, world: int = 1, unique_id: bytes | None = None):
        import ctypes as C
        from ._lib import u8p
        self.ctx, self.rank, self.world = ctx, rank, world
        if unique_id is None:
            buf = np.zeros(128, np.uint8)
            if rank == 0:
                rc = ctx.L.fdgpu_comm_unique_id(buf.ctypes.data_as(u8p))
                if rc:
                    raise RuntimeError("fdgpu_comm_unique_id failed: RCCL is not available")
            if world > 1:       # the id travels over the process group: device tensor under nccl (= RCCL), host tensor under gloo
                dv = _dev(torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
                t = torch.from_numpy(buf).to(dv)
                dist.broadcast(t, src=0)
                buf = t.cpu().numpy()
            unique_id = buf.tobytes()
        self.unique_id = unique_id
        idb = np.frombuffer(unique_id, np.uint8).copy()
        h = C.c_void_p()
        ctx.check(ctx.L.fdgpu_comm_init(ctx.h, idb.ctypes.data_as(u8p), rank, world, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.fdgpu_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        """(ncclAllReduce calls, ncclAllGather calls) this communicator has issued"""
        from ._lib import u64p
        a, g = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        self.ctx.check(self.ctx.L.fdgpu_comm_stats(self.h, a.ctypes.data_as(u64p), g.ctypes.data_as(u64p)))
        return int(a[0]), int(g[0])

    def single_index(self, local):
        """fdgpu_comm_single_index: every rank brings the resident sub-index of its id range -> (this rank's hash range of the database's single index,
        hashes before it, value bytes before it, total hashes, total value bytes).  Device to device (ncclSend / ncclRecv), no host staging."""
        import ctypes as C
        from ._lib import u64p
        from .api import FolddiscoIndex
        h = C.c_void_p()
        o = np.zeros(4, np.uint64)
        p = [o[k:k + 1].ctypes.data_as(u64p) for k in range(4)]
        self.ctx.check(self.ctx.L.fdgpu_comm_single_index(self.ctx.h, self.h, local.h, C.byref(h), *p))
        n_total = int(self.ctx.L.fdgpu_index_num_structures(h))      # the merged pieces' id ranges: every structure of the database
        return FolddiscoIndex(self.ctx, h, n_total, 0), int(o[0]), int(o[1]), int(o[2]), int(o[3])

    def allreduce_lengths(self, lens: np.ndarray) -> np.ndarray:
        from ._lib import u64p
        a = np.ascontiguousarray(lens, np.uint64).copy()
        self.ctx.check(self.ctx.L.fdgpu_allreduce_lengths(self.ctx.h, self.h, a.ctypes.data_as(u64p), len(a)))
        return a

    def sharded_count_query(self, index, queries, penalty_shard, total_structures: int, top_n: int = 0) -> list:
        """queries: list of (q_hash, q_node, q_edge_j).  -> per query the global ranking (REC_DTYPE), identical on every rank"""
        import ctypes as C
        from ._lib import CountRec, f32p, u32p, u64p
        qh = np.ascontiguousarray(np.concatenate([np.asarray(q[0], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
        qn = np.ascontiguousarray(np.concatenate([np.asarray(q[1], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
        qe = np.ascontiguousarray(np.concatenate([np.asarray(q[2], np.uint32) for q in queries]) if queries else np.zeros(0, np.uint32))
        q_off = np.concatenate([[0], np.cumsum([len(q[0]) for q in queries])]).astype(np.uint64)
        pen = np.ascontiguousarray(penalty_shard, np.float32)
        out, ooff = C.POINTER(CountRec)(), u64p()
        self.ctx.check(self.ctx.L.fdgpu_sharded_count_query(self.ctx.h, self.h, index.h, len(queries), q_off.ctypes.data_as(u64p), qh.ctypes.data_as(u32p),
                                                            qn.ctypes.data_as(u32p), qe.ctypes.data_as(u32p), pen.ctypes.data_as(f32p), int(total_structures),
                                                            int(top_n), C.byref(out), C.byref(ooff)))
        off = np.ctypeslib.as_array(ooff, shape=(len(queries) + 1,)).copy()
        n = int(off[-1])
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n, 1) * 20,))[: n * 20].copy().view(REC_DTYPE)
        self.ctx.L.fdgpu_free(out)
        self.ctx.L.fdgpu_free(ooff)
        return [arr[int(off[t]): int(off[t + 1])] for t in range(len(queries))]

    @staticmethod
    def _take_recs(ctx, out, ooff, T):
        off = np.ctypeslib.as_array(ooff, shape=(T + 1,)).copy()
        n = int(off[-1])
        arr = np.ctypeslib.as_array(ctypes_cast_u8(out), shape=(max(n, 1) * 20,))[: n * 20].copy().view(REC_DTYPE)
        ctx.L.fdgpu_free(out)
        ctx.L.fdgpu_free(ooff)
        return [arr[int(off[t]): int(off[t + 1])] for t in range(T)]

    def sharded_count_query_maps(self, index, qms, penalty_shard, total_structures: int, top_n: int = 0) -> list:
        """fdgpu_sharded_count_query_maps: the QueryMapResults of make_query_maps(index=None) scored against the sharded index — posting
        lengths all-reduced on the device (the maps' idf is rewritten from the global lengths), local selection, one device-to-device
        all-gather, global ranking on the device.  penalty_shard=None uses the index's resident penalty."""
        import ctypes as C
        from ._lib import CountRec, f32p, u64p
        T = len(qms)
        handles = (C.c_void_p * max(T, 1))(*[C.cast(q.handle, C.c_void_p) for q in qms])
        pen = None if penalty_shard is None else np.ascontiguousarray(penalty_shard, np.float32)
        out, ooff = C.POINTER(CountRec)(), u64p()
        self.ctx.check(self.ctx.L.fdgpu_sharded_count_query_maps(self.ctx.h, self.h, index.h, T, handles, None if pen is None else pen.ctypes.data_as(f32p),
                                                                 int(total_structures), int(top_n), C.byref(out), C.byref(ooff)))
        return Comm._take_recs(self.ctx, out, ooff, T)

    def sharded_retrieve(self, db_shard, first_id: int, resname_std_shard, cand_nids, qms, qbatch, q_structs, ca_distance_cutoff=1.0, node_count=2,
                         nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, partial_fit=False, hash_type=3, multiple_bins=None):
        """fdgpu_sharded_retrieve: cand_nids[t] = GLOBAL structure ids of query t's candidates in ranking order (the same on every rank).
        -> (matches MATCH_DTYPE[], match_off, residues int32[], res_off) like query.retrieve_batch(as_arrays=True) over the whole
        database; matches["cand"] = slot in cand_nids[t].  Identical on every rank."""
        import ctypes as C
        from ._lib import HashParams, MatchRec, QueryMap, u8p, u32p, u64p
        from .query import MATCH_DTYPE
        T = len(qms)
        cl = [np.ascontiguousarray(c, dtype=np.uint32) for c in cand_nids]
        cand_off = np.concatenate([[0], np.cumsum([len(c) for c in cl])]).astype(np.uint64)
        cand = np.ascontiguousarray(np.concatenate(cl) if cl else np.zeros(0, np.uint32))
        std = None if resname_std_shard is None else np.ascontiguousarray(resname_std_shard, np.uint8)
        qs = np.ascontiguousarray(q_structs, np.uint32)
        handles = (C.POINTER(QueryMap) * max(T, 1))(*[q.handle for q in qms])
        p = HashParams(nbin_dist, nbin_angle, dist_cutoff, hash_type, multiple_bins)
        mp, rp = C.POINTER(MatchRec)(), C.POINTER(C.c_int32)()
        mo, ro = u64p(), u64p()
        ctx = self.ctx
        ctx.check(ctx.L.fdgpu_sharded_retrieve(ctx.h, self.h, db_shard.h, int(first_id), None if std is None else std.ctypes.data_as(u8p), T,
                                               cand.ctypes.data_as(u32p), cand_off.ctypes.data_as(u64p), handles, qbatch.h, qs.ctypes.data_as(u32p), C.byref(p),
                                               ca_distance_cutoff, node_count, int(bool(partial_fit)), C.byref(mp), C.byref(mo), C.byref(rp), C.byref(ro)))
        moff = np.ctypeslib.as_array(mo, shape=(T + 1,)).copy()
        roff = np.ctypeslib.as_array(ro, shape=(T + 1,)).copy()
        nm, nr = int(moff[-1]), int(roff[-1])
        isz = MATCH_DTYPE.itemsize
        marr = np.ctypeslib.as_array(C.cast(mp, C.POINTER(C.c_uint8)), shape=(max(nm, 1) * isz,))[: nm * isz].copy().view(MATCH_DTYPE)
        rarr = np.ctypeslib.as_array(rp, shape=(max(nr, 1),))[:nr].copy()
        ctx.L.fdgpu_matches_free(mp, rp)
        ctx.L.fdgpu_free(mo)
        ctx.L.fdgpu_free(ro)
        return marr, moff, rarr, roff


def ctypes_cast_u8(p):
    import ctypes as C
    return C.cast(p, C.POINTER(C.c_uint8))


def merge_gathered(ctx, messages: np.ndarray, world: int, n_queries: int, top_n: int) -> list:
    """fdgpu_debug_merge_gathered: the device merge every rank runs after the all-gather, on hand-made messages (tests)"""
    import ctypes as C
    from ._lib import CountRec, u8p, u64p
    msg = np.ascontiguousarray(messages, np.uint8)
    out, ooff = C.POINTER(CountRec)(), u64p()
    ctx.check(ctx.L.fdgpu_debug_merge_gathered(ctx.h, int(world), int(n_queries), int(top_n), msg.ctypes.data_as(u8p), C.byref(out), C.byref(ooff)))
    return Comm._take_recs(ctx, out, ooff, n_queries)


def build_message(ctx, per_query_recs, top_n: int, status: int = 0) -> np.ndarray:
    """one rank's message of the device exchange (include/fdgpu.h, fdgpu_comm_message_bytes) from its ranked per-query records"""
    T = len(per_query_recs)
    mb = int(ctx.L.fdgpu_comm_message_bytes(T, top_n))
    m = np.zeros(mb, np.uint8)
    m[:16].view(np.uint32)[:] = (status, T, top_n, top_n + 1024)
    st = m[16:16 + 16 * T].view(np.uint32).reshape(T, 4)
    rec = m[16 + 16 * T:16 + 16 * T + 20 * T * top_n].view(REC_DTYPE).reshape(T, top_n) if T else None
    for t, r in enumerate(per_query_recs):
        k = min(len(r), top_n)
        st[t, 3] = len(r) if len(r) <= top_n + 1024 else k
        rec[t, :k] = r[:k]
    return m


def sharded_count_query_maps(ctx, index, qms, penalty_shard, total_structures: int, top_n: int, device=None, comm: "Comm | None" = None) -> list:
    """The batched sharded prefilter through the fused library entry points, over either transport: with an fdgpu Comm (RCCL, one GPU per
    rank) everything happens inside fdgpu_sharded_count_query_maps; without one (gloo: CPU tests, several ranks on one GPU) the two local
    halves of the same call run here with torch.distributed in between — fdgpu_query_maps_lengths -> all-reduce ->
    fdgpu_count_query_maps_top_global (idf from the global lengths, device selection of the local top_n) -> all-gather -> ranking."""
    import ctypes as C
    from ._lib import CountRec, f32p, u64p
    if comm is not None:
        return comm.sharded_count_query_maps(index, qms, penalty_shard, total_structures, top_n)
    T = len(qms)
    handles = (C.c_void_p * max(T, 1))(*[C.cast(q.handle, C.c_void_p) for q in qms])
    nq = int(sum(len(q.hash) for q in qms))
    lens = np.zeros(max(2 * nq, 1), np.uint64)
    ctx.check(ctx.L.fdgpu_query_maps_lengths(ctx.h, index.h, T, handles, lens.ctypes.data_as(u64p)))
    lens = np.ascontiguousarray(reduce_lengths(lens, device))
    pen = None if penalty_shard is None else np.ascontiguousarray(penalty_shard, np.float32)
    out, ooff = C.POINTER(CountRec)(), u64p()
    ctx.check(ctx.L.fdgpu_count_query_maps_top_global(ctx.h, index.h, T, handles, lens.ctypes.data_as(u64p), None if pen is None else pen.ctypes.data_as(f32p),
                                                      float(total_structures), int(top_n), C.byref(out), C.byref(ooff)))
    local = Comm._take_recs(ctx, out, ooff, T)
    return allgather_hits_many(local, device, top_n=top_n if top_n else None, ranked=bool(top_n))


def single_index_over_process_group(ctx, local, device=None):
    """The same exchange for launches without RCCL between the ranks (gloo: tests, several ranks on one GPU): bounds from rank 0, slices on the device, the
    pieces through torch.distributed as host arrays (this transport has no device path), merge of the received pieces on the device.  -> as Comm.single_index"""
    from .api import FolddiscoIndex, FolddiscoIndexSet
    if not _active():
        return local, 0, 0, local.num_hashes, local.value_len
    W, me = dist.get_world_size(), dist.get_rank()
    box = [local.range_bounds(W).tolist() if me == 0 else None]
    dist.broadcast_object_list(box, src=0)
    edge = [0] + [int(x) for x in box[0]] + [1 << 32]
    for j in range(1, W):
        edge[j] = max(edge[j], edge[j - 1])
    pieces = []
    for j in range(W):
        s = local.slice(edge[j], edge[j + 1])
        pieces.append((s.export(), int(local.first_id), int(local.n_structures)))
        del s
    got = [None] * W
    if dist.get_backend() == "nccl":      # (ProcessGroupNCCL has no gather: every rank takes every rank's piece j in turn and keeps its own range's — W x the bytes, a fall-back only)
        for j in range(W):
            box_j = [None] * W
            dist.all_gather_object(box_j, pieces[j])
            if me == j:
                got = box_j
            del box_j
    else:
        for j in range(W):      # piece j of every rank to rank j
            dist.gather_object(pieces[j], got if me == j else None, dst=j)
    parts = [FolddiscoIndex.load(ctx, h, o, v, n, first_id=f) for (v, h, o), f, n in got]
    rng = FolddiscoIndexSet(parts).merge() if len(parts) > 1 else parts[0]
    sizes = [None] * W
    dist.all_gather_object(sizes, (rng.num_hashes, rng.value_len))
    hb, vb = sum(x[0] for x in sizes[:me]), sum(x[1] for x in sizes[:me])
    return rng, hb, vb, sum(x[0] for x in sizes), sum(x[1] for x in sizes)
