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
            if world > 1:
                t = torch.from_numpy(buf)
                dist.broadcast(t, src=0)
                buf = t.numpy()
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
