// fd_comm.hip — the multi-GPU exchange steps of the query path behind the C ABI: one process per GPU, RCCL over xGMI.
//
// The index and the coordinates are sharded by structure id, so scoring and matching are local; three exchanges remain and all live here:
//   fdgpu_allreduce_lengths        posting lengths of the query's hashes summed over the shards — idf = log2(S / len) needs the length over
//                                  the WHOLE database (controller/query.rs:17-32, count_query.rs:130); ncclAllReduce on a device buffer
//   fdgpu_sharded_count_query[_maps]  local count_query with the candidate selection on the device -> ONE ncclAllGather of a fixed-stride
//                                  message per rank (selection state + ranked records, device to device, nothing staged on the host) ->
//                                  global selection and ranking on the device (k_comm_plan / k_comm_pack + the radix select and bitonic
//                                  sort of k_query.hip: idf descending, nid ascending, truncate — query_pdb.rs:404-411) -> one copy to the
//                                  host, identical on every rank.  Calls the device selection does not serve (top_n = 0: every touched
//                                  structure; top_n > 3072) exchange variable-length lists: counts first, then one padded payload.
//   fdgpu_sharded_retrieve         every candidate of the global ranking is matched on the rank that owns it (query_pdb.rs:415-452 per
//                                  shard), the fixed-size match records and residue lists are all-gathered and merged in candidate order
// Nothing is short-cut for a world of one: the collectives run (a 1-rank all-gather is a device copy inside RCCL), so a single-GPU test
// executes every line the N-rank path executes.  A rank whose local step fails still takes part in every collective of the call with an
// error status in its message, and all ranks return the error together — nobody is left waiting inside ncclAllGather.  What is NOT covered that
// way — a rank that cannot get an exchange buffer beyond the reserve made at fdgpu_comm_init, or whose HIP runtime fails between two
// collectives of a call — aborts the communicator (ncclCommAbort) and marks it dead: a sharded call that returns FDGPU_EHIP leaves the
// communicator unusable on every rank.
// RCCL is bound at run time (dlopen of librccl.so.1 — the copy torch already loaded when there is one), so libfdgpu.so itself has no
// link-time dependency on it; without RCCL the comm entry points fail with FDGPU_EHIP and say so.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <rccl/rccl.h>
#include "fdgpu_internal.h"

namespace {
struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;        // optional
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;      // optional: the single-index exchange (fdgpu_comm_single_index)
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
rccl_api &rccl() {
    static rccl_api A = [] {
        rccl_api a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.CommAbort = (decltype(a.CommAbort))dlsym(a.lib, "ncclCommAbort");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
        a.Send = (decltype(a.Send))dlsym(a.lib, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.lib, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.GetErrorString;
        return a;
    }();
    return A;
}

// device scratch of the global merge; owned by a communicator, or by a debug call
struct merge_bufs {
    fd_devbuf pack, off, sel, state, fin, flag;
    void release() { pack.release(); off.release(); sel.release(); state.release(); fin.release(); flag.release(); }
};
}  // namespace

struct fdgpu_comm {
    fdgpu_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    fd_devbuf send, recv, aux;      // aux: small per-call tables of the device-resident retrieval exchange
    merge_bufs mb;
    uint64_t n_sendrecv = 0;                        // point-to-point transfers issued (fdgpu_comm_single_index)
    uint64_t n_allreduce = 0, n_allgather = 0;      // collectives issued so far (fdgpu_comm_stats: tests check that a world of one runs them)
    bool dead = false;                              // aborted after a failure between collectives: every later call fails at once
};
// reserved at fdgpu_comm_init: the exchange buffers of ordinary calls (a batch of 128 queries x top 1000 is 2.6 MB per rank) exist before any
// rank-local work, so no rank can fail to allocate them halfway through a call while its peers already wait in a collective
static const size_t FD_COMM_RESERVE = (size_t)4 << 20;

#define FAIL_(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define NCHK(c, expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t _r = (expr);                                                                                      \
        if (_r != ncclSuccess) { (c)->err = std::string(#expr " -> ") + rccl().GetErrorString(_r); return FDGPU_EHIP; } \
    } while (0)
#define HCHK(c, expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) { (c)->err = std::string(#expr " -> ") + hipGetErrorString(_e); return FDGPU_EHIP; } \
    } while (0)

// A HIP failure BETWEEN the collectives of one call (an exchange buffer beyond the reserve that cannot grow, a failed copy): the peers may already
// wait for this rank inside a collective that it will never enter.  The communicator is aborted (ncclCommAbort: the peers' pending operation
// ends in an RCCL error instead of a wait without end where the library offers it) and marked dead — every later call on it fails at once.
static int comm_broken(fdgpu_ctx *c, fdgpu_comm *m, const std::string &what) {
    if (!m->dead) {
        m->dead = true;
        if (m->comm && rccl().CommAbort) { (void)rccl().CommAbort(m->comm); m->comm = nullptr; }
    }
    c->err = what + " — the communicator was aborted (a rank failed between the collectives of a call); create a new one";
    return FDGPU_EHIP;
}
#define HCHK_C(c, m, expr)                                                                                             \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) return comm_broken((c), (m), std::string(#expr " -> ") + hipGetErrorString(_e));         \
    } while (0)

extern "C" int fdgpu_comm_unique_id(uint8_t id[FDGPU_COMM_ID_BYTES]) {
    if (!id) return FDGPU_EINVAL;
    if (!rccl().ok) return FDGPU_EHIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return FDGPU_EHIP;
    static_assert(sizeof(u.internal) == FDGPU_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, u.internal, FDGPU_COMM_ID_BYTES);
    return FDGPU_OK;
}

extern "C" int fdgpu_comm_init(fdgpu_ctx *c, const uint8_t id[FDGPU_COMM_ID_BYTES], int rank, int world, fdgpu_comm **out) { FD_LOCK(c);
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return FDGPU_EINVAL;
    *out = nullptr;
    if (!rccl().ok) FAIL_(c, FDGPU_EHIP, "RCCL is not available (librccl.so.1 could not be loaded): the multi-GPU entry points need it");
    fdgpu_comm *m = new (std::nothrow) fdgpu_comm();
    if (!m) return FDGPU_ENOMEM;
    m->ctx = c; m->rank = rank; m->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, FDGPU_COMM_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&m->comm, world, u, rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank -> ") + rccl().GetErrorString(r); delete m; return FDGPU_EHIP; }
    if (m->send.ensure(FD_COMM_RESERVE) != hipSuccess || m->recv.ensure(FD_COMM_RESERVE * (size_t)world) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "fdgpu_comm_init: exchange buffers";
        (void)rccl().CommDestroy(m->comm); m->send.release(); m->recv.release(); m->aux.release(); delete m;
        return FDGPU_EHIP;
    }
    *out = m;
    return FDGPU_OK;
}
extern "C" void fdgpu_comm_destroy(fdgpu_comm *m) {
    if (!m) return;
    FD_LOCK(m->ctx);
    if (m->comm && rccl().ok) (void)rccl().CommDestroy(m->comm);
    m->send.release(); m->recv.release(); m->aux.release(); m->mb.release();
    delete m;
}
extern "C" int fdgpu_comm_rank(const fdgpu_comm *m) { return m ? m->rank : -1; }
extern "C" int fdgpu_comm_world(const fdgpu_comm *m) { return m ? m->world : 0; }
extern "C" int fdgpu_comm_stats(const fdgpu_comm *m, uint64_t *n_allreduce, uint64_t *n_allgather) {
    if (!m) return FDGPU_EINVAL;
    if (n_allreduce) *n_allreduce = m->n_allreduce;
    if (n_allgather) *n_allgather = m->n_allgather;
    return FDGPU_OK;
}

// in-place sum over the ranks of n u64 on the device (stream-ordered, no synchronisation)
static int allreduce_dev(fdgpu_ctx *c, fdgpu_comm *m, uint64_t *dev, uint64_t n) {
    if (m->dead) FAIL_(c, FDGPU_EHIP, "communicator aborted by an earlier failure");
    if (!n) return FDGPU_OK;
    NCHK(c, rccl().AllReduce(dev, dev, n, ncclUint64, ncclSum, m->comm, c->stream));
    ++m->n_allreduce;
    return FDGPU_OK;
}
static int allgather_dev(fdgpu_ctx *c, fdgpu_comm *m, const void *send, void *recv, size_t bytes) {
    if (m->dead) FAIL_(c, FDGPU_EHIP, "communicator aborted by an earlier failure");
    NCHK(c, rccl().AllGather(send, recv, bytes, ncclUint8, m->comm, c->stream));
    ++m->n_allgather;
    return FDGPU_OK;
}

// lengths[k] <- sum over ranks (in place, host array)
extern "C" int fdgpu_allreduce_lengths(fdgpu_ctx *c, fdgpu_comm *m, uint64_t *lengths, uint64_t n) { FD_LOCK(c);
    if (!c || !m || (n && !lengths)) return FDGPU_EINVAL;
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    HCHK(c, m->send.ensure(n * 8));
    HCHK(c, hipMemcpyAsync(m->send.p, lengths, n * 8, hipMemcpyHostToDevice, st));
    int rc = allreduce_dev(c, m, m->send.as<uint64_t>(), n);
    if (rc) return rc;
    HCHK(c, hipMemcpyAsync(lengths, m->send.p, n * 8, hipMemcpyDeviceToHost, st));
    HCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

namespace {
inline uint64_t rank_key(const fd_count_rec &r) {   // idf descending, nid ascending (stable sort of nid-ordered input, query_pdb.rs:404-411)
    float f = r.idf + 0.0f;
    uint32_t b;
    memcpy(&b, &f, 4);
    const uint32_t ordered = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((uint64_t)(~ordered) << 32) | r.nid;
}
inline void rank_sort(fd_count_rec *a, fd_count_rec *b) {
    std::sort(a, b, [](const fd_count_rec &x, const fd_count_rec &y) { return rank_key(x) < rank_key(y); });
}

// ---- the fixed-stride message of the device path ------------------------------------------------------------------------------------
// rank r contributes, for a call with n_queries queries and a cut at top_n:
//   comm_hdr                      status (0 = fine, else the rank's FDGPU_E* code negated), n_queries, top_n, cap
//   sel_state[n_queries]          the selection state k_topn_* leave behind: .count = records the rank selected for the query
//   fd_count_rec[n_queries][top_n]  its ranked records, min(count, top_n) valid per query
// padded to 16 bytes.  fdgpu_comm_message_bytes gives the size; tests build such messages by hand (fdgpu_debug_merge_gathered).
struct comm_hdr { uint32_t status, n_queries, top_n, cap; };
struct sel_state { uint32_t thr_bin, above, thr22, count; };    // = topn_state of k_query.hip
inline size_t msg_bytes(uint64_t nq, uint32_t top_n) {
    return (sizeof(comm_hdr) + nq * sizeof(sel_state) + nq * (size_t)top_n * sizeof(fd_count_rec) + 15) & ~(size_t)15;
}
__device__ __forceinline__ const sel_state *msg_state(const uint8_t *recv, uint64_t mb, uint32_t r) { return (const sel_state *)(recv + r * mb + sizeof(comm_hdr)); }

// flags: 1 = a rank reported an error, 2 = a rank's selection overflowed (it should have resolved that locally), 4 = ranks disagree on the call
__global__ __launch_bounds__(256) void k_comm_plan(const uint8_t *__restrict__ recv, uint32_t W, uint64_t mb, uint32_t nq, uint32_t top_n, uint64_t *__restrict__ off,
                                                   uint32_t *__restrict__ flags) {
    for (uint32_t r = threadIdx.x; r < W; r += 256) {
        const comm_hdr *h = (const comm_hdr *)(recv + r * mb);
        if (h->status) atomicOr(flags, 1u);
        if (h->n_queries != nq || h->top_n != top_n) atomicOr(flags, 4u);
    }
    for (uint32_t t = threadIdx.x; t < nq; t += 256) {
        uint64_t tot = 0;
        for (uint32_t r = 0; r < W; ++r) {
            const uint32_t cnt = msg_state(recv, mb, r)[t].count, cap = ((const comm_hdr *)(recv + r * mb))->cap;
            if (cnt > cap) atomicOr(flags, 2u);
            tot += cnt < top_n ? cnt : top_n;
        }
        off[t + 1] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        off[0] = 0;
        for (uint32_t t = 0; t < nq; ++t) { acc += off[t + 1]; off[t + 1] = acc; }
    }
}
// query t's records of rank r -> packed[off[t] + (records of ranks before r)]: per query one contiguous list, ranks in order
__global__ __launch_bounds__(128) void k_comm_pack(const uint8_t *__restrict__ recv, uint64_t mb, uint32_t nq, uint32_t top_n, const uint64_t *__restrict__ off,
                                                   uint32_t *__restrict__ packed) {
    const uint32_t t = blockIdx.x, r = blockIdx.y;
    uint64_t base = off[t];
    for (uint32_t q = 0; q < r; ++q) { const uint32_t c = msg_state(recv, mb, q)[t].count; base += c < top_n ? c : top_n; }
    uint32_t cnt = msg_state(recv, mb, r)[t].count;
    cnt = cnt < top_n ? cnt : top_n;
    const uint32_t *src = (const uint32_t *)(recv + r * mb + sizeof(comm_hdr) + (size_t)nq * sizeof(sel_state)) + ((size_t)t * top_n) * 5;
    uint32_t *dst = packed + base * 5;
    for (uint32_t k = threadIdx.x; k < cnt * 5; k += 128) dst[k] = src[k];
}

// W gathered messages on the device -> per query the global ranking cut to top_n, on the host.  Everything between the gather and the one
// copy back runs on the device.  status_out: the first non-zero status any rank reported (the call then has no result).
int merge_gathered(fdgpu_ctx *c, merge_bufs &B, const uint8_t *recv, uint32_t W, uint64_t n_queries, uint32_t top_n, fd_count_rec **out, uint64_t **out_off) {
    hipStream_t st = c->stream;
    const size_t mb = msg_bytes(n_queries, top_n);
    const uint32_t cap = top_n + 1024;
    const size_t n_pack = std::max<size_t>((size_t)W * n_queries * top_n, 1);
    HCHK(c, B.pack.ensure(n_pack * sizeof(fd_count_rec)));
    HCHK(c, B.off.ensure((n_queries + 2) * 8));
    HCHK(c, B.sel.ensure(std::max<size_t>((size_t)n_queries * cap, 1) * sizeof(fd_count_rec)));
    HCHK(c, B.state.ensure(std::max<uint64_t>(n_queries, 1) * sizeof(sel_state)));
    HCHK(c, B.fin.ensure(std::max<size_t>((size_t)n_queries * top_n, 1) * sizeof(fd_count_rec)));
    HCHK(c, B.flag.ensure(64));
    const size_t topn_bytes = std::max<uint64_t>(n_queries, 1) * 2048 * 4;      // the radix select's histogram table: zeroed when (re)allocated, left zero
    if (c->ws[WS_CQ_TOPN].cap < topn_bytes) {
        HCHK(c, c->ws[WS_CQ_TOPN].ensure(topn_bytes));
        HCHK(c, hipMemsetAsync(c->ws[WS_CQ_TOPN].p, 0, c->ws[WS_CQ_TOPN].cap, st));
    }
    HCHK(c, hipMemsetAsync(B.flag.p, 0, 64, st));
    std::vector<uint32_t> flags(1, 0);
    std::vector<sel_state> state(std::max<uint64_t>(n_queries, 1));
    std::vector<fd_count_rec> fin(std::max<size_t>((size_t)n_queries * top_n, 1));
    if (n_queries) {
        hipLaunchKernelGGL(k_comm_plan, dim3(1), dim3(256), 0, st, recv, W, (uint64_t)mb, (uint32_t)n_queries, top_n, B.off.as<uint64_t>(), B.flag.as<uint32_t>());
        hipLaunchKernelGGL(k_comm_pack, dim3((unsigned)n_queries, W), dim3(128), 0, st, recv, (uint64_t)mb, (uint32_t)n_queries, top_n, B.off.as<uint64_t>(), B.pack.as<uint32_t>());
        fd_launch_cq_topn(B.pack.p, B.off.as<uint64_t>(), (uint32_t)n_queries, top_n, cap, B.sel.p, B.state.p, c->ws[WS_CQ_TOPN].as<uint32_t>(), st);
        fd_launch_cq_topn_sort(B.sel.p, cap, B.state.p, (uint32_t)n_queries, top_n, B.fin.p, st);
        HCHK(c, hipGetLastError());
        HCHK(c, hipMemcpyAsync(state.data(), B.state.p, n_queries * sizeof(sel_state), hipMemcpyDeviceToHost, st));
        HCHK(c, hipMemcpyAsync(fin.data(), B.fin.p, (size_t)n_queries * top_n * sizeof(fd_count_rec), hipMemcpyDeviceToHost, st));
    }
    HCHK(c, hipMemcpyAsync(flags.data(), B.flag.p, 4, hipMemcpyDeviceToHost, st));
    HCHK(c, hipStreamSynchronize(st));
    if (flags[0] & 4u) FAIL_(c, FDGPU_EINVAL, "sharded query: the ranks passed different batches (n_queries / top_n differ)");
    if (flags[0] & 3u) {
        std::vector<comm_hdr> hd(W);
        for (uint32_t r = 0; r < W; ++r) HCHK(c, hipMemcpy(&hd[r], recv + r * mb, sizeof(comm_hdr), hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < W; ++r)
            if (hd[r].status) { c->err = "sharded query: rank " + std::to_string(r) + " failed its local step (code " + std::to_string(-(int)hd[r].status) + ")"; return FDGPU_EHIP; }
        FAIL_(c, FDGPU_EHIP, "sharded query: a rank sent an overflowed selection");
    }
    bool overflow = false;
    uint64_t tot = 0;
    for (uint64_t t = 0; t < n_queries; ++t) { overflow = overflow || state[t].count > cap; tot += std::min<uint32_t>(state[t].count, top_n); }
    std::vector<uint64_t> hoff;
    std::vector<fd_count_rec> hpack;
    if (overflow) {      // more ties at a cut-off than the selection's slots hold: those queries are ranked here from the packed lists
        hoff.resize(n_queries + 1);
        HCHK(c, hipMemcpy(hoff.data(), B.off.p, (n_queries + 1) * 8, hipMemcpyDeviceToHost));
        hpack.resize(std::max<uint64_t>(hoff[n_queries], 1));
        if (hoff[n_queries]) HCHK(c, hipMemcpy(hpack.data(), B.pack.p, hoff[n_queries] * sizeof(fd_count_rec), hipMemcpyDeviceToHost));
        tot = 0;
        for (uint64_t t = 0; t < n_queries; ++t) tot += std::min<uint64_t>(hoff[t + 1] - hoff[t], top_n);
    }
    uint64_t *ooff = (uint64_t *)calloc(n_queries + 1, 8);
    fd_count_rec *rr = (fd_count_rec *)malloc(std::max<uint64_t>(tot, 1) * sizeof(fd_count_rec));
    if (!ooff || !rr) { free(ooff); free(rr); return FDGPU_ENOMEM; }
    uint64_t w = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        ooff[t] = w;
        if (overflow) {
            fd_count_rec *a = hpack.data() + hoff[t], *b = hpack.data() + hoff[t + 1];
            rank_sort(a, b);
            const uint64_t k = std::min<uint64_t>((uint64_t)(b - a), top_n);
            if (k) memcpy(rr + w, a, k * sizeof(fd_count_rec));
            w += k;
        } else {
            const uint64_t k = std::min<uint32_t>(state[t].count, top_n);
            if (k) memcpy(rr + w, fin.data() + (size_t)t * top_n, k * sizeof(fd_count_rec));
            w += k;
        }
    }
    ooff[n_queries] = w;
    *out = rr; *out_off = ooff;
    return FDGPU_OK;
}

// variable-length lists (top_n = 0 or beyond the device selection): per-query counts + a status word first, then one padded payload
int exchange_lists(fdgpu_ctx *c, fdgpu_comm *m, uint64_t n_queries, uint32_t top_n, int local_rc, fd_count_rec *loc, const uint64_t *loff, fd_count_rec **out,
                   uint64_t **out_off) {
    hipStream_t st = c->stream;
    const int W = m->world;
    std::vector<fd_count_rec> mine;
    std::vector<uint64_t> cnt(n_queries + 1, 0);          // [n_queries] = status
    cnt[n_queries] = local_rc ? (uint64_t)(uint32_t)(-local_rc) : 0;
    if (!local_rc)
        for (uint64_t t = 0; t < n_queries; ++t) {        // a rank never contributes more than top_n records per query
            fd_count_rec *a = loc + loff[t], *b = loc + loff[t + 1];
            rank_sort(a, b);
            const uint64_t keep = top_n ? std::min<uint64_t>(top_n, (uint64_t)(b - a)) : (uint64_t)(b - a);
            mine.insert(mine.end(), a, a + keep);
            cnt[t] = keep;
        }
    const size_t cb = (n_queries + 1) * 8;
    std::vector<uint64_t> all_cnt((size_t)W * (n_queries + 1));
    HCHK_C(c, m, m->send.ensure(cb));
    HCHK_C(c, m, m->recv.ensure(cb * W));
    HCHK_C(c, m, hipMemcpyAsync(m->send.p, cnt.data(), cb, hipMemcpyHostToDevice, st));
    int rc = allgather_dev(c, m, m->send.p, m->recv.p, cb);
    if (rc) return rc;
    HCHK_C(c, m, hipMemcpyAsync(all_cnt.data(), m->recv.p, cb * W, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    for (int r = 0; r < W; ++r)
        if (all_cnt[(size_t)r * (n_queries + 1) + n_queries]) {      // every rank sees the same statuses: all return here, none enters the second gather
            if (r != m->rank || !local_rc) c->err = "sharded query: rank " + std::to_string(r) + " failed its local step (code -" + std::to_string(all_cnt[(size_t)r * (n_queries + 1) + n_queries]) + ")";
            return local_rc ? local_rc : FDGPU_EHIP;
        }
    uint64_t stride = 1;
    for (int r = 0; r < W; ++r) {
        uint64_t tot = 0;
        for (uint64_t t = 0; t < n_queries; ++t) tot += all_cnt[(size_t)r * (n_queries + 1) + t];
        stride = std::max(stride, tot);
    }
    const size_t bytes = stride * sizeof(fd_count_rec);
    HCHK_C(c, m, m->send.ensure(bytes));
    HCHK_C(c, m, m->recv.ensure(bytes * W));
    if (!mine.empty()) HCHK_C(c, m, hipMemcpyAsync(m->send.p, mine.data(), mine.size() * sizeof(fd_count_rec), hipMemcpyHostToDevice, st));
    if ((rc = allgather_dev(c, m, m->send.p, m->recv.p, bytes))) return rc;
    std::vector<fd_count_rec> all(stride * W);
    HCHK_C(c, m, hipMemcpyAsync(all.data(), m->recv.p, bytes * W, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    uint64_t *ooff = (uint64_t *)calloc(n_queries + 1, 8);
    if (!ooff) return FDGPU_ENOMEM;
    std::vector<fd_count_rec> res;
    std::vector<uint64_t> base((size_t)W, 0);
    for (uint64_t t = 0; t < n_queries; ++t) {
        const size_t s0 = res.size();
        for (int r = 0; r < W; ++r) {
            const uint64_t k = all_cnt[(size_t)r * (n_queries + 1) + t];
            const fd_count_rec *src = all.data() + (size_t)r * stride + base[r];
            res.insert(res.end(), src, src + k);
            base[r] += k;
        }
        rank_sort(res.data() + s0, res.data() + res.size());
        if (top_n && res.size() - s0 > top_n) res.resize(s0 + top_n);
        ooff[t + 1] = res.size();
    }
    fd_count_rec *o = (fd_count_rec *)malloc(std::max<size_t>(res.size(), 1) * sizeof(fd_count_rec));
    if (!o) { free(ooff); return FDGPU_ENOMEM; }
    if (!res.empty()) memcpy(o, res.data(), res.size() * sizeof(fd_count_rec));
    *out = o; *out_off = ooff;
    return FDGPU_OK;
}

inline bool device_path(uint32_t top_n) { return top_n > 0 && top_n + 1024 <= 4096; }      // the cut k_topn_sort serves (fdgpu_api.hip: dense_topn)

// the exchange after the local scoring: local_rc / dev / (loc, loff) describe what this rank has
int exchange(fdgpu_ctx *c, fdgpu_comm *m, uint64_t n_queries, uint32_t top_n, int local_rc, const fd_cq_dev_out &dev, fd_count_rec *loc, const uint64_t *loff,
             fd_count_rec **out, uint64_t **out_off) {
    if (!device_path(top_n)) return exchange_lists(c, m, n_queries, top_n, local_rc, loc, loff, out, out_off);
    hipStream_t st = c->stream;
    const int W = m->world;
    const size_t mb = msg_bytes(n_queries, top_n);
    HCHK_C(c, m, m->send.ensure(mb));
    HCHK_C(c, m, m->recv.ensure(mb * W));
    uint8_t *snd = m->send.as<uint8_t>();
    comm_hdr hd{local_rc ? (uint32_t)(-local_rc) : 0u, (uint32_t)n_queries, top_n, top_n + 1024};
    std::vector<uint8_t> host_msg;
    if (!local_rc && dev.got) {       // the selection state and the ranked records go from where the kernels left them into the message
        HCHK_C(c, m, hipMemcpyAsync(snd, &hd, sizeof hd, hipMemcpyHostToDevice, st));
        if (n_queries) {
            HCHK_C(c, m, hipMemcpyAsync(snd + sizeof hd, dev.state, n_queries * sizeof(sel_state), hipMemcpyDeviceToDevice, st));
            HCHK_C(c, m, hipMemcpyAsync(snd + sizeof hd + n_queries * sizeof(sel_state), dev.recs, (size_t)n_queries * top_n * sizeof(fd_count_rec), hipMemcpyDeviceToDevice, st));
        }
    } else {                          // a call the device selection did not serve on this rank (empty shard, wide accumulators, overflow), or an error
        host_msg.assign(mb, 0);
        memcpy(host_msg.data(), &hd, sizeof hd);
        if (!local_rc && loc)
            for (uint64_t t = 0; t < n_queries; ++t) {
                fd_count_rec *a = loc + loff[t], *b = loc + loff[t + 1];
                rank_sort(a, b);
                const uint32_t k = (uint32_t)std::min<uint64_t>((uint64_t)(b - a), top_n);
                sel_state s{0, 0, 0, k};
                memcpy(host_msg.data() + sizeof hd + t * sizeof s, &s, sizeof s);
                if (k) memcpy(host_msg.data() + sizeof hd + n_queries * sizeof s + (size_t)t * top_n * sizeof(fd_count_rec), a, (size_t)k * sizeof(fd_count_rec));
            }
        HCHK_C(c, m, hipMemcpyAsync(snd, host_msg.data(), mb, hipMemcpyHostToDevice, st));
    }
    int rc = allgather_dev(c, m, snd, m->recv.p, mb);
    if (!rc) rc = merge_gathered(c, m->mb, m->recv.as<uint8_t>(), (uint32_t)W, n_queries, top_n, out, out_off);
    if (!host_msg.empty()) (void)hipStreamSynchronize(st);     // host_msg outlives its copy on every path
    return local_rc ? local_rc : rc;
}
}  // namespace

extern "C" uint64_t fdgpu_comm_message_bytes(uint64_t n_queries, uint32_t top_n) { return msg_bytes(n_queries, top_n); }

// The global merge alone, on `world` messages laid out as the all-gather leaves them (host array, world * fdgpu_comm_message_bytes): what
// every rank runs after the gather.  Lets a single-GPU test drive the multi-rank unpack / select / rank code with hand-made contributions.
extern "C" int fdgpu_debug_merge_gathered(fdgpu_ctx *c, uint32_t world, uint64_t n_queries, uint32_t top_n, const uint8_t *messages, fd_count_rec **out,
                                          uint64_t **out_off) { FD_LOCK(c);
    if (!c || !world || !messages || !out || !out_off || !device_path(top_n)) return FDGPU_EINVAL;
    *out = nullptr; *out_off = nullptr;
    const size_t mb = msg_bytes(n_queries, top_n);
    fd_devbuf recv;
    merge_bufs B;
    hipError_t e = recv.ensure(mb * world);
    if (e == hipSuccess) e = hipMemcpyAsync(recv.p, messages, mb * world, hipMemcpyHostToDevice, c->stream);
    int rc = e == hipSuccess ? merge_gathered(c, B, recv.as<uint8_t>(), world, n_queries, top_n, out, out_off) : FDGPU_EHIP;
    (void)hipStreamSynchronize(c->stream);
    recv.release(); B.release();
    return rc;
}

// The sharded prefilter of a batch of queries.  Every rank passes the SAME queries (q_off / q_hash / q_node / q_edge_j as in
// fdgpu_count_query_batch, without idf) and its own shard `ix` + penalty (n_structures(ix) entries).  Result, identical on every rank:
// per query the global ranking truncated to top_n (0 = all touched structures), nid = global structure id.
extern "C" int fdgpu_sharded_count_query(fdgpu_ctx *c, fdgpu_comm *m, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off, const uint32_t *q_hash,
                                         const uint32_t *q_node, const uint32_t *q_edge_j, const float *penalty, uint64_t total_structures, uint32_t top_n,
                                         fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    if (!c || !m || !ix || !out || !out_off || !q_off) return FDGPU_EINVAL;
    *out = nullptr; *out_off = nullptr;
    const uint64_t nq = q_off[n_queries];
    if (nq && (!q_hash || !q_node || !q_edge_j)) return FDGPU_EINVAL;
    hipStream_t st = c->stream;
    // 1. posting lengths over the whole database: counted and summed on the device -> idf per query hash; absent hashes drop out
    //    (count_query.rs:121-130)
    std::vector<uint64_t> lens(std::max<uint64_t>(nq, 1), 0);
    uint64_t *dl = nullptr;
    int local_rc = fd_posting_lengths_dev(c, ix, q_hash, nq, &dl);
    if (local_rc) {     // this rank still takes part in the sum — with zeros, from its own buffer or the communicator's reserve — and reports its status in the gather
        if (!dl && nq * 8 <= m->send.cap) dl = m->send.as<uint64_t>();
        if (!dl || hipMemsetAsync(dl, 0, std::max<uint64_t>(nq, 1) * 8, st) != hipSuccess) return comm_broken(c, m, "sharded_count_query: no buffer for the posting lengths");
    }
    int rc = allreduce_dev(c, m, dl, nq);
    if (rc) return rc;
    if (nq) HCHK_C(c, m, hipMemcpyAsync(lens.data(), dl, nq * 8, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    std::vector<uint32_t> kh, kn, ke;
    std::vector<float> kidf;
    std::vector<uint64_t> koff(n_queries + 1, 0);
    const float Sf = (float)total_structures;
    for (uint64_t t = 0; t < n_queries; ++t) {
        for (uint64_t k = q_off[t]; k < q_off[t + 1]; ++k) {
            if (!lens[k]) continue;
            kh.push_back(q_hash[k]); kn.push_back(q_node[k]); ke.push_back(q_edge_j[k]);
            kidf.push_back(log2f(Sf / (float)lens[k]));
        }
        koff[t + 1] = kh.size();
    }
    if (kh.empty()) { kh.push_back(0); kn.push_back(0); ke.push_back(0); kidf.push_back(0.0f); }
    // 2. local scoring, candidate selection on the device where it applies (the ranked records then stay there)
    fd_count_rec *loc = nullptr;
    uint64_t *loff = nullptr;
    fd_cq_dev_out dev;
    if (!local_rc) local_rc = fd_count_query_batch_impl(c, ix, n_queries, koff.data(), kh.data(), kn.data(), ke.data(), kidf.data(), penalty, top_n, &loc, &loff, true,
                                                        device_path(top_n) ? &dev : nullptr);
    if (!local_rc && dev.got && dev.overflow)     // ties beyond the selection's slots: this rank ranks its full lists instead
        local_rc = fd_count_query_batch_impl(c, ix, n_queries, koff.data(), kh.data(), kn.data(), ke.data(), kidf.data(), penalty, top_n, &loc, &loff, false, nullptr), dev.got = false;
    // 3. + 4. all-gather and global ranking
    rc = exchange(c, m, n_queries, top_n, local_rc, dev, loc, loff, out, out_off);
    fdgpu_free(loc); free(loff);
    return rc;
}

// The same for query maps handed over as fdgpu_make_query_map[_batch] returned them (index = NULL there: a shard's lengths mean nothing):
// ONE all-reduce carries the lengths of the maps' hash[] (scoring idf, count_query.rs:181-200) and of their primary_hash[] (the maps' own
// idf[] — the retrieval's subgraph idf, query.rs:283-288 — is rewritten in place from the global lengths), then as above.  The sharded
// sibling of fdgpu_count_query_maps_top.
extern "C" int fdgpu_sharded_count_query_maps(fdgpu_ctx *c, fdgpu_comm *m, const fdgpu_index *ix, uint64_t n_queries, fd_query_map *const *qms, const float *penalty,
                                              uint64_t total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    if (!c || !m || !ix || !out || !out_off || (n_queries && !qms)) return FDGPU_EINVAL;
    *out = nullptr; *out_off = nullptr;
    for (uint64_t t = 0; t < n_queries; ++t) if (!qms[t]) return FDGPU_EINVAL;
    hipStream_t st = c->stream;
    std::vector<uint32_t> h;
    const uint64_t nq = fd_maps_hashes(n_queries, qms, h);
    std::vector<uint64_t> lens(std::max<uint64_t>(2 * nq, 1), 0);
    uint64_t *dl = nullptr;
    int local_rc = fd_posting_lengths_dev(c, ix, h.data(), 2 * nq, &dl);
    if (local_rc) {     // as above: zeros into the sum, the status into the gather
        if (!dl && 2 * nq * 8 <= m->send.cap) dl = m->send.as<uint64_t>();
        if (!dl || hipMemsetAsync(dl, 0, std::max<uint64_t>(2 * nq, 1) * 8, st) != hipSuccess) return comm_broken(c, m, "sharded_count_query_maps: no buffer for the posting lengths");
    }
    int rc = allreduce_dev(c, m, dl, 2 * nq);
    if (rc) return rc;
    if (nq) HCHK_C(c, m, hipMemcpyAsync(lens.data(), dl, 2 * nq * 8, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    fd_count_rec *loc = nullptr;
    uint64_t *loff = nullptr;
    fd_cq_dev_out dev;
    if (!local_rc) local_rc = fd_count_query_maps_len(c, ix, n_queries, qms, lens.data(), nq ? lens.data() + nq : nullptr, penalty, (float)total_structures, top_n, &loc,
                                                      &loff, device_path(top_n) ? &dev : nullptr);
    if (!local_rc && dev.got && dev.overflow) {   // more ties at a cut than the device selection holds: the compacting path ranks and trims to top_n on this rank
        fd_count_rec *l2 = nullptr; uint64_t *o2 = nullptr;
        local_rc = fd_count_query_maps_len(c, ix, n_queries, qms, lens.data(), nullptr, penalty, (float)total_structures, top_n, &l2, &o2, nullptr, nullptr, nullptr, false);
        loc = l2; loff = o2; dev.got = false;
    }
    rc = exchange(c, m, n_queries, top_n, local_rc, dev, loc, loff, out, out_off);
    fdgpu_free(loc); free(loff);
    return rc;
}

// What every rank runs on the gathered retrieval payloads: rank r's block (stride `bytes`) = its match records, then (at mbytes) its residue
// ints, both in (query, slot, component) order; all_cnt[r][t] = its matches of query t.  Per query the ranks' matches are merged by candidate
// slot (stable: a slot belongs to one rank).  Outputs as fdgpu_retrieve_batch returns them.
static int merge_retrieved(int W, uint64_t n_queries, const uint64_t *all_cnt, const uint8_t *all, size_t bytes, size_t mbytes, const uint64_t *nres_per,
                           fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off) {
    uint64_t tot_m = 0, tot_r = 0;
    for (int r = 0; r < W; ++r)
        for (uint64_t t = 0; t < n_queries; ++t) { tot_m += all_cnt[(size_t)r * (n_queries + 1) + t]; tot_r += all_cnt[(size_t)r * (n_queries + 1) + t] * nres_per[t]; }
    fd_match_rec *om = (fd_match_rec *)malloc(std::max<uint64_t>(tot_m, 1) * sizeof(fd_match_rec));
    int32_t *orr = (int32_t *)malloc(std::max<uint64_t>(tot_r, 1) * 4);
    uint64_t *omo = (uint64_t *)calloc(n_queries + 1, 8), *oro = (uint64_t *)calloc(n_queries + 1, 8);
    if (!om || !orr || !omo || !oro) { free(om); free(orr); free(omo); free(oro); return FDGPU_ENOMEM; }
    std::vector<uint64_t> mbase((size_t)W, 0), rbase((size_t)W, 0);
    struct Ref { uint32_t slot, rank; uint64_t mi, ri; };
    std::vector<Ref> refs;
    uint64_t wm = 0, wr = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        refs.clear();
        for (int r = 0; r < W; ++r) {
            const uint64_t k = all_cnt[(size_t)r * (n_queries + 1) + t];
            const fd_match_rec *src = (const fd_match_rec *)(all + (size_t)r * bytes) + mbase[r];
            for (uint64_t z = 0; z < k; ++z) refs.push_back({src[z].cand, (uint32_t)r, mbase[r] + z, rbase[r] + z * nres_per[t]});
            mbase[r] += k; rbase[r] += k * nres_per[t];
        }
        std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.slot < b.slot; });
        omo[t] = wm; oro[t] = wr;
        for (const Ref &f : refs) {
            om[wm++] = ((const fd_match_rec *)(all + (size_t)f.rank * bytes))[f.mi];
            if (nres_per[t]) memcpy(orr + wr, (const int32_t *)(all + (size_t)f.rank * bytes + mbytes) + f.ri, nres_per[t] * 4);
            wr += nres_per[t];
        }
    }
    omo[n_queries] = wm; oro[n_queries] = wr;
    *matches = om; *match_off = omo; *residues = orr; *res_off = oro;
    return FDGPU_OK;
}
// ---- one on-disk index from N ranks (Option A; fd_shard_index.hip has the transport-free pieces) --------------------------------
// Every rank brings the resident sub-index of its id range; it gets back the resident index of ITS HASH RANGE over ALL structures and where that
// range sits in the database's single index: hashes / value bytes before it and the totals — what fdgpu_index_save_part needs.  Steps: hash bounds of
// about equal posting bytes from rank 0's index (all-gather, rank 0's row used), N slices per rank, an all-gather of the slices' sizes, ONE group of
// ncclSend / ncclRecv — piece j of every rank to rank j, device to device over xGMI: the (hash, id) payload of SURVEY §8e, here already varint-encoded,
// ~1.5 bytes per posting instead of 8 —, fdgpu_index_merge of the N received pieces (their id ranges ascend with the source rank), an all-gather of
// the ranges' sizes.  A rank that fails between the collectives aborts the communicator (comm_broken) so that its peers do not wait for ever.
extern "C" int fdgpu_comm_single_index(fdgpu_ctx *c, fdgpu_comm *m, const fdgpu_index *local, fdgpu_index **range_index, uint64_t *hashes_before, uint64_t *value_before,
                                       uint64_t *total_hashes, uint64_t *total_value) { FD_LOCK(c);
    if (!c || !m || !local || !range_index || !hashes_before || !value_before || !total_hashes || !total_value) return FDGPU_EINVAL;
    *range_index = nullptr;
    if (m->dead) FAIL_(c, FDGPU_EHIP, "communicator aborted by an earlier failure");
    if (!rccl().Send || !rccl().Recv || !rccl().GroupStart || !rccl().GroupEnd) FAIL_(c, FDGPU_EHIP, "this RCCL has no ncclSend / ncclRecv");
    const int W = m->world, me = m->rank;
    if (W > 64) FAIL_(c, FDGPU_ERANGE, "single index: at most 64 ranks (fdgpu_index_merge takes 64 parts)");
    hipStream_t st = c->stream;
    // 1. hash bounds: rank 0's
    std::vector<uint32_t> bnd((size_t)std::max(W - 1, 1), 0xffffffffu), all_bnd((size_t)W * std::max(W - 1, 1));
    int rc = fdgpu_index_range_bounds(c, local, (uint32_t)W, bnd.data());
    const size_t bb = bnd.size() * 4;
    HCHK_C(c, m, m->send.ensure(bb));
    HCHK_C(c, m, m->recv.ensure(bb * W));
    HCHK_C(c, m, hipMemcpyAsync(m->send.p, bnd.data(), bb, hipMemcpyHostToDevice, st));
    int rg = allgather_dev(c, m, m->send.p, m->recv.p, bb);
    if (rg) return rg;
    HCHK_C(c, m, hipMemcpyAsync(all_bnd.data(), m->recv.p, bb * W, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    std::vector<uint64_t> edge((size_t)W + 1, 0);
    for (int j = 1; j < W; ++j) edge[j] = std::max<uint64_t>(edge[j - 1], all_bnd[j - 1]);      // rank 0's row, kept monotone
    edge[W] = 1ull << 32;
    // 2. this rank's slices
    std::vector<fdgpu_index *> sl((size_t)W, nullptr), got((size_t)W, nullptr);
    auto drop = [&]() { for (auto *x : sl) fdgpu_index_destroy(x); for (auto *x : got) fdgpu_index_destroy(x); };
    for (int j = 0; j < W && !rc; ++j) rc = fdgpu_index_slice(c, local, edge[j], edge[j + 1], &sl[j]);
    // 3. sizes: [W][4] {hashes, value bytes, postings, has last ids} of the slices + {first id, structures, status}
    const size_t mw = (size_t)4 * W + 3;
    std::vector<uint64_t> meta(mw, 0), all_meta(mw * W);
    for (int j = 0; j < W && !rc; ++j) { meta[4 * j] = sl[j]->n_hashes; meta[4 * j + 1] = sl[j]->value_len; meta[4 * j + 2] = sl[j]->n_postings; meta[4 * j + 3] = sl[j]->last_ids ? 1 : 0; }
    meta[4 * W] = local->first_id; meta[4 * W + 1] = local->n_structures; meta[4 * W + 2] = rc ? (uint64_t)(uint32_t)(-rc) : 0;
    hipError_t he = m->send.ensure(mw * 8);
    if (he == hipSuccess) he = m->recv.ensure(mw * 8 * W);
    if (he == hipSuccess) he = hipMemcpyAsync(m->send.p, meta.data(), mw * 8, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) { drop(); return comm_broken(c, m, std::string("single index: ") + hipGetErrorString(he)); }
    if ((rg = allgather_dev(c, m, m->send.p, m->recv.p, mw * 8))) { drop(); return rg; }
    he = hipMemcpyAsync(all_meta.data(), m->recv.p, mw * 8 * W, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) { drop(); return comm_broken(c, m, std::string("single index: ") + hipGetErrorString(he)); }
    for (int r = 0; r < W; ++r)
        if (all_meta[mw * r + 4 * W + 2]) {
            if (r != me || !rc) c->err = "single index: rank " + std::to_string(r) + " failed to slice its sub-index";
            drop();
            return rc ? rc : FDGPU_EHIP;
        }
    // 4. the pieces this rank receives: piece r = rank r's slice of this rank's hash range, with rank r's id range
    for (int r = 0; r < W; ++r) {
        const uint64_t *mr = &all_meta[mw * r + 4 * me];
        fdgpu_index *g = new (std::nothrow) fdgpu_index();
        if (!g) { drop(); return comm_broken(c, m, "single index: out of memory"); }
        got[r] = g;
        g->ctx = c; g->n_hashes = mr[0]; g->value_len = mr[1]; g->n_postings = mr[2]; g->first_id = all_meta[mw * r + 4 * W]; g->n_structures = all_meta[mw * r + 4 * W + 1];
        hipError_t e;
        g->hashes = (uint32_t *)c->pool_alloc(std::max<uint64_t>(g->n_hashes, 1) * 4, &e); g->cap_hashes = c->last_cap;
        if (e == hipSuccess) { g->offsets = (uint64_t *)c->pool_alloc((g->n_hashes + 1) * 8, &e); g->cap_offsets = c->last_cap; }
        if (e == hipSuccess) { g->value = (uint8_t *)c->pool_alloc(g->value_len + 16, &e); g->cap_value = c->last_cap; }
        if (e == hipSuccess && mr[3]) { g->last_ids = (uint32_t *)c->pool_alloc(std::max<uint64_t>(g->n_hashes, 1) * 4, &e); g->cap_last = c->last_cap; }
        if (e != hipSuccess) { drop(); return comm_broken(c, m, std::string("single index: ") + hipGetErrorString(e)); }
    }
    ncclResult_t nr = rccl().GroupStart();
    for (int r = 0; r < W && nr == ncclSuccess; ++r) {
        const fdgpu_index *s = sl[r];
        fdgpu_index *g = got[r];
        if (s->n_hashes) {
            nr = rccl().Send(s->hashes, s->n_hashes * 4, ncclUint8, r, m->comm, st);
            if (nr == ncclSuccess && s->last_ids) nr = rccl().Send(s->last_ids, s->n_hashes * 4, ncclUint8, r, m->comm, st);
        }
        if (nr == ncclSuccess) nr = rccl().Send(s->offsets, (s->n_hashes + 1) * 8, ncclUint8, r, m->comm, st);
        if (nr == ncclSuccess && s->value_len) nr = rccl().Send(s->value, s->value_len, ncclUint8, r, m->comm, st);
        if (nr == ncclSuccess && g->n_hashes) {
            nr = rccl().Recv(g->hashes, g->n_hashes * 4, ncclUint8, r, m->comm, st);
            if (nr == ncclSuccess && g->last_ids) nr = rccl().Recv(g->last_ids, g->n_hashes * 4, ncclUint8, r, m->comm, st);
        }
        if (nr == ncclSuccess) nr = rccl().Recv(g->offsets, (g->n_hashes + 1) * 8, ncclUint8, r, m->comm, st);
        if (nr == ncclSuccess && g->value_len) nr = rccl().Recv(g->value, g->value_len, ncclUint8, r, m->comm, st);
        m->n_sendrecv += 2;
    }
    const ncclResult_t ne = rccl().GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr == ncclSuccess) he = hipStreamSynchronize(st);
    if (nr != ncclSuccess || he != hipSuccess) {
        drop();
        return comm_broken(c, m, std::string("single index exchange: ") + (nr != ncclSuccess ? rccl().GetErrorString(nr) : hipGetErrorString(he)));
    }
    for (auto *&x : sl) { fdgpu_index_destroy(x); x = nullptr; }
    // 5. the N pieces of this rank's hash range, concatenated per hash on the device
    fdgpu_index *range = nullptr;
    rc = fdgpu_index_merge(c, (const fdgpu_index *const *)got.data(), (uint64_t)W, &range);
    for (auto *&x : got) { fdgpu_index_destroy(x); x = nullptr; }
    // 6. where the range sits
    uint64_t mine[3] = {range ? range->n_hashes : 0, range ? range->value_len : 0, rc ? (uint64_t)(uint32_t)(-rc) : 0};
    std::vector<uint64_t> all3((size_t)3 * W);
    he = hipMemcpyAsync(m->send.p, mine, 24, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) { fdgpu_index_destroy(range); return comm_broken(c, m, std::string("single index: ") + hipGetErrorString(he)); }
    if ((rg = allgather_dev(c, m, m->send.p, m->recv.p, 24))) { fdgpu_index_destroy(range); return rg; }
    he = hipMemcpyAsync(all3.data(), m->recv.p, (size_t)24 * W, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) { fdgpu_index_destroy(range); return comm_broken(c, m, std::string("single index: ") + hipGetErrorString(he)); }
    uint64_t hb = 0, vb = 0, ht = 0, vt = 0;
    for (int r = 0; r < W; ++r) {
        if (all3[3 * r + 2]) {
            if (r != me || !rc) c->err = "single index: rank " + std::to_string(r) + " failed to merge its hash range";
            fdgpu_index_destroy(range);
            return rc ? rc : FDGPU_EHIP;
        }
        if (r < me) { hb += all3[3 * r]; vb += all3[3 * r + 1]; }
        ht += all3[3 * r]; vt += all3[3 * r + 1];
    }
    *range_index = range; *hashes_before = hb; *value_before = vb; *total_hashes = ht; *total_value = vt;
    return FDGPU_OK;
}

// match records of the local retrieval (device-resident, ordered by (query, local slot, component)) -> the message: the same records with cand =
// the slot in the query's GLOBAL candidate list.  sizeof(fd_match_rec) = 39 words.
__global__ void k_comm_pack_matches(const uint32_t *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ match_off, const uint64_t *__restrict__ lc_off,
                                    const uint32_t *__restrict__ slot_of, uint32_t n_queries, uint32_t *__restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t lo = 0, hi = n_queries;          // the query t with match_off[t] <= k < match_off[t + 1]
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (match_off[mid] <= k) lo = mid; else hi = mid; }
    constexpr uint32_t W = sizeof(fd_match_rec) / 4;
    const uint32_t *src = recs + k * W;
    uint32_t *dst = out + k * W;
    dst[0] = slot_of[lc_off[lo] + src[0]];
    for (uint32_t w = 1; w < W; ++w) dst[w] = src[w];
}
// The unpack + merge of fdgpu_sharded_retrieve alone, on hand-made contributions of `world` ranks (tests drive ragged, empty and failing ranks
// through it on one GPU): counts[r * (n_queries + 1) + t] = rank r's matches of query t, counts[r * (n_queries + 1) + n_queries] = its status
// (0 = fine); rank_matches[r] / rank_residues[r] = its records (cand = slot in the query's GLOBAL candidate list) and residue ints in (query,
// slot, component) order; nres_per[t] = residue ints per match of query t.  A non-zero status of any rank: FDGPU_EHIP, like the real call.
extern "C" int fdgpu_debug_merge_retrieved(fdgpu_ctx *c, uint32_t world, uint64_t n_queries, const uint64_t *counts, const fd_match_rec *const *rank_matches,
                                           const int32_t *const *rank_residues, const uint64_t *nres_per, fd_match_rec **matches, uint64_t **match_off,
                                           int32_t **residues, uint64_t **res_off) { FD_LOCK(c);
    if (!c || !world || !counts || !rank_matches || !rank_residues || (n_queries && !nres_per) || !matches || !match_off || !residues || !res_off) return FDGPU_EINVAL;
    *matches = nullptr; *match_off = nullptr; *residues = nullptr; *res_off = nullptr;
    for (uint32_t r = 0; r < world; ++r)
        if (counts[(size_t)r * (n_queries + 1) + n_queries]) { c->err = "sharded retrieve: rank " + std::to_string(r) + " failed its local step"; return FDGPU_EHIP; }
    uint64_t max_m = 1, max_r = 1;
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t tm = 0, tr = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { tm += counts[(size_t)r * (n_queries + 1) + t]; tr += counts[(size_t)r * (n_queries + 1) + t] * nres_per[t]; }
        max_m = std::max(max_m, tm); max_r = std::max(max_r, tr);
    }
    const size_t mbytes = max_m * sizeof(fd_match_rec), rbytes = ((max_r * 4 + 15) & ~(size_t)15), bytes = mbytes + rbytes;
    std::vector<uint8_t> all(bytes * world, 0);        // the padded payload layout of the all-gather
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t tm = 0, tr = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { tm += counts[(size_t)r * (n_queries + 1) + t]; tr += counts[(size_t)r * (n_queries + 1) + t] * nres_per[t]; }
        if (tm && !rank_matches[r]) return FDGPU_EINVAL;
        if (tr && !rank_residues[r]) return FDGPU_EINVAL;
        if (tm) memcpy(all.data() + (size_t)r * bytes, rank_matches[r], tm * sizeof(fd_match_rec));
        if (tr) memcpy(all.data() + (size_t)r * bytes + mbytes, rank_residues[r], tr * 4);
    }
    return merge_retrieved((int)world, n_queries, counts, all.data(), bytes, mbytes, nres_per, matches, match_off, residues, res_off);
}

// Sharded retrieval: query t's candidates cand_nid[cand_off[t] .. cand_off[t+1]) are GLOBAL structure ids in ranking order (what
// fdgpu_sharded_count_query[_maps] returned, cut to the number of structures to match); this rank holds the coordinates of structures
// first_id .. first_id + n_structures(db) - 1 and matches the candidates inside that range (fdgpu_retrieve_batch on its shard), then the
// ranks all-gather their match records and residue lists (counts first, then one padded payload) and every rank returns what
// fdgpu_retrieve_batch returns for the unsharded database: matches ordered by candidate slot (slot = position in the query's global
// list), components of a candidate in graph.rs:43-45 order.  Replaces the par_iter over candidates of query_pdb.rs:415-452.
extern "C" int fdgpu_sharded_retrieve(fdgpu_ctx *c, fdgpu_comm *m, const fdgpu_batch *db, uint64_t first_id, const uint8_t *resname_std, uint64_t n_queries,
                                      const uint32_t *cand_nid, const uint64_t *cand_off, const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct,
                                      const fd_hash_params *p, float ca_distance_cutoff, uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches,
                                      uint64_t **match_off, int32_t **residues, uint64_t **res_off) { FD_LOCK(c);
    if (!c || !m || !db || !qb || !p || !matches || !match_off || !residues || !res_off || !cand_off || (n_queries && (!qms || !q_struct))) return FDGPU_EINVAL;
    *matches = nullptr; *match_off = nullptr; *residues = nullptr; *res_off = nullptr;
    if (cand_off[n_queries] && !cand_nid) return FDGPU_EINVAL;
    for (uint64_t t = 0; t < n_queries; ++t) if (!qms[t]) return FDGPU_EINVAL;
    hipStream_t st = c->stream;
    const int W = m->world;
    const uint64_t S = db->n_struct;
    // 1. the candidates this rank owns, as local structure indices; slot_of[k] = their slot in the query's global list
    std::vector<uint32_t> lc, slot_of;
    std::vector<uint64_t> lc_off(n_queries + 1, 0);
    for (uint64_t t = 0; t < n_queries; ++t) {
        for (uint64_t k = cand_off[t]; k < cand_off[t + 1]; ++k)
            if (cand_nid[k] >= first_id && cand_nid[k] - first_id < S) { lc.push_back((uint32_t)(cand_nid[k] - first_id)); slot_of.push_back((uint32_t)(k - cand_off[t])); }
        lc_off[t + 1] = lc.size();
    }
    if (lc.empty()) lc.push_back(0);
    // 2. local retrieval; the device glue (motif-sized queries) leaves its ordered records and residue lists in HBM (D.got): they are gathered from
    // there — nothing of the payload touches the host before the gathered result is copied out
    fd_match_rec *lm = nullptr; uint64_t *lmo = nullptr; int32_t *lr = nullptr; uint64_t *lro = nullptr;
    fd_rb_dev_out D;
    int local_rc = fd_retrieve_batch_dev(c, db, resname_std, n_queries, lc.data(), lc_off.data(), qms, qb, q_struct, p, ca_distance_cutoff, node_count, partial_fit, &lm, &lmo,
                                         &lr, &lro, &D);
    struct Rel { fd_match_rec *&a; uint64_t *&b; int32_t *&cc; uint64_t *&d; ~Rel() { fdgpu_matches_free(a, cc); free(b); free(d); } } rel{lm, lmo, lr, lro};
    // 3. counts: per query the number of matches, + status; residue ints per match = 2 * n_indices of the query (same on every rank)
    std::vector<uint64_t> cnt(n_queries + 1, 0), nres_per(n_queries, 0);
    for (uint64_t t = 0; t < n_queries; ++t) nres_per[t] = 2 * qms[t]->n_indices;
    cnt[n_queries] = local_rc ? (uint64_t)(uint32_t)(-local_rc) : 0;
    uint64_t my_m = 0, my_r = 0;
    if (!local_rc)
        for (uint64_t t = 0; t < n_queries; ++t) {
            cnt[t] = lmo[t + 1] - lmo[t];
            if (!D.got) for (uint64_t k = lmo[t]; k < lmo[t + 1]; ++k) lm[k].cand = slot_of[lc_off[t] + lm[k].cand];      // local slot -> slot in the global list
            my_m += cnt[t]; my_r += cnt[t] * nres_per[t];
        }
    const size_t cb = (n_queries + 1) * 8;
    std::vector<uint64_t> all_cnt((size_t)W * (n_queries + 1));
    HCHK_C(c, m, m->send.ensure(cb));
    HCHK_C(c, m, m->recv.ensure(cb * W));
    HCHK_C(c, m, hipMemcpyAsync(m->send.p, cnt.data(), cb, hipMemcpyHostToDevice, st));
    int rc = allgather_dev(c, m, m->send.p, m->recv.p, cb);
    if (rc) return rc;
    HCHK_C(c, m, hipMemcpyAsync(all_cnt.data(), m->recv.p, cb * W, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    for (int r = 0; r < W; ++r)
        if (all_cnt[(size_t)r * (n_queries + 1) + n_queries]) {
            if (r != m->rank || !local_rc) c->err = "sharded retrieve: rank " + std::to_string(r) + " failed its local step (code -" + std::to_string(all_cnt[(size_t)r * (n_queries + 1) + n_queries]) + ")";
            return local_rc ? local_rc : FDGPU_EHIP;
        }
    // 4. payload: [match records | residue ints], each part padded to the largest contribution
    uint64_t max_m = 1, max_r = 1;
    for (int r = 0; r < W; ++r) {
        uint64_t tm = 0, tr = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { tm += all_cnt[(size_t)r * (n_queries + 1) + t]; tr += all_cnt[(size_t)r * (n_queries + 1) + t] * nres_per[t]; }
        max_m = std::max(max_m, tm); max_r = std::max(max_r, tr);
    }
    const size_t mbytes = max_m * sizeof(fd_match_rec), rbytes = ((max_r * 4 + 15) & ~(size_t)15), bytes = mbytes + rbytes;
    HCHK_C(c, m, m->send.ensure(bytes));
    HCHK_C(c, m, m->recv.ensure(bytes * W));
    if (D.got) {
        // device to device: records into the message with their slots rewritten to the query's GLOBAL list (k_comm_pack_matches: one thread per record,
        // its query by bisection in the offsets), residue ints by a plain copy
        if (D.n_recs != my_m || D.n_res != my_r) return comm_broken(c, m, "sharded retrieve: the device-resident result disagrees with its offsets");
        const size_t tb = (slot_of.size() + 1) * 4 + 2 * (n_queries + 1) * 8;
        HCHK_C(c, m, m->aux.ensure(tb));
        std::vector<uint8_t> tabs(tb);
        memcpy(tabs.data(), lmo, (n_queries + 1) * 8);
        memcpy(tabs.data() + (n_queries + 1) * 8, lc_off.data(), (n_queries + 1) * 8);
        if (!slot_of.empty()) memcpy(tabs.data() + 2 * (n_queries + 1) * 8, slot_of.data(), slot_of.size() * 4);
        HCHK_C(c, m, hipMemcpyAsync(m->aux.p, tabs.data(), tb, hipMemcpyHostToDevice, st));
        if (my_m) {
            const uint64_t *d_mo = m->aux.as<uint64_t>(), *d_lo = d_mo + (n_queries + 1);
            hipLaunchKernelGGL(k_comm_pack_matches, dim3((unsigned)((my_m + 255) / 256)), dim3(256), 0, st, (const uint32_t *)D.recs, my_m, d_mo, d_lo,
                               (const uint32_t *)(d_lo + (n_queries + 1)), (uint32_t)n_queries, m->send.as<uint32_t>());
            HCHK_C(c, m, hipGetLastError());
        }
        if (my_r) HCHK_C(c, m, hipMemcpyAsync(m->send.as<uint8_t>() + mbytes, D.residues, my_r * 4, hipMemcpyDeviceToDevice, st));
        HCHK_C(c, m, hipStreamSynchronize(st));      // tabs leaves scope
    } else {      // the host glue ran (whole-structure queries, --partial-fit, a candidate beyond the device glue's limits): its records are host arrays
        if (my_m) HCHK_C(c, m, hipMemcpyAsync(m->send.p, lm, my_m * sizeof(fd_match_rec), hipMemcpyHostToDevice, st));
        if (my_r) HCHK_C(c, m, hipMemcpyAsync(m->send.as<uint8_t>() + mbytes, lr, my_r * 4, hipMemcpyHostToDevice, st));
    }
    if ((rc = allgather_dev(c, m, m->send.p, m->recv.p, bytes))) return rc;
    std::vector<uint8_t> all(bytes * W);
    HCHK_C(c, m, hipMemcpyAsync(all.data(), m->recv.p, bytes * W, hipMemcpyDeviceToHost, st));
    HCHK_C(c, m, hipStreamSynchronize(st));
    // 5. merge per query by candidate slot (a slot belongs to one rank, whose matches arrive in slot / component order)
    fd_match_rec *om = nullptr; int32_t *orr = nullptr; uint64_t *omo = nullptr, *oro = nullptr;
    int mrc = merge_retrieved(W, n_queries, all_cnt.data(), all.data(), bytes, mbytes, nres_per.data(), &om, &omo, &orr, &oro);
    if (mrc) return mrc;
    *matches = om; *match_off = omo; *residues = orr; *res_off = oro;
    return FDGPU_OK;
}
