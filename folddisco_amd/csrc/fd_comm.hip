// fd_comm.hip — the multi-GPU exchange steps of the query path behind the C ABI: one process per GPU, RCCL over xGMI.
//
// The index is sharded by structure id, so scoring is local; two exchanges remain and both live here:
//   fdgpu_allreduce_lengths       posting lengths of the query's hashes summed over the shards — idf = log2(S / len) needs the
//                                 length over the WHOLE database (controller/query.rs:17-32, count_query.rs:130)
//   fdgpu_sharded_count_query     local count_query (top-N preselected on the device) -> ncclAllGather of the candidate records
//                                 (sizes first, then one padded payload, both from / into device memory) -> global ranking
//                                 (idf descending, nid ascending, truncate: query_pdb.rs:404-411) on every rank
// This is what the reference's query workflow would call per batch of queries (cli/workflows/query_pdb.rs:376-452) instead of the
// single-index count_query.  RCCL is bound at run time (dlopen of librccl.so.1 — the copy torch already loaded when there is one), so
// libfdgpu.so itself has no link-time dependency on it; without RCCL the comm entry points fail with FDGPU_EHIP and say so.
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
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.GetErrorString;
        return a;
    }();
    return A;
}
}  // namespace

struct fdgpu_comm {
    fdgpu_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    fd_devbuf send, recv;
};

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
    *out = m;
    return FDGPU_OK;
}
extern "C" void fdgpu_comm_destroy(fdgpu_comm *m) {
    if (!m) return;
    FD_LOCK(m->ctx);
    if (m->comm && rccl().ok) (void)rccl().CommDestroy(m->comm);
    m->send.release(); m->recv.release();
    delete m;
}
extern "C" int fdgpu_comm_rank(const fdgpu_comm *m) { return m ? m->rank : -1; }
extern "C" int fdgpu_comm_world(const fdgpu_comm *m) { return m ? m->world : 0; }

// lengths[k] <- sum over ranks (in place, host array)
extern "C" int fdgpu_allreduce_lengths(fdgpu_ctx *c, fdgpu_comm *m, uint64_t *lengths, uint64_t n) { FD_LOCK(c);
    if (!c || !m || (n && !lengths)) return FDGPU_EINVAL;
    if (!n || m->world == 1) return FDGPU_OK;
    hipStream_t st = c->stream;
    HCHK(c, m->send.ensure(n * 8));
    HCHK(c, hipMemcpyAsync(m->send.p, lengths, n * 8, hipMemcpyHostToDevice, st));
    NCHK(c, rccl().AllReduce(m->send.p, m->send.p, n, ncclUint64, ncclSum, m->comm, st));
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
}  // namespace

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
    // 1. posting lengths over the whole database -> idf per query hash; absent hashes drop out (count_query.rs:121-130)
    std::vector<uint64_t> lens(std::max<uint64_t>(nq, 1));
    int rc = fdgpu_posting_lengths(c, ix, q_hash, nq, lens.data());
    if (rc) return rc;
    if ((rc = fdgpu_allreduce_lengths(c, m, lens.data(), nq))) return rc;
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
    // 2. local scoring with the device-side top-N preselection
    fd_count_rec *loc = nullptr;
    uint64_t *loff = nullptr;
    rc = fdgpu_count_query_batch_top(c, ix, n_queries, koff.data(), kh.data(), kn.data(), ke.data(), kidf.data(), penalty, top_n, &loc, &loff);
    if (rc) return rc;
    // a rank never contributes more than top_n records per query
    std::vector<fd_count_rec> mine;
    std::vector<uint64_t> cnt(n_queries, 0);
    for (uint64_t t = 0; t < n_queries; ++t) {
        fd_count_rec *a = loc + loff[t], *b = loc + loff[t + 1];
        std::sort(a, b, [](const fd_count_rec &x, const fd_count_rec &y) { return rank_key(x) < rank_key(y); });
        const uint64_t keep = top_n ? std::min<uint64_t>(top_n, (uint64_t)(b - a)) : (uint64_t)(b - a);
        mine.insert(mine.end(), a, a + keep);
        cnt[t] = keep;
    }
    free(loc); free(loff);
    const int W = m->world;
    std::vector<uint64_t> all_cnt((size_t)W * n_queries);
    std::vector<fd_count_rec> all;
    uint64_t stride = 0;
    if (W == 1) {
        all_cnt = cnt; all = mine; stride = mine.size();
    } else {
        // 3. all-gather: the per-query counts of every rank, then the records padded to the longest contribution
        hipStream_t st = c->stream;
        HCHK(c, m->send.ensure(std::max<size_t>(n_queries * 8, 8)));
        HCHK(c, m->recv.ensure(std::max<size_t>((size_t)W * n_queries * 8, 8)));
        HCHK(c, hipMemcpyAsync(m->send.p, cnt.data(), n_queries * 8, hipMemcpyHostToDevice, st));
        NCHK(c, rccl().AllGather(m->send.p, m->recv.p, n_queries, ncclUint64, m->comm, st));
        HCHK(c, hipMemcpyAsync(all_cnt.data(), m->recv.p, (size_t)W * n_queries * 8, hipMemcpyDeviceToHost, st));
        HCHK(c, hipStreamSynchronize(st));
        for (int r = 0; r < W; ++r) {
            uint64_t tot = 0;
            for (uint64_t t = 0; t < n_queries; ++t) tot += all_cnt[(size_t)r * n_queries + t];
            stride = std::max(stride, tot);
        }
        stride = std::max<uint64_t>(stride, 1);
        const size_t bytes = stride * sizeof(fd_count_rec);
        HCHK(c, m->send.ensure(bytes));
        HCHK(c, m->recv.ensure(bytes * W));
        if (!mine.empty()) HCHK(c, hipMemcpyAsync(m->send.p, mine.data(), mine.size() * sizeof(fd_count_rec), hipMemcpyHostToDevice, st));
        NCHK(c, rccl().AllGather(m->send.p, m->recv.p, bytes, ncclUint8, m->comm, st));
        all.resize(stride * W);
        HCHK(c, hipMemcpyAsync(all.data(), m->recv.p, bytes * W, hipMemcpyDeviceToHost, st));
        HCHK(c, hipStreamSynchronize(st));
    }
    // 4. global ranking per query
    uint64_t *ooff = (uint64_t *)calloc(n_queries + 1, 8);
    if (!ooff) return FDGPU_ENOMEM;
    std::vector<fd_count_rec> res;
    std::vector<uint64_t> base((size_t)W, 0);
    for (uint64_t t = 0; t < n_queries; ++t) {
        const size_t s0 = res.size();
        for (int r = 0; r < W; ++r) {
            const uint64_t k = all_cnt[(size_t)r * n_queries + t];
            const fd_count_rec *src = all.data() + (size_t)r * stride + base[r];
            res.insert(res.end(), src, src + k);
            base[r] += k;
        }
        std::sort(res.begin() + s0, res.end(), [](const fd_count_rec &x, const fd_count_rec &y) { return rank_key(x) < rank_key(y); });
        if (top_n && res.size() - s0 > top_n) res.resize(s0 + top_n);
        ooff[t + 1] = res.size();
    }
    fd_count_rec *o = (fd_count_rec *)malloc(std::max<size_t>(res.size(), 1) * sizeof(fd_count_rec));
    if (!o) { free(ooff); return FDGPU_ENOMEM; }
    if (!res.empty()) memcpy(o, res.data(), res.size() * sizeof(fd_count_rec));
    *out = o; *out_off = ooff;
    return FDGPU_OK;
}
