// fd_api_count.hip — C ABI of libfdgpu.so, seam S3 (include/fdgpu.h): posting lookup, count_query in its forms (single, batch, query maps, the
// tiled motif path of k_qtile.hip / k_qscore32.hip, whole-structure slices) and the candidate selection — the orchestration of k_query.hip and the
// k_qt* kernels.  Reference: src/index/indextable.rs:53-86,421-463, src/controller/count_query.rs:82-220, src/cli/workflows/query_pdb.rs:404-411.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <thread>
#include "fdgpu_internal.h"
#include "fd_api_common.h"

// ---- S3 ---------------------------------------------------------------------------------------------------------------
// posting lengths of nq query hashes (host array) left ON THE DEVICE in the context's WS_MISC1 (u64 [nq]); the sharded query all-reduces
// them there (fd_comm.hip).  No synchronisation.
int fd_posting_lengths_dev(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t **dev_lengths) {
    hipStream_t st = c->stream;
    HIPCHK(c, c->ws[WS_MISC0].ensure(std::max<uint64_t>(nq, 1) * 4));
    HIPCHK(c, c->ws[WS_MISC1].ensure(std::max<uint64_t>(nq, 1) * 8));
    *dev_lengths = c->ws[WS_MISC1].as<uint64_t>();
    if (!nq) return FDGPU_OK;
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, q_hash, nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, c->ws[WS_CQ_KIDX].ensure(nq * 8));
    HIPCHK(c, c->ws[WS_CQ_NSEG].ensure(nq * 4));
    static const bool lens_cache = [] { const char *e = getenv("FDGPU_LENS_CACHE"); return !(e && e[0] == '0'); }();
    if (lens_cache && ix->n_hashes) {
        // the index remembers the length of every list (4 bytes per hash, one pass over the value bytes on the first request)
        {
            std::lock_guard<std::mutex> lk(ix->lens_mu);
            if (!ix->lens) {
                uint32_t *l = nullptr;
                HIPCHK(c, hipMalloc((void **)&l, ix->n_hashes * 4));
                fd_launch_index_lens(ix->offsets, ix->value, ix->n_hashes, l, st);
                hipError_t le = hipGetLastError();
                if (le == hipSuccess) le = hipStreamSynchronize(st);      // other contexts read it from their own streams
                if (le != hipSuccess) { (void)hipFree(l); c->err = std::string("posting lengths of the index: ") + hipGetErrorString(le); return FDGPU_EHIP; }
                ix->lens = l;
            }
        }
        fd_launch_posting_lookup(ix->hashes, ix->offsets, ix->lens, ix->n_hashes, c->ws[WS_MISC0].as<uint32_t>(), nq, c->ws[WS_MISC1].as<uint64_t>(),
                                 c->ws[WS_CQ_NSEG].as<uint32_t>(), c->ws[WS_CQ_KIDX].as<long long>(), st);
        HIPCHK(c, hipGetLastError());
        return FDGPU_OK;
    }
    HIPCHK(c, c->ws[WS_CQ_WSTART].ensure((nq + 2) * 8));
    HIPCHK(c, c->ws[WS_SCANTMP].ensure(fd_scan_tmp_elems(nq) * 8 + 64));
    HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
    fd_launch_posting_lengths(ix->hashes, ix->offsets, ix->value, ix->n_hashes, c->ws[WS_MISC0].as<uint32_t>(), nq, c->ws[WS_MISC1].as<uint64_t>(),
                              c->ws[WS_CQ_KIDX].as<long long>(), c->ws[WS_CQ_NSEG].as<uint32_t>(), c->ws[WS_CQ_WSTART].as<uint64_t>(),
                              c->ws[WS_SCANTMP].as<uint64_t>(), c->ws[WS_TOTAL].as<uint64_t>(), st);
    HIPCHK(c, hipGetLastError());
    return FDGPU_OK;
}
// lengths and the number of CQ_SEG-byte scoring segments of every hash (what k_cq_plan will find again), one synchronisation
// land_out != null (with kidx): when the page-locked block can be had the arrays STAY there — *land_out = [lengths u64 x nq | kidx i64 x nq | segs u32 x nq],
// valid until the context's block 3 is used again; lengths / segs / kidx are then untouched (reading the block twice — once to copy it out, once to use the
// copy — was 0.3 ms of a whole-structure query's 178 k lookups)
int fd_posting_lengths_segs(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t *lengths, uint32_t *segs, long long *kidx,
                            const uint8_t **land_out) {
    if (land_out) *land_out = nullptr;
    if (!nq) return FDGPU_OK;
    uint64_t *d = nullptr;
    int rc = fd_posting_lengths_dev(c, ix, q_hash, nq, &d);
    if (rc) return rc;
    // the three arrays land in one page-locked block (a pageable destination makes every copy a staged, blocking one)
    uint8_t *land = (uint8_t *)c->host_pinned(3, nq * 20);
    if (!land) {
        HIPCHK(c, hipMemcpyAsync(lengths, d, nq * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(segs, c->ws[WS_CQ_NSEG].p, nq * 4, hipMemcpyDeviceToHost, c->stream));
        if (kidx) HIPCHK(c, hipMemcpyAsync(kidx, c->ws[WS_CQ_KIDX].p, nq * 8, hipMemcpyDeviceToHost, c->stream));     // both length paths leave the list positions there
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return FDGPU_OK;
    }
    HIPCHK(c, hipMemcpyAsync(land, d, nq * 8, hipMemcpyDeviceToHost, c->stream));
    if (kidx) HIPCHK(c, hipMemcpyAsync(land + nq * 8, c->ws[WS_CQ_KIDX].p, nq * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(land + nq * 16, c->ws[WS_CQ_NSEG].p, nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (land_out && kidx) { *land_out = land; return FDGPU_OK; }
    memcpy(lengths, land, nq * 8);
    if (kidx) memcpy(kidx, land + nq * 8, nq * 8);
    memcpy(segs, land + nq * 16, nq * 4);
    return FDGPU_OK;
}
extern "C" int fdgpu_posting_lengths(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t *lengths) { FD_LOCK(c);
    if (!c || !ix || (nq && (!q_hash || !lengths))) return FDGPU_EINVAL;
    if (!nq) return FDGPU_OK;
    uint64_t *d = nullptr;
    int rc = fd_posting_lengths_dev(c, ix, q_hash, nq, &d);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(lengths, d, nq * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDGPU_OK;
}

void fd_launch_posting_bytes(const uint32_t *hashes, const uint64_t *offsets, uint64_t H, const uint32_t *q_hash, uint64_t nq, uint64_t *bytes, hipStream_t st);
extern "C" int fdgpu_posting_bytes(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t *bytes) { FD_LOCK(c);
    if (!c || !ix || (nq && (!q_hash || !bytes))) return FDGPU_EINVAL;
    if (!nq) return FDGPU_OK;
    hipStream_t st = c->stream;
    HIPCHK(c, c->ws[WS_MISC0].ensure(nq * 4));
    HIPCHK(c, c->ws[WS_MISC1].ensure(nq * 8));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, q_hash, nq * 4, hipMemcpyHostToDevice, st));
    fd_launch_posting_bytes(ix->hashes, ix->offsets, ix->n_hashes, c->ws[WS_MISC0].as<uint32_t>(), nq, c->ws[WS_MISC1].as<uint64_t>(), st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bytes, c->ws[WS_MISC1].p, nq * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

// get_entries for many hashes: ids of hash k = (*ids)[(*ids_off)[k] .. (*ids_off)[k+1])
extern "C" int fdgpu_get_entries(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint32_t **ids, uint64_t **ids_off) { FD_LOCK(c);
    if (!c || !ix || !ids || !ids_off || (nq && !q_hash)) return FDGPU_EINVAL;
    *ids = nullptr; *ids_off = nullptr;
    uint64_t *off = (uint64_t *)calloc(nq + 1, 8);
    if (!off) return FDGPU_ENOMEM;
    std::vector<uint64_t> lens(std::max<uint64_t>(nq, 1));
    int rc = fdgpu_posting_lengths(c, ix, q_hash, nq, lens.data());
    if (rc) { free(off); return rc; }
    for (uint64_t k = 0; k < nq; ++k) off[k + 1] = off[k] + lens[k];
    const uint64_t tot = off[nq];
    uint32_t *out = (uint32_t *)malloc(std::max<uint64_t>(tot, 1) * 4);
    if (!out) { free(off); return FDGPU_ENOMEM; }
    if (tot) {
        hipStream_t st = c->stream;
        hipError_t e = c->ws[WS_MISC0].ensure(nq * 4);
        if (e == hipSuccess) e = c->ws[WS_MISC1].ensure((nq + 1) * 8);
        if (e == hipSuccess) e = c->ws[WS_MISC2].ensure(tot * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws[WS_MISC0].p, q_hash, nq * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws[WS_MISC1].p, off, (nq + 1) * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            fd_launch_get_entries(ix->hashes, ix->offsets, ix->value, ix->n_hashes, c->ws[WS_MISC0].as<uint32_t>(), nq, c->ws[WS_MISC1].as<uint64_t>(),
                                  c->ws[WS_MISC2].as<uint32_t>(), st);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, c->ws[WS_MISC2].p, tot * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { free(off); free(out); c->err = std::string("get_entries: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    }
    *ids = out; *ids_off = off;
    return FDGPU_OK;
}

// plan (list positions, CQ_SEG-byte segments) + segment-parallel scoring of the query hashes in A (k_query.hip)
// idf of a query hash in the accumulators' fixed point (2^-22; count_query.rs:181-200 sums f32 in hash-map order, here the sum is exact
// and order-independent to 2.4e-7 per addend)
static inline uint64_t fd_idf_fix(float idf) {
    const double v = (double)idf;
    return (v > 0.0 && v < 1.0e6) ? (uint64_t)(v * 4194304.0 + 0.5) : 0ull;
}
// The rows of one query in (node, partner) order with their metadata word: idf (2^-22 fixed point) << 2 | last row of its node << 1 |
// last row of its edge.  -> false when an idf does not fit the packed accumulator (>= 32).
static bool fd_cq_rows(const uint32_t *q_hash, const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf, uint64_t a, uint64_t b,
                       std::vector<uint32_t> &rows_hash, std::vector<unsigned long long> &rows_meta, const long long *q_kidx = nullptr,
                       std::vector<long long> *rows_kidx = nullptr) {
    std::vector<uint64_t> ord(b - a);
    for (uint64_t k = a; k < b; ++k) ord[k - a] = k;
    bool small_ids = true, in_order = true;
    for (uint64_t k = a; k < b && small_ids; ++k) small_ids = q_node[k] < 65536u && q_edge_j[k] < 65536u;
    // a query map lists its entries pair by pair in row-major (i, j) order (query.rs:231-329): a whole-structure query's 10^5 rows arrive sorted
    for (uint64_t k = a + 1; k < b && in_order; ++k) in_order = q_node[k - 1] < q_node[k] || (q_node[k - 1] == q_node[k] && q_edge_j[k - 1] <= q_edge_j[k]);
    if (in_order) {
    } else if (b - a > 2048 && small_ids) {
        // a whole-structure query has ~10^5 rows: (node, partner) order by a stable LSD radix sort of node << 16 | partner (a comparison
        // sort of the index array took several milliseconds of the prefilter)
        std::vector<uint64_t> tmp(ord.size());
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 8 * pass;
            size_t cnt[257] = {0};
            auto key = [&](uint64_t k) { return ((q_node[k] << 16) | q_edge_j[k]) >> sh & 255u; };
            for (uint64_t k : ord) ++cnt[key(k) + 1];
            for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
            for (uint64_t k : ord) tmp[cnt[key(k)]++] = k;
            ord.swap(tmp);
        }
    } else
    std::stable_sort(ord.begin(), ord.end(), [&](uint64_t x, uint64_t y) {
        return q_node[x] != q_node[y] ? q_node[x] < q_node[y] : q_edge_j[x] < q_edge_j[y];
    });
    bool fits = true;
    const size_t n = ord.size(), base = rows_hash.size();
    rows_hash.resize(base + n); rows_meta.resize(base + n);
    const bool with_k = q_kidx && rows_kidx;
    if (with_k) rows_kidx->resize(base + n);
    uint32_t *const oh = rows_hash.data() + base;
    unsigned long long *const om = rows_meta.data() + base;
    long long *const ok = with_k ? rows_kidx->data() + base : nullptr;
    const uint64_t *const od = ord.data();
    for (size_t z = 0; z < n; ++z) {
        const uint64_t k = od[z];
        const uint64_t fix = fd_idf_fix(q_idf[k]);
        fits = fits && fix < (1ull << 27);
        const bool last = z + 1 == n;
        const bool node_end = last || q_node[od[z + 1]] != q_node[k];
        const bool edge_end = node_end || q_edge_j[od[z + 1]] != q_edge_j[k];
        oh[z] = q_hash[k];
        if (ok) ok[z] = q_kidx[k];
        om[z] = ((unsigned long long)fix << 2) | (node_end ? 2ull : 0ull) | (edge_end ? 1ull : 0ull);
    }
    return fits;
}
static int cq_score(fdgpu_ctx *c, const cq_args &A, int64_t known_segments = -1) {
    hipStream_t st = c->stream;
    HIPCHK(c, c->ws[WS_CQ_KIDX].ensure(A.nq * 8));
    HIPCHK(c, c->ws[WS_CQ_NSEG].ensure(A.nq * 4));
    HIPCHK(c, c->ws[WS_CQ_WSTART].ensure((A.nq + 2) * 8));
    HIPCHK(c, c->ws[WS_SCANTMP].ensure(fd_scan_tmp_elems(A.nq) * 8 + 64));
    HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
    fd_launch_cq_plan(A, c->ws[WS_CQ_KIDX].as<long long>(), c->ws[WS_CQ_NSEG].as<uint32_t>(), st);
    fd_exclusive_scan<uint32_t>(c->ws[WS_CQ_NSEG].as<uint32_t>(), A.nq, c->ws[WS_CQ_WSTART].as<uint64_t>(), c->ws[WS_SCANTMP].as<uint64_t>(),
                                c->ws[WS_TOTAL].as<uint64_t>(), st);
    HIPCHK(c, hipGetLastError());
    uint64_t W = (uint64_t)known_segments;      // the caller knows the work count (query maps remember their hashes' segments): no round trip
    if (known_segments < 0) {
        int rc = d2h_u64(c, c->ws[WS_TOTAL].as<uint64_t>(), &W);
        if (rc) return rc;
    }
    HIPCHK(c, c->ws[WS_CQ_SEGSUM].ensure(std::max<uint64_t>(W, 1) * 4));
    fd_launch_cq_seg(A, c->ws[WS_CQ_KIDX].as<long long>(), c->ws[WS_CQ_WSTART].as<uint64_t>(), c->ws[WS_CQ_SEGSUM].as<uint32_t>(), W, W > 0, st);
    return FDGPU_OK;
}

extern "C" int fdgpu_count_query(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, const uint32_t *q_node, const uint32_t *q_edge_j,
                                 const float *q_idf, uint64_t nq, const float *penalty, fd_count_rec **out, uint64_t *n_out) { FD_LOCK(c);
    if (!c || !ix || !out || !n_out || (nq && (!q_hash || !q_node || !q_edge_j || !q_idf)) || (ix->n_structures && !penalty && !ix->penalty)) return FDGPU_EINVAL;
    *out = nullptr; *n_out = 0;
    reset_timings(c);
    hipStream_t st = c->stream;
    const uint64_t S = ix->n_structures;
    if (S == 0 || nq == 0) { *out = (fd_count_rec *)malloc(sizeof(fd_count_rec)); return *out ? FDGPU_OK : FDGPU_ENOMEM; }
    if (S >= 0xffffffe0ull) FAIL(c, FDGPU_ERANGE, "too many structures");
    std::vector<uint32_t> rows_hash;
    std::vector<unsigned long long> rows_meta;
    rows_hash.reserve(nq); rows_meta.reserve(nq);
    const bool packed = fd_cq_rows(q_hash, q_node, q_edge_j, q_idf, 0, nq, rows_hash, rows_meta) && nq < (1ull << 18);
    const uint32_t words = (uint32_t)((S + 31) / 32);
    // workspace
    HIPCHK(c, c->ws[WS_MISC0].ensure(nq * 4));   // query hashes in row order
    HIPCHK(c, c->ws[WS_MISC3].ensure(nq * 8));   // row metadata
    HIPCHK(c, c->ws[WS_COUNTS].ensure(S * 4));   // match counts (wide form)
    HIPCHK(c, c->ws[WS_SEGOFF].ensure((S + 2) * 8));  // (count, idf sum)
    HIPCHK(c, c->ws[WS_KEYS_B].ensure((size_t)nq * words * 4));   // occupancy rows
    HIPCHK(c, c->ws[WS_IDS_A].ensure(S * 4));    // node counts
    HIPCHK(c, c->ws[WS_IDS_B].ensure(S * 4));    // edge counts
    HIPCHK(c, c->ws[WS_MISC4].ensure(S + 8));    // flags
    HIPCHK(c, c->ws[WS_TILE_BO].ensure((S + 2) * 8));  // positions
    HIPCHK(c, c->ws[WS_MISC5].ensure(S * 4));    // penalty
    HIPCHK(c, c->ws[WS_SCANTMP].ensure(fd_scan_tmp_elems(S) * 8 + 64));
    HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, rows_hash.data(), nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC3].p, rows_meta.data(), nq * 8, hipMemcpyHostToDevice, st));
    if (penalty) HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC5].p, penalty, S * 4, hipMemcpyHostToDevice, st));
    const float *d_penalty = penalty ? c->ws[WS_MISC5].as<float>() : ix->penalty;
    cq_args A;
    A.hashes = ix->hashes; A.offsets = ix->offsets; A.value = ix->value; A.H = ix->n_hashes;
    A.q_hash = c->ws[WS_MISC0].as<uint32_t>(); A.nq = nq;
    A.hash_bits = c->ws[WS_KEYS_B].as<uint32_t>(); A.row_meta = c->ws[WS_MISC3].as<unsigned long long>();
    A.match = c->ws[WS_COUNTS].as<uint32_t>(); A.idf = c->ws[WS_SEGOFF].as<unsigned long long>(); A.packed = packed ? 1 : 0;
    A.words = words; A.first_id = (uint32_t)ix->first_id; A.S = (uint32_t)S;
    {
        StageTimer t(c, "cq_accumulate", 0);
        int rs = cq_score(c, A);
        if (rs) return rs;
    }
    std::vector<uint64_t> slices;        // outlives the asynchronous copy below (the stream is synchronised before this function returns)
    {
        StageTimer t(c, "cq_finalize", (uint64_t)nq * words * 4 + S * 16);
        // thousands of rows (whole-structure queries): S / 4096 workgroups of word columns do not fill the chip — cut the rows into ~32
        // slices at node boundaries
        if (nq >= 4096) {
            const uint64_t per = (nq + 31) / 32;
            slices.push_back(0);
            for (uint64_t r = 0; r + 1 < nq; ++r)
                if ((rows_meta[r] & 2ull) && r + 1 - slices.back() >= per) slices.push_back(r + 1);
            slices.push_back(nq);
            HIPCHK(c, c->ws[WS_TILE_B].ensure(slices.size() * 8));
            HIPCHK(c, hipMemcpyAsync(c->ws[WS_TILE_B].p, slices.data(), slices.size() * 8, hipMemcpyHostToDevice, st));
        }
        fd_launch_cq_rows_finalize(A, nullptr, 1, slices.empty() ? nullptr : c->ws[WS_TILE_B].as<uint64_t>(), slices.empty() ? 0u : (uint32_t)slices.size() - 1,
                                   c->ws[WS_IDS_A].as<uint32_t>(), c->ws[WS_IDS_B].as<uint32_t>(), c->ws[WS_MISC4].as<uint8_t>(), nq, st);
        fd_exclusive_scan<uint8_t>(c->ws[WS_MISC4].as<uint8_t>(), S, c->ws[WS_TILE_BO].as<uint64_t>(), c->ws[WS_SCANTMP].as<uint64_t>(),
                                   c->ws[WS_TOTAL].as<uint64_t>(), st);
    }
    HIPCHK(c, hipGetLastError());
    uint64_t n = 0;
    int rc = d2h_u64(c, c->ws[WS_TOTAL].as<uint64_t>(), &n);
    if (rc) return rc;
    fd_count_rec *r = (fd_count_rec *)malloc(std::max<uint64_t>(n, 1) * sizeof(fd_count_rec));
    if (!r) return FDGPU_ENOMEM;
    HIPCHK(c, c->ws[WS_TILE_HO].ensure(std::max<uint64_t>(n, 1) * sizeof(fd_count_rec)));
    fd_launch_cq_compact(packed ? nullptr : A.match, A.idf, c->ws[WS_IDS_A].as<uint32_t>(), c->ws[WS_IDS_B].as<uint32_t>(), c->ws[WS_MISC4].as<uint8_t>(),
                         c->ws[WS_TILE_BO].as<uint64_t>(), d_penalty, (uint32_t)S, (uint32_t)ix->first_id, c->ws[WS_TILE_HO].p, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n) e = hipMemcpyAsync(r, c->ws[WS_TILE_HO].p, n * sizeof(fd_count_rec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(r); c->err = std::string("count_query: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    *out = r; *n_out = n;
    return FDGPU_OK;
}

// The index's checkpoint table for tiled scoring (k_qtile.hip), made on first use for the id range the index has then: sizes per list ->
// exclusive scan -> one sequential decode of every list long enough to hold entries.  -> FDGPU_OK with the table published, FDGPU_ENOMEM
// when it does not fit (remembered: the caller keeps the occupancy-row path for this index).
static int fd_index_checkpoints(fdgpu_ctx *c, const fdgpu_index *ix) {
    std::lock_guard<std::mutex> lk(ix->lens_mu);
    if (ix->ck_meta && ix->ck_first == ix->first_id && ix->ck_S == ix->n_structures) return FDGPU_OK;
    // a failure is remembered for the id range it happened with (a changed range is a new table of another size) and retried every 64th request:
    // one transient hipMalloc failure must not switch the tiled path off for the life of the index
    if (ix->ck_failed && ix->ck_first == ix->first_id && ix->ck_S == ix->n_structures && (++ix->ck_fail_skips & 63)) return FDGPU_ENOMEM;
    hipStream_t st = c->stream;
    const uint64_t H = ix->n_hashes, S = ix->n_structures;
    if (!H || !S) return FDGPU_ENOMEM;
    if (ix->ck_meta) {
        // the table of the previous id range: other contexts that share the index (query lanes, one context per host thread) may still have
        // k_qt_plan / k_qt_score in flight on THEIR streams reading it — drain the whole device, not only this context's stream, before the free
        (void)hipDeviceSynchronize();
        (void)hipFree(ix->ck_meta); (void)hipFree(ix->ck_ent); ix->ck_meta = nullptr; ix->ck_ent = nullptr;
    }
    ix->ck_failed = false;
    const uint32_t NC = (uint32_t)((S + (1u << QT_CELL_LOG2) - 1) >> QT_CELL_LOG2);
    hipError_t e = c->ws[WS_QT_COUNT].ensure(H * 4);
    if (e == hipSuccess) e = c->ws[WS_QT_RANGES].ensure((H + 2) * 8);
    if (e == hipSuccess) e = c->ws[WS_SCANTMP].ensure(fd_scan_tmp_elems(H) * 8 + 64);
    if (e == hipSuccess) e = c->ws[WS_TOTAL].ensure(64);
    unsigned long long *meta = nullptr;
    void *ent = nullptr;
    uint64_t n_ent = 0;
    if (e == hipSuccess) {
        fd_launch_ck_count(ix->offsets, H, NC, c->ws[WS_QT_COUNT].as<uint32_t>(), st);
        fd_exclusive_scan<uint32_t>(c->ws[WS_QT_COUNT].as<uint32_t>(), H, c->ws[WS_QT_RANGES].as<uint64_t>(), c->ws[WS_SCANTMP].as<uint64_t>(),
                                    c->ws[WS_TOTAL].as<uint64_t>(), st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&n_ent, c->ws[WS_TOTAL].p, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMalloc((void **)&meta, H * 8);
    if (e == hipSuccess) e = hipMalloc(&ent, std::max<uint64_t>(n_ent, 1) * 8);
    if (e == hipSuccess) {
        fd_launch_ck_fill(ix->offsets, ix->value, H, NC, (uint32_t)S, (uint32_t)ix->first_id, c->ws[WS_QT_RANGES].as<uint64_t>(), meta, ent, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // other contexts read the table from their own streams
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (meta) (void)hipFree(meta);
        if (ent) (void)hipFree(ent);
        ix->ck_failed = true; ix->ck_first = ix->first_id; ix->ck_S = S;
        return FDGPU_ENOMEM;
    }
    ix->ck_meta = meta; ix->ck_ent = ent; ix->ck_n = n_ent; ix->ck_first = ix->first_id; ix->ck_S = S;
    return FDGPU_OK;
}

// batched count_query: queries [q_off[t], q_off[t+1]) of the concatenated hash arrays; results of query t are
// (*out)[(*out_off)[t] .. (*out_off)[t+1])
// idf descending, ties by ascending structure id (the candidate ranking of query_pdb.rs:404-411), cut to top_n; -> records kept
static uint64_t fd_rank_trim(fd_count_rec *r, uint64_t n, uint32_t top_n) {
    auto key = [](const fd_count_rec &x) {
        float v = x.idf + 0.0f;
        uint32_t b; memcpy(&b, &v, 4);
        const uint32_t o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        return ((uint64_t)(~o) << 32) | x.nid;
    };
    std::sort(r, r + n, [&](const fd_count_rec &a, const fd_count_rec &b) { return key(a) < key(b); });
    return std::min<uint64_t>(n, top_n);
}
// dev != null: a call whose candidate selection runs on the device (dense_topn below) leaves its result THERE — dev->recs = [n_queries][top_n]
// ranked records, dev->state = the per-query selection state (count = records selected) — and returns without host records (dev->got); the
// sharded query all-gathers those buffers (fd_comm.hip).  dev->overflow: more ties at a cut-off than the selection holds (the caller
// takes the compacting path together with the other ranks).  Calls the device selection does not serve return host records as usual.
int fd_count_query_batch_impl(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off, const uint32_t *q_hash,
                              const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf, const float *penalty, uint32_t top_n,
                              fd_count_rec **out, uint64_t **out_off, bool allow_dense, fd_cq_dev_out *dev, int64_t known_segments, const long long *known_kidx,
                              const uint64_t *known_len) {
    if (!c || !ix || !out || !out_off || !q_off || (ix->n_structures && !penalty && !ix->penalty)) return FDGPU_EINVAL;
    *out = nullptr; *out_off = nullptr;
    reset_timings(c);
    hipStream_t st = c->stream;
    const uint64_t S = ix->n_structures, nq = q_off[n_queries];
    const bool cq_trace = getenv("FDGPU_TRACE") != nullptr;       // host-side stage stamps on stderr (measurement aid)
    const auto cq_t0 = std::chrono::steady_clock::now();
    auto cq_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - cq_t0).count(); };
    uint64_t *ooff = (uint64_t *)calloc(n_queries + 1, 8);
    if (!ooff) return FDGPU_ENOMEM;
    if (dev) { dev->got = false; dev->overflow = false; }
    if (S == 0 || nq == 0 || n_queries == 0) { *out = (fd_count_rec *)malloc(sizeof(fd_count_rec)); *out_off = ooff; return *out ? FDGPU_OK : FDGPU_ENOMEM; }
    if (!q_hash || !q_node || !q_edge_j || !q_idf) { free(ooff); return FDGPU_EINVAL; }
    if (S >= 0xffffffe0ull || n_queries * S >= (1ull << 34)) { free(ooff); FAIL(c, FDGPU_ERANGE, "count_query_batch: n_queries x n_structures too large; split the batch"); }
    // occupancy rows: the query hashes of the whole batch, per query in (node, partner) order
    std::vector<uint32_t> rows_hash;
    std::vector<unsigned long long> rows_meta;
    std::vector<long long> rows_kidx;       // the rows' list positions when the caller knows them (query maps made against this index)
    (void)known_len;                        // (lengths are per input row: only their sum matters below, no permutation needed)
    rows_hash.reserve(nq); rows_meta.reserve(nq);
    if (known_kidx) rows_kidx.reserve(nq);
    bool packed = true, sums_fit32 = true;
    uint64_t max_rows = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        max_rows = std::max<uint64_t>(max_rows, q_off[t + 1] - q_off[t]);
        packed = packed && (q_off[t + 1] - q_off[t]) < (1ull << 18);
        packed = fd_cq_rows(q_hash, q_node, q_edge_j, q_idf, q_off[t], q_off[t + 1], rows_hash, rows_meta, known_kidx, &rows_kidx) && packed;
        // 32-bit accumulators serve the batch when no row adds nothing (touched <=> sum != 0) and no query's idf units can reach 2^32
        unsigned long long units = 0;
        for (uint64_t r = q_off[t]; r < q_off[t + 1]; ++r) { const unsigned long long fix = rows_meta[r] >> 2; units += fix; sums_fit32 = sums_fit32 && fix != 0ull; }
        sums_fit32 = sums_fit32 && units < (1ull << 32);
    }
    if (cq_trace) fprintf(stderr, "[count_query] %llu rows of %llu queries in order at %.3f ms\n", (unsigned long long)nq, (unsigned long long)n_queries, cq_ms());
    const uint32_t words = (uint32_t)((S + 31) / 32);
    const uint64_t QS = n_queries * S;
    const bool dense_topn = allow_dense && packed && top_n > 0 && top_n + 1024 <= 4096;
    // one query with thousands of rows (whole-structure mode): its rows are walked in slices that add into the dense results; motif
    // queries with a selection skip the dense results altogether: scores per tile of structures in LDS (k_qtile.hip), or, when the
    // index has no checkpoint table, ranking keys from occupancy rows (k_cq_rows_keys); records only for the survivors either way
    const bool sliced = n_queries == 1 && nq >= 4096;
    const bool keys_only = dense_topn && !sliced;
    const bool qtile_on = [] { const char *e = getenv("FDGPU_QTILE"); return !(e && e[0] == '0'); }();      // 0: occupancy rows (read per call: tests compare the two)
    // k_qscore32.hip (32-bit sums, planned slot stream; needs the rows' posting lengths for the stream's bound): FDGPU_QT32=0 keeps the 64-bit kernel
    // (read per call: tests compare the two), FDGPU_QT32=15 takes tiles of 2^15 structures (one workgroup per CU) instead of 2^14 (two per CU)
    const int qt32_env = [] { const char *e = getenv("FDGPU_QT32"); return e ? atoi(e) : 14; }();
    // (queries of more than 128 rows keep the 64-bit kernel as well: their idf units reach 2^32 on any real index, and k_qt_rows' small tables are sized for 128)
    bool qt32 = keys_only && qtile_on && qt32_env != 0 && sums_fit32 && known_len && max_rows <= 128;
    const uint32_t qt_tl2 = qt32 ? (qt32_env == 15 ? 15u : qt32_env == 13 ? 13u : 14u)
                                 : [] { const char *e = getenv("FDGPU_QT_TILE"); return e && atoi(e) == 13 ? 13u : 14u; }();      // structures per tile (measurement)
    const uint32_t NT = (uint32_t)((S + (1u << qt_tl2) - 1) >> qt_tl2);
    bool tiled = keys_only && qtile_on && max_rows <= QT_MAX_ROWS && nq * (S >> QT_CELL_LOG2) < (1ull << 31) && fd_index_checkpoints(c, ix) == FDGPU_OK;
    qt32 = qt32 && tiled;
    const uint32_t qt_wpr = (1u << (qt_tl2 - 8)) + 2u;      // windows a row can need in a tile: < 2 bytes per posting of a tile, windows at least half full, + its pieces' tails
    // one query of thousands of rows (a whole structure as the query) with a selection: the same tiles, the rows cut into slices (k_qt_score<BIG>)
    const uint32_t NT14 = (uint32_t)((S + (1u << 14) - 1) >> 14);
    bool tiled_big = dense_topn && sliced && qtile_on && !tiled && nq < (1ull << 18) && nq * NT14 < (1ull << 31) && fd_index_checkpoints(c, ix) == FDGPU_OK;
    const uint32_t big_slices = (uint32_t)std::min<uint64_t>(32, nq), big_wpr = (uint32_t)((nq + 31) / 32), big_cap = top_n + 1024;
    hipError_t e = hipSuccess;
    auto need = [&](int w, size_t bytes) { if (e == hipSuccess) e = c->ws[w].ensure(bytes); };
    need(WS_MISC0, nq * 4); need(WS_MISC3, nq * 8); need(WS_TILE_H, (n_queries + 1) * 8); need(WS_MISC5, S * 4); need(WS_TOTAL, 64);
    // the decoded stream of the tiled path (k_qt_rows): sized from the rows' posting lengths when the caller knows them — a list of n ids is at most
    // n x (bytes of the largest id) bytes, a slot holds 16 of them, and every (row, cell) piece ends in one partly filled slot
    uint64_t stream_cap = 0;
    const bool qt_stream = [] { const char *e = getenv("FDGPU_QT_STREAM"); return !(e && e[0] == '0'); }();      // 0: pass B decodes the lists again (tests, measurement)
    if (tiled && (qt32 || (qt_stream && known_len && max_rows * (1u << (qt_tl2 - QT_CELL_LOG2)) <= (uint64_t)QT_MAXB * (qt_tl2 == 14 ? 512 : 256)))) {
        const uint64_t top_id = ix->first_id + S, vb = top_id < (1ull << 7) ? 1 : top_id < (1ull << 14) ? 2 : top_id < (1ull << 21) ? 3 : top_id < (1ull << 28) ? 4 : 5;
        const uint64_t NCc = (S + (1u << QT_CELL_LOG2) - 1) >> QT_CELL_LOG2;
        // a piece per (row, cell) where the list has an entry per cell; a list with entries 2^j cells apart is cut into pieces of < 96 bytes on
        // average that every tile they span decodes once: at most 6 slots x tiles on top of its bytes
        uint64_t slots = 0;
        for (uint64_t r = 0; r < nq; ++r) slots += (known_len[r] * vb + 15) / 16 + NCc + 6ull * NT + 8;
        // (the planned stream pads its windows: a piece that would straddle a 64-slot boundary starts the next window — windows stay at least half full)
        if (qt32) slots = 2 * slots + 64ull * n_queries * NT;
        if (slots < (1ull << 31)) stream_cap = slots + 1024;
        else qt32 = false;
        // FDGPU_QT_STREAM_CAP=n (tests): a stream of n records — the tiles that do not fit raise the selection's overflow flag and the call is ranked by the compacting path
        if (const char *e = getenv("FDGPU_QT_STREAM_CAP")) { const long v = atol(e); if (v > 0 && (uint64_t)v < stream_cap) stream_cap = (uint64_t)v; }
    }
    auto need_rows = [&]() {      // the occupancy-row path's scratch
        need(WS_COUNTS, packed ? 64 : QS * 4); need(WS_SEGOFF, QS * 8 + 16);
        need(WS_KEYS_B, (size_t)nq * words * 4);
        need(WS_IDS_A, QS * 4); need(WS_IDS_B, QS * 4); need(WS_MISC4, QS + 8); need(WS_TILE_BO, (QS + 2) * 8);
        need(WS_SCANTMP, fd_scan_tmp_elems(QS) * 8 + 64);
    };
    if (tiled) {
        if (stream_cap) { need(WS_QT_STREAM, stream_cap * 34 + 64); need(WS_QT_STAB, (size_t)n_queries * NT * QT_MAXB * 8 + 64); }
        need(WS_CQ_KIDX, nq * 8); need(WS_CQ_NSEG, nq * 4);
        need(WS_QT_RANGES, (size_t)nq * ((S + (1u << QT_CELL_LOG2) - 1) >> QT_CELL_LOG2) * 16); need(WS_QT_COMPACT, ((size_t)n_queries * NT << qt_tl2) * 8);
        need(WS_QT_COUNT, (size_t)n_queries * NT * 4); need(WS_QT_AUX, n_queries * sizeof(qt_aux) + 256);
        if (qt32) {       // pieces in WS_QT_RANGES ([nq x NT x cells per tile] >= the ranges table: sized below), their first slots, the window tables, the heads
            const size_t ent = (size_t)nq * NT << (qt_tl2 - QT_CELL_LOG2);
            need(WS_QT_RANGES, ent * 16); need(WS_QT_PIECEP, ent * 4);
            need(WS_QT_WIN, ((size_t)nq * NT * qt_wpr + 2 * (size_t)n_queries * NT) * 4); need(WS_QT_HEAD, (size_t)n_queries * NT * 16);
        }
    } else if (tiled_big) {
        need(WS_CQ_KIDX, nq * 8); need(WS_CQ_NSEG, nq * 4);
        need(WS_QT_RANGES, (size_t)nq * NT14 * 16); need(WS_QT_COMPACT, ((size_t)NT14 << 14) * 8); need(WS_QT_COUNT, (size_t)NT14 * 4); need(WS_QT_AUX, sizeof(qt_aux) + 256);
        need(WS_QT_PARTIAL, ((size_t)big_slices * NT14 << 14) * 8);
        need(WS_QT_SURV, ((size_t)NT14 * 512 * 2 + NT14 + big_cap + 2 * big_wpr) * 4 + (big_slices + 2) * 8 + 64);
        need(WS_QT_ROWBITS, (size_t)big_cap * big_wpr * 4);
    } else need_rows();
    if (e != hipSuccess && (tiled || tiled_big)) {      // the tiled path's scratch did not fit (ranges, first-touch lists, decoded stream): the occupancy-row path instead
        (void)hipGetLastError();
        e = hipSuccess; tiled = false; tiled_big = false; qt32 = false; stream_cap = 0;
        need_rows();
    }
    if (e != hipSuccess) { free(ooff); c->err = std::string("count_query_batch workspace: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    // the rows' inputs: hashes, metadata, the queries' row ranges and — when the caller knows them — the rows' list positions.  The tiled motif path
    // sends them up in ONE block packed in page-locked memory ([metadata | list positions | row ranges | hashes] in ws[WS_MISC3]): one copy launch, not four
    uint32_t *d_qhash = c->ws[WS_MISC0].as<uint32_t>();
    unsigned long long *d_meta = c->ws[WS_MISC3].as<unsigned long long>();
    uint64_t *d_qoff = c->ws[WS_TILE_H].as<uint64_t>();
    long long *d_kidx = c->ws[WS_CQ_KIDX].as<long long>();
    const bool have_k = known_kidx && rows_kidx.size() == nq;
    bool k_up = false;      // the list positions are on the device already
    if (tiled) {
        const size_t o_k = nq * 8, o_off = 2 * nq * 8, o_h = o_off + (n_queries + 1) * 8, up_bytes = o_h + nq * 4;
        uint8_t *up = c->ws[WS_MISC3].ensure(up_bytes + 64) == hipSuccess ? (uint8_t *)c->host_pinned(5, up_bytes) : nullptr;
        if (up) {
            memcpy(up, rows_meta.data(), nq * 8);
            if (have_k) memcpy(up + o_k, rows_kidx.data(), nq * 8);
            memcpy(up + o_off, q_off, (n_queries + 1) * 8);
            memcpy(up + o_h, rows_hash.data(), nq * 4);
            uint8_t *d_blk = c->ws[WS_MISC3].as<uint8_t>();
            d_meta = (unsigned long long *)d_blk; d_kidx = (long long *)(d_blk + o_k); d_qoff = (uint64_t *)(d_blk + o_off); d_qhash = (uint32_t *)(d_blk + o_h);
            (void)hipMemcpyAsync(d_blk, up, up_bytes, hipMemcpyHostToDevice, st);
            k_up = have_k;
        } else {
            (void)hipGetLastError();
            if (c->ws[WS_MISC3].ensure(nq * 8) != hipSuccess) { free(ooff); c->err = "count_query_batch workspace"; return FDGPU_EHIP; }
            d_meta = c->ws[WS_MISC3].as<unsigned long long>();
        }
    }
    if (d_meta == c->ws[WS_MISC3].as<unsigned long long>() && d_qhash == c->ws[WS_MISC0].as<uint32_t>()) {
        (void)hipMemcpyAsync(d_qhash, rows_hash.data(), nq * 4, hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(d_meta, rows_meta.data(), nq * 8, hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(d_qoff, q_off, (n_queries + 1) * 8, hipMemcpyHostToDevice, st);
    }
    if (penalty) (void)hipMemcpyAsync(c->ws[WS_MISC5].p, penalty, S * 4, hipMemcpyHostToDevice, st);
    const float *d_penalty = penalty ? c->ws[WS_MISC5].as<float>() : ix->penalty;
    cq_args A;
    A.hashes = ix->hashes; A.offsets = ix->offsets; A.value = ix->value; A.H = ix->n_hashes;
    A.q_hash = d_qhash; A.nq = nq;
    A.hash_bits = c->ws[WS_KEYS_B].as<uint32_t>(); A.row_meta = d_meta;
    A.match = c->ws[WS_COUNTS].as<uint32_t>(); A.idf = c->ws[WS_SEGOFF].as<unsigned long long>(); A.packed = packed ? 1 : 0;
    A.words = words; A.first_id = (uint32_t)ix->first_id; A.S = (uint32_t)S;
    qt_args T;
    if (tiled) {
        T.value = ix->value; T.offsets = ix->offsets; T.ck_meta = ix->ck_meta; T.ck_ent = (const uint2 *)ix->ck_ent;
        T.kidx = d_kidx; T.row_meta = A.row_meta; T.q_rows = d_qoff; T.penalty = d_penalty;
        T.nq = (uint32_t)nq; T.n_queries = (uint32_t)n_queries; T.S = (uint32_t)S; T.first_id = (uint32_t)ix->first_id; T.NT = NT; T.tile_log2 = qt_tl2;
        T.NC = (uint32_t)((S + (1u << QT_CELL_LOG2) - 1) >> QT_CELL_LOG2);
        T.ranges = c->ws[WS_QT_RANGES].as<uint4>(); T.c_nid = c->ws[WS_QT_COMPACT].as<uint32_t>(); T.c_key = T.c_nid + ((size_t)n_queries * NT << qt_tl2);
        T.ccount = c->ws[WS_QT_COUNT].as<uint32_t>();
        T.ghist = nullptr; T.state = nullptr; T.aux = c->ws[WS_QT_AUX].as<qt_aux>(); T.out = nullptr; T.cap = 0;
        T.stream_ids = nullptr; T.stream_row = nullptr; T.stream_tab = nullptr; T.stream_used = nullptr; T.stream_cap = 0;
        if (stream_cap) {
            uint8_t *sb = c->ws[WS_QT_STREAM].as<uint8_t>();
            T.stream_ids = sb; T.stream_row = (uint16_t *)(sb + stream_cap * 32); T.stream_cap = (uint32_t)stream_cap;
            T.stream_tab = c->ws[WS_QT_STAB].as<uint2>(); T.stream_used = (uint32_t *)(c->ws[WS_QT_STAB].as<uint8_t>() + (size_t)n_queries * NT * QT_MAXB * 8);
            if (!qt32) (void)hipMemsetAsync(T.stream_used, 0, 4, st);       // (the 32-bit path scans the tiles' windows instead of claiming records: k_qt_bases)
        }
        T.plan_log2 = QT_CELL_LOG2; T.slices = nullptr; T.n_slices = 0; T.partial = nullptr; T.g_bm = T.g_rank = T.g_tcount = T.g_nid = T.g_rowbits = nullptr;
        T.g_wpr = 0; T.g_eend = T.g_nend = nullptr;
        T.dbg = nullptr;
        T.pieces = nullptr; T.piece_p = nullptr; T.win = nullptr; T.heads = nullptr; T.win_per_row = 0; T.top_n = top_n; T.max_rows = (uint32_t)max_rows;
        if (qt32) {
            T.pieces = c->ws[WS_QT_RANGES].as<uint4>(); T.piece_p = c->ws[WS_QT_PIECEP].as<uint32_t>(); T.win = c->ws[WS_QT_WIN].as<uint32_t>();
            T.heads = c->ws[WS_QT_HEAD].as<uint4>(); T.win_per_row = qt_wpr;
        }
        if (getenv("FDGPU_QT_DBG")) {       // phase durations of the tile kernels (measurement aid)
            T.dbg = (unsigned long long *)(c->ws[WS_QT_AUX].as<uint8_t>() + n_queries * sizeof(qt_aux));
            (void)hipMemsetAsync(T.dbg, 0, 256, st);
        }
    }
    std::vector<uint64_t> big_sl;        // row slices of the large-query path and the rows that end an edge / a node (outlive their asynchronous copies)
    std::vector<uint32_t> big_ends;
    if (tiled_big) {
        T.value = ix->value; T.offsets = ix->offsets; T.ck_meta = ix->ck_meta; T.ck_ent = (const uint2 *)ix->ck_ent;
        T.kidx = d_kidx; T.row_meta = A.row_meta; T.q_rows = d_qoff; T.penalty = d_penalty;
        T.nq = (uint32_t)nq; T.n_queries = 1; T.S = (uint32_t)S; T.first_id = (uint32_t)ix->first_id; T.NT = NT14; T.tile_log2 = 14; T.plan_log2 = 14;
        T.NC = (uint32_t)((S + (1u << QT_CELL_LOG2) - 1) >> QT_CELL_LOG2);
        T.ranges = c->ws[WS_QT_RANGES].as<uint4>(); T.c_nid = c->ws[WS_QT_COMPACT].as<uint32_t>(); T.c_key = T.c_nid + ((size_t)NT14 << 14);
        T.ccount = c->ws[WS_QT_COUNT].as<uint32_t>();
        T.ghist = nullptr; T.state = nullptr; T.aux = c->ws[WS_QT_AUX].as<qt_aux>(); T.out = nullptr; T.cap = big_cap; T.dbg = nullptr;
        T.stream_ids = nullptr; T.stream_row = nullptr; T.stream_tab = nullptr; T.stream_used = nullptr; T.stream_cap = 0;
        T.pieces = nullptr; T.piece_p = nullptr; T.win = nullptr; T.heads = nullptr; T.win_per_row = 0; T.top_n = top_n; T.max_rows = (uint32_t)max_rows;
        // slices of roughly equal posting counts: a row's list holds ~ S / 2^idf ids (idf = log2(S / length), its fixed-point image is in the metadata)
        std::vector<double> w(nq);
        double tot = 0;
        for (uint64_t r = 0; r < nq; ++r) {       // 2^-idf to a few percent: the integer part as an exponent field, the fraction linearly
            const unsigned long long fix = rows_meta[r] >> 2;
            const uint64_t eb = (uint64_t)(1023 - (int)std::min<unsigned long long>(fix >> 22, 1000ull)) << 52;
            double p2;
            memcpy(&p2, &eb, 8);
            w[r] = (1.0 - 0.5 * (double)(fix & 4194303ull) / 4194304.0) * p2 + 1e-7;
            tot += w[r];
        }
        big_sl.push_back(0);
        double acc = 0;
        for (uint64_t r = 0; r < nq; ++r) {
            acc += w[r];
            if (big_sl.size() < big_slices && acc >= tot * (double)big_sl.size() / big_slices && r + 1 < nq) big_sl.push_back(r + 1);
        }
        big_sl.push_back(nq);
        T.n_slices = (uint32_t)big_sl.size() - 1;
        big_ends.assign((size_t)2 * big_wpr, 0u);
        for (uint64_t r = 0; r < nq; ++r) {
            if (rows_meta[r] & 1ull) big_ends[r >> 5] |= 1u << (r & 31u);
            if (rows_meta[r] & 2ull) big_ends[big_wpr + (r >> 5)] |= 1u << (r & 31u);
        }
        uint32_t *sv = c->ws[WS_QT_SURV].as<uint32_t>();
        T.g_bm = sv; T.g_rank = sv + (size_t)NT14 * 512; T.g_tcount = T.g_rank + (size_t)NT14 * 512; T.g_nid = T.g_tcount + NT14;
        uint32_t *d_ends = T.g_nid + big_cap;
        T.g_eend = d_ends; T.g_nend = d_ends + big_wpr;
        uint64_t *d_sl = (uint64_t *)(((uintptr_t)(d_ends + 2 * big_wpr) + 63) & ~(uintptr_t)63);
        T.slices = d_sl;
        T.partial = c->ws[WS_QT_PARTIAL].as<unsigned long long>(); T.g_rowbits = c->ws[WS_QT_ROWBITS].as<uint32_t>(); T.g_wpr = big_wpr;
        (void)hipMemcpyAsync(d_ends, big_ends.data(), big_ends.size() * 4, hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(d_sl, big_sl.data(), big_sl.size() * 8, hipMemcpyHostToDevice, st);
    }
    std::vector<uint64_t> slices;        // outlives its asynchronous copy (every path below synchronises the stream before returning)
    if (!tiled && !tiled_big) {
        StageTimer t(c, "cq_batch", 0);
        int rs = cq_score(c, A, known_segments);
        if (rs) { free(ooff); return rs; }
        if (sliced) {      // slices at node boundaries (see fdgpu_count_query)
            const uint64_t per = (nq + 31) / 32;
            slices.push_back(0);
            for (uint64_t r = 0; r + 1 < nq; ++r)
                if ((rows_meta[r] & 2ull) && r + 1 - slices.back() >= per) slices.push_back(r + 1);
            slices.push_back(nq);
            hipError_t es = c->ws[WS_TILE_B].ensure(slices.size() * 8);
            if (es == hipSuccess) es = hipMemcpyAsync(c->ws[WS_TILE_B].p, slices.data(), slices.size() * 8, hipMemcpyHostToDevice, st);
            if (es != hipSuccess) slices.clear();
        }
        if (!keys_only)
            fd_launch_cq_rows_finalize(A, d_qoff, (uint32_t)n_queries, slices.empty() ? nullptr : c->ws[WS_TILE_B].as<uint64_t>(),
                                       slices.empty() ? 0u : (uint32_t)slices.size() - 1, c->ws[WS_IDS_A].as<uint32_t>(), c->ws[WS_IDS_B].as<uint32_t>(),
                                       c->ws[WS_MISC4].as<uint8_t>(), max_rows, st);
        if (!dense_topn)
            fd_exclusive_scan<uint8_t>(c->ws[WS_MISC4].as<uint8_t>(), QS, c->ws[WS_TILE_BO].as<uint64_t>(), c->ws[WS_SCANTMP].as<uint64_t>(),
                                       c->ws[WS_TOTAL].as<uint64_t>(), st);
    }
    if (dense_topn) {
        // candidate selection straight from the dense accumulators (k_topn_*_dense + k_topn_sort): no flag scan, no compaction of every
        // touched structure, one synchronisation instead of three
        const uint32_t cap = top_n + 1024;
        hipError_t e2 = c->ws[WS_KEYS_A].ensure((size_t)n_queries * cap * sizeof(fd_count_rec));
        if (e2 == hipSuccess) e2 = c->ws[WS_TILE_HO].ensure((size_t)n_queries * top_n * sizeof(fd_count_rec));
        const size_t topn_bytes = (size_t)n_queries * 2048 * 4;
        if (e2 == hipSuccess && c->ws[WS_CQ_TOPN].cap < topn_bytes) {
            e2 = c->ws[WS_CQ_TOPN].ensure(topn_bytes);
            if (e2 == hipSuccess) e2 = hipMemsetAsync(c->ws[WS_CQ_TOPN].p, 0, c->ws[WS_CQ_TOPN].cap, st);
        }
        if (e2 == hipSuccess) e2 = c->ws[WS_MISC2].ensure(n_queries * 16);
        std::vector<uint32_t> tstate((size_t)n_queries * 4);
        // the ranked records land in the caller's array (page-locked, pooled) at a stride of top_n and are closed up in place afterwards
        fd_count_rec *rr = dev ? nullptr : (fd_count_rec *)fd_out_alloc(std::max<uint64_t>((uint64_t)n_queries * top_n, 1) * sizeof(fd_count_rec), true);
        if (!dev && !rr) { free(ooff); return FDGPU_ENOMEM; }
        if (e2 == hipSuccess && tiled) {
            T.ghist = c->ws[WS_CQ_TOPN].as<uint32_t>(); T.state = c->ws[WS_MISC2].as<qt_state>(); T.out = c->ws[WS_KEYS_A].p; T.cap = cap;
            // (the rows' list positions are an input like their hashes: uploaded before the timed stage)
            if (have_k && !k_up) (void)hipMemcpyAsync(d_kidx, rows_kidx.data(), nq * 8, hipMemcpyHostToDevice, st);
            {
                StageTimer t(c, "cq_batch", 0);
                if (!have_k) fd_launch_cq_plan(A, d_kidx, c->ws[WS_CQ_NSEG].as<uint32_t>(), st);
                if (qt32) { fd_launch_qt_layout(T, st); fd_launch_qt_score32(T, st); }
                else { fd_launch_qt_plan(T, st); fd_launch_qt_score(T, st); }
            }
            StageTimer t(c, "cq_topn", 0);
            fd_launch_qt_select(T, top_n, c->ws[WS_TILE_HO].p, st);
            if (T.dbg && qt32) {
                unsigned long long d[32];
                if (hipMemcpyAsync(d, T.dbg, 256, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                    const double wg = (double)NT * (double)n_queries * 100.0;      // ticks of 10 ns -> us per workgroup
                    fprintf(stderr, "[qt32] set-up %.2f decode (first wavefront) %.2f wait %.2f keys %.2f cut %.2f emit %.2f us/WG, %.1f windows/WG (%u x %llu WGs)\n",
                            d[0] / wg, d[1] / wg, d[2] / wg, d[3] / wg, d[4] / wg, d[5] / wg, d[17] / ((double)NT * (double)n_queries), NT, (unsigned long long)n_queries);
                }
            } else if (T.dbg) {
                unsigned long long d[32];
                if (hipMemcpyAsync(d, T.dbg, 256, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                    const double wg = (double)NT * (double)n_queries * 100.0;      // ticks of 10 ns -> us per workgroup
                    fprintf(stderr, "[qt] A: setup %.2f plan %.2f units %.2f decode %.2f wait %.2f final %.2f | B: setup %.2f plan %.2f units %.2f decode %.2f wait %.2f records %.2f us/WG (%u x %llu WGs)\n",
                            d[0] / wg, d[1] / wg, d[2] / wg, d[3] / wg, d[4] / wg, d[5] / wg, d[8] / wg, d[9] / wg, d[10] / wg, d[11] / wg, d[12] / wg, d[13] / wg, NT,
                            (unsigned long long)n_queries);
                    for (int z = 0; z < 2; ++z) {
                        const unsigned long long *e = d + 16 + 8 * z;
                        const double nw = (double)NT * (double)n_queries;
                        fprintf(stderr, "[qt] %c unit loop: units/WG %.1f steps/WG %.1f, wave time mean %.2f us, slowest wave mean %.2f us\n", z ? 'B' : 'A', e[0] / nw, e[1] / nw,
                                e[3] / (nw * (qt_tl2 == 14 ? 16 : 8) * 100.0), e[2] / (nw * 100.0));
                    }
                }
            }
        } else if (e2 == hipSuccess && tiled_big) {
            T.ghist = c->ws[WS_CQ_TOPN].as<uint32_t>(); T.state = c->ws[WS_MISC2].as<qt_state>(); T.out = c->ws[WS_KEYS_A].p;
            if (have_k && !k_up) (void)hipMemcpyAsync(d_kidx, rows_kidx.data(), nq * 8, hipMemcpyHostToDevice, st);
            {
                StageTimer t(c, "cq_batch", 0);
                if (!have_k) fd_launch_cq_plan(A, d_kidx, c->ws[WS_CQ_NSEG].as<uint32_t>(), st);
                fd_launch_qt_plan(T, st);
                fd_launch_qt_big_score(T, st);
            }
            StageTimer t(c, "cq_topn", 0);
            fd_launch_qt_big_select(T, top_n, c->ws[WS_TILE_HO].p, st);
        } else if (e2 == hipSuccess) {
            StageTimer t(c, "cq_topn", 0);
            if (keys_only)      // keys in the compaction's position buffer, unused on this path
                fd_launch_cq_topn_dense(A, d_qoff, d_penalty, c->ws[WS_TILE_BO].as<uint32_t>(), (uint32_t)n_queries, top_n, cap,
                                        c->ws[WS_KEYS_A].p, c->ws[WS_MISC2].p, c->ws[WS_CQ_TOPN].as<uint32_t>(), st);
            else
                fd_launch_cq_topn_acc(A, d_penalty, c->ws[WS_IDS_A].as<uint32_t>(), c->ws[WS_IDS_B].as<uint32_t>(), (uint32_t)n_queries, top_n, cap,
                                      c->ws[WS_KEYS_A].p, c->ws[WS_MISC2].p, c->ws[WS_CQ_TOPN].as<uint32_t>(), st);
            fd_launch_cq_topn_sort(c->ws[WS_KEYS_A].p, cap, c->ws[WS_MISC2].p, (uint32_t)n_queries, top_n, c->ws[WS_TILE_HO].p, st);
        }
        if (cq_trace) fprintf(stderr, "[count_query] launched at %.3f ms\n", cq_ms());
        if (dev && dev->while_running && *dev->while_running) (*dev->while_running)();
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(tstate.data(), c->ws[WS_MISC2].p, n_queries * 16, hipMemcpyDeviceToHost, st);
        if (e2 == hipSuccess && !dev) e2 = hipMemcpyAsync(rr, c->ws[WS_TILE_HO].p, (size_t)n_queries * top_n * sizeof(fd_count_rec), hipMemcpyDeviceToHost, st);
        fd_count_rec *head = nullptr;      // the caller's candidates (fdgpu_query_batch: match_top per query) ride on the same wait
        if (e2 == hipSuccess && dev && dev->head_n) {
            const uint32_t mt = std::min(dev->head_n, top_n);
            head = (fd_count_rec *)c->host_pinned(3, (size_t)n_queries * mt * sizeof(fd_count_rec));
            if (head) e2 = hipMemcpy2DAsync(head, (size_t)mt * sizeof(fd_count_rec), c->ws[WS_TILE_HO].p, (size_t)top_n * sizeof(fd_count_rec), (size_t)mt * sizeof(fd_count_rec), n_queries,
                                            hipMemcpyDeviceToHost, st);
        }
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        if (e2 == hipSuccess) e2 = hipGetLastError();
        if (e2 != hipSuccess) { free(ooff); fdgpu_free(rr); c->err = std::string("count_query_batch: ") + hipGetErrorString(e2); return FDGPU_EHIP; }
        if (cq_trace) fprintf(stderr, "[count_query] records on the host at %.3f ms\n", cq_ms());
        bool overflow = false;
        uint64_t tot = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { overflow = overflow || tstate[4 * t + 3] > cap; tot += std::min<uint32_t>(tstate[4 * t + 3], top_n); }
        if (dev) {        // the ranked selection stays where it is
            free(ooff);
            dev->counts.resize(n_queries);
            for (uint64_t t = 0; t < n_queries; ++t) dev->counts[t] = tstate[4 * t + 3];
            dev->head = head;
            dev->got = true; dev->overflow = overflow; dev->recs = c->ws[WS_TILE_HO].p; dev->state = c->ws[WS_MISC2].p; dev->top_n = top_n; dev->cap = cap;
            return FDGPU_OK;
        }
        if (overflow) {   // more ties at the cut-off than the selection's slots hold: the compacting path ranks that call
            free(ooff); fdgpu_free(rr);
            return fd_count_query_batch_impl(c, ix, n_queries, q_off, q_hash, q_node, q_edge_j, q_idf, penalty, top_n, out, out_off, false, nullptr, known_segments);
        }
        (void)tot;
        uint64_t w = 0;
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint64_t m = std::min<uint32_t>(tstate[4 * t + 3], top_n);
            ooff[t] = w;
            if (m && w != t * top_n) memmove(rr + w, rr + (size_t)t * top_n, (size_t)m * sizeof(fd_count_rec));      // w <= t * top_n: forward
            w += m;
        }
        ooff[n_queries] = w;
        *out = rr; *out_off = ooff;
        return FDGPU_OK;
    }
    uint64_t n = 0;
    int rc = d2h_u64(c, c->ws[WS_TOTAL].as<uint64_t>(), &n);
    if (rc) { free(ooff); return rc; }
    const bool select = top_n > 0 && n > (uint64_t)top_n * n_queries;   // worth preselecting on the device
    fd_count_rec *r = select ? nullptr : (fd_count_rec *)malloc(std::max<uint64_t>(n, 1) * sizeof(fd_count_rec));
    if (!select && !r) { free(ooff); return FDGPU_ENOMEM; }
    e = c->ws[WS_TILE_HO].ensure(std::max<uint64_t>(n, 1) * sizeof(fd_count_rec));
    if (e == hipSuccess) e = c->ws[WS_TILE_PO].ensure((n_queries + 1) * 8 + 64);
    if (e == hipSuccess) {
        fd_launch_cq_compact_batch(A, c->ws[WS_IDS_A].as<uint32_t>(), c->ws[WS_IDS_B].as<uint32_t>(), c->ws[WS_MISC4].as<uint8_t>(),
                                   c->ws[WS_TILE_BO].as<uint64_t>(), d_penalty, QS, c->ws[WS_TILE_HO].p, st);
        // out_off[t] = scan position at t * S
        std::vector<uint64_t> idx(n_queries + 1);
        for (uint64_t t = 0; t <= n_queries; ++t) idx[t] = t * S;
        e = c->ws[WS_MISC1].ensure((n_queries + 1) * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws[WS_MISC1].p, idx.data(), (n_queries + 1) * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            fd_launch_gather_u64(c->ws[WS_TILE_BO].as<uint64_t>(), c->ws[WS_MISC1].as<uint64_t>(), n_queries + 1, c->ws[WS_TILE_PO].as<uint64_t>(), st);
            e = hipMemcpyAsync(ooff, c->ws[WS_TILE_PO].p, (n_queries + 1) * 8, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // idx must outlive the copy
    }
    if (e == hipSuccess && select) {
        // per-query preselection of the top_n by idf on the device (k_cq_topn); a query whose threshold bin overflows the
        // fixed-stride output falls back to its full list
        const uint32_t cap = top_n + 1024;
        const bool dev_sort = cap <= 4096;          // k_topn_sort ranks and cuts on the device: only top_n records per query cross the bus
        const uint32_t stride = dev_sort ? top_n : cap;
        std::vector<uint32_t> cnt(n_queries);
        std::vector<fd_count_rec> sel((size_t)n_queries * stride);
        e = c->ws[WS_KEYS_A].ensure((size_t)n_queries * cap * sizeof(fd_count_rec));
        if (e == hipSuccess && dev_sort) e = c->ws[WS_KEYS_B].ensure((size_t)n_queries * top_n * sizeof(fd_count_rec));
        const size_t topn_bytes = (size_t)n_queries * 2048 * 4;
        if (e == hipSuccess && c->ws[WS_CQ_TOPN].cap < topn_bytes) {     // histogram table: zeroed when (re)allocated, the kernels leave it zero
            e = c->ws[WS_CQ_TOPN].ensure(topn_bytes);
            if (e == hipSuccess) e = hipMemsetAsync(c->ws[WS_CQ_TOPN].p, 0, c->ws[WS_CQ_TOPN].cap, st);
        }
        if (e == hipSuccess) e = c->ws[WS_MISC2].ensure(n_queries * 16);
        std::vector<uint32_t> tstate((size_t)n_queries * 4);
        if (e == hipSuccess) {
            fd_launch_cq_topn(c->ws[WS_TILE_HO].p, c->ws[WS_TILE_PO].as<uint64_t>(), (uint32_t)n_queries, top_n, cap, c->ws[WS_KEYS_A].p, c->ws[WS_MISC2].p,
                              c->ws[WS_CQ_TOPN].as<uint32_t>(), st);
            if (dev_sort) fd_launch_cq_topn_sort(c->ws[WS_KEYS_A].p, cap, c->ws[WS_MISC2].p, (uint32_t)n_queries, top_n, c->ws[WS_KEYS_B].p, st);
            e = hipMemcpyAsync(tstate.data(), c->ws[WS_MISC2].p, n_queries * 16, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(sel.data(), dev_sort ? c->ws[WS_KEYS_B].p : c->ws[WS_KEYS_A].p, sel.size() * sizeof(fd_count_rec), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { free(ooff); c->err = std::string("count_query_batch: ") + hipGetErrorString(e); return FDGPU_EHIP; }
        for (uint64_t t = 0; t < n_queries; ++t) cnt[t] = tstate[4 * t + 3];
        uint64_t tot = 0;
        for (uint64_t t = 0; t < n_queries; ++t) tot += cnt[t] <= cap ? (dev_sort ? std::min<uint32_t>(cnt[t], top_n) : cnt[t]) : (ooff[t + 1] - ooff[t]);
        r = (fd_count_rec *)malloc(std::max<uint64_t>(tot, 1) * sizeof(fd_count_rec));
        if (!r) { free(ooff); return FDGPU_ENOMEM; }
        std::vector<uint64_t> noff(n_queries + 1, 0);
        for (uint64_t t = 0; t < n_queries; ++t) {
            if (cnt[t] <= cap) {
                uint64_t m = dev_sort ? std::min<uint32_t>(cnt[t], top_n) : cnt[t];
                memcpy(r + noff[t], sel.data() + (size_t)t * stride, (size_t)m * sizeof(fd_count_rec));
                if (!dev_sort) m = fd_rank_trim(r + noff[t], m, top_n);
                noff[t + 1] = noff[t] + m;
            } else {
                uint64_t m = ooff[t + 1] - ooff[t];
                if (hipMemcpy(r + noff[t], (const fd_count_rec *)c->ws[WS_TILE_HO].p + ooff[t], m * sizeof(fd_count_rec), hipMemcpyDeviceToHost) != hipSuccess) {
                    free(r); free(ooff); c->err = "count_query_batch: fallback copy failed"; return FDGPU_EHIP;
                }
                noff[t + 1] = noff[t] + fd_rank_trim(r + noff[t], m, top_n);
            }
        }
        memcpy(ooff, noff.data(), (n_queries + 1) * 8);
        *out = r; *out_off = ooff;
        return FDGPU_OK;
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(r, c->ws[WS_TILE_HO].p, n * sizeof(fd_count_rec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { free(r); free(ooff); c->err = std::string("count_query_batch: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    if (top_n > 0) {     // few records: ranked and cut on the host, same contract as the device selection
        uint64_t w = 0;
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint64_t a = ooff[t], m = fd_rank_trim(r + a, ooff[t + 1] - a, top_n);
            if (w != a) memmove(r + w, r + a, m * sizeof(fd_count_rec));
            ooff[t] = w; w += m;
        }
        ooff[n_queries] = w;
    }
    *out = r; *out_off = ooff;
    return FDGPU_OK;
}
extern "C" int fdgpu_count_query_batch(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off, const uint32_t *q_hash,
                                       const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf, const float *penalty,
                                       fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    return fd_count_query_batch_impl(c, ix, n_queries, q_off, q_hash, q_node, q_edge_j, q_idf, penalty, 0, out, out_off, true, nullptr);
}
// as above, but per query only the top_n records come back, RANKED as the candidate selection of query_pdb.rs:404-411 ranks them (idf
// descending, ties by ascending structure id): radix selection + LDS bitonic sort on the device, top_n records per query over the bus
extern "C" int fdgpu_count_query_batch_top(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off, const uint32_t *q_hash,
                                           const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf, const float *penalty,
                                           uint32_t top_n, fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    return fd_count_query_batch_impl(c, ix, n_queries, q_off, q_hash, q_node, q_edge_j, q_idf, penalty, top_n, out, out_off, true, nullptr);
}

// The length penalty nres^(-lp) of the index's structures (count_query.rs:200), kept on the device: count queries may then pass
// penalty = NULL instead of uploading S floats per call.  penalty = NULL drops the resident copy.
extern "C" int fdgpu_index_set_penalty(fdgpu_ctx *c, fdgpu_index *ix, const float *penalty) { FD_LOCK(c);
    if (!c || !ix) return FDGPU_EINVAL;
    if (ix->penalty) { (void)hipFree(ix->penalty); ix->penalty = nullptr; }
    if (!penalty || !ix->n_structures) return FDGPU_OK;
    HIPCHK(c, hipMalloc((void **)&ix->penalty, ix->n_structures * 4));
    HIPCHK(c, hipMemcpy(ix->penalty, penalty, ix->n_structures * 4, hipMemcpyHostToDevice));
    return FDGPU_OK;
}

// count_query for the query maps fdgpu_make_query_map[_batch] returned, without a round trip through the caller: the entries of
// every map (hash, (qi, qj)) are scored with idf = log2(total_structures / posting length) of the hash ITSELF (count_query.rs:181-200;
// the idf inside the map belongs to the pair's observed hash and feeds the retrieval's subgraph idf instead); hashes the index does
// not hold are dropped.  Output as fdgpu_count_query_batch_top.
// every map's hash[] (all queries), then every map's primary_hash[]: the 2 * sum(n) hashes whose posting lengths a sharded query needs
// over the WHOLE database
uint64_t fd_maps_hashes(uint64_t n_queries, const fd_query_map *const *qms, std::vector<uint32_t> &h) {
    uint64_t nq = 0;
    for (uint64_t t = 0; t < n_queries; ++t) nq += qms[t]->n;
    h.assign(std::max<uint64_t>(2 * nq, 1), 0);
    uint64_t at = 0;
    for (uint64_t t = 0; t < n_queries; ++t) { if (qms[t]->n) memcpy(&h[at], qms[t]->hash, qms[t]->n * 4); at += qms[t]->n; }
    for (uint64_t t = 0; t < n_queries; ++t) { if (qms[t]->n) memcpy(&h[at], qms[t]->primary_hash, qms[t]->n * 4); at += qms[t]->n; }
    return nq;
}
// scoring of query maps with the posting lengths given: len[0 .. nq) belong to the maps' hash[] in order (idf = log2f(total / len), absent
// hashes drop out); with primary_len the maps' own idf[] (the retrieval's subgraph idf, query.rs:283-288) is rewritten from the lengths
// of primary_hash[] — what a sharded index needs, whose make_query_map saw one shard only
int fd_count_query_maps_len(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, const uint64_t *len,
                            const uint64_t *primary_len, const float *penalty, float total_structures, uint32_t top_n, fd_count_rec **out,
                            uint64_t **out_off, fd_cq_dev_out *dev, const uint32_t *seg, const long long *kidx, bool allow_dense) {
    uint64_t nq = 0;
    int64_t W = seg ? 0 : -1;
    for (uint64_t t = 0; t < n_queries; ++t) nq += qms[t]->n;
    std::vector<uint64_t> q_off(n_queries + 1, 0);
    std::vector<uint32_t> qh, qn, qe;
    std::vector<float> qi;
    std::vector<long long> qk;
    std::vector<uint64_t> ql;       // the kept rows' posting lengths (local to this index only when the lengths are: the tiled path sizes its stream from them)
    // (filled through plain pointers: a whole-structure query is 10^5 rows, six push_backs each were a third of this loop)
    qh.resize(nq + 1); qn.resize(nq + 1); qe.resize(nq + 1); qi.resize(nq + 1); ql.resize(nq + 1);
    if (kidx) qk.resize(nq + 1);
    uint32_t *const p_h = qh.data(), *const p_n = qn.data(), *const p_e = qe.data();
    float *const p_i = qi.data();
    uint64_t *const p_l = ql.data();
    long long *const p_k = kidx ? qk.data() : nullptr;
    uint64_t at = 0, w = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        const fd_query_map *m = qms[t];
        const uint32_t *const mh = m->hash, *const mqi = m->qi, *const mqj = m->qj;
        float *const midf = m->idf;
        const uint64_t mn = m->n;
        for (uint64_t k = 0; k < mn; ++k, ++at) {
            if (primary_len) midf[k] = primary_len[at] ? log2f(total_structures / (float)primary_len[at]) : 0.0f;
            const uint64_t L = len[at];
            if (!L) continue;
            p_h[w] = mh[k]; p_n[w] = mqi[k]; p_e[w] = mqj[k];
            if (p_k) p_k[w] = kidx[at];
            p_l[w] = L;
            if (seg) W += seg[at];
            ++w;
        }
        q_off[t + 1] = w;
    }
    // idf of the kept rows: f32 like the reference's (total / len).log2().  log2f is ~8 ns a call: the 10^5 rows of a whole-structure query in eight parts on
    // the context's helper threads (0.6 ms of the call's host time on one)
    auto idf_rows = [&](uint64_t a, uint64_t b) { for (uint64_t r = a; r < b; ++r) p_i[r] = log2f(total_structures / (float)p_l[r]); };
    if (w < 16384) idf_rows(0, w);
    else {
        const unsigned nt = 8;
        std::atomic<unsigned> part(0);
        const std::function<void()> wk = [&]() { for (;;) { const unsigned k = part.fetch_add(1); if (k >= nt) break; idf_rows(w * k / nt, w * (k + 1) / nt); } };
        c->host_pool.run(c->small_par(nt), wk);
    }
    qh.resize(w); qn.resize(w); qe.resize(w); qi.resize(w); ql.resize(w);
    if (kidx) qk.resize(w);
    if (qh.empty()) { qh.push_back(0); qn.push_back(0); qe.push_back(0); qi.push_back(0.0f); W = seg ? 0 : -1; qk.clear(); ql.clear(); }
    // (the lengths bound the LOCAL lists only when they are this index's own: the caller that passes kidx made the maps against it)
    return fd_count_query_batch_impl(c, ix, n_queries, q_off.data(), qh.data(), qn.data(), qe.data(), qi.data(), penalty, top_n, out, out_off, allow_dense, dev, W,
                                     kidx && qk.size() == qh.size() ? qk.data() : nullptr, kidx && ql.size() == qh.size() ? ql.data() : nullptr);
}
extern "C" int fdgpu_count_query_maps_top(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, const float *penalty,
                                          float total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    return fd_count_query_maps_top_impl(c, ix, n_queries, qms, penalty, total_structures, top_n, out, out_off, nullptr);
}
int fd_count_query_maps_top_impl(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, const float *penalty, float total_structures,
                                 uint32_t top_n, fd_count_rec **out, uint64_t **out_off, fd_cq_dev_out *dev, bool allow_dense) { FD_LOCK(c);
    if (!c || !ix || !out || !out_off || (n_queries && !qms)) return FDGPU_EINVAL;
    uint64_t nq = 0;
    for (uint64_t t = 0; t < n_queries; ++t) { if (!qms[t]) return FDGPU_EINVAL; nq += qms[t]->n; }
    std::vector<uint32_t> h(std::max<uint64_t>(nq, 1));
    std::vector<uint64_t> len(std::max<uint64_t>(nq, 1), 0);
    std::vector<uint32_t> seg(std::max<uint64_t>(nq, 1), 0);
    std::vector<long long> kidx(std::max<uint64_t>(nq, 1), -1);
    bool remembered = nq > 0;       // maps made against THIS index carry their hashes' posting lengths, segment counts and list positions
    for (uint64_t t = 0; t < n_queries; ++t) remembered = remembered && (!qms[t]->n || (qms[t]->post_len && qms[t]->post_seg && qms[t]->post_index_uid == ix->uid));
    bool have_kidx = remembered;
    for (uint64_t t = 0; t < n_queries; ++t) have_kidx = have_kidx && (!qms[t]->n || qms[t]->post_kidx);
    uint64_t at = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        if (qms[t]->n) {
            if (have_kidx) memcpy(&kidx[at], qms[t]->post_kidx, qms[t]->n * 8);
            if (remembered) { memcpy(&len[at], qms[t]->post_len, qms[t]->n * 8); memcpy(&seg[at], qms[t]->post_seg, qms[t]->n * 4); }
            else memcpy(&h[at], qms[t]->hash, qms[t]->n * 4);
        }
        at += qms[t]->n;
    }
    int rc = !remembered && nq && ix->n_structures ? fd_posting_lengths_segs(c, ix, h.data(), nq, len.data(), seg.data()) : FDGPU_OK;
    if (rc) return rc;
    return fd_count_query_maps_len(c, ix, n_queries, qms, len.data(), nullptr, penalty, total_structures, top_n, out, out_off, dev, seg.data(),
                                   have_kidx ? kidx.data() : nullptr, allow_dense);
}
// The two halves of the sharded form for hosts that bring their own transport (MPI, gloo, ...): the LOCAL posting lengths of the maps'
// hash[] and primary_hash[] (2 * sum(n) values, fd_maps_hashes order) — the caller sums them over the ranks — and the scoring of the
// local shard with those GLOBAL lengths (maps' idf[] rewritten from the primary lengths).  With RCCL: fdgpu_sharded_count_query_maps.
extern "C" int fdgpu_query_maps_lengths(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, uint64_t *lengths) { FD_LOCK(c);
    if (!c || !ix || (n_queries && !qms) || !lengths) return FDGPU_EINVAL;
    for (uint64_t t = 0; t < n_queries; ++t) if (!qms[t]) return FDGPU_EINVAL;
    std::vector<uint32_t> h;
    const uint64_t nq = fd_maps_hashes(n_queries, qms, h);
    return nq ? fdgpu_posting_lengths(c, ix, h.data(), 2 * nq, lengths) : FDGPU_OK;
}
extern "C" int fdgpu_count_query_maps_top_global(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, fd_query_map *const *qms, const uint64_t *global_lengths,
                                                 const float *penalty, float total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off) { FD_LOCK(c);
    if (!c || !ix || !out || !out_off || (n_queries && !qms)) return FDGPU_EINVAL;
    uint64_t nq = 0;
    for (uint64_t t = 0; t < n_queries; ++t) { if (!qms[t]) return FDGPU_EINVAL; nq += qms[t]->n; }
    if (nq && !global_lengths) return FDGPU_EINVAL;
    const uint64_t zero = 0;
    return fd_count_query_maps_len(c, ix, n_queries, qms, nq ? global_lengths : &zero, nq ? global_lengths + nq : nullptr, penalty, total_structures, top_n,
                                   out, out_off, nullptr);
}

