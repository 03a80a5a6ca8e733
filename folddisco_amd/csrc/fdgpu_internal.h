// fdgpu_internal.h — host-side objects behind the opaque handles of include/fdgpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/fdgpu.h"
#include "../../include/fdgpu_debug.h"
#include "fd_device.h"
#include "fd_geom_other.h"

#include <condition_variable>
#include <thread>
// A context's host helper threads (the per-candidate glue of whole-structure retrievals, fd_host_query.hip): started once, woken per job — spawning
// nineteen std::threads per pass cost ~1 ms of a 10 ms retrieval.  run(n, f): f() on n - 1 pool threads and on the caller, returns when all are done.
struct fd_host_pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void()> *job = nullptr;
    uint64_t gen = 0;
    unsigned want = 0, taken = 0, active = 0;
    bool stop = false;
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()> *f = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (gen != seen && taken < want); });
                if (stop) return;
                seen = gen; ++taken; f = job;
            }
            (*f)();
            {
                std::lock_guard<std::mutex> lk(mu);
                --active;
            }
            cv_done.notify_one();
        }
    }
    void run(unsigned n, const std::function<void()> &f) {
        if (n <= 1) { f(); return; }
        const unsigned helpers = n - 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (th.size() < helpers) th.emplace_back([this] { loop(); });
            job = &f; want = helpers; taken = 0; active = helpers; ++gen;
        }
        cv.notify_all();
        f();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
        job = nullptr;
    }
    ~fd_host_pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};

struct fd_devbuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        // small buffers grow geometrically (query-sized scratch would otherwise be re-allocated for every slightly larger
        // query: hipFree synchronises the device); big ones take what they need
        size_t want = bytes + (bytes < ((size_t)4 << 30) ? bytes / 16 : (size_t)0) + 256;      // the sort buffers of a 2^34-key call: no 6 % on 70 GB
        if (cap && cap < (64u << 20) && want < cap + cap / 2) want = cap + cap / 2;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct fd_timing_entry { const char *name; hipEvent_t ev0, ev1; uint64_t bytes; };

enum {
    WS_COUNTS, WS_CURSOR, WS_SEGOFF, WS_SCANTMP, WS_TOTAL, WS_KEYS_A, WS_IDS_A, WS_KEYS_B, WS_IDS_B, WS_GHIST, WS_TOT,
    WS_TILE_B, WS_TILE_H, WS_TILE_P, WS_TILE_BO, WS_TILE_HO /* ranked records of a scoring call: fdgpu_query_batch copies them out on the side stream WHILE the
    retrieval runs — no retrieval stage may ensure() or write this buffer (checked there) */, WS_TILE_PO, WS_MISC0, WS_MISC1, WS_MISC2, WS_MISC3, WS_MISC4, WS_MISC5, WS_FRAMES,
    WS_CQ_KIDX, WS_CQ_NSEG, WS_CQ_WSTART, WS_CQ_SEGSUM, WS_CQ_TOPN,
    WS_QT_RANGES, WS_QT_COMPACT, WS_QT_COUNT, WS_QT_AUX, WS_RS_PLAN, WS_RS_REC, WS_RS_RECRES, WS_QT_PARTIAL, WS_QT_SURV, WS_QT_ROWBITS, WS_QT_STREAM, WS_QT_STAB, WS_QT_PIECEP, WS_QT_WIN, WS_QT_HEAD, WS_RS_SPLIT, WS_MP_QSET,
    WS_CA_PERM, WS_OK_PERM, WS_AA_PERM, WS_SEG_TAB,
    WS_RS_TAB, WS_RS_SEG, WS_RS_OUT, WS_RS_RES, WS_RS_KX, WS_RS_KY, WS_RS_KOFF, WS_RS_SOL, WS_RS_CNT, WS_RS_GQ, WS_MP_ACT, WS_MP_Q,
    WS_COUNT
};

struct fdgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    fd_devbuf ws[WS_COUNT];
    bool timing = false;
    // Every entry point that takes the context locks it: concurrent callers (the reference calls its seams from rayon workers,
    // controller/mod.rs:291, query_pdb.rs:348,415) are safe and run one at a time per context — one stream, one scratch set;
    // callers that want overlap create one context per thread.  Recursive: entry points call each other.
    mutable std::recursive_mutex mu;
    int host_libm_matches = -1;   // 1: this host's libm agrees with the glibc generation the device arithmetic restates, 0: it does not
    unsigned long long *spec_miss = nullptr;   // device counter: pairs the speculative torsion path re-evaluated exactly
    std::vector<fd_timing_entry> timings;
    std::vector<hipEvent_t> event_pool;
    size_t event_used = 0;
    // pinned staging of large device-to-host copies (fd_d2h_big in fdgpu_api.hip): FD_PIN_SLOTS buffers of FD_PIN_BYTES, made on first use
    // pinned host buffers that outlive a call (the packed candidate pairs of a whole-structure retrieval: 2 x ~100 MB per call — as
    // malloc'd blocks their first-touch page faults and their munmap cost more than the copy)
    hipStream_t side_stream = nullptr;      // second stream of the fused query call: the ranked records travel to the host while the retrieval's kernels run
    void *hbuf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // 0, 1: retrieval; 2, 3: landing / upload blocks of the query-map stage; 4: its pair features; 5: their (i, j) upload
    size_t hbuf_cap[6] = {0, 0, 0, 0, 0, 0};
    void *host_pinned(int k, size_t bytes) {
        if (hbuf_cap[k] >= bytes) return hbuf[k];
        if (hbuf[k]) (void)hipHostFree(hbuf[k]);
        hbuf[k] = nullptr; hbuf_cap[k] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&hbuf[k], want, hipHostMallocDefault) != hipSuccess) { hbuf[k] = nullptr; return nullptr; }
        hbuf_cap[k] = want;
        return hbuf[k];
    }
    void *pin[16] = {};
    hipEvent_t pin_ev[16] = {};
    // device blocks of destroyed indices, reused by the next build (steady-state builds do not call
    // hipMalloc/hipFree, which would serialise the stream)
    struct pooled { void *p; size_t cap; };
    std::vector<pooled> pool;
    void *pool_alloc(size_t bytes, hipError_t *err) {
        *err = hipSuccess;
        size_t best = (size_t)-1;
        for (size_t k = 0; k < pool.size(); ++k)
            if (pool[k].cap >= bytes && pool[k].cap <= bytes + bytes / 4 + 4096 && (best == (size_t)-1 || pool[k].cap < pool[best].cap)) best = k;
        if (best != (size_t)-1) { void *p = pool[best].p; last_cap = pool[best].cap; pool.erase(pool.begin() + best); return p; }
        void *p = nullptr;
        size_t want = bytes + bytes / 32 + 256;
        *err = hipMalloc(&p, want);
        if (*err != hipSuccess) {  // drop the cache and retry once
            pool_drop();
            (void)hipGetLastError();
            *err = hipMalloc(&p, want);
            if (*err != hipSuccess) return nullptr;
        }
        last_cap = want;
        return p;
    }
    void pool_drop() { for (auto &b : pool) (void)hipFree(b.p); pool.clear(); }
    void pool_free(void *p, size_t cap) {
        if (!p) return;
        if (pool.size() >= 96) { (void)hipFree(p); return; }   // a shard built as 8 sub-indices + their merge cycles through ~40 blocks per step
        pool.push_back({p, cap});
    }
    size_t last_cap = 0;
    bool counted = false;         // fdgpu_create got as far as a device + stream (live-context count behind fdgpu_trim at the last destroy)
    const void *mp_bintab_at = nullptr; size_t mp_bintab_cap = 0;      // the WS_MP_Q allocation whose first 256 bytes hold the pair drain's bin tables
    fd_host_pool host_pool;       // helper threads of the host glue (made on first use)
    bool is_lane = false;         // a query lane's private context: its short host loops stay on the lane's thread (the other lanes' queries are the parallelism)
    // helper threads for a short host loop of a large query: n on a caller's own context, 1 (inline) on a lane
    unsigned small_par(unsigned n) const { return is_lane ? 1u : std::min(n, std::max(1u, std::thread::hardware_concurrency())); }
    void *lanes = nullptr;        // fd_lanes.hip: sibling contexts + worker threads behind fdgpu_query_batch_submit / _wait (made on first use)
};
void fd_lanes_destroy(fdgpu_ctx *c);

// HIP-event stage timer (active only after fdgpu_enable_timing(ctx, 1))
static inline hipEvent_t fd_next_event(fdgpu_ctx *c) {
    if (c->event_used == c->event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->event_pool.push_back(e);
    }
    return c->event_pool[c->event_used++];
}
struct StageTimer {
    fdgpu_ctx *c;
    size_t idx = (size_t)-1;
    StageTimer(fdgpu_ctx *ctx, const char *name, uint64_t bytes) : c(ctx) {
        if (!c || !c->timing) return;
        fd_timing_entry t;
        t.name = name; t.bytes = bytes; t.ev0 = fd_next_event(c); t.ev1 = fd_next_event(c);
        if (!t.ev0 || !t.ev1) return;
        (void)hipEventRecord(t.ev0, c->stream);
        c->timings.push_back(t);
        idx = c->timings.size() - 1;
    }
    ~StageTimer() { if (idx != (size_t)-1) (void)hipEventRecord(c->timings[idx].ev1, c->stream); }
};

// the current HIP device is per host thread: a caller thread that never selected the context's device gets it here
#define FD_LOCK(c) std::unique_lock<std::recursive_mutex> _fd_lk; if (c) { _fd_lk = std::unique_lock<std::recursive_mutex>((c)->mu); (void)hipSetDevice((c)->device); }

struct fdgpu_batch {
    fdgpu_ctx *ctx = nullptr;
    bool owns = false;
    uint64_t n_struct = 0, n_res = 0;
    uint32_t n_work = 0;
    // device
    float *n_xyz = nullptr, *ca_xyz = nullptr, *cb_xyz = nullptr;
    uint8_t *aa = nullptr, *cb_valid = nullptr, *hash_ok = nullptr;
    uint32_t *res_off = nullptr, *wi_struct = nullptr, *wi_i0 = nullptr;
    std::vector<uint64_t> h_res_off;  // host copy
    fd_batch_view view() const {
        fd_batch_view v;
        v.n_xyz = n_xyz; v.ca_xyz = ca_xyz; v.cb_xyz = cb_xyz; v.aa = aa; v.hash_ok = hash_ok; v.res_off = res_off;
        v.wi_struct = wi_struct; v.wi_i0 = wi_i0; v.n_struct = (uint32_t)n_struct; v.n_work = n_work;
        return v;
    }
};

#include <atomic>
static inline uint64_t fd_next_index_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
struct fdgpu_index {
    fdgpu_ctx *ctx = nullptr;
    uint64_t n_hashes = 0, value_len = 0, n_postings = 0, n_structures = 0, first_id = 0;
    uint32_t *hashes = nullptr;   // device [H]
    uint64_t *offsets = nullptr;  // device [H+1]
    uint8_t *value = nullptr;     // device [value_len]
    uint32_t *last_ids = nullptr; // device [H] last structure id of every list (written by the encoder; the device merge re-bases the next
                                  // part's first delta against it); null for an index that was loaded — computed on demand
    size_t cap_hashes = 0, cap_offsets = 0, cap_value = 0, cap_last = 0;
    mutable uint32_t *lens = nullptr;   // device [H] posting length of every list, made on the first length request (fd_posting_lengths_dev)
    mutable std::mutex lens_mu;         // several contexts (streams) may share one index: the first one fills lens, complete before it is published
    uint64_t uid = fd_next_index_uid();   // identifies the index in what query maps remember about it (an address can be reused)
    float *penalty = nullptr;     // device [n_structures] length penalty set with fdgpu_index_set_penalty (count queries may then pass NULL)
    // checkpoints of the posting lists (k_qtile.hip): where a list can be entered at a boundary of structure ids.  Derived, device-only,
    // made on the first tiled scoring call for the (first_id, n_structures) the index has then (lens_mu guards them like lens)
    mutable unsigned long long *ck_meta = nullptr;   // [H] first entry of the list | stride log2 << 56
    mutable void *ck_ent = nullptr;                  // uint2 entries {byte offset in the list, id before}
    mutable uint64_t ck_n = 0, ck_first = 0, ck_S = 0;
    mutable bool ck_failed = false;                  // the table did not fit for the id range (ck_first, ck_S): the tiled path is off until the range changes or a retry succeeds
    mutable uint32_t ck_fail_skips = 0;
};

// batched scoring with the ranked selection left on the device (fdgpu_api.hip; consumed by the sharded query, fd_comm.hip)
struct fd_cq_dev_out {
    bool got = false, overflow = false; const void *recs = nullptr; const void *state = nullptr; uint32_t top_n = 0, cap = 0;
    std::vector<uint32_t> counts;                          // got: records selected per query (before the cut to top_n)
    const std::function<void()> *while_running = nullptr;  // host work of the caller, run once between the scoring launches and the wait for them
    uint32_t head_n = 0;                                   // in: the first head_n records of every query's ranking are wanted on the host as well ...
    const fd_count_rec *head = nullptr;                    // ... got: [n_queries][min(head_n, top_n)] in the context's page-locked block 3 (same wait as the state), or null
};
int fd_count_query_batch_impl(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off, const uint32_t *q_hash,
                              const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf, const float *penalty, uint32_t top_n,
                              fd_count_rec **out, uint64_t **out_off, bool allow_dense, fd_cq_dev_out *dev, int64_t known_segments = -1,
                              const long long *known_kidx = nullptr, const uint64_t *known_len = nullptr);
int fd_posting_lengths_dev(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t **dev_lengths);
uint64_t fd_maps_hashes(uint64_t n_queries, const fd_query_map *const *qms, std::vector<uint32_t> &h);
int fd_count_query_maps_len(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, const uint64_t *len,
                            const uint64_t *primary_len, const float *penalty, float total_structures, uint32_t top_n, fd_count_rec **out,
                            uint64_t **out_off, fd_cq_dev_out *dev, const uint32_t *seg = nullptr, const long long *kidx = nullptr, bool allow_dense = true);
int fd_posting_lengths_segs(fdgpu_ctx *c, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t nq, uint64_t *lengths, uint32_t *segs, long long *kidx = nullptr,
                            const uint8_t **land_out = nullptr);
// fdgpu_count_query_maps_top with the device-resident form of its result (dev != null: see fd_count_query_batch_impl)
int fd_count_query_maps_top_impl(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, const float *penalty, float total_structures,
                                 uint32_t top_n, fd_count_rec **out, uint64_t **out_off, fd_cq_dev_out *dev, bool allow_dense = true);
// (allow_dense = false: straight to the compacting path — what a caller passes when the device selection of a first attempt overflowed; the
// overflow is deterministic, a second attempt on the same path would overflow again before falling back)

// fdgpu_retrieve_batch with its ordered result left ON THE DEVICE (the sharded retrieval gathers it from there, fd_comm.hip): when the device glue
// produced the records (motif-sized queries), got = true, recs / residues point into the context's workspaces (valid until its next retrieval),
// *matches / *residues of the call stay NULL and only the per-query offsets come back on the host.  got = false: the host glue ran, host arrays as usual.
struct fd_rb_dev_out { bool got = false; const void *recs = nullptr; const int32_t *residues = nullptr; uint64_t n_recs = 0, n_res = 0; };
int fd_retrieve_batch_dev(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand, const uint64_t *cand_off,
                          const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct, const fd_hash_params *p, float ca_distance_cutoff,
                          uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off, fd_rb_dev_out *dev);
hipError_t fd_d2h_to_file(fdgpu_ctx *c, int fd, uint64_t file_off, const void *src, size_t bytes, std::atomic<int> *io_err);      // fdgpu_api.hip: device array -> file region through the pinned slots
void *fd_out_alloc(size_t bytes, bool pinned = false);      // result arrays of the hot query paths: recycled blocks (fdgpu_api.hip); released with fdgpu_free like any other output

// kernels / launchers implemented in the k_*.hip files
void fd_launch_selfcheck(const fd_quant &q, uint32_t *out, hipStream_t st);
void fd_launch_hash_ok(const uint8_t *aa, const uint8_t *cb_valid, uint8_t *ok, uint64_t n, hipStream_t st);
void fd_launch_pair_count(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st);
void fd_launch_pair_emit(const fd_batch_view &B, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor, uint32_t *keys,
                         uint32_t *ids, uint32_t first_id, hipStream_t st);
void fd_launch_frames(const fd_batch_view &B, uint64_t n_res, void *frames, hipStream_t st);
void fd_launch_pair_count2(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st);
void fd_launch_pair_emit2(const fd_batch_view &B, const void *frames, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor,
                          uint32_t *keys, void *ids, bool ids16, uint32_t first_id, hipStream_t st);
int fd_radix_sort_pairs16(uint32_t *keys_a, uint16_t *vals_a, uint32_t *keys_b, uint16_t *vals_b, uint64_t n, int key_bits, uint32_t *ghist,
                          uint64_t *tot, hipStream_t st, fdgpu_ctx *timing_ctx = nullptr);
void fd_launch_aa_check(const fd_batch_view &B, uint64_t n_res, unsigned long long *wide_flag, hipStream_t st);
void fd_launch_frames_perm(const fd_batch_view &B, void *frames, float *ca_perm, uint8_t *ok_perm, uint8_t *aa_perm, unsigned long long *wide_flag, hipStream_t st);
void fd_launch_pair_count_msd(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st);
void fd_launch_pair_emit_msd(const fd_batch_view &B, const void *frames, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor, uint32_t *keys,
                             uint16_t *ids, hipStream_t st);
uint32_t fd_rs_seg_num_tiles(uint64_t n, uint32_t n_seg);
uint64_t fd_rs_seg_tot_words(uint64_t n, uint32_t n_seg);
size_t fd_rs_seg_tab_bytes(uint64_t n, uint32_t n_seg);
int fd_radix_sort_pairs16_seg(uint32_t *keys_a, uint16_t *vals_a, uint32_t *keys_b, uint16_t *vals_b, uint64_t n, const uint64_t *seg_off, uint64_t stride,
                              uint32_t n_seg, int shift0, int passes, uint32_t *ghist, uint64_t *tot, void *seg_tab, hipStream_t st, fdgpu_ctx *timing_ctx = nullptr,
                              unsigned long long *overflow = nullptr);      // overflow: set when a (bucket, digit) run reaches 2^32 keys (the result is then invalid)
void fd_launch_row_count(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *row_cnt, float cutoff, hipStream_t st);
void fd_launch_row_emit(const fd_batch_view &B, const fd_hash_consts &C, const uint64_t *row_off, uint32_t *keys, float cutoff, uint32_t *ids,
                        uint32_t first_id, hipStream_t st, int ids_partner = 0);
template <typename TIn>
void fd_exclusive_scan(const TIn *in, uint64_t n, uint64_t *out, uint64_t *chunk_tmp, uint64_t *total_dev, hipStream_t st);
uint64_t fd_scan_tmp_elems(uint64_t n);
uint32_t fd_rs_num_tiles(uint64_t n);
void fd_rs_set_variant(int v);
int fd_radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint64_t n, int key_bits, uint32_t *ghist,
                        uint64_t *tot, hipStream_t st, fdgpu_ctx *timing_ctx = nullptr);
uint32_t fd_enc_num_tiles(uint64_t n);
void fd_launch_enc_sizes(const uint32_t *keys, const void *ids, int codec, uint32_t first_id, uint64_t n, uint32_t *tb, uint32_t *th, uint32_t *tp,
                         const uint64_t *seg_off, uint64_t S, uint64_t *bo, hipStream_t st);
void fd_launch_enc_write(const uint32_t *keys, const void *ids, int codec, uint32_t first_id, uint64_t n, const uint64_t *tbo, const uint64_t *tho,
                         uint8_t *value, uint32_t *hashes, uint64_t *offsets, uint32_t *last_ids, const uint64_t *total_bytes_dev, uint64_t H,
                         const uint64_t *bo, hipStream_t st);
void fd_launch_uniq_flags(const uint32_t *keys, const uint32_t *ids, uint64_t n, uint8_t *flags, hipStream_t st);
void fd_launch_compact(const uint32_t *keys, const uint8_t *flags, const uint64_t *pos, uint64_t n, uint32_t *out, hipStream_t st);
void fd_launch_gather_u64(const uint64_t *src, const uint64_t *idx, uint64_t n, uint64_t *dst, hipStream_t st);

// k_query.hip
struct cq_args {
    const uint32_t *hashes; const uint64_t *offsets; const uint8_t *value; uint64_t H;
    const uint32_t *q_hash; uint64_t nq;       // the query hashes of the call, per query sorted by (node, partner): row k of hash_bits belongs to q_hash[k]
    uint32_t *hash_bits;                       // [nq][words] occupancy: bit s of row k = structure s holds hash k
    const unsigned long long *row_meta;        // [nq] idf (2^-22 fixed point) << 2 | last row of its node << 1 | last row of its edge
    uint32_t *match;                           // [queries][S] match counts — wide form only
    unsigned long long *idf;                   // [queries][S] idf sums (2^-22 units); packed form: count << 46 | sum
    int packed;
    uint32_t words; uint32_t first_id; uint32_t S;
};

// k_qtile.hip: tiled scoring of motif batches (posting lists entered per tile of structures, scores in LDS)
#define QT_CELL_LOG2 11                    /* checkpoint granule: 2,048 structure ids — the piece of a list one (row, tile) work unit is cut into */
/* structures per workgroup (tile): 2^13 (64 KB of LDS accumulators, two workgroups per CU) or 2^14; qt_args.tile_log2 */
#define QT_BINS 2048
#define QT_MAX_ROWS 1024                   /* rows per query the survivors' row bits are laid out for */
#define QT_MAXB 16                         /* row batches of a (query, tile) the decoded stream keeps a record of */
struct qt_state { uint32_t thr_bin, above, thr_key, count; };      // same 16 bytes as topn_state / sel_state: consumers read .count
struct qt_aux { uint32_t need_l2, shift2, edge, pad; };
struct qt_args {
    const uint8_t *value; const uint64_t *offsets;
    const unsigned long long *ck_meta; const uint2 *ck_ent;
    const long long *kidx;                 // [nq] list of every row's hash in the index, -1 = absent
    const unsigned long long *row_meta;    // [nq] idf (2^-22 fixed point) << 2 | last row of its node << 1 | last row of its edge
    const uint64_t *q_rows;                // [n_queries + 1] row ranges of the queries
    const float *penalty;                  // [S]
    uint32_t nq, n_queries, S, first_id, NT, NC, tile_log2, plan_log2;      // plan_log2: structure ids per plan granule (QT_CELL_LOG2, or tile_log2 for one large query)
    uint4 *ranges;                         // [NC][nq] byte range of (row, cell): first byte lo / hi, bytes (0: decoded with an earlier cell of the tile), id before the first posting
    uint32_t *c_nid, *c_key;               // [n_queries][NT][tile] each: structure and ranking key of the touched structures (any order; two arrays:
                                           // the structure is written when it is first met, the key at the tile's end — full lines either way)
    uint32_t *ccount;                      // [n_queries][NT] entries of compact
    uint32_t *ghist;                       // [n_queries][QT_BINS], zero on entry and left zero
    qt_state *state; qt_aux *aux;          // [n_queries]
    void *out; uint32_t cap;               // fd_count_rec [n_queries][cap]
    // the decoded stream (motif batches, optional): pass A leaves every 16-byte slot of a tile's posting ranges as sixteen 16-bit structure ids
    // + its row, pass B (k_qt_rows) tests them against the survivors instead of decoding the lists a second time
    void *stream_ids; uint16_t *stream_row;        // [stream_cap][16] u16 (0xffff: none), [stream_cap]
    uint2 *stream_tab;                     // [n_queries][NT][QT_MAXB] first record and records of a (query, tile, row batch)
    uint32_t *stream_used; uint32_t stream_cap;    // records claimed so far (zero on entry), records the buffers hold
    unsigned long long *dbg;               // optional (FDGPU_QT_DBG): [16] phase durations summed over the workgroups
    // the 32-bit path (k_qscore32.hip): k_qt_layout leaves, per (query, tile), its non-empty pieces laid out as a stream of 16-byte slots in
    // 64-slot WINDOWS (a piece of <= 64 slots never straddles a window), k_qt_score32 gives every wavefront whole windows
    uint4 *pieces;                         // [nq x NC] region of (query, tile): {first byte lo, first byte hi (16 bits) | row << 16, bytes, id before}
    uint32_t *piece_p;                     // [nq x NC] first slot of the piece inside its (query, tile) stream
    uint32_t *win;                         // per (query, tile) [rows x win_per_row + 2]: first piece that ends behind the window's first slot | QT_WIN_CONT
    uint4 *heads;                          // [n_queries][NT] {pieces, windows, first record of the decoded stream, flags (1: a table did not hold the tile)}
    uint32_t win_per_row, top_n;
    uint32_t max_rows;                     // rows of the batch's largest query (0: unknown) — k_qt_rows picks its table sizes by it
    // one query of ~10^5 rows (k_qt_score<..., BIG>): row slices, per-slice sums, the survivors' bitmap / slots / row bits
    const uint64_t *slices; uint32_t n_slices;     // [n_slices + 1] row boundaries
    unsigned long long *partial;           // [n_slices][NT][tile] count << 46 | idf sum
    uint32_t *g_bm, *g_rank, *g_tcount, *g_nid;    // [NT][tile / 32] survivors and their slots, [NT] survivors per tile, [cap] structure of a slot
    uint32_t *g_rowbits; uint32_t g_wpr;   // [cap][g_wpr] (row, survivor) bits
    const uint32_t *g_eend, *g_nend;       // [g_wpr] rows that end an edge / a node
};
void fd_launch_ck_count(const uint64_t *offsets, uint64_t H, uint32_t NC, uint32_t *cnt, hipStream_t st);
void fd_launch_ck_fill(const uint64_t *offsets, const uint8_t *value, uint64_t H, uint32_t NC, uint32_t S, uint32_t first_id, const uint64_t *ent_off,
                       unsigned long long *meta, void *ent, hipStream_t st);
void fd_launch_qt_plan(const qt_args &A, hipStream_t st);
void fd_launch_qt_score(const qt_args &A, hipStream_t st);
void fd_launch_qt_big_score(const qt_args &A, hipStream_t st);
void fd_launch_qt_big_select(const qt_args &A, uint32_t top_n, void *sorted, hipStream_t st);
void fd_launch_qt_select(const qt_args &A, uint32_t top_n, void *sorted, hipStream_t st);
#define QT_WIN_CONT 0x80000000u            /* the window begins inside a piece of more than 64 slots: decoded by the wavefront that decoded the window before */
void fd_launch_qt_layout(const qt_args &A, hipStream_t st);
void fd_launch_qt_score32(const qt_args &A, hipStream_t st);

void fd_launch_posting_lengths(const uint32_t *hashes, const uint64_t *offsets, const uint8_t *value, uint64_t H, const uint32_t *q_hash,
                               uint64_t nq, uint64_t *lengths, long long *kidx, uint32_t *nseg, uint64_t *wstart, uint64_t *scan_tmp, uint64_t *total,
                               hipStream_t st);
void fd_launch_index_lens(const uint64_t *offsets, const uint8_t *value, uint64_t H, uint32_t *lens, hipStream_t st);
void fd_launch_posting_lookup(const uint32_t *hashes, const uint64_t *offsets, const uint32_t *lens, uint64_t H, const uint32_t *q_hash, uint64_t nq, uint64_t *lengths,
                              uint32_t *nseg, long long *kidx, hipStream_t st);
void fd_launch_cq_plan(const cq_args &A, long long *kidx, uint32_t *nseg, hipStream_t st);
void fd_launch_cq_seg(const cq_args &A, const long long *kidx, const uint64_t *wstart, uint32_t *segsum, uint64_t n_items, bool split,
                      hipStream_t st);
void fd_launch_cq_rows_finalize(const cq_args &A, const uint64_t *q_rows, uint32_t n_queries, const uint64_t *slices, uint32_t n_slices, uint32_t *node_cnt,
                                uint32_t *edge_cnt, uint8_t *flags, uint64_t max_rows_per_query, hipStream_t st);
void fd_launch_cq_compact(const uint32_t *match, const unsigned long long *idf, const uint32_t *node_cnt, const uint32_t *edge_cnt,
                          const uint8_t *flags, const uint64_t *pos, const float *penalty, uint32_t S, uint32_t first_id, void *out,
                          hipStream_t st);

// k_match.hip
struct mp_query_dev {   // per-query table of the pair scan (many queries in one launch)
    uint32_t qh_off, n_hashes;   // sorted unique hashes: q_hashes[qh_off .. +n_hashes)
    uint32_t aad_off, n_aad;     // grouped aa_dist_map: aad_dist / aad_qi [aad_off .. +n_aad); start table aad_start[1025 * query]
    uint32_t aa1_mask, aa2_mask; // residue types of the hashes' first / second residue (prefilter_amino_acid)
    int use_prefilter;
    float ca_window;
    uint32_t qs_off, qs_mask;    // more than MP_QH_LDS hashes: an open-addressing set of them, qset[qs_off .. + qs_mask + 1) (0xffffffff = empty); qs_mask = 0: none
};
#define MP_QH_LDS 1024          /* query hashes a drain copies into LDS for its membership test; larger sets are probed in their hash table (qset) */
#define MP_SUBQ_STRIDE 16u      /* u64 words between the pair queue's sub-queue counters (one 128-byte line each) */
struct mp_args {
    fd_batch_view B;
    fd_hash_consts C;
    float cutoff;
    const uint32_t *cand; uint32_t n_cand;
    const uint32_t *wi_cand; const uint32_t *wi_i0; const uint32_t *wi_query; uint32_t n_work;
    const uint32_t *wi_j0; uint32_t j_span;   // j_span != 0: a work item scans partner residues [wi_j0, wi_j0 + j_span) only (few, long candidates)
    const mp_query_dev *qtab;
    const uint32_t *qset;        // the large queries' hash sets (null: none)
    const uint8_t *resname_std;
    uint32_t aa1_mask, aa2_mask;
    int use_prefilter;
    const uint32_t *q_hashes; uint32_t n_hashes;
    const uint32_t *aad_start;   // [1025] start of the (aa_i * 32 + aa_j) group in aad_dist / aad_qi (stable order)
    const float *aad_dist; const uint32_t *aad_qi; uint32_t n_aad;
    const uint32_t *iv_start; const float2 *iv;   // optional (queries too large for the LDS copy of their distances): merged pass intervals per group, [1025 * query] offsets into iv
    const uint32_t *iv_grp;                       // with iv_start: [1024 * query] (group's first interval - query's first) << 8 | min(count, 255): the scan's LDS copy
    float ca_window;
    unsigned long long *n_found, *n_cands;
    const uint4 *cinfo; const uint32_t *act;      // optional (items written on the device): per candidate {first residue, end, active residues | full << 31, first list entry}, the lists
    unsigned long long *dbg;       // FDGPU_MP_DBG: [8] live items, their ticks, early exits, their ticks, chunks drained, their ticks, partners visited, pairs queued; else null
    // the queue between k_mp_scan and k_mp_drain: chunks of up to 64 packed pairs in 64 sub-queues of cap_subq chunks (sub-queue s: chunks
    // [s * cap_subq, ...), q_cnt[s] claimed — beyond cap_subq: counted only), their headers {slot, pairs | query << 8, first residue, end}.  In chunk
    // order v (sub-queue by sub-queue): the drain's results per pair, the chunk's {candidate-pair records, found triples}, their first positions
    unsigned long long *q_cnt; uint4 *chunk_hdr; uint32_t *chunk_ij; uint32_t cap_subq;
    uint32_t *res_h, *res_meta; float *res_d; uint2 *chunk_cnt; ulonglong2 *chunk_base;
    const uint32_t *bintab;        // [64] fd_fill_bintab's table + its clamped copy at [32, 59) (fdgpu_api.hip fills it once per workspace)
    int compact;                   // every query of the launch observes <= 1,024 distances and needs no interval table: the scan's small LDS layout
    fd_pair_rec *found; fd_cand_rec *cands;
    unsigned long long cap_found, cap_cands;   // records the buffers hold (EMIT counts beyond them without writing)
    uint32_t n_cfg;                // --multiple-bins: bin pairs to hash every surviving pair with (1 = the single configuration in C)
    fd_quant qk[8];                // their quantisers (qk[0] == C.q)
    uint32_t mode;                 // bit 0: emit found triples, bit 1: emit candidate pairs
    const uint32_t *cj_mask;       // optional: only partner residues j whose bit mask_off[slot] + (j - r0) is set are scanned
    const uint32_t *mask_off;      // [n_cand] first bit of every candidate slot
    // mode bit 5 (rescue votes on the device, second scan of a large query): instead of leaving the kernel, a candidate pair (qi, i, j) adds one
    // to votes[vt_off[slot] + ((cj_comp[bit of j] - 1) * vt_qs[slot] + qi) * n_residues(slot) + i] — the table retrieve.rs:498-511 builds per component
    uint32_t *votes; const uint64_t *vt_off; const uint32_t *vt_qs; const uint8_t *cj_comp;
    const float *sd_dist; const uint32_t *sd_qi;      // optional, vote mode: every group's observed (distance, query residue) list sorted by distance
};
// rescue votes of a large query's second scan (fd_match_pairs_multi, mode bit 5).  In: per marked partner residue (bit position as in cj_mask) the
// 1-based ordinal of the component that mapped it, per slot the first counter and the query's residue count, the table's size in counters.
// Out: per (slot, ordinal, query residue) row of the table {largest count, number of target residues holding it, one of them}.
// host block of a pair scan's work items and query tables (fd_match_pairs_multi builds it on the first call, reuses it on the next)
struct fd_mp_tables {
    std::vector<uint32_t> blk; size_t o[13] = {0}; size_t nw = 0; bool want_iv = false; uint32_t j_span = 0; bool valid = false;
    const uint32_t *data = nullptr; size_t words = 0;      // the packed block: blk, or (tables not kept by the caller) the context's pinned staging buffer
    bool dev_items = false; size_t o_wb = 0;                // work items written on the device from [first item | query] per candidate at o_wb (one-off blocks)
    uint64_t qset_slots = 0, qset_max_hashes = 0;           // slots of the large queries' hash sets (mp_query_dev.qs_off / qs_mask), their largest hash count
};
// work items of the pair scan: candidate k (structure cand[k], first item wbase[k], query cq[k]) -> one item per (64-residue tile, span of j_span partners)
void fd_launch_mp_items(const uint32_t *db_res_off, const uint32_t *cand, uint32_t n_cand, const uint32_t *wbase, const uint32_t *cq, uint32_t j_span, uint32_t *wc,
                        uint32_t *wi, uint32_t *wq, uint32_t *wj, hipStream_t st, const uint8_t *aa, const uint8_t *hash_ok, const uint8_t *resname_std, int tert,
                        const struct mp_query_dev *qtab, void *cinfo /* uint4 [n_cand] or null */, uint32_t *act /* [64 x items] */);
struct fd_vote_row { uint32_t mx, nmx, arg; };
struct fd_vote_plan {
    const uint8_t *cj_comp; uint64_t n_bits;
    const uint64_t *vt_off; const uint32_t *vt_qs; uint64_t n_counters;
    const uint64_t *row_off; const uint32_t *row_len; uint64_t n_rows;      // every row's first counter and length (the slot's residue count)
    fd_vote_row *rows;                                                       // [n_rows] host, filled by the call
    // optional: the queries' observed-distance lists in the scan's group layout (aad_start offsets), every group sorted by distance — the
    // entries inside a pair's window are then one contiguous run (fl(d - x) is monotone in x) instead of a walk over the whole group
    const float *sd_dist; const uint32_t *sd_qi; uint64_t n_sd;
};
void fd_launch_vote_rows(const uint32_t *votes, const uint64_t *row_off, const uint32_t *row_len, uint64_t n_rows, fd_vote_row *out, hipStream_t st);
void fd_launch_match_pairs(const mp_args &A, hipStream_t st);
void fd_launch_mp_qset_build(const mp_query_dev *qtab, uint32_t n_queries, uint32_t max_hashes, const uint32_t *q_hashes, uint32_t *qset, hipStream_t st);
void fd_launch_found_key_ij(const fd_pair_rec *f, uint64_t n, uint32_t *key, uint32_t *val, hipStream_t st);
void fd_launch_found_key_slot(const fd_pair_rec *f, const uint32_t *val, uint64_t n, uint32_t *key, hipStream_t st);
void fd_launch_found_gather(const fd_pair_rec *f, const uint32_t *val, uint64_t n, fd_pair_rec *out, hipStream_t st);
void fd_launch_pack_cands(const fd_cand_rec *c, uint64_t n, uint32_t *key, uint32_t *val, hipStream_t st);
void fd_launch_kabsch(const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot, float *tran, hipStream_t st, uint64_t n_points = 0);
void fd_launch_metrics(const float *ref, const float *mov, const uint64_t *off, uint64_t n, const float *rot, const float *tran, const float *d0, float *out,
                       hipStream_t st, uint64_t n_points = 0);      // n_points: the problems' points in all (0 = not known) — small problems run four to a wavefront
void fd_launch_lms_qcp(const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot, float *tran, uint32_t *core_len,
                       uint8_t *flags, uint32_t *order, hipStream_t st);
void fd_launch_cq_compact_batch(const cq_args &A, const uint32_t *node_cnt, const uint32_t *edge_cnt, const uint8_t *flags, const uint64_t *pos,
                                const float *penalty, uint64_t total, void *out, hipStream_t st);
void fd_launch_cq_topn(const void *recs, const uint64_t *off, uint32_t n_queries, uint32_t top_n, uint32_t cap, void *out, void *state, uint32_t *ghist,
                       hipStream_t st);
void fd_launch_cq_topn_acc(const cq_args &A, const float *penalty, const uint32_t *node_cnt, const uint32_t *edge_cnt, uint32_t n_queries, uint32_t top_n,
                           uint32_t cap, void *out, void *state, uint32_t *ghist, hipStream_t st);
void fd_launch_cq_topn_dense(const cq_args &A, const uint64_t *q_rows, const float *penalty, uint32_t *keys, uint32_t n_queries, uint32_t top_n, uint32_t cap,
                             void *out, void *state, uint32_t *ghist, hipStream_t st);
void fd_launch_cq_topn_sort(const void *sel, uint32_t cap, const void *state, uint32_t n_queries, uint32_t top_n, void *out, hipStream_t st);
void fd_launch_get_entries(const uint32_t *hashes, const uint64_t *offsets, const uint8_t *value, uint64_t H, const uint32_t *q_hash, uint64_t nq,
                           const uint64_t *out_off, uint32_t *out, hipStream_t st);
// k_retrieve.hip: retrieval glue on the device (graph -> components -> votes -> assignment -> rescue -> superposition problems)
struct rs_query_dev {
    uint32_t qh_off, n_hashes;   // sorted unique hashes of the query: hashes[qh_off ..), kfirst / sym parallel to them
    uint32_t map_off;            // its query-map entries: map_qi / map_qj / map_idf [map_off ..)
    uint32_t idx_off, n_idx;     // all_query_indices
    uint32_t q_size;             // 1 + largest query residue index any entry names
    uint32_t q_res0;             // first residue of the query structure in the query batch
    uint32_t pad;
};
struct rs_match_dev { uint32_t slot, ci, same, res_pos, prob0, prob1; float idf; uint32_t ord; };      // ord: the record's place among its slot's records (components ascend)
#define RS_CNT_STRIDE 512      // u64 words between the slots kernel's three allocation counters
struct rs_args {
    const fd_pair_rec *found; const fd_cand_rec *cands;
    const uint32_t *seg_f, *seg_c, *perm_f, *perm_c;     // per-slot segments of the (unordered) scan output
    const uint32_t *cand, *slot_q;                       // slot -> structure of the database batch, slot -> query
    const uint32_t *order;                               // launch order of the slots, heaviest first (k_rs_order; null: slot order)
    uint32_t *slot_matches;                              // records every slot wrote (zeroed before the launch; null: not kept)
    unsigned long long *dbg;                             // FDGPU_RS_DBG: phase clocks summed over the slots (8 counters), else null
    uint4 *dbg_slot;                                     // FDGPU_RS_DBG: per slot {found triples, candidate pairs, components, 100 MHz ticks}, else null
    const uint32_t *db_res_off; const float *db_ca, *db_cb, *q_ca, *q_cb;
    const rs_query_dev *qt;
    const uint32_t *hashes, *kfirst; const uint8_t *sym;
    const uint32_t *map_qi, *map_qj; const float *map_idf;
    const uint32_t *indices;
    const float *d0tab;                                  // d0_scale of metrics.rs:117-123 by point count (host powf)
    uint32_t node_count;
    uint32_t node_cap;                                   // graph nodes per candidate the kernel accepts (64 lanes; tests lower it to reach the fallback)
    unsigned long long *counters;                        // [0] matches, [RS_CNT_STRIDE] problems << 40 | points, [2 * RS_CNT_STRIDE] residue ints: 4 KB apart —
                                                         // on one cache line the three returning atomics of 10^4 records queued behind each other (27 us per slot)
    uint32_t *flags;                                     // bit 0: a slot beyond the kernel's limits, bit 1: an output buffer too small
    rs_match_dev *matches; int32_t *residues; float *kx, *ky; uint64_t *koff; float *d0;
    uint32_t *gq, *gr;                                   // per residue pair of a superposition problem (= two points, [CA, CB]): query / target residue, absolute in qb / db —
                                                         // k_rs_points gathers the coordinates into kx / ky afterwards, with every pair its own thread (a slot is one serial wavefront)
    uint64_t cap_matches, cap_res, cap_prob, cap_pts;
    // the split form (k_rs_setup + k_rs_comp; null sp_work: k_rs_slots alone): what a slot's set-up leaves for its components
    uint4 *sp_edges;                                     // [found triples] per edge, at its slot's segment: nodes | symmetric | in the map, query residues, idf
    uint32_t *sp_nodes;                                  // [slots][64] residue of node v
    unsigned long long *sp_comps;                        // [slots][128] the components' node sets, in the reference's order
    uint4 *sp_head;                                      // [slots][2] {triples, components, nodes, -}, {first record lo / hi, first residue int lo / hi}
    uint2 *sp_work;                                      // [cap_matches] (slot, component) of record k
    uint4 *sp_np;                                        // [cap_matches] {problems, points, assigned pairs, rescued-list pairs} of record k
    uint32_t *sp_gq, *sp_gr;                             // [cap_matches][128] its residue pairs (query / target residue), first problem then second
    uint32_t *sp_big, *sp_big_n;                         // [slots] the slots left to k_rs_slots, their number (zero on entry)
    const uint32_t *n_listed;                            // k_rs_slots: slots in order[] (null: the grid)
};
void fd_launch_rs_group(const fd_pair_rec *found, uint64_t nf, const fd_cand_rec *cands, uint64_t nc, uint32_t n_cand, uint32_t *cnt, uint32_t *seg, uint32_t *cur,
                        uint32_t *perm_f, uint32_t *perm_c, hipStream_t st);
void fd_launch_rs_slots(const rs_args &A, uint32_t n_cand, hipStream_t st);
void fd_launch_rs_points(const rs_args &A, uint64_t n_prob, uint64_t n_points, hipStream_t st);      // the [CA, CB] point lists of the problems k_rs_slots described and koff[n_prob] (before k_superpose / k_metrics)
void fd_launch_rs_records(const void *matches, const void *plan, uint64_t n, const float *rmsd, const float *rot, const float *tran, const float *met,
                          const int32_t *residues, void *out, int32_t *out_res, hipStream_t st);
// the same with the order made on the device: slot_matches -> per-slot bases, per-query offsets (match_off / res_off, n_queries + 1 each) -> every
// source record gathered into its final place.  scratch: (2 * n_cand + 2) words
void fd_launch_rs_records_dev(const void *matches, uint64_t n, const uint32_t *slot_matches, uint32_t n_cand, const uint64_t *cand_off, const uint32_t *slot_q,
                              const rs_query_dev *qt, uint32_t n_queries, uint32_t *scratch, uint64_t *match_off, uint64_t *res_off, const float *rmsd,
                              const float *rot, const float *tran, const float *met, const int32_t *residues, void *out, int32_t *out_res, hipStream_t st);
int fd_match_pairs_multi(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const fd_match_query *qs,
                         const uint32_t *cand, const uint64_t *cand_off, const fd_hash_params *p, fd_pair_rec **found, uint64_t *n_found,
                         fd_cand_rec **cands, uint64_t *n_cands, uint32_t mode = 3, const uint32_t *cj_mask = nullptr,
                         const uint32_t *mask_off = nullptr, uint64_t mask_words = 0, uint32_t **pk_key = nullptr, uint32_t **pk_val = nullptr,
                         fd_vote_plan *votes = nullptr, struct fd_mp_tables *tables = nullptr, const std::function<void()> *while_scanning = nullptr);
// (while_scanning: host work of the caller that does not need the scan's result — run once, between the scan's launch and the wait for it)
