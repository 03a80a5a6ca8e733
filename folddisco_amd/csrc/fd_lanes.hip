// fd_lanes.hip — the non-blocking form of fdgpu_query_batch: submit / wait.
//
// The reference drives its query body from many rayon workers (src/cli/workflows/query_pdb.rs:348 `queries.into_par_iter()`, :415 the
// retrieval's par_iter_mut): while one worker prepares tables or formats records, another one's kernels... there are none, it is a CPU
// program — but the SHAPE of the host is "many queries in flight".  A host with ONE thread per GPU gets the same overlap from this file:
// fdgpu_query_batch_submit hands a batch of queries to one of a few LANES and returns; fdgpu_query_batch_wait collects it.  A lane is a
// private sibling context of the caller's (its own HIP stream, its own workspaces, page-locked landing blocks and result pool — "two sets
// of pooled scratch" and more) driven by a library thread through the very same fdgpu_query_batch, so a batch's host-side gaps (table
// building, the waits for counts, the result copies) are filled by the kernels of the batches in the other lanes, and the results are
// bit for bit those of the blocking call (tests compare them).  The resident index, its checkpoint table and the coordinate batch are
// shared by all lanes (read-only on the query path).
#include "fdgpu_internal.h"

#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

#define FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)

struct fdgpu_query_job {
    // inputs: the small per-query arrays are copied at submit (the caller may reuse its buffers at once); index, batches, resname_std and
    // penalty are borrowed until the wait returns
    const fdgpu_index *ix; const fdgpu_batch *db; const uint8_t *resname_std; const fdgpu_batch *qb;
    uint64_t n_queries;
    std::vector<uint32_t> q_struct, q_index, n_subs;
    std::vector<uint64_t> q_off;
    std::vector<std::vector<uint8_t>> subs_store;
    std::vector<const uint8_t *> subs;
    bool have_subs = false, have_nsubs = false;
    std::vector<float> dist_thr, angle_thr;
    fd_hash_params p;
    float total_structures; const float *penalty; uint32_t top_n, match_top; float ca_cut; uint32_t node_count;
    // outputs
    std::vector<fd_query_map *> maps;
    fd_count_rec *recs = nullptr; uint64_t *rec_off = nullptr; fd_match_rec *matches = nullptr; uint64_t *match_off = nullptr; int32_t *residues = nullptr; uint64_t *res_off = nullptr;
    int rc = FDGPU_OK; std::string err;
    bool done = false;
    fdgpu_ctx *owner = nullptr;
};

struct fd_lane_pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<fdgpu_query_job *> queue;
    std::vector<std::thread> threads;
    std::vector<fdgpu_ctx *> ctxs;
    bool stop = false;
    uint64_t in_flight = 0;
};

static void lane_main(fd_lane_pool *P, fdgpu_ctx *lc) {
    for (;;) {
        fdgpu_query_job *j = nullptr;
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv_work.wait(lk, [&] { return P->stop || !P->queue.empty(); });
            if (P->queue.empty()) return;      // stop, and nothing left to run
            j = P->queue.front();
            P->queue.pop_front();
        }
        j->maps.assign(std::max<uint64_t>(j->n_queries, 1), nullptr);
        j->rc = fdgpu_query_batch(lc, j->ix, j->db, j->resname_std, j->qb, j->n_queries, j->q_struct.data(), j->q_off.data(), j->q_index.data(),
                                  j->have_subs ? j->subs.data() : nullptr, j->have_nsubs ? j->n_subs.data() : nullptr, j->dist_thr.data(), j->dist_thr.size(),
                                  j->angle_thr.data(), j->angle_thr.size(), &j->p, j->total_structures, j->penalty, j->top_n, j->match_top, j->ca_cut, j->node_count,
                                  j->maps.data(), &j->recs, &j->rec_off, &j->matches, &j->match_off, &j->residues, &j->res_off);
        if (j->rc) j->err = fdgpu_last_error(lc);
        {
            std::lock_guard<std::mutex> lk(P->mu);
            j->done = true;
        }
        P->cv_done.notify_all();
    }
}

static uint32_t fd_default_lanes() {
    const char *e = getenv("FDGPU_QUERY_LANES");
    long v = e ? atol(e) : 6;      // measured at 542,000 structures, batches of 128 (lanes x batches in flight, warmed): 4x6 150 k queries/s, 4x12 154 k, 5x8 149 k, 6x10 159 k,
                                   // 8x16 161 k (HISTORY.md); a lane is a sibling context: its own stream, scratch (~1 GB at this size) and worker thread
    return (uint32_t)std::min<long>(std::max<long>(v, 1), 8);
}

// lanes of a context: made on the first submit (or by fdgpu_query_lanes), torn down by fdgpu_destroy
static int fd_lanes_ensure(fdgpu_ctx *c, uint32_t want) {
    fd_lane_pool *P = (fd_lane_pool *)c->lanes;
    if (!P) { P = new (std::nothrow) fd_lane_pool(); if (!P) return FDGPU_ENOMEM; c->lanes = P; }
    while (P->ctxs.size() < want) {
        fdgpu_ctx *lc = nullptr;
        const int rc = fdgpu_create(c->device, &lc);
        if (rc) { c->err = std::string("query lane: ") + (lc ? fdgpu_last_error(lc) : "out of memory"); if (lc) fdgpu_destroy(lc); return rc; }
        lc->is_lane = true;
        P->ctxs.push_back(lc);
        P->threads.emplace_back(lane_main, P, lc);
    }
    return FDGPU_OK;
}

void fd_lanes_destroy(fdgpu_ctx *c) {
    fd_lane_pool *P = (fd_lane_pool *)c->lanes;
    if (!P) return;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->stop = true;       // jobs already queued still run: their tickets may be waited for by another thread
    }
    P->cv_work.notify_all();
    for (auto &t : P->threads) t.join();
    for (fdgpu_ctx *lc : P->ctxs) fdgpu_destroy(lc);
    delete P;
    c->lanes = nullptr;
}

extern "C" int fdgpu_query_lanes(fdgpu_ctx *c, uint32_t n_lanes) { FD_LOCK(c);
    if (!c) return FDGPU_EINVAL;
    if (n_lanes == 0) return c->lanes ? (int)((fd_lane_pool *)c->lanes)->ctxs.size() : 0;
    if (n_lanes > 8) FAIL(c, FDGPU_EINVAL, "at most 8 query lanes");
    const int rc = fd_lanes_ensure(c, n_lanes);
    return rc ? rc : (int)((fd_lane_pool *)c->lanes)->ctxs.size();
}

extern "C" int fdgpu_query_batch_submit(fdgpu_ctx *c, const fdgpu_index *ix, const fdgpu_batch *db, const uint8_t *resname_std, const fdgpu_batch *qb, uint64_t n_queries,
                                        const uint32_t *q_struct, const uint64_t *q_off, const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs,
                                        const float *dist_thr, uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p, float total_structures,
                                        const float *penalty, uint32_t top_n, uint32_t match_top, float ca_distance_cutoff, uint32_t node_count, fdgpu_query_job **job) { FD_LOCK(c);
    if (!c || !job) return FDGPU_EINVAL;
    *job = nullptr;
    if (!ix || !db || !qb || !p || !q_off || (n_queries && !q_struct) || (n_dist && !dist_thr) || (n_angle && !angle_thr_deg)) FAIL(c, FDGPU_EINVAL, "query_batch_submit: null argument");
    if (!c->lanes || ((fd_lane_pool *)c->lanes)->ctxs.empty()) {
        const int rc = fd_lanes_ensure(c, fd_default_lanes());
        if (rc) return rc;
    }
    fd_lane_pool *P = (fd_lane_pool *)c->lanes;
    std::unique_ptr<fdgpu_query_job> j(new (std::nothrow) fdgpu_query_job());
    if (!j) return FDGPU_ENOMEM;
    j->owner = c; j->ix = ix; j->db = db; j->resname_std = resname_std; j->qb = qb; j->n_queries = n_queries;
    j->q_struct.assign(q_struct, q_struct + n_queries);
    j->q_off.assign(q_off, q_off + n_queries + 1);
    const uint64_t nt = q_off[n_queries];
    if (nt && !q_index) FAIL(c, FDGPU_EINVAL, "query_batch_submit: null q_index");
    j->q_index.assign(q_index, q_index + nt);
    j->q_struct.reserve(1); j->q_index.reserve(1);      // data() of an empty vector may be null; the blocking call checks q_struct only when n_queries != 0
    if (n_subs) { j->have_nsubs = true; j->n_subs.assign(n_subs, n_subs + nt); }
    if (subs) {
        j->have_subs = true;
        j->subs.assign(std::max<uint64_t>(nt, 1), nullptr);
        j->subs_store.resize(nt);
        for (uint64_t k = 0; k < nt; ++k)
            if (subs[k]) {
                const uint32_t n = n_subs ? n_subs[k] : 0;
                j->subs_store[k].assign(subs[k], subs[k] + std::max<uint32_t>(n, 1));
                j->subs[k] = j->subs_store[k].data();
            }
    }
    j->dist_thr.assign(dist_thr, dist_thr + n_dist);
    j->angle_thr.assign(angle_thr_deg, angle_thr_deg + n_angle);
    j->dist_thr.reserve(1); j->angle_thr.reserve(1);
    j->p = *p;
    j->total_structures = total_structures; j->penalty = penalty; j->top_n = top_n; j->match_top = match_top; j->ca_cut = ca_distance_cutoff; j->node_count = node_count;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        if (P->stop) FAIL(c, FDGPU_EINVAL, "query_batch_submit: the context is being destroyed");
        P->queue.push_back(j.get());
        ++P->in_flight;
    }
    P->cv_work.notify_one();
    *job = j.release();
    return FDGPU_OK;
}

extern "C" int fdgpu_query_batch_wait(fdgpu_ctx *c, fdgpu_query_job *job, fd_query_map **maps, fd_count_rec **recs, uint64_t **rec_off, fd_match_rec **matches,
                                      uint64_t **match_off, int32_t **residues, uint64_t **res_off) {
    // no FD_LOCK: a wait must not keep other threads from submitting to (or waiting on) the same context
    if (!c || !job || job->owner != c || !c->lanes) return FDGPU_EINVAL;
    fd_lane_pool *P = (fd_lane_pool *)c->lanes;
    {
        std::unique_lock<std::mutex> lk(P->mu);
        P->cv_done.wait(lk, [&] { return job->done; });
        --P->in_flight;
    }
    const int rc = job->rc;
    const bool want = maps && recs && rec_off && matches && match_off && residues && res_off;
    if (!rc && want) {
        for (uint64_t t = 0; t < job->n_queries; ++t) maps[t] = job->maps[t];
        *recs = job->recs; *rec_off = job->rec_off; *matches = job->matches; *match_off = job->match_off; *residues = job->residues; *res_off = job->res_off;
    } else {
        if (!rc) {      // the caller does not take the results (or passed a null output): release them
            for (uint64_t t = 0; t < job->n_queries; ++t) fdgpu_query_map_free(job->maps[t]);
            fdgpu_free(job->recs); fdgpu_free(job->rec_off); fdgpu_free(job->matches); fdgpu_free(job->match_off); fdgpu_free(job->residues); fdgpu_free(job->res_off);
        } else {
            std::lock_guard<std::recursive_mutex> lk(c->mu);
            c->err = job->err;
        }
    }
    delete job;
    return rc ? rc : (want ? FDGPU_OK : FDGPU_EINVAL);
}
