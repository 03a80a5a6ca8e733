// fd_api_match.hip — C ABI of libfdgpu.so, seam S4 (include/fdgpu.h): the pair scan over candidates, superposition, metrics and LMS-QCP — the
// orchestration of k_match.hip.  Reference: src/controller/retrieve.rs:52-156, src/structure/kabsch.rs:157-554, src/structure/metrics.rs:62-251,
// src/structure/lms_qcp.rs.
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

// found triples in the reference's scan order — (slot, i, j), several bin pairs of one (i, j) in emission order — from the kernel's
// append order: counting sort by slot (stable), then the slots' runs sorted independently on host threads (a whole-structure query
// returns ~10^5 triples for a handful of slots: one 20 ms std::stable_sort otherwise)
static void fd_sort_found(fd_pair_rec *f, uint64_t n, uint64_t n_cand) {
    if (n < 2) return;
    auto by_ij = [](const fd_pair_rec &a, const fd_pair_rec &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; };
    if (n < 4096 || n_cand == 0) {
        std::stable_sort(f, f + n, [&](const fd_pair_rec &a, const fd_pair_rec &b) { return a.cand != b.cand ? a.cand < b.cand : by_ij(a, b); });
        return;
    }
    std::vector<uint64_t> start(n_cand + 2, 0);
    for (uint64_t k = 0; k < n; ++k) ++start[std::min<uint64_t>(f[k].cand, n_cand) + 1];
    for (uint64_t s = 0; s <= n_cand; ++s) start[s + 1] += start[s];
    std::vector<fd_pair_rec> tmp(n);
    {
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (uint64_t k = 0; k < n; ++k) tmp[cur[std::min<uint64_t>(f[k].cand, n_cand)]++] = f[k];
    }
    std::atomic<uint64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const uint64_t s = next.fetch_add(1);
            if (s > n_cand) break;
            std::stable_sort(tmp.begin() + start[s], tmp.begin() + start[s + 1], by_ij);
        }
    };
    const unsigned T = (unsigned)std::min<uint64_t>(std::min<uint64_t>(16, std::max(1u, std::thread::hardware_concurrency())), n_cand + 1);
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    memcpy(f, tmp.data(), n * sizeof(fd_pair_rec));
}

// ---- S4 ---------------------------------------------------------------------------------------------------------------
// Pair scan for MANY queries in one launch: query t scans the candidates cand[cand_off[t] .. cand_off[t+1]); the records carry the
// GLOBAL slot (position in cand) and come back sorted by (slot, i, j).
int fd_match_pairs_multi(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const fd_match_query *qs,
                         const uint32_t *cand, const uint64_t *cand_off, const fd_hash_params *p, fd_pair_rec **found, uint64_t *n_found,
                         fd_cand_rec **cands, uint64_t *n_cands, uint32_t mode, const uint32_t *cj_mask, const uint32_t *mask_off,
                         uint64_t mask_words, uint32_t **pk_key, uint32_t **pk_val, fd_vote_plan *votes, fd_mp_tables *tables, const std::function<void()> *while_scanning) {
    if (!c || !db || !p || !found || !n_found || !cands || !n_cands || !cand_off || (n_queries && !qs)) return FDGPU_EINVAL;
    if ((mode & 32u) && (!votes || !cj_mask || !mask_off || (mode & 3u))) return FDGPU_EINVAL;
    const uint64_t n_cand = cand_off[n_queries];
    if (n_cand && !cand) return FDGPU_EINVAL;
    *found = nullptr; *cands = nullptr; *n_found = 0; *n_cands = 0;
    if (!fd_hash_type_supported(p->hash_type)) FAIL(c, FDGPU_EINVAL, "hash_type: only the encodings over the (d_CA, d_CB, theta, tau1, tau2) descriptor are built (0, 1, 3, 7, 8)");
    reset_timings(c);
    hipStream_t st = c->stream;
    const bool mp_trace = getenv("FDGPU_TRACE") != nullptr;
    const auto mp_t0 = std::chrono::steady_clock::now();
    auto mp_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - mp_t0).count(); };
    // work items and query tables (below) depend on the queries and candidates only: a caller that scans the same candidates twice (the two
    // scans of a large query) passes a fd_mp_tables and the second call reuses the host block
    fd_mp_tables tb_local;
    fd_mp_tables &TB = tables ? *tables : tb_local;
    const bool dev_items = !tables && n_cand < (1ull << 24) && !(getenv("FDGPU_MP_ITEMS") && getenv("FDGPU_MP_ITEMS")[0] == '0');      // 0: host-built items (tests)
    auto build_tables = [&]() -> int {
    // work items: (query, candidate slot, 64-residue i-tile); a handful of long candidates (whole-structure queries: the top 20)
    // would leave most of the chip idle, so the partner residues are split into spans as well until ~2000 wavefronts exist
    uint64_t n_tiles = 0;
    for (uint64_t k = 0; k < n_cand; ++k) {
        if (cand[k] >= db->n_struct) { c->err = "match_pairs: candidate id outside the batch"; return FDGPU_EINVAL; }
        n_tiles += (db->h_res_off[cand[k] + 1] - db->h_res_off[cand[k]] + FD_WAVE - 1) / FD_WAVE;
    }
    // many candidates (a batch of motif queries): spans of 256 partners cap the longest work items — the scan launch ends with its slowest wavefront, and a
    // wavefront walks its partners one at a time (128 queries x 32 candidates, k_mp_scan: 93 us at 512, 68 at 256, 60 at 128 — where the drains, one per
    // partly filled chunk, have grown by as much)
    // large queries (whole-structure: the window test passes nearly every pair inside the cutoff, so a work item's time is its number of close
    // pairs x one descriptor + hash each): while scan and drain were one kernel, a diagonal block of 64 x 128 residues was 128 drains on ONE wavefront and
    // spans of 32 partners measured best (5.1 ms at 128, 4.4 at 64, 3.7 at 32).  With the drains in their own launch (k_mp_drain, a wavefront per chunk) the
    // scan's work items only test and queue: 128 partners per item (first + second scan of the top 20 of a 300-residue query: 1.64 + 1.45 ms at 32,
    // 1.32 + 1.19 at 64, 1.21 + 1.02 at 128, 1.24 + 1.00 at 256)
    uint64_t max_aad_q = 0;
    for (uint64_t t = 0; t < n_queries; ++t) max_aad_q = std::max<uint64_t>(max_aad_q, qs[t].n_aad);
    uint32_t j_span = !n_tiles ? 0u : max_aad_q > 4096 ? 128u : n_tiles < 256 ? 64u : n_tiles < 1024 ? 128u : 256u;
    if (const char *js = getenv("FDGPU_MP_JSPAN")) if (n_tiles && atoi(js) >= 32) j_span = (uint32_t)atoi(js) & ~63u ? (uint32_t)atoi(js) & ~31u : 32u;      // (measurement aid)
    TB.j_span = j_span;
    std::vector<uint32_t> wc, wi, wq, wj;
    // a one-off block (a batch of motif queries: 18 k work items per 128 queries) gets its work items written on the DEVICE (k_mp_items): the host
    // sends every candidate's first item and query (8 bytes per candidate instead of 16 per item) and skips the loop below
    std::vector<uint32_t> wbase, cq;
    size_t n_wi = 0;
    {
        if (dev_items) { wbase.resize(n_cand + 1); cq.resize(std::max<uint64_t>(n_cand, 1)); }
        for (uint64_t k = 0; k < n_cand; ++k) {
            const uint64_t len = db->h_res_off[cand[k] + 1] - db->h_res_off[cand[k]];
            if (dev_items) wbase[k] = (uint32_t)n_wi;
            n_wi += ((len + FD_WAVE - 1) / FD_WAVE) * (j_span ? (len + j_span - 1) / j_span : (len ? 1 : 0));
        }
        // (k_mp_items keeps a candidate's first entry of the active-residue list as the 32-bit word 64 x first item)
        if (dev_items && n_wi >= (1ull << 26)) FAIL(c, FDGPU_ERANGE, "match_pairs: more than 2^26 work items in one call; split the candidates");
        if (dev_items) {
            wbase[n_cand] = (uint32_t)n_wi;
            for (uint64_t t = 0; t < n_queries; ++t) for (uint64_t k = cand_off[t]; k < cand_off[t + 1]; ++k) cq[k] = (uint32_t)t;
        } else { wc.reserve(n_wi); wi.reserve(n_wi); wq.reserve(n_wi); wj.reserve(n_wi); }
    }
    if (!dev_items)
    for (uint64_t t = 0; t < n_queries; ++t)
        for (uint64_t k = cand_off[t]; k < cand_off[t + 1]; ++k) {
            uint64_t r0 = db->h_res_off[cand[k]], r1 = db->h_res_off[cand[k] + 1];
            for (uint64_t r = r0; r < r1; r += FD_WAVE)
                for (uint64_t j0 = r0; j0 < r1; j0 += j_span ? j_span : (r1 - r0)) {
                    wc.push_back((uint32_t)k); wi.push_back((uint32_t)r); wq.push_back((uint32_t)t); wj.push_back((uint32_t)j0);
                }
        }
    // per-query tables: sorted hash set, residue-type masks, aa_dist_map grouped by (aa_i, aa_j) — stable order inside a group =
    // the observed-list order the reference emits in
    std::vector<mp_query_dev> qtab(std::max<uint64_t>(n_queries, 1));
    uint64_t qset_slots = 0, qset_max_hashes = 0;
    std::vector<uint32_t> all_hashes, all_start, all_qi;
    std::vector<float> all_dist;
    all_start.assign((size_t)1025 * n_queries, 0u);      // one start table per query, filled in place (a batch of 512 motif queries: 2 MB)
    std::vector<uint32_t> cur(1024);
    for (uint64_t t = 0; t < n_queries; ++t) {
        const fd_match_query *q = &qs[t];
        mp_query_dev &Q = qtab[t];
        Q.qh_off = (uint32_t)all_hashes.size(); Q.n_hashes = (uint32_t)q->n_hashes;
        Q.qs_off = 0; Q.qs_mask = 0;
        if (q->n_hashes > MP_QH_LDS && qset_slots + 4 * q->n_hashes < (1ull << 31)) {      // a hash set for the drains' membership test (k_mp_qset_build fills it)
            uint64_t slots = 2048;
            while (slots < 2 * q->n_hashes) slots <<= 1;
            Q.qs_off = (uint32_t)qset_slots; Q.qs_mask = (uint32_t)(slots - 1);
            qset_slots += slots; qset_max_hashes = std::max<uint64_t>(qset_max_hashes, q->n_hashes);
        }
        all_hashes.insert(all_hashes.end(), q->hashes, q->hashes + q->n_hashes);
        Q.aa1_mask = Q.aa2_mask = 0;
        for (uint64_t k = 0; k < q->n_hashes; ++k) {
            uint32_t a1, a2;
            fd_hash_aa_pair(p->hash_type, q->hashes[k], &a1, &a2);
            Q.aa1_mask |= 1u << (a1 & 31u); Q.aa2_mask |= 1u << (a2 & 31u);
        }
        // TertiaryInteraction / Hybrid hashes carry no residue types: the reference's prefilter unwraps a None there
        // (retrieve.rs:576) and panics for queries of <= 200 hashes; every pair is scanned instead
        const bool no_aa = p->hash_type == FD_HASH_TERTIARY || p->hash_type == FD_HASH_HYBRID;
        Q.use_prefilter = no_aa ? 0 : q->use_aa_prefilter; Q.ca_window = q->ca_distance_cutoff;
        uint32_t *cnt = &all_start[(size_t)1025 * t];
        if (q->n_aad <= 256) {
            // a motif query's dozen observed distances: (group << 16 | e) keys sorted (= stable by e), the 1,025-entry start table written as
            // a few runs — counting into it and scanning it cost 1,024 dependent adds per query, 128 times per batch
            uint32_t keys[256];
            uint32_t nk = 0;
            for (uint64_t e = 0; e < q->n_aad; ++e)
                if (q->aad_aa1[e] < 32 && q->aad_aa2[e] < 32) keys[nk++] = ((q->aad_aa1[e] * 32u + q->aad_aa2[e]) << 16) | (uint32_t)e;
            std::sort(keys, keys + nk);
            Q.aad_off = (uint32_t)all_dist.size(); Q.n_aad = nk;
            all_qi.resize(Q.aad_off + nk); all_dist.resize(Q.aad_off + nk);
            uint32_t g_next = 0;       // start[g] for g < g_next is written
            for (uint32_t k = 0; k < nk; ++k) {
                const uint32_t g = keys[k] >> 16, e = keys[k] & 0xffffu;
                if (g >= g_next) { std::fill(cnt + g_next, cnt + g + 1, k); g_next = g + 1; }
                all_qi[Q.aad_off + k] = q->aad_qi[e]; all_dist[Q.aad_off + k] = q->aad_dist[e];
            }
            std::fill(cnt + g_next, cnt + 1025, nk);
            continue;
        }
        for (uint64_t e = 0; e < q->n_aad; ++e)
            if (q->aad_aa1[e] < 32 && q->aad_aa2[e] < 32) ++cnt[q->aad_aa1[e] * 32u + q->aad_aa2[e] + 1];   // residue type 255 never passes get_single_feature
        for (int k = 0; k < 1024; ++k) cnt[k + 1] += cnt[k];
        Q.aad_off = (uint32_t)all_dist.size(); Q.n_aad = cnt[1024];
        all_qi.resize(Q.aad_off + Q.n_aad); all_dist.resize(Q.aad_off + Q.n_aad);
        memcpy(cur.data(), cnt, 1024 * 4);
        for (uint64_t e = 0; e < q->n_aad; ++e)
            if (q->aad_aa1[e] < 32 && q->aad_aa2[e] < 32) {
                uint32_t k = Q.aad_off + cur[q->aad_aa1[e] * 32u + q->aad_aa2[e]]++;
                all_qi[k] = q->aad_qi[e]; all_dist[k] = q->aad_dist[e];
            }
    }
    // queries whose observed-distance lists do not fit the kernel's LDS copy (whole-structure queries: ~10^2 distances per residue-type
    // pair): per group the union of the float intervals {d : |d - x| < window} over its observed x, merged — the scan's window test reads
    // one or two intervals instead of walking the list.  Exact: fl(d - x) is monotone in d, so the set of passing d of one x is an interval
    // of floats whose ends are found by stepping from x -/+ window to the last float that still passes.
    bool want_iv = false;
    for (uint64_t t = 0; t < n_queries; ++t) want_iv = want_iv || qtab[t].n_aad > 1024u;
    std::vector<uint32_t> iv_start;
    std::vector<float> iv_lohi;      // lo, hi interleaved (float2 on the device)
    if (want_iv) {
        iv_start.assign((size_t)1025 * n_queries, 0);
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint32_t *stt = &all_start[1025 * t];
            const float *base = all_dist.data() + qtab[t].aad_off;
            const float w = qtab[t].ca_window;
            // the 1,024 groups are independent: eight parts of 128 groups on the context's helper threads when the query observes enough distances to pay
            // for waking them (a whole-structure query: ~3·10^4 observed distances, six nextafterf each), stitched in group order afterwards
            constexpr unsigned NP = 8;
            std::vector<float> part_lohi[NP];
            uint32_t g_cnt[1024];
            auto groups = [&](unsigned part) {
                std::vector<std::pair<float, float>> tmp;
                std::vector<float> &out = part_lohi[part];
                for (int g = 128 * (int)part; g < 128 * ((int)part + 1); ++g) {
                    const size_t before = out.size();
                    tmp.clear();
                    for (uint32_t e = stt[g]; e < stt[g + 1]; ++e) {
                        const float x = base[e];
                        if (!(fabsf(x - x) < w)) continue;                  // window <= 0 or NaN: nothing passes
                        float hi = x + w, lo = x - w;
                        while (!(fabsf(hi - x) < w)) hi = nextafterf(hi, -INFINITY);
                        for (float n2 = nextafterf(hi, INFINITY); fabsf(n2 - x) < w; n2 = nextafterf(hi, INFINITY)) hi = n2;
                        while (!(fabsf(lo - x) < w)) lo = nextafterf(lo, INFINITY);
                        for (float n2 = nextafterf(lo, -INFINITY); fabsf(n2 - x) < w; n2 = nextafterf(lo, -INFINITY)) lo = n2;
                        tmp.emplace_back(lo, hi);
                    }
                    std::sort(tmp.begin(), tmp.end());
                    for (size_t k = 0; k < tmp.size();) {
                        float lo = tmp[k].first, hi = tmp[k].second;
                        size_t z = k + 1;
                        while (z < tmp.size() && tmp[z].first <= nextafterf(hi, INFINITY)) { hi = std::max(hi, tmp[z].second); ++z; }
                        out.push_back(lo); out.push_back(hi);
                        k = z;
                    }
                    g_cnt[g] = (uint32_t)((out.size() - before) / 2);
                }
            };
            if (stt[1024] - stt[0] < 8192u) { for (unsigned k = 0; k < NP; ++k) groups(k); }
            else {
                std::atomic<unsigned> next_part(0);
                const std::function<void()> wk = [&]() { for (;;) { const unsigned k = next_part.fetch_add(1); if (k >= NP) break; groups(k); } };
                c->host_pool.run(c->small_par(NP), wk);
            }
            uint32_t run = (uint32_t)(iv_lohi.size() / 2);
            for (unsigned k = 0; k < NP; ++k) iv_lohi.insert(iv_lohi.end(), part_lohi[k].begin(), part_lohi[k].end());      // the parts in group order
            for (int g = 0; g < 1024; ++g) { iv_start[1025 * t + g] = run; run += g_cnt[g]; }
            iv_start[1025 * t + 1024] = run;
        }
    }
    const size_t nw = dev_items ? n_wi : wc.size(), na = all_dist.size(), nh = all_hashes.size();
    TB.nw = nw; TB.want_iv = want_iv;
    // one packed host block -> one H2D copy: [cand | wc | wi | wq | hashes | start tables | dist | qi | qtab]
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t o_cand = 0, o_wc = o_cand + up4(n_cand), o_wi = o_wc + up4(nw), o_wq = o_wi + up4(nw), o_wj = o_wq + up4(nw), o_h = o_wj + up4(nw),
                 o_st = o_h + up4(nh), o_d = o_st + up4(all_start.size()), o_qi = o_d + up4(na), o_qt = o_qi + up4(na),
                 o_ivs = o_qt + up4(n_queries * (sizeof(mp_query_dev) / 4)), o_iv = o_ivs + (want_iv ? up4(iv_start.size()) : 0),
                 o_iv1 = o_iv + (want_iv ? up4(iv_lohi.size()) : 0), o_wb = o_iv1 + (want_iv ? 1024 * n_queries : 0),
                 words = o_wb + (dev_items ? up4(n_cand + 1) + up4(n_cand) : 0) + 4;
    const size_t offs[13] = {o_cand, o_wc, o_wi, o_wq, o_wj, o_h, o_st, o_d, o_qi, o_qt, o_ivs, o_iv, o_iv1};
    memcpy(TB.o, offs, sizeof offs);
    TB.o_wb = dev_items ? o_wb : 0; TB.dev_items = dev_items;
    // a caller that keeps the tables (two scans of a large query) gets them in a vector; a one-off block (a batch of motif queries: ~3.5 MB
    // per 512 queries) is packed straight into the context's pinned staging buffer — the copy below is then a DMA, not a staged pageable copy
    uint32_t *blk = tables ? nullptr : (uint32_t *)c->host_pinned(0, words * 4);
    if (!blk) { TB.blk.assign(words, 0); blk = TB.blk.data(); }
    TB.data = blk; TB.words = words;
    if (n_cand) memcpy(&blk[o_cand], cand, n_cand * 4);
    if (dev_items) { memcpy(&blk[o_wb], wbase.data(), (n_cand + 1) * 4); if (n_cand) memcpy(&blk[o_wb + up4(n_cand + 1)], cq.data(), n_cand * 4); }
    else if (nw) { memcpy(&blk[o_wc], wc.data(), nw * 4); memcpy(&blk[o_wi], wi.data(), nw * 4); memcpy(&blk[o_wq], wq.data(), nw * 4); memcpy(&blk[o_wj], wj.data(), nw * 4); }
    if (nh) memcpy(&blk[o_h], all_hashes.data(), nh * 4);
    if (!all_start.empty()) memcpy(&blk[o_st], all_start.data(), all_start.size() * 4);
    if (na) { memcpy(&blk[o_d], all_dist.data(), na * 4); memcpy(&blk[o_qi], all_qi.data(), na * 4); }
    if (want_iv) {
        memcpy(&blk[o_ivs], iv_start.data(), iv_start.size() * 4);
        if (!iv_lohi.empty()) memcpy(&blk[o_iv], iv_lohi.data(), iv_lohi.size() * 4);
        // the scan's LDS copy: per query and group (first interval relative to the query's first) << 8 | number of intervals (a window of 1 A
        // leaves 1-5 disjoint intervals per group of a 300-residue query, ~700 in all; 255+ intervals of one group: the count saturates and
        // the scan walks that group in global memory)
        for (uint64_t t = 0; t < n_queries; ++t)
            for (int g = 0; g < 1024; ++g) {
                const uint32_t v_lo = iv_start[1025 * t + g], v_hi = iv_start[1025 * t + g + 1], rel = v_lo - iv_start[1025 * t];
                blk[o_iv1 + 1024 * t + g] = (std::min<uint32_t>(rel, 0xffffffu) << 8) | std::min<uint32_t>(v_hi - v_lo, 255u);
            }
    }
    if (n_queries) memcpy(&blk[o_qt], qtab.data(), n_queries * sizeof(mp_query_dev));
    TB.qset_slots = qset_slots; TB.qset_max_hashes = qset_max_hashes;
    TB.valid = true;
    return FDGPU_OK;
    };
    if (!TB.valid) { const int rcb = build_tables(); if (rcb) return rcb; }
    const size_t o_cand = TB.o[0], o_wc = TB.o[1], o_wi = TB.o[2], o_wq = TB.o[3], o_wj = TB.o[4], o_h = TB.o[5], o_st = TB.o[6], o_d = TB.o[7], o_qi = TB.o[8],
                 o_qt = TB.o[9], o_ivs = TB.o[10], o_iv = TB.o[11], o_iv1 = TB.o[12], nw = TB.nw, words = TB.words;
    const bool want_iv = TB.want_iv;
    const uint32_t j_span = TB.j_span;
    if (mp_trace) fprintf(stderr, "[match_pairs] tables at %.3f ms (%zu work items)\n", mp_ms(), nw);
    HIPCHK(c, c->ws[WS_MISC0].ensure(words * 4));
    HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
    uint8_t *d_std = nullptr;
    if (resname_std) {
        HIPCHK(c, c->ws[WS_MISC5].ensure(std::max<uint64_t>(db->n_res, 1)));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC5].p, resname_std, db->n_res, hipMemcpyHostToDevice, st));
        d_std = c->ws[WS_MISC5].as<uint8_t>();
    }
    const uint4 *d_cinfo = nullptr;
    const uint32_t *d_act = nullptr;
    if (TB.dev_items) {       // the item arrays [o_wc, o_h) are not sent: the device writes them — and, per candidate, the list of its active residues
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, TB.data, o_wc * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].as<uint32_t>() + o_h, TB.data + o_h, (words - o_h) * 4, hipMemcpyHostToDevice, st));
        uint32_t *d = c->ws[WS_MISC0].as<uint32_t>();
        const size_t ci_bytes = ((size_t)n_cand * 16 + 255) & ~(size_t)255;
        HIPCHK(c, c->ws[WS_MP_ACT].ensure(ci_bytes + (size_t)64 * std::max<size_t>(nw, 1) * 4));
        d_cinfo = c->ws[WS_MP_ACT].as<uint4>();
        d_act = (const uint32_t *)(c->ws[WS_MP_ACT].as<uint8_t>() + ci_bytes);
        fd_launch_mp_items(db->res_off, d + o_cand, (uint32_t)n_cand, d + TB.o_wb, d + TB.o_wb + ((n_cand + 1 + 3) & ~(size_t)3), j_span, d + o_wc, d + o_wi, d + o_wq, d + o_wj, st,
                           db->aa, db->hash_ok, d_std, fd_make_consts_cfg(p, 0).q.type == FD_HASH_TERTIARY ? 1 : 0, (const mp_query_dev *)(d + o_qt), (void *)d_cinfo,
                           (uint32_t *)d_act);
    } else
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, TB.data, words * 4, hipMemcpyHostToDevice, st));
    const uint32_t *dblk = c->ws[WS_MISC0].as<uint32_t>();
    mp_args A;
    memset(&A, 0, sizeof A);
    A.mode = mode;
    if (cj_mask && mask_off) {   // partner-residue filter (second pass of a large query's retrieval)
        HIPCHK(c, c->ws[WS_MISC1].ensure((mask_words + n_cand + 2) * 4));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, cj_mask, mask_words * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].as<uint32_t>() + mask_words, mask_off, n_cand * 4, hipMemcpyHostToDevice, st));
        A.cj_mask = c->ws[WS_MISC1].as<uint32_t>(); A.mask_off = A.cj_mask + mask_words;
    }
    fd_vote_row *d_rows = nullptr;
    if (mode & 32u) {   // rescue votes stay on the device: counters in ws[WS_IDS_A], their tables in ws[WS_IDS_B]
        const uint64_t nb = votes->n_bits, nr = votes->n_rows;
        auto up8 = [](uint64_t x) { return (x + 7) & ~(uint64_t)7; };
        const uint64_t o_off = 0, o_roff = o_off + up8(n_cand * 8), o_rows = o_roff + up8(nr * 8), o_qs = o_rows + up8(nr * sizeof(fd_vote_row)),
                       o_rlen = o_qs + up8(n_cand * 4), o_comp = o_rlen + up8(nr * 4), o_sdd = o_comp + up8(nb), o_sdq = o_sdd + up8(votes->n_sd * 4),
                       bytes = o_sdq + up8(votes->n_sd * 4) + 8;
        HIPCHK(c, c->ws[WS_IDS_A].ensure(std::max<uint64_t>(votes->n_counters, 1) * 4));
        HIPCHK(c, c->ws[WS_IDS_B].ensure(bytes));
        uint8_t *base = c->ws[WS_IDS_B].as<uint8_t>();
        HIPCHK(c, hipMemsetAsync(c->ws[WS_IDS_A].p, 0, std::max<uint64_t>(votes->n_counters, 1) * 4, st));
        HIPCHK(c, hipMemcpyAsync(base + o_off, votes->vt_off, n_cand * 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(base + o_qs, votes->vt_qs, n_cand * 4, hipMemcpyHostToDevice, st));
        if (nr) {
            HIPCHK(c, hipMemcpyAsync(base + o_roff, votes->row_off, nr * 8, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_rlen, votes->row_len, nr * 4, hipMemcpyHostToDevice, st));
        }
        if (nb) HIPCHK(c, hipMemcpyAsync(base + o_comp, votes->cj_comp, nb, hipMemcpyHostToDevice, st));
        A.votes = c->ws[WS_IDS_A].as<uint32_t>(); A.vt_off = (const uint64_t *)(base + o_off); A.vt_qs = (const uint32_t *)(base + o_qs);
        A.cj_comp = base + o_comp;
        if (votes->sd_dist && votes->sd_qi && votes->n_sd) {
            HIPCHK(c, hipMemcpyAsync(base + o_sdd, votes->sd_dist, votes->n_sd * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_sdq, votes->sd_qi, votes->n_sd * 4, hipMemcpyHostToDevice, st));
            A.sd_dist = (const float *)(base + o_sdd); A.sd_qi = (const uint32_t *)(base + o_sdq);
        }
        d_rows = (fd_vote_row *)(base + o_rows);
    }
    if (!fd_multiple_bins_valid(p)) FAIL(c, FDGPU_EINVAL, "multiple_bins: at most 8 (dist, angle) bin pairs, no zero counts");
    A.B = db->view(); A.C = fd_make_consts_cfg(p, 0); A.cutoff = p->dist_cutoff;
    A.n_cfg = fd_num_bin_configs(p);
    for (uint32_t k = 0; k < A.n_cfg; ++k) A.qk[k] = fd_make_consts_cfg(p, k).q;
    A.cand = dblk + o_cand; A.n_cand = (uint32_t)n_cand;
    A.wi_cand = dblk + o_wc; A.wi_i0 = dblk + o_wi; A.wi_query = dblk + o_wq; A.n_work = (uint32_t)nw;
    A.wi_j0 = dblk + o_wj; A.j_span = j_span;
    A.resname_std = d_std;
    A.q_hashes = dblk + o_h; A.aad_start = dblk + o_st; A.aad_dist = (const float *)(dblk + o_d); A.aad_qi = dblk + o_qi;
    A.iv_start = want_iv ? dblk + o_ivs : nullptr; A.iv = want_iv ? (const float2 *)(dblk + o_iv) : nullptr;
    A.iv_grp = want_iv ? dblk + o_iv1 : nullptr;
    A.qtab = (const mp_query_dev *)(dblk + o_qt);
    A.qset = nullptr;
    if ((mode & 1u) && TB.qset_slots) {      // found triples wanted and some query holds more hashes than a drain stages in LDS: their hash sets
        HIPCHK(c, c->ws[WS_MP_QSET].ensure(TB.qset_slots * 4));
        HIPCHK(c, hipMemsetAsync(c->ws[WS_MP_QSET].p, 0xff, TB.qset_slots * 4, st));
        fd_launch_mp_qset_build(A.qtab, (uint32_t)n_queries, (uint32_t)TB.qset_max_hashes, dblk + o_h, c->ws[WS_MP_QSET].as<uint32_t>(), st);
        A.qset = c->ws[WS_MP_QSET].as<uint32_t>();
    }
    A.n_found = c->ws[WS_TOTAL].as<unsigned long long>(); A.n_cands = A.n_found + 1; A.found = nullptr; A.cands = nullptr;
    const bool mp_dbg = getenv("FDGPU_MP_DBG") != nullptr;       // clocks and counts of the pair scan's work items and drains on stderr (measurement aid)
    HIPCHK(c, c->ws[WS_TOTAL].ensure(16384));
    // ws[WS_TOTAL] (u64): [0] found triples, [1] candidate pairs, [2] chunks drained, [4, 12) FDGPU_MP_DBG, from [16]: the 64 sub-queues' claimed chunks, one per 128-byte line
    A.n_found = c->ws[WS_TOTAL].as<unsigned long long>(); A.n_cands = A.n_found + 1; A.q_cnt = A.n_found + 16;
    A.dbg = mp_dbg ? A.n_found + 4 : nullptr;
    A.cinfo = d_cinfo; A.act = d_act;
    A.compact = !want_iv;
      // (want_iv: some query observes more than 1,024 distances)
    // one emitting pass into buffers sized by the previous calls; a pass that overflows only counts, the buffers grow and
    // the pass is repeated (the scan is deterministic up to record order, which is restored below)
    static_assert(MP_SUBQ_STRIDE == 16, "ws[WS_TOTAL] layout");
    std::vector<uint64_t> tot_v(16 + 64 * MP_SUBQ_STRIDE, 0);
    uint64_t *tot = tot_v.data();
    uint64_t q_max = 0;      // most chunks one sub-queue was asked for by the previous attempt
    for (int attempt = 0; attempt < 4; ++attempt) {
        uint64_t capf = c->ws[WS_KEYS_A].cap / sizeof(fd_pair_rec), capc = c->ws[WS_KEYS_B].cap / sizeof(fd_cand_rec);
        if (capf < 4096 || capf < tot[0]) { HIPCHK(c, c->ws[WS_KEYS_A].ensure(std::max<uint64_t>(2 * tot[0], 65536) * sizeof(fd_pair_rec))); }
        if (capc < 4096 || capc < tot[1]) { HIPCHK(c, c->ws[WS_KEYS_B].ensure(std::max<uint64_t>(2 * tot[1], 65536) * sizeof(fd_cand_rec))); }
        A.found = c->ws[WS_KEYS_A].as<fd_pair_rec>(); A.cands = c->ws[WS_KEYS_B].as<fd_cand_rec>();
        A.cap_found = c->ws[WS_KEYS_A].cap / sizeof(fd_pair_rec); A.cap_cands = c->ws[WS_KEYS_B].cap / sizeof(fd_cand_rec);
        {
            // the chunk queue between the scan and the drains, ws[WS_MP_Q]: [256 B bin tables | per chunk: 16 B header, 256 B pairs, 3 x 256 B results,
            // 8 B totals, 16 B positions], 64 sub-queues.  A work item rarely queues more than two chunks; a launch that asks for more than a sub-queue
            // holds counts them and is repeated with the queue it asked for
            const uint64_t per = 16 + 256 + 768 + 8 + 16, have = c->ws[WS_MP_Q].cap > 256 ? (c->ws[WS_MP_Q].cap - 256) / (per * 64) : 0;
            uint64_t want = std::max<uint64_t>(have, std::max<uint64_t>(64, (nw + nw / 2) / 64 + 32));
            if (q_max > have) want = std::max<uint64_t>(want, q_max + q_max / 4 + 16);
            if (want > have) HIPCHK(c, c->ws[WS_MP_Q].ensure(256 + want * per * 64));
            if (c->ws[WS_MP_Q].p != c->mp_bintab_at || c->ws[WS_MP_Q].cap != c->mp_bintab_cap) {      // (a new allocation: the tables once)
                uint32_t t64[64];
                memset(t64, 0, sizeof t64);
                fd_fill_bintab(t64);
                for (int k = 0; k < FD_BINTAB_WORDS; ++k) t64[32 + k] = t64[k];
                for (int m = 0; m < 4; ++m)
                    for (int k = 0; k < 4; ++k) { const uint32_t v = t64[32 + 7 + 5 * m + k]; t64[32 + 7 + 5 * m + k] = v > 0x7f7fffffu ? 0x7f7fffffu : v; }
                HIPCHK(c, hipMemcpyAsync(c->ws[WS_MP_Q].p, t64, sizeof t64, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipStreamSynchronize(st));
                c->mp_bintab_at = c->ws[WS_MP_Q].p; c->mp_bintab_cap = c->ws[WS_MP_Q].cap;
            }
            const uint64_t capq = std::min<uint64_t>((c->ws[WS_MP_Q].cap - 256) / (per * 64), 0x1ffffffu), nch = capq * 64;
            uint8_t *qb = c->ws[WS_MP_Q].as<uint8_t>() + 256;
            A.cap_subq = (uint32_t)capq;
            A.bintab = c->ws[WS_MP_Q].as<uint32_t>();
            A.chunk_base = (ulonglong2 *)qb; qb += nch * 16;
            A.chunk_hdr = (uint4 *)qb; qb += nch * 16;
            A.chunk_cnt = (uint2 *)qb; qb += nch * 8;
            A.chunk_ij = (uint32_t *)qb; qb += nch * 256;
            A.res_h = (uint32_t *)qb; qb += nch * 256;
            A.res_meta = (uint32_t *)qb; qb += nch * 256;
            A.res_d = (float *)qb;
        }
        HIPCHK(c, hipMemsetAsync(c->ws[WS_TOTAL].p, 0, tot_v.size() * 8, st));
        // a repeated launch drains its chunks again: the rescue votes of the attempt before (atomic adds of the chunks that did fit) must not count twice
        if (attempt && (mode & 32u)) HIPCHK(c, hipMemsetAsync(c->ws[WS_IDS_A].p, 0, std::max<uint64_t>(votes->n_counters, 1) * 4, st));
        {
            StageTimer t(c, "match_pairs", 0);
            fd_launch_match_pairs(A, st);
        }
        if (mp_dbg) {
            unsigned long long d[8];
            if (hipMemcpyAsync(d, A.dbg, 64, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                const double live = (double)std::max<unsigned long long>(d[0], 1), us = 0.01;      // 100 MHz ticks
                fprintf(stderr, "[mp] %llu work items: %llu live (%.2f us each; %.1f partners visited, %.1f pairs queued per item), %llu early exits (%.2f us each); %llu chunks drained (%.2f us each)\n",
                        (unsigned long long)nw, d[0], d[1] * us / live, d[6] / live, d[7] / live, d[2], d[2] ? d[3] * us / (double)d[2] : 0.0, d[4], d[4] ? d[5] * us / (double)d[4] : 0.0);
            }
        }
        HIPCHK(c, hipGetLastError());
        if (attempt == 0 && while_scanning && *while_scanning) (*while_scanning)();      // before the copy: one into pageable memory waits for the stream
        HIPCHK(c, hipMemcpyAsync(tot, c->ws[WS_TOTAL].p, tot_v.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        q_max = 0;
        for (int k = 0; k < 64; ++k) q_max = std::max<uint64_t>(q_max, tot[16 + MP_SUBQ_STRIDE * k]);
        if (q_max > A.cap_subq) {      // the drains saw a part of the pairs only: their counts mean nothing
            if (attempt == 3) FAIL(c, FDGPU_ERANGE, "match_pairs: the pair queue did not fit after regrowing");
            tot[0] = tot[1] = 0;
            continue;
        }
        if (tot[0] <= A.cap_found && tot[1] <= A.cap_cands) break;
        if (attempt == 3) FAIL(c, FDGPU_ERANGE, "match_pairs: output did not fit after regrowing");
    }
    if (mp_trace) fprintf(stderr, "[match_pairs] scan done at %.3f ms (found %llu, cands %llu)\n", mp_ms(), (unsigned long long)tot[0], (unsigned long long)tot[1]);
    if (mode & 32u) {   // the rows of the vote table: (largest count, how many hold it, which) per (slot, component, query residue)
        const uint8_t *base = c->ws[WS_IDS_B].as<uint8_t>();
        const uint64_t nr = votes->n_rows;
        auto up8 = [](uint64_t x) { return (x + 7) & ~(uint64_t)7; };
        const uint64_t o_roff = up8(n_cand * 8), o_rows = o_roff + up8(nr * 8), o_rlen = o_rows + up8(nr * sizeof(fd_vote_row)) + up8(n_cand * 4);
        fd_launch_vote_rows(A.votes, (const uint64_t *)(base + o_roff), (const uint32_t *)(base + o_rlen), nr, d_rows, st);
        HIPCHK(c, hipGetLastError());
        if (nr) HIPCHK(c, hipMemcpyAsync(votes->rows, d_rows, nr * sizeof(fd_vote_row), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (mp_trace) fprintf(stderr, "[match_pairs] vote rows done at %.3f ms (%llu rows, %llu counters)\n", mp_ms(), (unsigned long long)nr, (unsigned long long)votes->n_counters);
        return FDGPU_OK;
    }
    // mode bit 4: the records stay on the device (ws[WS_KEYS_A] = found triples, ws[WS_KEYS_B] = candidate pairs, in append order) for
    // the device-side retrieval glue (k_retrieve.hip); only the counts return
    if (mode & 16u) { *n_found = tot[0]; *n_cands = tot[1]; return FDGPU_OK; }
    // mode bit 3 (with pk_key / pk_val): the candidate pairs come back packed — key = slot << 16 | partner residue j, value =
    // query residue << 16 | residue i — and sorted by key on the device: half the bytes over PCIe and no bucketing on the host
    // (the rescue walks the pairs of one partner residue at a time).  Needs slots, residues and query residues below 2^16.
    const char *pm_env = getenv("FDGPU_PACK_MIN");      // tests force the packed form on small inputs
    const uint64_t pack_min = pm_env ? strtoull(pm_env, nullptr, 10) : (1ull << 18);   // a motif query's few thousand pairs are cheaper as they are
    const bool packed = (mode & 8u) && pk_key && pk_val && n_cand < 65536 && tot[1] >= pack_min;
    if (pk_key) *pk_key = nullptr;
    if (pk_val) *pk_val = nullptr;
    if (packed) {
        const uint64_t n = tot[1];
        // the packed pairs land in pinned buffers the CONTEXT keeps (valid until the next packed scan on this context; never freed by the caller)
        uint32_t *hk = (uint32_t *)c->host_pinned(0, std::max<uint64_t>(n, 1) * 4), *hv = (uint32_t *)c->host_pinned(1, std::max<uint64_t>(n, 1) * 4);
        fd_pair_rec *hf2 = (fd_pair_rec *)malloc(std::max<uint64_t>(tot[0], 1) * sizeof(fd_pair_rec));
        if (!hk || !hv || !hf2) { free(hf2); return FDGPU_ENOMEM; }
        if (n) {
            HIPCHK(c, c->ws[WS_MISC2].ensure(n * 4)); HIPCHK(c, c->ws[WS_MISC3].ensure(n * 4));
            HIPCHK(c, c->ws[WS_MISC4].ensure(n * 4)); HIPCHK(c, c->ws[WS_MISC5].ensure(n * 4));
            HIPCHK(c, c->ws[WS_GHIST].ensure((size_t)256 * std::max<uint32_t>(fd_rs_num_tiles(n), 1) * 4));
            HIPCHK(c, c->ws[WS_TOT].ensure((256 + (size_t)(fd_rs_num_tiles(n) / 128 + 2) * 256) * 8));
            uint32_t *ka = c->ws[WS_MISC2].as<uint32_t>(), *va = c->ws[WS_MISC3].as<uint32_t>(), *kb = c->ws[WS_MISC4].as<uint32_t>(),
                     *vb = c->ws[WS_MISC5].as<uint32_t>();
            fd_launch_pack_cands(A.cands, n, ka, va, st);
            int bits = 17;
            while (bits < 32 && (1ull << (bits - 16)) < std::max<uint64_t>(n_cand, 2)) ++bits;
            const int cur = sort_pairs(c, ka, va, kb, vb, n, bits);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(hk, cur ? kb : ka, n * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(hv, cur ? vb : va, n * 4, hipMemcpyDeviceToHost, st));
        }
        if (tot[0]) HIPCHK(c, hipMemcpyAsync(hf2, A.found, tot[0] * sizeof(fd_pair_rec), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (mp_trace) fprintf(stderr, "[match_pairs] packed copy done at %.3f ms\n", mp_ms());
        fd_sort_found(hf2, tot[0], n_cand);
        if (mp_trace) fprintf(stderr, "[match_pairs] found sorted at %.3f ms\n", mp_ms());
        *found = hf2; *n_found = tot[0]; *cands = nullptr; *n_cands = n; *pk_key = hk; *pk_val = hv;
        return FDGPU_OK;
    }
    // many found triples and no candidate pairs to keep (first scan of a large query: ~10^5 triples, most of them in the slot of the query's own
    // structure): (slot, i, j) order is made on the device — two stable radix sorts over the record index — instead of one host thread's
    // stable_sort of that slot.  FDGPU_FOUND_SORT=host: the host form (tests)
    bool sorted_on_device = false;
    {
        uint64_t max_len = 0;
        for (uint64_t k = 0; k < n_cand; ++k) max_len = std::max<uint64_t>(max_len, db->h_res_off[cand[k] + 1] - db->h_res_off[cand[k]]);
        const char *fs_env = getenv("FDGPU_FOUND_SORT");
        const uint64_t fs_min = fs_env && !strcmp(fs_env, "device") ? 2 : 32768;
        if (tot[0] >= fs_min && tot[1] == 0 && n_cand < 65536 && max_len < 65536 && !(fs_env && !strcmp(fs_env, "host"))) {
            const uint64_t n = tot[0];
            HIPCHK(c, c->ws[WS_MISC2].ensure(n * 4)); HIPCHK(c, c->ws[WS_MISC3].ensure(n * 4));
            HIPCHK(c, c->ws[WS_MISC4].ensure(n * 4)); HIPCHK(c, c->ws[WS_MISC5].ensure(n * 4));
            HIPCHK(c, c->ws[WS_IDS_A].ensure(n * sizeof(fd_pair_rec)));
            HIPCHK(c, c->ws[WS_GHIST].ensure((size_t)256 * std::max<uint32_t>(fd_rs_num_tiles(n), 1) * 4));
            HIPCHK(c, c->ws[WS_TOT].ensure((256 + (size_t)(fd_rs_num_tiles(n) / 128 + 2) * 256) * 8));
            uint32_t *ka = c->ws[WS_MISC2].as<uint32_t>(), *va = c->ws[WS_MISC3].as<uint32_t>(), *kb = c->ws[WS_MISC4].as<uint32_t>(), *vb = c->ws[WS_MISC5].as<uint32_t>();
            fd_launch_found_key_ij(A.found, n, ka, va, st);
            int cur = sort_pairs(c, ka, va, kb, vb, n, 32);
            uint32_t *k1 = cur ? kb : ka, *v1 = cur ? vb : va, *k2 = cur ? ka : kb, *v2 = cur ? va : vb;
            fd_launch_found_key_slot(A.found, v1, n, k1, st);
            int bits = 1;
            while (bits < 16 && (1ull << bits) < std::max<uint64_t>(n_cand, 2)) ++bits;
            cur = sort_pairs(c, k1, v1, k2, v2, n, bits);
            fd_launch_found_gather(A.found, cur ? v2 : v1, n, c->ws[WS_IDS_A].as<fd_pair_rec>(), st);
            HIPCHK(c, hipGetLastError());
            sorted_on_device = true;
        }
    }
    fd_pair_rec *hf = (fd_pair_rec *)malloc(std::max<uint64_t>(tot[0], 1) * sizeof(fd_pair_rec));
    fd_cand_rec *hc = (fd_cand_rec *)malloc(std::max<uint64_t>(tot[1], 1) * sizeof(fd_cand_rec));
    if (!hf || !hc) { free(hf); free(hc); return FDGPU_ENOMEM; }
    hipError_t e = hipSuccess;
    if (sorted_on_device) e = hipMemcpyAsync(hf, c->ws[WS_IDS_A].p, tot[0] * sizeof(fd_pair_rec), hipMemcpyDeviceToHost, st);
    if (!sorted_on_device && tot[0]) e = hipMemcpyAsync(hf, A.found, tot[0] * sizeof(fd_pair_rec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && tot[1]) e = hipMemcpyAsync(hc, A.cands, tot[1] * sizeof(fd_cand_rec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(hf); free(hc); c->err = std::string("match_pairs: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    // restore the reference's scan order (row-major over the prefilter sets, retrieve.rs:146-153): the
    // kernel appends with atomics, one contiguous run per (i, j) in observed-list order
    if (mp_trace) fprintf(stderr, "[match_pairs] copy done at %.3f ms\n", mp_ms());
    if (!sorted_on_device) fd_sort_found(hf, tot[0], n_cand);
    if (mp_trace) fprintf(stderr, "[match_pairs] found sorted at %.3f ms%s\n", mp_ms(), sorted_on_device ? " (on the device)" : "");
    // mode bit 2: the caller buckets the candidate pairs itself and does not depend on their order (the rescue only counts them)
    if (!(mode & 4u)) std::stable_sort(hc, hc + tot[1], [](const fd_cand_rec &a, const fd_cand_rec &b) {
        if (a.cand != b.cand) return a.cand < b.cand;
        if (a.i != b.i) return a.i < b.i;
        return a.j < b.j;
    });
    *found = hf; *n_found = tot[0]; *cands = hc; *n_cands = tot[1];
    return FDGPU_OK;
}
extern "C" int fdgpu_match_pairs(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, const uint32_t *cand, uint64_t n_cand,
                                 const fd_match_query *q, const fd_hash_params *p, fd_pair_rec **found, uint64_t *n_found,
                                 fd_cand_rec **cands, uint64_t *n_cands) { FD_LOCK(c);
    if (!q) return FDGPU_EINVAL;
    const uint64_t off[2] = {0, n_cand};
    return fd_match_pairs_multi(c, db, resname_std, 1, q, cand, off, p, found, n_found, cands, n_cands);
}

// Similarity metrics of n superpositions on the device (k_metrics): problem k compares ref[off[k] .. off[k+1]) (fixed points) with
// rot[k] * mov[...] + tran[k]; metrics[5k ..] = {tm_score, gdt_ts, gdt_ha, chamfer, hausdorff} (src/structure/metrics.rs:62-251).
extern "C" int fdgpu_metrics_batch(fdgpu_ctx *c, const float *ref, const float *mov, const uint64_t *off, uint64_t n, const float *rot, const float *tran,
                                   float *metrics) { FD_LOCK(c);
    if (!c || (n && (!ref || !mov || !off || !rot || !tran || !metrics))) return FDGPU_EINVAL;
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    const uint64_t npts = off[n];
    std::vector<float> d0(n);
    for (uint64_t k = 0; k < n; ++k) {   // d0_scale (metrics.rs:117-123) with the host's powf, like the reference
        const uint64_t len = off[k + 1] - off[k];
        d0[k] = len > 21 ? 1.24f * powf((float)len - 15.0f, 1.0f / 3.0f) - 1.8f : 0.5f;
    }
    HIPCHK(c, c->ws[WS_MISC0].ensure(std::max<uint64_t>(npts, 1) * 12));
    HIPCHK(c, c->ws[WS_MISC1].ensure(std::max<uint64_t>(npts, 1) * 12));
    HIPCHK(c, c->ws[WS_MISC2].ensure((n + 1) * 8));
    HIPCHK(c, c->ws[WS_MISC3].ensure(n * 4));
    HIPCHK(c, c->ws[WS_MISC4].ensure(n * 36));
    HIPCHK(c, c->ws[WS_MISC5].ensure(n * 12));
    HIPCHK(c, c->ws[WS_TILE_PO].ensure(n * 20));
    if (npts) {
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, ref, npts * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, mov, npts * 12, hipMemcpyHostToDevice, st));
    }
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC2].p, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC3].p, d0.data(), n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC4].p, rot, n * 36, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC5].p, tran, n * 12, hipMemcpyHostToDevice, st));
    fd_launch_metrics(c->ws[WS_MISC0].as<float>(), c->ws[WS_MISC1].as<float>(), c->ws[WS_MISC2].as<uint64_t>(), n, c->ws[WS_MISC4].as<float>(),
                      c->ws[WS_MISC5].as<float>(), c->ws[WS_MISC3].as<float>(), c->ws[WS_TILE_PO].as<float>(), st, npts);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(metrics, c->ws[WS_TILE_PO].p, n * 20, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

extern "C" int fdgpu_kabsch_batch(fdgpu_ctx *c, const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot,
                                  float *tran) { FD_LOCK(c);
    if (!c || (n && (!x || !y || !off || !rmsd || !rot || !tran))) return FDGPU_EINVAL;
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    uint64_t npts = off[n];
    const bool tr = getenv("FDGPU_TRACE") != nullptr;
    auto k0 = std::chrono::steady_clock::now();
    if (tr) { (void)hipStreamSynchronize(st); fprintf(stderr, "[kabsch] entry sync %.3f ms, %llu problems, %llu points\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - k0).count(), (unsigned long long)n, (unsigned long long)npts); }
    HIPCHK(c, c->ws[WS_MISC0].ensure(std::max<uint64_t>(npts, 1) * 12));
    HIPCHK(c, c->ws[WS_MISC1].ensure(std::max<uint64_t>(npts, 1) * 12));
    HIPCHK(c, c->ws[WS_MISC2].ensure((n + 1) * 8));
    HIPCHK(c, c->ws[WS_MISC3].ensure(n * 4));
    HIPCHK(c, c->ws[WS_MISC4].ensure(n * 36));
    HIPCHK(c, c->ws[WS_MISC5].ensure(n * 12));
    if (npts) {
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, x, npts * 12, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, y, npts * 12, hipMemcpyHostToDevice, st));
    }
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC2].p, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    fd_launch_kabsch(c->ws[WS_MISC0].as<float>(), c->ws[WS_MISC1].as<float>(), c->ws[WS_MISC2].as<uint64_t>(), n, c->ws[WS_MISC3].as<float>(),
                     c->ws[WS_MISC4].as<float>(), c->ws[WS_MISC5].as<float>(), st, npts);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rmsd, c->ws[WS_MISC3].p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(rot, c->ws[WS_MISC4].p, n * 36, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(tran, c->ws[WS_MISC5].p, n * 12, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (tr) fprintf(stderr, "[kabsch] total %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - k0).count());
    return FDGPU_OK;
}

// --partial-fit: LmsQcpSuperimposer with its default parameters (src/structure/lms_qcp.rs), one wavefront per problem
extern "C" int fdgpu_lms_qcp_batch(fdgpu_ctx *c, const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot,
                                   float *tran, uint32_t *core_len, uint32_t *core) { FD_LOCK(c);
    if (!c || (n && (!x || !y || !off || !rmsd || !rot || !tran))) return FDGPU_EINVAL;
    if (!n) return FDGPU_OK;
    for (uint64_t k = 0; k < n; ++k)
        if (off[k + 1] < off[k] + 3 || off[k + 1] - off[k] > 0xffffffffull) {   // the reference asserts >= 3 pairs (lms_qcp.rs:84)
            c->err = "fdgpu_lms_qcp_batch: every problem needs at least 3 point pairs";
            return FDGPU_EINVAL;
        }
    hipStream_t st = c->stream;
    const uint64_t npts = off[n];
    HIPCHK(c, c->ws[WS_MISC0].ensure(npts * 12));
    HIPCHK(c, c->ws[WS_MISC1].ensure(npts * 12));
    HIPCHK(c, c->ws[WS_MISC2].ensure((n + 1) * 8));
    HIPCHK(c, c->ws[WS_MISC3].ensure(n * 8));       // rmsd f32[n] | core_len u32[n]
    HIPCHK(c, c->ws[WS_MISC4].ensure(n * 48));      // rot f32[9n] | tran f32[3n]
    HIPCHK(c, c->ws[WS_MISC5].ensure(npts * 5));    // order u32[npts] | flags u8[npts]
    float *d_rmsd = c->ws[WS_MISC3].as<float>();
    uint32_t *d_core = (uint32_t *)(d_rmsd + n);
    float *d_rot = c->ws[WS_MISC4].as<float>(), *d_tran = d_rot + 9 * n;
    uint32_t *d_order = c->ws[WS_MISC5].as<uint32_t>();
    uint8_t *d_flags = (uint8_t *)(d_order + npts);
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, x, npts * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, y, npts * 12, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC2].p, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    fd_launch_lms_qcp(c->ws[WS_MISC0].as<float>(), c->ws[WS_MISC1].as<float>(), c->ws[WS_MISC2].as<uint64_t>(), n, d_rmsd, d_rot, d_tran, d_core,
                      d_flags, d_order, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rmsd, d_rmsd, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(rot, d_rot, n * 36, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(tran, d_tran, n * 12, hipMemcpyDeviceToHost, st));
    if (core_len) HIPCHK(c, hipMemcpyAsync(core_len, d_core, n * 4, hipMemcpyDeviceToHost, st));
    if (core) HIPCHK(c, hipMemcpyAsync(core, d_order, npts * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

