/* fdo_bench.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Multi-threaded driver used only by bench.py's cpu_baseline leg: the per-structure
 * hash + sort + dedup stage of controller/mod.rs:289-348 fanned out over structures with
 * OpenMP (the reference uses rayon par_iter at the same place), followed by the serial
 * count -> allocate -> fill -> finish table build. */
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

/* Hash every structure (parallel over structures) into CSR of sorted-unique lists.  Every thread hashes into its own scratch buffer
 * and appends the sorted-unique list to its own arena (both only grow): no allocation per structure — the per-structure malloc / free
 * of >= 128 KB blocks this replaced went through mmap / munmap, whose TLB shootdowns made 256 threads no faster than 64.  The arenas are
 * concatenated in parallel. */
#include <omp.h>
#include <malloc.h>
/* sort_unstable + dedup (controller/mod.rs:343-345) as an LSD radix sort into a caller-owned buffer: glibc's qsort mallocs a merge
 * buffer of the list's size per call — above the mmap threshold for a 300-residue structure, i.e. one more mmap / munmap (and its TLB
 * shootdown across all threads) per structure */
static uint64_t sort_dedup_radix(uint32_t *v, uint64_t n, uint32_t *tmp) {
    if (n == 0) return 0;
    uint32_t *a = v, *b = tmp;
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 8 * pass;
        uint64_t cnt[257] = {0};
        for (uint64_t k = 0; k < n; ++k) ++cnt[((a[k] >> sh) & 255u) + 1];
        if (cnt[((a[0] >> sh) & 255u) + 1] == n) continue;      /* all keys share this digit */
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (uint64_t k = 0; k < n; ++k) b[cnt[(a[k] >> sh) & 255u]++] = a[k];
        uint32_t *t = a; a = b; b = t;
    }
    uint64_t m = 1;
    v[0] = a[0];
    for (uint64_t k = 1; k < n; ++k)
        if (a[k] != v[m - 1]) v[m++] = a[k];      /* a == v: in-place compaction; a == tmp: copy back while compacting (m <= k) */
    return m;
}
int fdo_hash_structure_buf(const fdo_structure *s, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff, uint32_t **buf, uint64_t *cap_io,
                           uint64_t *n_out);
/* threads of the OpenMP regions below (OMP_NUM_THREADS is read once, when the runtime starts: a bench that times several thread counts
 * in one process has to say so here) */
void fdo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int fdo_hash_batch(const fdo_structure *const *structs, uint64_t S, uint64_t nbin_dist, uint64_t nbin_angle,
                   float dist_cutoff, uint32_t **out_hashes, uint64_t **out_off) {
    uint64_t *cnt = (uint64_t *)calloc(S + 1, sizeof *cnt);
    uint64_t *where = (uint64_t *)calloc(S ? S : 1, sizeof *where);      /* offset of the structure's list inside its thread's arena */
    int *owner = (int *)calloc(S ? S : 1, sizeof *owner);
    int T = omp_get_max_threads();
    mallopt(M_MMAP_THRESHOLD, 1 << 30);     /* the per-thread buffers below grow inside the arenas, not through mmap */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    uint32_t **arena = (uint32_t **)calloc((size_t)T, sizeof *arena);
    uint64_t *order = (uint64_t *)malloc((S ? S : 1) * sizeof *order);
    {   /* counting sort of the structures by residue count, descending */
        int64_t mx = 0;
        for (uint64_t id = 0; id < S; ++id) mx = structs[id]->n > mx ? structs[id]->n : mx;
        uint64_t *cn = (uint64_t *)calloc((size_t)mx + 2, sizeof *cn);
        for (uint64_t id = 0; id < S; ++id) ++cn[mx - structs[id]->n + 1];
        for (int64_t k = 0; k <= mx; ++k) cn[k + 1] += cn[k];
        for (uint64_t id = 0; id < S; ++id) order[cn[mx - structs[id]->n]++] = id;
        free(cn);
    }
#pragma omp parallel
    {
        const int tid = omp_get_thread_num();
        uint32_t *scratch = NULL, *mine = NULL, *stmp = NULL;
        uint64_t cap = 0, acap = 0, an = 0, tcap = 0;
#pragma omp for schedule(dynamic, 1)
        for (int64_t k = 0; k < (int64_t)S; ++k) {
            const int64_t id = (int64_t)order[k];      /* longest structures first: the loop ends with its slowest item (cost ~ residues^2) */
            uint64_t n = 0;
            fdo_hash_structure_buf(structs[id], nbin_dist, nbin_angle, dist_cutoff, &scratch, &cap, &n);
            if (n > tcap) { tcap = cap; stmp = (uint32_t *)realloc(stmp, tcap * sizeof *stmp); }
            n = sort_dedup_radix(scratch, n, stmp);
            if (an + n > acap) { acap = (an + n) * 2 + (1u << 20); mine = (uint32_t *)realloc(mine, acap * sizeof *mine); }
            memcpy(mine + an, scratch, n * sizeof *mine);
            where[id] = an; owner[id] = tid; cnt[id + 1] = n;
            an += n;
        }
        arena[tid] = mine;
        free(scratch); free(stmp);
    }
    for (uint64_t id = 0; id < S; ++id) cnt[id + 1] += cnt[id];
    uint32_t *all = (uint32_t *)malloc((cnt[S] ? cnt[S] : 1) * sizeof *all);
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < (int64_t)S; ++id)
        memcpy(all + cnt[id], arena[owner[id]] + where[id], (cnt[id + 1] - cnt[id]) * sizeof *all);
    for (int t = 0; t < T; ++t) free(arena[t]);
    free(arena); free(where); free(owner); free(order);
    *out_hashes = all;
    *out_off = cnt;
    return 0;
}

/* Table build with the reference's lock-free ownership partition (controller/mod.rs:349-358,
 * 427-437): T workers each scan the whole (hash, id) stream and handle only the hashes they own.
 * The reference assigns ownership by hash % T; here ownership is by table page ((hash >> 6) % T) so
 * that two workers never allocate the same page — same idea, same cost structure (every worker
 * reads every key). */
fdo_index *fdo_build_index_from_lists_mt(const uint32_t *hashes, const uint64_t *off, uint64_t S, int T) {
    fdo_index *ix = fdo_index_new(30);
    if (T < 1) T = 1;
    for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel num_threads(T)
        {
            uint32_t tid = (uint32_t)omp_get_thread_num(), nt = (uint32_t)omp_get_num_threads();
            for (uint64_t id = 0; id < S; ++id)
                for (uint64_t k = off[id]; k < off[id + 1]; ++k) {
                    uint32_t h = hashes[k];
                    if ((h >> 6) % nt != tid) continue;
                    if (pass == 0) fdo_index_count_single_entry(ix, h, id);
                    else fdo_index_add_single_entry(ix, h, id);
                }
        }
        if (pass == 0) fdo_index_allocate_entries(ix);
    }
    fdo_index_finish(ix);
    return ix;
}

/* Query half of bench.py's cpu_baseline: Q motif queries against an index, fanned out over queries with OpenMP exactly where the
 * reference fans out (`queries.into_par_iter()`, cli/workflows/query_pdb.rs:348): per query make_query_map (query.rs:208-329,
 * two get_entries per observed hash), count_query (count_query.rs:82-220), stable sort by idf descending + truncate
 * (query_pdb.rs:404-411), retrieval_wrapper on the first match_top candidates (retrieve.rs:364-552; the reference re-reads and
 * re-parses the candidate's file there — here the packed coordinates are handed over, parse time excluded on both sides).
 * Query t = residues q_res[q_off[t] .. q_off[t+1]) (0-based indices) of database structure q_struct[t].
 * stage_s[0..2] = thread-seconds spent in make_query_map / count_query / retrieval, summed over the threads.
 * Returns the wall time in seconds. */
#include <stdio.h>
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static fdo_structure *struct_at(const uint64_t *res_off, const float *n_xyz, const float *ca_xyz, const float *cb_xyz, const uint8_t *aa, uint64_t s) {
    uint64_t a = res_off[s], b = res_off[s + 1];
    return fdo_structure_from_packed((int32_t)(b - a), n_xyz + 3 * a, ca_xyz + 3 * a, cb_xyz + 3 * a, NULL, aa + a, NULL);
}
typedef struct { float idf; uint64_t nid; uint64_t pos; } rank_t;
static int rank_cmp(const void *x, const void *y) {
    const rank_t *a = (const rank_t *)x, *b = (const rank_t *)y;
    if (a->idf > b->idf) return -1;
    if (a->idf < b->idf) return 1;
    return a->pos < b->pos ? -1 : (a->pos > b->pos ? 1 : 0);   /* stable */
}
double fdo_query_bench(const fdo_index *ix, const uint64_t *nres, uint64_t S, const uint64_t *res_off, const float *n_xyz, const float *ca_xyz,
                       const float *cb_xyz, const uint8_t *aa, uint64_t n_queries, const uint64_t *q_struct, const uint64_t *q_off,
                       const uint64_t *q_res, uint64_t top_n, uint64_t match_top, int n_threads, uint64_t *n_hits, uint64_t *n_matches,
                       double stage_s[3]) {
    uint64_t hits = 0, matches = 0;
    double s0 = 0, s1 = 0, s2 = 0;
    const float dthr[1] = {0.5f}, athr[1] = {5.0f};
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : hits, matches, s0, s1, s2)
    for (int64_t t = 0; t < (int64_t)n_queries; ++t) {
        double a = now_s();
        fdo_structure *qs = struct_at(res_off, n_xyz, ca_xyz, cb_xyz, aa, q_struct[t]);
        char buf[4096]; size_t w = 0;
        for (uint64_t k = q_off[t]; k < q_off[t + 1] && w + 32 < sizeof buf; ++k)
            w += (size_t)snprintf(buf + w, sizeof buf - w, "%sA%llu", k > q_off[t] ? "," : "", (unsigned long long)(q_res[k] + 1));
        fdo_query_spec *sp = fdo_parse_query_string(buf, 'A');
        fdo_query_map *m = fdo_make_query_map(qs, sp, 0, 0, dthr, 1, athr, 1, 20.0f, 0, ix, (float)S);
        fdo_query_spec_free(sp);
        double b = now_s();
        fdo_count_result *res = NULL;
        uint64_t n = fdo_count_query(m, ix, nres, S, -1.0f, 0.5f, &res);
        rank_t *rk = (rank_t *)malloc((n ? n : 1) * sizeof *rk);
        for (uint64_t k = 0; k < n; ++k) { rk[k].idf = res[k].idf; rk[k].nid = res[k].nid; rk[k].pos = k; }
        qsort(rk, n, sizeof *rk, rank_cmp);
        uint64_t keep = n < top_n ? n : top_n;
        double c = now_s();
        uint64_t nm = 0;
        for (uint64_t k = 0; k < keep && k < match_top; ++k) {
            fdo_structure *tg = struct_at(res_off, n_xyz, ca_xyz, cb_xyz, aa, rk[k].nid);
            fdo_retrieval *r = fdo_retrieve(tg, qs, m, 2, 0, 0, 20.0f, 1.0f);
            nm += r->n_matches;
            fdo_retrieval_free(r);
            fdo_structure_free(tg);
        }
        double d = now_s();
        hits += keep; matches += nm; s0 += b - a; s1 += c - b; s2 += d - c;
        free(rk); fdo_free(res); fdo_query_map_free(m); fdo_structure_free(qs);
    }
    double dt = now_s() - t0;
    if (n_hits) *n_hits = hits;
    if (n_matches) *n_matches = matches;
    if (stage_s) { stage_s[0] = s0; stage_s[1] = s1; stage_s[2] = s2; }
    return dt;
}
