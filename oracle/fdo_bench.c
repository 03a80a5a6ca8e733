/* fdo_bench.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Multi-threaded driver used only by bench.py's cpu_baseline leg: the per-structure
 * hash + sort + dedup stage of controller/mod.rs:289-348 fanned out over structures with
 * OpenMP (the reference uses rayon par_iter at the same place), followed by the serial
 * count -> allocate -> fill -> finish table build. */
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

/* Hash every structure (parallel over structures) into CSR of sorted-unique lists. */
int fdo_hash_batch(const fdo_structure *const *structs, uint64_t S, uint64_t nbin_dist, uint64_t nbin_angle,
                   float dist_cutoff, uint32_t **out_hashes, uint64_t **out_off) {
    uint32_t **lists = (uint32_t **)calloc(S ? S : 1, sizeof *lists);
    uint64_t *cnt = (uint64_t *)calloc(S + 1, sizeof *cnt);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t id = 0; id < (int64_t)S; ++id) {
        uint32_t *h = NULL;
        uint64_t n = 0;
        fdo_hash_structure(structs[id], nbin_dist, nbin_angle, dist_cutoff, &h, &n);
        n = fdo_sort_dedup_u32(h, n);
        lists[id] = h;
        cnt[id + 1] = n;
    }
    for (uint64_t id = 0; id < S; ++id) cnt[id + 1] += cnt[id];
    uint32_t *all = (uint32_t *)malloc((cnt[S] ? cnt[S] : 1) * sizeof *all);
    for (uint64_t id = 0; id < S; ++id) {
        memcpy(all + cnt[id], lists[id], (cnt[id + 1] - cnt[id]) * sizeof *all);
        free(lists[id]);
    }
    free(lists);
    *out_hashes = all;
    *out_off = cnt;
    return 0;
}

/* Table build with the reference's lock-free ownership partition (controller/mod.rs:349-358,
 * 427-437): T workers each scan the whole (hash, id) stream and handle only the hashes they own.
 * The reference assigns ownership by hash % T; here ownership is by table page ((hash >> 6) % T) so
 * that two workers never allocate the same page — same idea, same cost structure (every worker
 * reads every key). */
#include <omp.h>
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
