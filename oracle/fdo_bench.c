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
