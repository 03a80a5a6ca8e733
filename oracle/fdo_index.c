/* fdo_index.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates src/index/indextable.rs (dense count -> prefix -> fill -> prune, delta + 7-bit
 * varint codec, .offset file), src/controller/mod.rs:274-441 (two-pass build),
 * src/index/lookup.rs:17-58, src/cli/config.rs:66-97.
 *
 * The reference allocates dense 2^30-entry `offsets`/`last_id` vectors.  The same dense
 * semantics are kept here, but the tables are paged (64-entry pages allocated on first
 * touch) so that the test-suite does not need 16 GiB; the observable result (value bytes,
 * sparse hashes, offsets) is identical by construction because every operation is the same
 * per-hash read-modify-write in the same order. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include "fd_oracle.h"

#define PAGE_BITS 6
#define PAGE_SIZE (1u << PAGE_BITS)

typedef struct { uint64_t off[PAGE_SIZE]; uint64_t last[PAGE_SIZE]; } page_t;

struct fdo_index {
    uint32_t bits;
    uint64_t total_hashes;
    page_t **pages; /* dense phase */
    uint64_t n_pages;
    uint8_t *entries;
    uint64_t total_entries;
    /* sparse phase */
    uint64_t H;
    uint32_t *hashes;
    uint64_t *offsets; /* H+1 */
    int loaded;
};

static page_t *get_page(fdo_index *ix, uint32_t hash) {
    uint64_t p = hash >> PAGE_BITS;
    if (!ix->pages[p]) {
        page_t *pg = (page_t *)malloc(sizeof *pg);
        memset(pg->off, 0, sizeof pg->off);
        memset(pg->last, 0xff, sizeof pg->last); /* usize::MAX */
        ix->pages[p] = pg;
    }
    return ix->pages[p];
}

/* indextable.rs:23-41 */
fdo_index *fdo_index_new(uint32_t hash_bits) {
    fdo_index *ix = (fdo_index *)calloc(1, sizeof *ix);
    ix->bits = hash_bits;
    ix->total_hashes = 1ull << hash_bits;
    ix->n_pages = (ix->total_hashes + PAGE_SIZE - 1) >> PAGE_BITS;
    ix->pages = (page_t **)calloc(ix->n_pages, sizeof(page_t *));
    return ix;
}

static unsigned ilog2_u64(uint64_t v) { return 63u - (unsigned)__builtin_clzll(v); }

static uint64_t entry_len(uint64_t last, uint64_t id) {
    if (last == UINT64_MAX) return id == 0 ? 1 : 1 + ilog2_u64(id) / 7;
    return 1 + ilog2_u64(id - last) / 7; /* id == last would panic in Rust (ilog2(0)); never happens after dedup */
}

/* indextable.rs:88-105 */
void fdo_index_count_single_entry(fdo_index *ix, uint32_t hash, uint64_t id) {
    page_t *pg = get_page(ix, hash);
    uint32_t k = hash & (PAGE_SIZE - 1);
    pg->off[k] += entry_len(pg->last[k], id);
    pg->last[k] = id;
}

/* indextable.rs:204-237: exclusive prefix, allocate value buffer, reset last_id */
void fdo_index_allocate_entries(fdo_index *ix) {
    uint64_t total = 0;
    for (uint64_t p = 0; p < ix->n_pages; ++p) {
        page_t *pg = ix->pages[p];
        if (!pg) continue; /* all-zero page: offsets[i] = total for each i, total unchanged */
        for (uint32_t k = 0; k < PAGE_SIZE; ++k) {
            uint64_t cur = pg->off[k];
            pg->off[k] = total;
            total += cur;
            pg->last[k] = UINT64_MAX;
        }
    }
    ix->total_entries = total;
    ix->entries = (uint8_t *)calloc(total ? total : 1, 1);
}

/* indextable.rs:397-418 */
uint64_t fdo_split_by_seven_bits(uint64_t id, uint8_t out[10]) {
    uint64_t len = 0;
    while (id > 0) {
        uint8_t b = (uint8_t)(id & 0x7f);
        id >>= 7;
        out[len++] = id > 0 ? (uint8_t)(b | 0x80) : b;
    }
    if (len == 0) out[len++] = 0;
    return len;
}

/* indextable.rs:171-202 */
void fdo_index_add_single_entry(fdo_index *ix, uint32_t hash, uint64_t id) {
    page_t *pg = get_page(ix, hash);
    uint32_t k = hash & (PAGE_SIZE - 1);
    uint64_t count = entry_len(pg->last[k], id);
    uint64_t offset = pg->off[k];
    pg->off[k] += count;
    uint64_t prev = pg->last[k];
    pg->last[k] = id;
    uint64_t v = prev == UINT64_MAX ? id : id - prev;
    uint8_t buf[10];
    uint64_t nb = fdo_split_by_seven_bits(v, buf);
    memcpy(ix->entries + offset, buf, nb);
}

/* indextable.rs:239-295: after the fill, offsets[h] == end of list h; shifting right by one
 * (wrapup) turns them back into starts, then prune keeps hashes with start < end. */
void fdo_index_finish(fdo_index *ix) {
    uint64_t H = 0;
    for (uint64_t p = 0; p < ix->n_pages; ++p) {
        page_t *pg = ix->pages[p];
        if (!pg) continue;
        for (uint32_t k = 0; k < PAGE_SIZE; ++k)
            if (pg->last[k] != UINT64_MAX) ++H;
    }
    ix->H = H;
    ix->hashes = (uint32_t *)malloc((H ? H : 1) * sizeof(uint32_t));
    ix->offsets = (uint64_t *)malloc((H + 1) * sizeof(uint64_t));
    ix->offsets[0] = 0;
    uint64_t m = 0, prev_end = 0;
    for (uint64_t p = 0; p < ix->n_pages; ++p) {
        page_t *pg = ix->pages[p];
        if (!pg) continue;
        for (uint32_t k = 0; k < PAGE_SIZE; ++k) {
            uint64_t end = pg->off[k]; /* end of list (p,k) after fill */
            if (end > prev_end) {      /* start < end */
                ix->hashes[m] = (uint32_t)((p << PAGE_BITS) | k);
                ix->offsets[++m] = end;
            }
            prev_end = end;
        }
        free(pg);
        ix->pages[p] = NULL;
    }
    ix->H = m;
    free(ix->pages);
    ix->pages = NULL;
}

/* indextable.rs:239-264 (value file) + :297-326 (.offset) */
int fdo_index_save(const fdo_index *ix, const char *prefix) {
    FILE *f = fopen(prefix, "wb");
    if (!f) return -1;
    if (ix->total_entries && fwrite(ix->entries, 1, ix->total_entries, f) != ix->total_entries) { fclose(f); return -1; }
    fclose(f);
    size_t L = strlen(prefix);
    char *p = (char *)malloc(L + 16);
    sprintf(p, "%s.offset", prefix);
    f = fopen(p, "wb");
    free(p);
    if (!f) return -1;
    uint64_t H = ix->H;
    fwrite(&H, 8, 1, f);
    fwrite(ix->hashes, 4, H, f);
    fwrite(ix->offsets, 8, H + 1, f);
    fclose(f);
    return 0;
}

static uint8_t *read_file(const char *path, uint64_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return NULL; }
    fclose(f);
    *len = (uint64_t)n;
    return b;
}

/* indextable.rs:331-394 */
fdo_index *fdo_index_load(const char *prefix) {
    size_t L = strlen(prefix);
    char *p = (char *)malloc(L + 16);
    sprintf(p, "%s.offset", prefix);
    uint64_t olen = 0;
    uint8_t *ob = read_file(p, &olen);
    if (!ob) { free(p); return NULL; }
    sprintf(p, "%s.value", prefix);
    struct stat st;
    uint64_t vlen = 0;
    uint8_t *vb = stat(p, &st) == 0 ? read_file(p, &vlen) : read_file(prefix, &vlen);
    free(p);
    if (!vb) { free(ob); return NULL; }
    uint64_t H;
    memcpy(&H, ob, 8);
    if (olen < 8 + H * 4 + (H + 1) * 8) { free(ob); free(vb); return NULL; }
    fdo_index *ix = (fdo_index *)calloc(1, sizeof *ix);
    ix->loaded = 1;
    ix->H = H;
    ix->hashes = (uint32_t *)malloc((H ? H : 1) * 4);
    ix->offsets = (uint64_t *)malloc((H + 1) * 8);
    memcpy(ix->hashes, ob + 8, H * 4);
    memcpy(ix->offsets, ob + 8 + H * 4, (H + 1) * 8);
    ix->entries = vb;
    ix->total_entries = vlen;
    free(ob);
    return ix;
}

void fdo_index_free(fdo_index *ix) {
    if (!ix) return;
    if (ix->pages) {
        for (uint64_t p = 0; p < ix->n_pages; ++p) free(ix->pages[p]);
        free(ix->pages);
    }
    free(ix->entries); free(ix->hashes); free(ix->offsets);
    free(ix);
}
uint64_t fdo_index_num_hashes(const fdo_index *ix) { return ix->H; }
const uint32_t *fdo_index_hashes(const fdo_index *ix) { return ix->hashes; }
const uint64_t *fdo_index_offsets(const fdo_index *ix) { return ix->offsets; }
const uint8_t *fdo_index_values(const fdo_index *ix) { return ix->entries; }
uint64_t fdo_index_value_len(const fdo_index *ix) { return ix->total_entries; }

/* indextable.rs:44-86 + :421-463: binary search, slice, varint-delta decode */
uint64_t fdo_index_get_entries(const fdo_index *ix, uint32_t hash, uint64_t **ids) {
    *ids = NULL;
    uint64_t lo = 0, hi = ix->H;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (ix->hashes[mid] < hash) lo = mid + 1; else hi = mid;
    }
    if (lo >= ix->H || ix->hashes[lo] != hash) return 0;
    uint64_t start = ix->offsets[lo], end = ix->offsets[lo + 1];
    if (end <= start) return 0;
    uint64_t *out = (uint64_t *)malloc((end - start) * sizeof(uint64_t));
    uint64_t n = 0, prev = UINT64_MAX, acc = 0;
    unsigned shift = 0;
    for (uint64_t k = start; k < end; ++k) {
        uint8_t b = ix->entries[k];
        acc |= (uint64_t)(b & 0x7f) << shift;
        if (b & 0x80) { shift += 7; continue; }
        prev = prev == UINT64_MAX ? acc : prev + acc;
        out[n++] = prev;
        acc = 0;
        shift = 0;
    }
    *ids = out;
    return n;
}

/* controller/mod.rs:274-441, single-threaded: pass 1 counts, pass 2 fills; both re-hash */
fdo_index *fdo_build_index(const fdo_structure *const *structs, uint64_t S, uint64_t nbin_dist,
                           uint64_t nbin_angle, float dist_cutoff, uint64_t max_residue, uint64_t *nres,
                           float *plddt) {
    fdo_index *ix = fdo_index_new(fdo_get_hash_type() == 3 ? 30 : 32);   /* encoding_bits (geometry/core.rs:79-93); pages are allocated on touch */
    for (int pass = 0; pass < 2; ++pass) {
        for (uint64_t id = 0; id < S; ++id) {
            const fdo_structure *s = structs[id];
            if ((uint64_t)s->num_residues_raw > max_residue) { /* mod.rs:313-318: id kept, no hashes */
                if (pass == 0) { if (nres) nres[id] = 0; if (plddt) plddt[id] = 0.0f; }
                continue;
            }
            if (pass == 0) {
                if (nres) nres[id] = (uint64_t)s->n;
                if (plddt) plddt[id] = fdo_avg_plddt(s);
            }
            uint32_t *h = NULL;
            uint64_t n = 0;
            fdo_hash_structure(s, nbin_dist, nbin_angle, dist_cutoff, &h, &n);
            n = fdo_sort_dedup_u32(h, n);
            for (uint64_t k = 0; k < n; ++k) {
                if (pass == 0) fdo_index_count_single_entry(ix, h[k], id);
                else fdo_index_add_single_entry(ix, h[k], id);
            }
            free(h);
        }
        if (pass == 0) fdo_index_allocate_entries(ix);
    }
    fdo_index_finish(ix);
    return ix;
}

fdo_index *fdo_build_index_from_lists(const uint32_t *hashes, const uint64_t *off, uint64_t S) {
    fdo_index *ix = fdo_index_new(fdo_get_hash_type() == 3 ? 30 : 32);   /* encoding_bits (geometry/core.rs:79-93); pages are allocated on touch */
    for (uint64_t id = 0; id < S; ++id)
        for (uint64_t k = off[id]; k < off[id + 1]; ++k) fdo_index_count_single_entry(ix, hashes[k], id);
    fdo_index_allocate_entries(ix);
    for (uint64_t id = 0; id < S; ++id)
        for (uint64_t k = off[id]; k < off[id + 1]; ++k) fdo_index_add_single_entry(ix, hashes[k], id);
    fdo_index_finish(ix);
    return ix;
}

/* Rust `{}` Display for f32: shortest round-trip digits, never exponent form, integral values
 * without a fraction, "NaN"/"inf"/"-inf" (SURVEY §8c (5)). */
int fdo_format_f32_display(float v, char *buf, size_t cap) {
    if (v != v) return snprintf(buf, cap, "NaN");
    if (isinf(v)) return snprintf(buf, cap, v > 0 ? "inf" : "-inf");
    char tmp[64];
    int prec;
    for (prec = 1; prec <= 9; ++prec) { /* shortest %.{prec}e that round-trips */
        snprintf(tmp, sizeof tmp, "%.*e", prec - 1, (double)v);
        if (strtof(tmp, NULL) == v) break;
    }
    /* tmp = d.ddddde[+-]XX -> expand to plain decimal */
    char digits[16];
    int nd = 0, neg = 0;
    const char *p = tmp;
    if (*p == '-') { neg = 1; ++p; }
    for (; *p && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int ex = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd; /* strip trailing zeros of the mantissa */
    char out[96];
    int o = 0;
    if (neg) out[o++] = '-';
    if (ex < 0) {
        out[o++] = '0'; out[o++] = '.';
        for (int k = 0; k < -ex - 1; ++k) out[o++] = '0';
        for (int k = 0; k < nd; ++k) out[o++] = digits[k];
    } else {
        for (int k = 0; k <= ex; ++k) out[o++] = k < nd ? digits[k] : '0';
        if (nd > ex + 1) {
            out[o++] = '.';
            for (int k = ex + 1; k < nd; ++k) out[o++] = digits[k];
        }
    }
    out[o] = 0;
    return snprintf(buf, cap, "%s", out);
}

/* lookup.rs:35-56: id \t tid \t nres \t plddt \t db_key */
int fdo_save_lookup(const char *path, const char *const *tids, const uint64_t *nres, const float *plddt,
                    const uint64_t *db_key, uint64_t S) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    char fb[64];
    for (uint64_t i = 0; i < S; ++i) {
        fdo_format_f32_display(plddt[i], fb, sizeof fb);
        fprintf(f, "%llu\t%s\t%llu\t%s\t%llu\n", (unsigned long long)i, tids[i], (unsigned long long)nres[i], fb,
                (unsigned long long)(db_key ? db_key[i] : i));
    }
    fclose(f);
    return 0;
}

/* cli/config.rs:66-97 (toml 0.8 serialisation of IndexConfig without foldcomp/multiple_bin) */
int fdo_save_type(const char *path, uint64_t chunk_size, float grid_width, uint64_t max_residue,
                  uint64_t nbin_angle, uint64_t nbin_dist) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    char fb[64];
    fdo_format_f32_display(grid_width, fb, sizeof fb);
    int has_dot = strchr(fb, '.') != NULL;
    fprintf(f, "chunk_size = %llu\ngrid_width = %s%s\nhash_type = \"PDBTrRosetta\"\ninput_format = \"PDB\"\n"
               "max_residue = %llu\nnum_bin_angle = %llu\nnum_bin_dist = %llu\n",
            (unsigned long long)chunk_size, fb, has_dot ? "" : ".0", (unsigned long long)max_residue,
            (unsigned long long)nbin_angle, (unsigned long long)nbin_dist);
    fclose(f);
    return 0;
}

/* index over BORROWED arrays (bench.py's query cpu_baseline: the arrays are the GPU build's export, which the parity tests pin
 * byte for byte to fdo_build_index); release with fdo_index_free_borrowed — the arrays stay the caller's */
fdo_index *fdo_index_borrow(const uint32_t *hashes, const uint64_t *offsets, uint64_t H, const uint8_t *values, uint64_t vlen) {
    fdo_index *ix = (fdo_index *)calloc(1, sizeof *ix);
    ix->loaded = 1;
    ix->H = H;
    ix->hashes = (uint32_t *)hashes;
    ix->offsets = (uint64_t *)offsets;
    ix->entries = (uint8_t *)values;
    ix->total_entries = vlen;
    return ix;
}
void fdo_index_free_borrowed(fdo_index *ix) { free(ix); }
