/* fdo_query.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates src/controller/query.rs:17-384 (parse_query_string, make_query_map and helpers)
 * and src/controller/count_query.rs:82-273 (count_query, build_node_groups).
 *
 * Iteration-order note: the reference keeps the query map in an FxHashMap, so the order in
 * which query hashes are visited (and hence the f32 summation order of idf) follows
 * hashbrown's bucket order.  This restatement fixes a canonical order instead: hashes in
 * first-insertion order, node groups by ascending node index.  All integer outputs are
 * order-independent; idf sums are compared with a relative tolerance. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

/* ------------------------------------------------------------------ parse_query_string */
static int is_alpha(int c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

/* utils/convert.rs:223-262 */
static int one_letter_to_u8_vec(char c, uint8_t out[20]) {
    static const char *std1 = "ARNDCQEGHILKMFPSTWYV";
    const char *p = strchr(std1, c);
    if (p && c) { out[0] = (uint8_t)(p - std1); return 1; }
    int n = 0;
    switch (c) {
        case 'B': out[0] = 2; out[1] = 3; return 2;
        case 'Z': out[0] = 5; out[1] = 6; return 2;
        case 'X': case 'x': for (n = 0; n < 20; ++n) out[n] = (uint8_t)n; return 20;
        case 'J': out[0] = 9; out[1] = 10; return 2;
        case 'U': out[0] = 4; return 1;
        case 'O': out[0] = 11; return 1;
        case 'p': out[0] = 1; out[1] = 8; out[2] = 11; return 3;
        case 'n': out[0] = 3; out[1] = 6; return 2;
        case 'h': { static const uint8_t v[] = {2, 5, 15, 16, 18}; memcpy(out, v, 5); return 5; }
        case 'b': { static const uint8_t v[] = {0, 4, 7, 9, 10, 12, 13, 14, 19}; memcpy(out, v, 9); return 9; }
        case 'a': { static const uint8_t v[] = {8, 13, 17, 18}; memcpy(out, v, 4); return 4; }
        default: out[0] = 255; return 1;
    }
}

static void spec_push(fdo_query_spec *q, uint64_t *cap, uint8_t chain, uint64_t serial, const uint8_t *subs,
                      int64_t n_subs) {
    if (q->n == *cap) {
        *cap = *cap ? *cap * 2 : 16;
        q->chain = (uint8_t *)realloc(q->chain, *cap);
        q->serial = (uint64_t *)realloc(q->serial, *cap * sizeof(uint64_t));
        q->subs = (uint8_t **)realloc(q->subs, *cap * sizeof(uint8_t *));
        q->n_subs = (uint64_t *)realloc(q->n_subs, *cap * sizeof(uint64_t));
    }
    q->chain[q->n] = chain;
    q->serial[q->n] = serial;
    if (n_subs < 0) { q->subs[q->n] = NULL; q->n_subs[q->n] = 0; }
    else {
        q->subs[q->n] = (uint8_t *)malloc((size_t)(n_subs > 0 ? n_subs : 1));
        memcpy(q->subs[q->n], subs, (size_t)n_subs);
        q->n_subs[q->n] = (uint64_t)n_subs;
    }
    q->n++;
}

static int parse_u64_str(const char *s, size_t len, uint64_t *out) {
    if (len == 0) return 0;
    size_t k = 0;
    if (s[0] == '+') k = 1;
    if (k == len) return 0;
    uint64_t v = 0;
    for (; k < len; ++k) {
        if (s[k] < '0' || s[k] > '9') return 0;
        v = v * 10 + (uint64_t)(s[k] - '0');
    }
    *out = v;
    return 1;
}

/* query.rs:331-384. Returns NULL where the reference would panic (malformed number). */
fdo_query_spec *fdo_parse_query_string(const char *qs, uint8_t default_chain) {
    fdo_query_spec *q = (fdo_query_spec *)calloc(1, sizeof *q);
    uint64_t cap = 0;
    if (!qs || !*qs) return q;
    if (!is_alpha(default_chain)) default_chain = 'A';
    size_t L = strlen(qs);
    char *buf = (char *)malloc(L + 1);
    size_t m = 0;
    for (size_t k = 0; k < L; ++k)
        if (qs[k] != ' ') buf[m++] = qs[k];
    buf[m] = 0;
    size_t pos = 0;
    for (;;) {
        size_t end = pos;
        while (end < m && buf[end] != ',') ++end;
        const char *seg = buf + pos;
        size_t sl = end - pos;
        uint8_t chain = default_chain;
        if (sl > 0 && is_alpha((unsigned char)seg[0])) { chain = (uint8_t)seg[0]; ++seg; --sl; }
        /* split_once(':') */
        size_t colon = 0;
        while (colon < sl && seg[colon] != ':') ++colon;
        uint8_t subs[512];
        int64_t n_subs = -1;
        size_t rl = sl;
        if (colon < sl) {
            rl = colon;
            n_subs = 0;
            for (size_t k = colon + 1; k < sl; ++k) {
                if (!is_alpha((unsigned char)seg[k])) continue; /* is_aa_group_char */
                uint8_t tmp[20];
                int c = one_letter_to_u8_vec(seg[k], tmp);
                if (n_subs + c > (int64_t)sizeof subs) break;
                memcpy(subs + n_subs, tmp, (size_t)c);
                n_subs += c;
            }
        }
        size_t dash = 0;
        while (dash < rl && seg[dash] != '-') ++dash;
        if (dash < rl) {
            uint64_t a, b;
            if (!parse_u64_str(seg, dash, &a) || !parse_u64_str(seg + dash + 1, rl - dash - 1, &b)) {
                free(buf); fdo_query_spec_free(q); return NULL;
            }
            for (uint64_t r = a; r <= b; ++r) spec_push(q, &cap, chain, r, subs, n_subs);
        } else {
            uint64_t a;
            if (!parse_u64_str(seg, rl, &a)) { free(buf); fdo_query_spec_free(q); return NULL; }
            spec_push(q, &cap, chain, a, subs, n_subs);
        }
        if (end >= m) break;
        pos = end + 1;
    }
    free(buf);
    return q;
}

void fdo_query_spec_free(fdo_query_spec *q) {
    if (!q) return;
    for (uint64_t k = 0; k < q->n; ++k) free(q->subs[k]);
    free(q->chain); free(q->serial); free(q->subs); free(q->n_subs);
    free(q);
}

/* ------------------------------------------------------------------------ make_query_map */
typedef struct { fdo_query_map *m; uint64_t cap, aad_cap; } qm_builder;

static int qm_find(const fdo_query_map *m, uint32_t h) {
    for (uint64_t k = 0; k < m->n; ++k)
        if (m->hash[k] == h) return 1;
    return 0;
}
static void insert_hash(qm_builder *b, uint32_t h, uint64_t qi, uint64_t qj, int is_primary, float idf) {
    fdo_query_map *m = b->m;
    if (qm_find(m, h)) return;
    if (m->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        m->hash = (uint32_t *)realloc(m->hash, b->cap * sizeof(uint32_t));
        m->qi = (uint64_t *)realloc(m->qi, b->cap * sizeof(uint64_t));
        m->qj = (uint64_t *)realloc(m->qj, b->cap * sizeof(uint64_t));
        m->is_primary = (uint8_t *)realloc(m->is_primary, b->cap);
        m->idf = (float *)realloc(m->idf, b->cap * sizeof(float));
    }
    m->hash[m->n] = h; m->qi[m->n] = qi; m->qj[m->n] = qj;
    m->is_primary[m->n] = (uint8_t)is_primary; m->idf[m->n] = idf;
    m->n++;
}
/* query.rs:53-84 insert_binned_hash: first insert wins; with multiple_bin every bin pair inserts its own hash (either count 0 ->
 * the encoding's defaults) */
static void insert_binned_hash(qm_builder *b, const float *feature, uint64_t qi, uint64_t qj, uint64_t nbin_dist,
                               uint64_t nbin_angle, int is_primary, float idf) {
    uint64_t mb[16];
    const uint64_t nmb = fdo_multiple_bins(mb);
    if (nmb) {
        for (uint64_t k = 0; k < nmb; ++k) insert_hash(b, fdo_hash_any(feature, mb[2 * k], mb[2 * k + 1]), qi, qj, is_primary, idf);
        return;
    }
    insert_hash(b, fdo_hash_any(feature, nbin_dist, nbin_angle), qi, qj, is_primary, idf);
}

/* query.rs:179-206 */
static void expand_and_insert(qm_builder *b, const int *idxs, int n_idx, const float *thr, uint64_t n_thr,
                              float *near, float *far, uint64_t qi, uint64_t qj, uint64_t nbd, uint64_t nba,
                              float idf) {
    for (uint64_t t = 0; t < n_thr; ++t) {
        float delta = thr[t];
        for (int k = 0; k < n_idx; ++k) {
            int idx = idxs[k];
            near[idx] -= delta;
            far[idx] += delta;
            insert_binned_hash(b, near, qi, qj, nbd, nba, 0, idf);
            insert_binned_hash(b, far, qi, qj, nbd, nba, 0, idf);
            near[idx] += delta;
            far[idx] -= delta;
        }
    }
}

/* query.rs:208-329 */
fdo_query_map *fdo_make_query_map(const fdo_structure *qs, const fdo_query_spec *spec_in, uint64_t nbin_dist,
                                  uint64_t nbin_angle, const float *dist_thr, uint64_t n_dist_thr,
                                  const float *angle_thr, uint64_t n_angle_thr, float dist_cutoff,
                                  int serial_query, const fdo_index *index, float total_structures) {
    fdo_query_map *m = (fdo_query_map *)calloc(1, sizeof *m);
    qm_builder b = {m, 0, 0};
    /* resolve query residues; empty spec => all residues (query.rs:226-234) */
    uint64_t nq = spec_in && spec_in->n ? spec_in->n : (uint64_t)qs->n;
    m->indices = (uint64_t *)malloc((nq ? nq : 1) * sizeof(uint64_t));
    /* substitution_map: index -> subs (later insert overrides) */
    const uint8_t **sub_of = (const uint8_t **)calloc((size_t)(qs->n > 0 ? qs->n : 1), sizeof(uint8_t *));
    uint64_t *nsub_of = (uint64_t *)calloc((size_t)(qs->n > 0 ? qs->n : 1), sizeof(uint64_t));
    uint8_t *has_sub = (uint8_t *)calloc((size_t)(qs->n > 0 ? qs->n : 1), 1);
    for (uint64_t k = 0; k < nq; ++k) {
        int64_t idx;
        const uint8_t *subs = NULL;
        uint64_t ns = 0;
        int hs = 0;
        if (spec_in && spec_in->n) {
            idx = serial_query ? (int64_t)spec_in->serial[k] : fdo_get_index(qs, spec_in->chain[k], spec_in->serial[k]);
            if (spec_in->subs[k]) { subs = spec_in->subs[k]; ns = spec_in->n_subs[k]; hs = 1; }
        } else {
            idx = fdo_get_index(qs, qs->chain[k], qs->serial[k]);
            if (serial_query) idx = (int64_t)qs->serial[k];
        }
        if (idx < 0) continue;
        m->indices[m->n_indices++] = (uint64_t)idx;
        if (hs && idx < qs->n) { sub_of[idx] = subs; nsub_of[idx] = ns; has_sub[idx] = 1; }
    }
    const float RADS_PER_DEG = 3.14159274101257324f / 180.0f; /* f32::to_radians */
    float *athr = (float *)malloc((n_angle_thr ? n_angle_thr : 1) * sizeof(float));
    for (uint64_t t = 0; t < n_angle_thr; ++t) athr[t] = angle_thr[t] * RADS_PER_DEG;

    float feature[9] = {0}, near[9], far[9];
    uint64_t K = m->n_indices;
    for (uint64_t i = 0; i < K; ++i) {
        for (uint64_t j = 0; j < K; ++j) {
            if (i == j) continue;
            uint64_t ri = m->indices[i], rj = m->indices[j];
            if (ri >= (uint64_t)qs->n || rj >= (uint64_t)qs->n) continue; /* reference would panic */
            if (!fdo_pair_feature(qs, (int64_t)ri, (int64_t)rj, dist_cutoff, feature)) continue;
            memcpy(near, feature, sizeof near);
            memcpy(far, feature, sizeof far);
            /* observed (aa pair, CA distance) list: core.rs:462-477 */
            {
                float dx = qs->ca_xyz[3 * ri] - qs->ca_xyz[3 * rj], dy = qs->ca_xyz[3 * ri + 1] - qs->ca_xyz[3 * rj + 1],
                      dz = qs->ca_xyz[3 * ri + 2] - qs->ca_xyz[3 * rj + 2];
                float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (d <= 20.0f) {
                    if (m->n_aad == b.aad_cap) {
                        b.aad_cap = b.aad_cap ? b.aad_cap * 2 : 64;
                        m->aad_aa1 = (uint8_t *)realloc(m->aad_aa1, b.aad_cap);
                        m->aad_aa2 = (uint8_t *)realloc(m->aad_aa2, b.aad_cap);
                        m->aad_dist = (float *)realloc(m->aad_dist, b.aad_cap * sizeof(float));
                        m->aad_qi = (uint64_t *)realloc(m->aad_qi, b.aad_cap * sizeof(uint64_t));
                    }
                    m->aad_aa1[m->n_aad] = qs->aa[ri]; m->aad_aa2[m->n_aad] = qs->aa[rj];
                    m->aad_dist[m->n_aad] = d; m->aad_qi[m->n_aad] = ri;
                    m->n_aad++;
                }
            }
            /* observed hash + idf (query.rs:17-32, 283-288) */
            uint32_t oh = fdo_hash_any(feature, nbin_dist, nbin_angle);
            float idf = 0.0f;
            if (index) {
                uint64_t *ids = NULL;
                uint64_t cnt = fdo_index_get_entries(index, oh, &ids);
                free(ids);
                if (cnt > 0) idf = log2f(total_structures / (float)cnt);
            }
            insert_binned_hash(&b, feature, ri, rj, nbin_dist, nbin_angle, 1, idf);
            /* apply_substitutions (query.rs:86-156), feature = feature_near; only encodings with residue fields
             * (amino_acid_index, controller/feature.rs:260-267: not TertiaryInteraction / Hybrid) */
            const uint32_t ht = fdo_get_hash_type();
            const int has_aa = ht != 5 && ht != 6;
            if (!has_aa) {
            } else if (has_sub[ri]) {
                for (uint64_t a = 0; a < nsub_of[ri]; ++a) {
                    float tmp[9];
                    memcpy(tmp, near, sizeof tmp);
                    tmp[0] = (float)sub_of[ri][a];
                    insert_binned_hash(&b, tmp, ri, rj, nbin_dist, nbin_angle, 0, idf);
                }
                if (has_sub[rj]) {
                    float o0 = near[0], o1 = near[1];
                    for (uint64_t a = 0; a < nsub_of[ri]; ++a)
                        for (uint64_t c = 0; c < nsub_of[rj]; ++c) {
                            near[0] = (float)sub_of[ri][a];
                            near[1] = (float)sub_of[rj][c];
                            insert_binned_hash(&b, near, ri, rj, nbin_dist, nbin_angle, 0, idf);
                            near[0] = o0; near[1] = o1;
                        }
                }
            } else if (has_sub[rj]) {
                for (uint64_t c = 0; c < nsub_of[rj]; ++c) {
                    float tmp[9];
                    memcpy(tmp, near, sizeof tmp);
                    tmp[1] = (float)sub_of[rj][c];
                    insert_binned_hash(&b, tmp, ri, rj, nbin_dist, nbin_angle, 0, idf);
                }
            }
            /* dist_index / angle_index of the encoding (controller/feature.rs:269-291): theta only for the two PDBMotif forms —
             * PDBMotif's degree-valued theta is shifted by the threshold in radians, as in the reference */
            static const int d23[2] = {2, 3}, d2[1] = {2}, d7[1] = {7};
            static const int a456[3] = {4, 5, 6}, a37[5] = {3, 4, 5, 6, 7}, a345[3] = {3, 4, 5}, a06[7] = {0, 1, 2, 3, 4, 5, 6}, a48[5] = {4, 5, 6, 7, 8};
            const int *di = d23, *ai = a456;
            int ndi = 2, nai = 3;
            if (ht <= 1) nai = 1;
            else if (ht == 2) { di = d2; ndi = 1; ai = a37; nai = 5; }
            else if (ht == 4) { di = d2; ndi = 1; ai = a345; nai = 3; }
            else if (ht == 5) { di = d7; ndi = 1; ai = a06; nai = 7; }
            else if (ht == 6) { ai = a48; nai = 5; }
            expand_and_insert(&b, di, ndi, dist_thr, n_dist_thr, near, far, ri, rj, nbin_dist, nbin_angle, idf);
            expand_and_insert(&b, ai, nai, athr, n_angle_thr, near, far, ri, rj, nbin_dist, nbin_angle, idf);
        }
    }
    free(athr); free(sub_of); free(nsub_of); free(has_sub);
    return m;
}

void fdo_query_map_free(fdo_query_map *m) {
    if (!m) return;
    free(m->hash); free(m->qi); free(m->qj); free(m->is_primary); free(m->idf); free(m->indices);
    free(m->aad_aa1); free(m->aad_aa2); free(m->aad_dist); free(m->aad_qi);
    free(m);
}

/* ---------------------------------------------------------------------------- count_query */
typedef struct { uint64_t qi, qj; uint32_t hash; uint64_t ord; } qent;
static int cmp_qent(const void *a, const void *b) {
    const qent *x = (const qent *)a, *y = (const qent *)b;
    if (x->qi != y->qi) return x->qi < y->qi ? -1 : 1;  /* node group */
    if (x->qj != y->qj) return x->qj < y->qj ? -1 : 1;  /* chunk.sort_by_key(edge), stable */
    return x->ord < y->ord ? -1 : x->ord > y->ord;
}

/* count_query.rs:82-220 */
uint64_t fdo_count_query(const fdo_query_map *m, const fdo_index *index, const uint64_t *nres, uint64_t S,
                         float freq_filter, float length_penalty, fdo_count_result **out) {
    *out = NULL;
    float lp = length_penalty; /* caller passes 0.5 for None */
    qent *q = (qent *)malloc((m->n ? m->n : 1) * sizeof *q);
    for (uint64_t k = 0; k < m->n; ++k) { q[k].qi = m->qi[k]; q[k].qj = m->qj[k]; q[k].hash = m->hash[k]; q[k].ord = k; }
    qsort(q, m->n, sizeof *q, cmp_qent);

    /* merged (across node groups) */
    uint8_t *found = (uint8_t *)calloc(S ? S : 1, 1);
    uint32_t *mc = (uint32_t *)calloc(S ? S : 1, 4), *ec = (uint32_t *)calloc(S ? S : 1, 4);
    uint16_t *nc = (uint16_t *)calloc(S ? S : 1, 2);
    float *idf_sum = (float *)calloc(S ? S : 1, 4);
    /* per node group locals */
    uint8_t *l_init = (uint8_t *)malloc(S ? S : 1), *edge_occ = (uint8_t *)malloc(S ? S : 1);
    uint32_t *l_mc = (uint32_t *)malloc((S ? S : 1) * 4), *l_ec = (uint32_t *)malloc((S ? S : 1) * 4);
    float *l_idf = (float *)malloc((S ? S : 1) * 4);

    uint64_t k = 0;
    while (k < m->n) {
        uint64_t node = q[k].qi, g_end = k;
        while (g_end < m->n && q[g_end].qi == node) ++g_end;
        memset(l_init, 0, S); memset(edge_occ, 0, S);
        memset(l_mc, 0, S * 4); memset(l_ec, 0, S * 4); memset(l_idf, 0, S * 4);
        int have_prev = 0;
        uint64_t prev_e = 0;
        for (uint64_t t = k; t < g_end; ++t) {
            if (!have_prev || prev_e != q[t].qj) {
                if (have_prev)
                    for (uint64_t nid = 0; nid < S; ++nid)
                        if (edge_occ[nid]) l_ec[nid] += 1;
                memset(edge_occ, 0, S);
                have_prev = 1;
                prev_e = q[t].qj;
            }
            uint64_t *ids = NULL;
            uint64_t cnt = fdo_index_get_entries(index, q[t].hash, &ids);
            if (freq_filter >= 0.0f && (float)cnt / (float)S > freq_filter) { free(ids); continue; }
            float idf = log2f((float)S / (float)cnt);
            for (uint64_t e = 0; e < cnt; ++e) {
                uint64_t v = ids[e];
                if (v >= S) continue;
                l_init[v] = 1;
                l_mc[v] += 1;
                l_idf[v] += idf;
                edge_occ[v] = 1;
            }
            free(ids);
        }
        for (uint64_t nid = 0; nid < S; ++nid)
            if (edge_occ[nid]) l_ec[nid] += 1;
        /* merge (count_query.rs:172-195), node groups visited in ascending node order */
        for (uint64_t nid = 0; nid < S; ++nid) {
            if (!l_init[nid]) continue;
            if (!found[nid]) { found[nid] = 1; mc[nid] = l_mc[nid]; idf_sum[nid] = l_idf[nid]; nc[nid] = 1; ec[nid] = l_ec[nid]; }
            else { mc[nid] += l_mc[nid]; idf_sum[nid] += l_idf[nid]; nc[nid] = (uint16_t)(nc[nid] + 1); ec[nid] += l_ec[nid]; }
        }
        k = g_end;
    }
    uint64_t n_out = 0;
    for (uint64_t nid = 0; nid < S; ++nid)
        if (found[nid] && mc[nid] > 0) ++n_out;
    fdo_count_result *r = (fdo_count_result *)malloc((n_out ? n_out : 1) * sizeof *r);
    uint64_t w = 0;
    for (uint64_t nid = 0; nid < S; ++nid) {
        if (!(found[nid] && mc[nid] > 0)) continue;
        r[w].nid = nid;
        r[w].total_match_count = mc[nid];
        r[w].node_count = nc[nid];
        r[w].edge_count = ec[nid];
        r[w].idf = idf_sum[nid] * powf((float)nres[nid], -lp);
        ++w;
    }
    free(q); free(found); free(mc); free(ec); free(nc); free(idf_sum);
    free(l_init); free(edge_occ); free(l_mc); free(l_ec); free(l_idf);
    *out = r;
    return n_out;
}
