/* fdo_retrieve.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates src/controller/retrieve.rs:52-156 (retrieve_with_prefilter), :364-552
 * (retrieval_wrapper), :563-602 (prefilter_amino_acid), :604-702
 * (map_query_and_retrieved_residues), :705-719 (calculate_subgraph_idf), :756-834
 * (rmsd_with_calpha_and_rottran, Kabsch branch) and src/controller/graph.rs:16-50.
 * petgraph 0.6.5 (Cargo.lock) tarjan_scc / kosaraju_scc(undirected) are restated as
 * Tarjan SCC + union-find components; only component *membership* is consumed
 * (graph.rs:43-45 sorts and dedups), edges are visited in insertion order. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

#define PREFILTER_AA_SKIPPING_SIZE 200
#define RESIDUE_RESCUE_COUNT_CUTOFF 2

typedef struct { uint64_t *v; uint64_t n, cap; } vec64;
static void v_push(vec64 *a, uint64_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 16; a->v = (uint64_t *)realloc(a->v, a->cap * 8); }
    a->v[a->n++] = x;
}

static int qm_lookup(const fdo_query_map *m, uint32_t h) {
    for (uint64_t k = 0; k < m->n; ++k)
        if (m->hash[k] == h) return (int)k;
    return -1;
}

/* ---- graph ------------------------------------------------------------------------ */
typedef struct {
    uint64_t nn;        /* nodes, weight = residue index, in first-appearance order */
    uint64_t *w;
    uint64_t ne;
    uint64_t *es, *et;  /* node indices */
    uint32_t *eh;
} graph_t;

static uint64_t node_of(graph_t *g, vec64 *w, uint64_t res) {
    for (uint64_t k = 0; k < w->n; ++k)
        if (w->v[k] == res) return k;
    v_push(w, res);
    return w->n - 1;
}

/* iterative Tarjan; comp[v] = scc id */
static void tarjan(const graph_t *g, int64_t *comp, uint64_t *ncomp) {
    uint64_t n = g->nn;
    /* adjacency (CSR) */
    uint64_t *deg = (uint64_t *)calloc(n + 1, 8), *adj = (uint64_t *)malloc((g->ne ? g->ne : 1) * 8);
    for (uint64_t e = 0; e < g->ne; ++e) deg[g->es[e] + 1]++;
    for (uint64_t v = 0; v < n; ++v) deg[v + 1] += deg[v];
    uint64_t *fill = (uint64_t *)malloc((n ? n : 1) * 8);
    memcpy(fill, deg, n * 8);
    for (uint64_t e = 0; e < g->ne; ++e) adj[fill[g->es[e]]++] = g->et[e];
    int64_t *idx = (int64_t *)malloc((n ? n : 1) * 8), *low = (int64_t *)malloc((n ? n : 1) * 8);
    uint8_t *on = (uint8_t *)calloc(n ? n : 1, 1);
    uint64_t *stk = (uint64_t *)malloc((n ? n : 1) * 8), *cs = (uint64_t *)malloc((n ? n : 1) * 8),
             *ci = (uint64_t *)malloc((n ? n : 1) * 8);
    for (uint64_t v = 0; v < n; ++v) { idx[v] = -1; comp[v] = -1; }
    int64_t counter = 0;
    uint64_t sp = 0, nc = 0;
    for (uint64_t root = 0; root < n; ++root) {
        if (idx[root] >= 0) continue;
        uint64_t cp = 0;
        cs[cp] = root; ci[cp] = deg[root]; ++cp;
        idx[root] = low[root] = counter++; stk[sp++] = root; on[root] = 1;
        while (cp) {
            uint64_t v = cs[cp - 1];
            if (ci[cp - 1] < deg[v + 1]) {
                uint64_t w = adj[ci[cp - 1]++];
                if (idx[w] < 0) {
                    idx[w] = low[w] = counter++; stk[sp++] = w; on[w] = 1;
                    cs[cp] = w; ci[cp] = deg[w]; ++cp;
                } else if (on[w] && idx[w] < low[v]) low[v] = idx[w];
            } else {
                if (low[v] == idx[v]) {
                    uint64_t w;
                    do { w = stk[--sp]; on[w] = 0; comp[w] = (int64_t)nc; } while (w != v);
                    ++nc;
                }
                --cp;
                if (cp && low[v] < low[cs[cp - 1]]) low[cs[cp - 1]] = low[v];
            }
        }
    }
    *ncomp = nc;
    free(deg); free(adj); free(fill); free(idx); free(low); free(on); free(stk); free(cs); free(ci);
}

static uint64_t uf_find(uint64_t *p, uint64_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

typedef struct { uint64_t *nodes; uint64_t n; } compo;
static int cmp_compo(const void *a, const void *b) {
    const compo *x = (const compo *)a, *y = (const compo *)b;
    uint64_t n = x->n < y->n ? x->n : y->n;
    for (uint64_t k = 0; k < n; ++k)
        if (x->nodes[k] != y->nodes[k]) return x->nodes[k] < y->nodes[k] ? -1 : 1;
    return x->n < y->n ? -1 : x->n > y->n;
}

/* graph.rs:29-50: SCCs ∪ WCCs with >= node_count nodes, each sorted, list sorted, dedup */
static compo *components(const graph_t *g, uint64_t node_count, uint64_t *n_out) {
    uint64_t n = g->nn;
    int64_t *scc = (int64_t *)malloc((n ? n : 1) * 8);
    uint64_t nscc = 0;
    tarjan(g, scc, &nscc);
    uint64_t *uf = (uint64_t *)malloc((n ? n : 1) * 8);
    for (uint64_t v = 0; v < n; ++v) uf[v] = v;
    for (uint64_t e = 0; e < g->ne; ++e) {
        uint64_t a = uf_find(uf, g->es[e]), b = uf_find(uf, g->et[e]);
        if (a != b) uf[a] = b;
    }
    compo *list = (compo *)calloc(2 * (n ? n : 1), sizeof *list);
    uint64_t nl = 0;
    for (int pass = 0; pass < 2; ++pass) {
        /* bucket nodes by label */
        for (uint64_t lab = 0; lab < n; ++lab) {
            uint64_t cnt = 0;
            for (uint64_t v = 0; v < n; ++v) {
                uint64_t l = pass == 0 ? (uint64_t)scc[v] : uf_find(uf, v);
                if (l == lab) ++cnt;
            }
            if (cnt < node_count || cnt == 0) continue;
            compo c;
            c.n = 0;
            c.nodes = (uint64_t *)malloc(cnt * 8);
            for (uint64_t v = 0; v < n; ++v) {
                uint64_t l = pass == 0 ? (uint64_t)scc[v] : uf_find(uf, v);
                if (l == lab) c.nodes[c.n++] = v;
            }
            list[nl++] = c;
        }
    }
    qsort(list, nl, sizeof *list, cmp_compo);
    uint64_t m = 0;
    for (uint64_t k = 0; k < nl; ++k) {
        if (m > 0 && cmp_compo(&list[k], &list[m - 1]) == 0) { free(list[k].nodes); continue; }
        list[m++] = list[k];
    }
    free(scc); free(uf);
    *n_out = m;
    return list;
}

/* ---- RMSD (retrieve.rs:756-834, Kabsch branch) ---------------------------------------- */
static float rmsd_ca_cb(const fdo_structure *q, const fdo_structure *t, const uint64_t *qi, const uint64_t *ti,
                        uint64_t n, float rot[9], float tran[3]) {
    float *x = (float *)malloc((n ? n : 1) * 6 * sizeof(float)), *y = (float *)malloc((n ? n : 1) * 6 * sizeof(float));
    for (uint64_t k = 0; k < n; ++k) {
        memcpy(y + 6 * k, q->ca_xyz + 3 * qi[k], 12);     /* reference (fixed) = query */
        memcpy(y + 6 * k + 3, q->cb_xyz + 3 * qi[k], 12);
        memcpy(x + 6 * k, t->ca_xyz + 3 * ti[k], 12);     /* coords (moving) = target */
        memcpy(x + 6 * k + 3, t->cb_xyz + 3 * ti[k], 12);
    }
    float r = fdo_kabsch(x, y, 2 * n, 2, rot, tran);
    free(x); free(y);
    return r;
}

static void match_init(fdo_match *mt, uint64_t n) {
    mt->n = n;
    mt->has = (uint8_t *)calloc(n ? n : 1, 1);
    mt->chain = (uint8_t *)calloc(n ? n : 1, 1);
    mt->serial = (uint64_t *)calloc(n ? n : 1, 8);
    mt->tindex = (int64_t *)malloc((n ? n : 1) * 8);
    for (uint64_t k = 0; k < n; ++k) mt->tindex[k] = -1;
}
static void match_set(fdo_match *mt, uint64_t pos, const fdo_structure *t, int64_t r) {
    if (r < 0) { mt->has[pos] = 0; mt->tindex[pos] = -1; return; }
    mt->has[pos] = 1; mt->chain[pos] = t->chain[r]; mt->serial[pos] = t->serial[r]; mt->tindex[pos] = r;
}
static int match_equal(const fdo_match *a, const fdo_match *b) {
    for (uint64_t k = 0; k < a->n; ++k) {
        if (a->has[k] != b->has[k]) return 0;
        if (a->has[k] && (a->chain[k] != b->chain[k] || a->serial[k] != b->serial[k])) return 0;
    }
    return 1;
}

fdo_retrieval *fdo_retrieve(const fdo_structure *t, const fdo_structure *qs, const fdo_query_map *m,
                            uint64_t node_count, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff,
                            float ca_distance_cutoff) {
    fdo_retrieval *R = (fdo_retrieval *)calloc(1, sizeof *R);
    vec64 fi = {0}, fj = {0}, fh = {0}, cq = {0}, cI = {0}, cJ = {0};

    /* prefilter_amino_acid (retrieve.rs:563-602): exact 3-letter name match */
    vec64 set1 = {0}, set2 = {0};
    /* TertiaryInteraction / Hybrid hashes carry no residue types: the reference's prefilter unwraps a None there (retrieve.rs:576)
     * and panics for queries of <= 200 hashes; the restatement scans every pair instead */
    if (m->n <= PREFILTER_AA_SKIPPING_SIZE && m->n > 0 && fdo_get_hash_type() != 5 && fdo_get_hash_type() != 6) {
        uint8_t seen1[256] = {0}, seen2[256] = {0};
        uint8_t *in1 = (uint8_t *)calloc((size_t)(t->n > 0 ? t->n : 1), 1), *in2 = (uint8_t *)calloc((size_t)(t->n > 0 ? t->n : 1), 1);
        for (uint64_t k = 0; k < m->n; ++k) {
            /* residue types from the encoding's own reverse_hash (retrieve.rs:574-577) */
            uint32_t hk = m->hash[k], ht = fdo_get_hash_type();
            uint8_t aa1, aa2;
            if (ht == 0) { aa1 = (uint8_t)((hk >> 20) & 0x1f); aa2 = (uint8_t)((hk >> 15) & 0x1f); }
            else if (ht == 2) { uint32_t pr = (hk >> 23) & 0x1ff; aa1 = (uint8_t)(pr / 20); aa2 = (uint8_t)(pr % 20); }
            else if (ht == 4) { aa1 = (uint8_t)((hk >> 27) & 0x1f); aa2 = (uint8_t)((hk >> 22) & 0x1f); }
            else if (ht == 1) { aa1 = (uint8_t)((hk >> 21) & 0x1f); aa2 = (uint8_t)((hk >> 16) & 0x1f); }
            else if (ht == 7 || ht == 8) { uint32_t pr = (hk >> 21) & 0x1ff; aa1 = (uint8_t)(pr / 20); aa2 = (uint8_t)(pr % 20); }
            else { aa1 = (uint8_t)((hk >> 25) & 0x1f); aa2 = (uint8_t)((hk >> 20) & 0x1f); }
            if (!seen1[aa1]) {
                seen1[aa1] = 1;
                const char *nm = fdo_map_u8_to_aa(aa1);
                for (int32_t i = 0; i < t->n; ++i)
                    if (!memcmp(t->resname + 3 * i, nm, 3)) in1[i] = 1;
            }
            if (!seen2[aa2]) {
                seen2[aa2] = 1;
                const char *nm = fdo_map_u8_to_aa(aa2);
                for (int32_t i = 0; i < t->n; ++i)
                    if (!memcmp(t->resname + 3 * i, nm, 3)) in2[i] = 1;
            }
        }
        for (int32_t i = 0; i < t->n; ++i) { if (in1[i]) v_push(&set1, (uint64_t)i); if (in2[i]) v_push(&set2, (uint64_t)i); }
        free(in1); free(in2);
    }
    int use_pref = set1.n > 0 && set2.n > 0;
    uint64_t n1 = use_pref ? set1.n : (uint64_t)t->n, n2 = use_pref ? set2.n : (uint64_t)t->n;

    /* retrieve_with_prefilter (retrieve.rs:52-156) */
    if (m->n > 0) {
        float feature[9] = {0};
        vec64 tmp_q = {0};
        for (uint64_t a = 0; a < n1; ++a) {
            uint64_t i = use_pref ? set1.v[a] : a;
            for (uint64_t b = 0; b < n2; ++b) {
                uint64_t j = use_pref ? set2.v[b] : b;
                float dx = t->ca_xyz[3 * i] - t->ca_xyz[3 * j], dy = t->ca_xyz[3 * i + 1] - t->ca_xyz[3 * j + 1],
                      dz = t->ca_xyz[3 * i + 2] - t->ca_xyz[3 * j + 2];
                float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (!(d <= dist_cutoff)) continue;
                uint8_t aa1 = t->aa[i], aa2 = t->aa[j];
                tmp_q.n = 0;
                for (uint64_t e = 0; e < m->n_aad; ++e)
                    if (m->aad_aa1[e] == aa1 && m->aad_aa2[e] == aa2 && fabsf(d - m->aad_dist[e]) < ca_distance_cutoff)
                        v_push(&tmp_q, m->aad_qi[e]);
                if (tmp_q.n == 0) continue;
                if (!fdo_pair_feature(t, (int64_t)i, (int64_t)j, dist_cutoff, feature)) continue;
                for (uint64_t e = 0; e < tmp_q.n; ++e) { v_push(&cq, tmp_q.v[e]); v_push(&cI, i); v_push(&cJ, j); }
                /* retrieve.rs:124-139: with multiple_bin one (i, j, hash) per bin pair whose hash the query holds */
                uint64_t mb[16];
                uint64_t nmb = fdo_multiple_bins(mb);
                for (uint64_t k = 0; k < (nmb ? nmb : 1); ++k) {
                    uint32_t h = nmb ? fdo_hash_cfg(feature, mb[2 * k], mb[2 * k + 1]) : fdo_hash_any(feature, nbin_dist, nbin_angle);
                    if (qm_lookup(m, h) >= 0) { v_push(&fi, i); v_push(&fj, j); v_push(&fh, h); }
                }
            }
        }
        free(tmp_q.v);
    }
    R->n_found = fi.n; R->found_i = fi.v; R->found_j = fj.v;
    R->found_hash = (uint32_t *)malloc((fh.n ? fh.n : 1) * 4);
    for (uint64_t k = 0; k < fh.n; ++k) R->found_hash[k] = (uint32_t)fh.v[k];
    free(fh.v);
    R->n_cand = cq.n; R->cand_qi = cq.v; R->cand_i = cI.v; R->cand_j = cJ.v;
    free(set1.v); free(set2.v);

    /* create_index_graph (graph.rs:16-26) */
    graph_t g = {0};
    vec64 w = {0};
    g.ne = R->n_found;
    g.es = (uint64_t *)malloc((g.ne ? g.ne : 1) * 8); g.et = (uint64_t *)malloc((g.ne ? g.ne : 1) * 8);
    g.eh = R->found_hash;
    for (uint64_t e = 0; e < g.ne; ++e) {
        g.es[e] = node_of(&g, &w, R->found_i[e]);
        g.et[e] = node_of(&g, &w, R->found_j[e]);
    }
    g.nn = w.n; g.w = w.v;
    uint64_t ncomp = 0;
    compo *comps = components(&g, node_count, &ncomp);

    R->n_matches = ncomp;
    R->from_hash = (fdo_match *)calloc(ncomp ? ncomp : 1, sizeof(fdo_match));
    R->processed = (fdo_match *)calloc(ncomp ? ncomp : 1, sizeof(fdo_match));

    uint64_t q_size = 0;
    for (uint64_t k = 0; k < m->n; ++k) { uint64_t mx = m->qi[k] > m->qj[k] ? m->qi[k] : m->qj[k]; if (mx + 1 > q_size) q_size = mx + 1; }
    if (q_size == 0) q_size = 1;
    uint8_t *sym = (uint8_t *)malloc(m->n ? m->n : 1);
    for (uint64_t k = 0; k < m->n; ++k) sym[k] = (uint8_t)fdo_hash_is_symmetric(m->hash[k]);

    for (uint64_t c = 0; c < ncomp; ++c) {
        /* subgraph (filter_map keeps insertion order of nodes/edges) */
        uint8_t *inc = (uint8_t *)calloc(g.nn ? g.nn : 1, 1);
        for (uint64_t k = 0; k < comps[c].n; ++k) inc[comps[c].nodes[k]] = 1;
        uint64_t sub_nodes = comps[c].n, r_size = 0;
        for (uint64_t k = 0; k < comps[c].n; ++k) if (g.w[comps[c].nodes[k]] + 1 > r_size) r_size = g.w[comps[c].nodes[k]] + 1;
        /* calculate_subgraph_idf (retrieve.rs:705-719) */
        float sub_idf = 0.0f;
        /* map_query_and_retrieved_residues (retrieve.rs:604-702) */
        uint8_t *counts = (uint8_t *)calloc(q_size * r_size, 1);
        uint8_t *best_c = (uint8_t *)calloc(q_size, 1);
        uint64_t *best_r = (uint64_t *)calloc(q_size, 8);
        for (uint64_t e = 0; e < g.ne; ++e) {
            if (!inc[g.es[e]] || !inc[g.et[e]]) continue;
            int qk = qm_lookup(m, g.eh[e]);
            if (qk < 0) continue;
            sub_idf += m->idf[qk];
            uint64_t qi = m->qi[qk], qj = m->qj[qk], ri = g.w[g.es[e]], rj = g.w[g.et[e]];
            uint64_t pq[2], pr[2];
            if (sym[qk]) {
                uint64_t q1 = qi < qj ? qi : qj, q2 = qi < qj ? qj : qi;
                uint64_t r1 = ri < rj ? ri : rj, r2 = ri < rj ? rj : ri;
                pq[0] = q1; pr[0] = r1; pq[1] = q2; pr[1] = r2;
            } else { pq[0] = qi; pr[0] = ri; pq[1] = qj; pr[1] = rj; }
            for (int z = 0; z < 2; ++z) {
                uint8_t *cc = &counts[pq[z] * r_size + pr[z]];
                if (*cc < 255) ++*cc;
                if (*cc > best_c[pq[z]] || (*cc == best_c[pq[z]] && pr[z] < best_r[pq[z]])) { best_c[pq[z]] = *cc; best_r[pq[z]] = pr[z]; }
            }
        }
        vec64 q_idx = {0}, r_idx = {0};
        uint8_t *q_used = (uint8_t *)calloc(q_size, 1), *r_used = (uint8_t *)calloc(r_size ? r_size : 1, 1);
        int done = 0;
        for (int bucket = 255; bucket >= 1 && !done; --bucket)
            for (uint64_t q = 0; q < q_size && !done; ++q) {
                if (best_c[q] != bucket) continue;
                uint64_t r = best_r[q];
                if (!q_used[q] && !r_used[r]) {
                    v_push(&q_idx, q); v_push(&r_idx, r); q_used[q] = 1; r_used[r] = 1;
                    if (q_idx.n == sub_nodes) done = 1;
                }
            }
        free(counts); free(best_c); free(best_r); free(q_used); free(r_used);

        /* residue assignment + rescue (retrieve.rs:430-516) */
        uint64_t NQ = m->n_indices;
        fdo_match *fh_m = &R->from_hash[c], *pr_m = &R->processed[c];
        match_init(fh_m, NQ);
        match_init(pr_m, NQ);
        vec64 qs_sc = {0}, rs_sc = {0};
        for (uint64_t pos = 0; pos < NQ; ++pos) {
            uint64_t qi = m->indices[pos];
            int64_t mapped = -1;
            for (uint64_t k = 0; k < q_idx.n; ++k) if (q_idx.v[k] == qi) { mapped = (int64_t)r_idx.v[k]; break; }
            if (mapped >= 0) {
                match_set(fh_m, pos, t, mapped);
                int64_t prev_pos = -1;
                for (uint64_t k = 0; k < rs_sc.n; ++k) if (rs_sc.v[k] == (uint64_t)mapped) { prev_pos = (int64_t)k; break; }
                if (prev_pos < 0) {
                    match_set(pr_m, pos, t, mapped);
                    v_push(&qs_sc, qi); v_push(&rs_sc, (uint64_t)mapped);
                } else {
                    match_set(pr_m, (uint64_t)prev_pos, t, -1); /* quirk: index into res_vec by scanned position */
                    match_set(pr_m, pos, t, mapped);
                    memmove(qs_sc.v + prev_pos, qs_sc.v + prev_pos + 1, (qs_sc.n - (uint64_t)prev_pos - 1) * 8); qs_sc.n--;
                    memmove(rs_sc.v + prev_pos, rs_sc.v + prev_pos + 1, (rs_sc.n - (uint64_t)prev_pos - 1) * 8); rs_sc.n--;
                    v_push(&qs_sc, qi); v_push(&rs_sc, (uint64_t)mapped);
                }
            } else {
                match_set(fh_m, pos, t, -1);
                /* count target residues j that pair (AA + CA window) with an already retrieved residue k */
                vec64 keys = {0}, vals = {0};
                for (uint64_t e = 0; e < R->n_cand; ++e) {
                    if (R->cand_qi[e] != qi) continue;
                    uint64_t j = R->cand_i[e], k2 = R->cand_j[e];
                    int in_ret = 0;
                    for (uint64_t z = 0; z < r_idx.n; ++z) if (r_idx.v[z] == k2) { in_ret = 1; break; }
                    if (!in_ret) continue;
                    uint64_t z;
                    for (z = 0; z < keys.n; ++z) if (keys.v[z] == j) { vals.v[z]++; break; }
                    if (z == keys.n) { v_push(&keys, j); v_push(&vals, 1); }
                }
                int rescued = 0;
                if (keys.n) {
                    uint64_t mx = 0, nmx = 0, arg = 0;
                    for (uint64_t z = 0; z < keys.n; ++z) if (vals.v[z] > mx) mx = vals.v[z];
                    for (uint64_t z = 0; z < keys.n; ++z) if (vals.v[z] == mx) { ++nmx; arg = keys.v[z]; }
                    int already = 0;
                    for (uint64_t z = 0; z < rs_sc.n; ++z) if (rs_sc.v[z] == arg) already = 1;
                    if (nmx == 1 && mx >= RESIDUE_RESCUE_COUNT_CUTOFF && !already) {
                        match_set(pr_m, pos, t, (int64_t)arg);
                        v_push(&qs_sc, qi); v_push(&rs_sc, arg);
                        rescued = 1;
                    }
                }
                if (!rescued) match_set(pr_m, pos, t, -1);
                free(keys.v); free(vals.v);
            }
        }
        fh_m->idf = sub_idf; pr_m->idf = sub_idf;
        fh_m->rmsd = rmsd_ca_cb(qs, t, q_idx.v, r_idx.v, q_idx.n, fh_m->rot, fh_m->tran);
        if (match_equal(pr_m, fh_m)) {
            pr_m->rmsd = fh_m->rmsd; memcpy(pr_m->rot, fh_m->rot, sizeof fh_m->rot); memcpy(pr_m->tran, fh_m->tran, sizeof fh_m->tran);
        } else {
            pr_m->rmsd = rmsd_ca_cb(qs, t, qs_sc.v, rs_sc.v, qs_sc.n, pr_m->rot, pr_m->tran);
        }
        free(inc); free(q_idx.v); free(r_idx.v); free(qs_sc.v); free(rs_sc.v);
    }
    /* retrieve.rs:537-550 */
    R->max_matching_node_count = 0;
    R->min_rmsd_with_max_match = 0.0f;
    for (uint64_t c = 0; c < ncomp; ++c) {
        uint64_t cnt = 0;
        for (uint64_t k = 0; k < R->processed[c].n; ++k) cnt += R->processed[c].has[k];
        if (cnt > R->max_matching_node_count) { R->max_matching_node_count = cnt; R->min_rmsd_with_max_match = R->processed[c].rmsd; }
        else if (cnt == R->max_matching_node_count && R->processed[c].rmsd < R->min_rmsd_with_max_match) R->min_rmsd_with_max_match = R->processed[c].rmsd;
    }
    for (uint64_t c = 0; c < ncomp; ++c) free(comps[c].nodes);
    free(comps); free(sym); free(g.es); free(g.et); free(g.w);
    return R;
}

static void match_free(fdo_match *mt) { free(mt->has); free(mt->chain); free(mt->serial); free(mt->tindex); }
void fdo_retrieval_free(fdo_retrieval *r) {
    if (!r) return;
    for (uint64_t c = 0; c < r->n_matches; ++c) { match_free(&r->from_hash[c]); match_free(&r->processed[c]); }
    free(r->from_hash); free(r->processed);
    free(r->found_i); free(r->found_j); free(r->found_hash); free(r->cand_qi); free(r->cand_i); free(r->cand_j);
    free(r);
}
