// fd_host_query.hip — host-side (C++) half of the retrieval around the GPU kernels, and the fused query call:
//   retrieval             (src/controller/retrieve.rs:364-552)                pair scan + Kabsch on the GPU,
//                          graph components (src/controller/graph.rs:16-50), residue voting
//                          (retrieve.rs:604-702) and rescue (:479-515) on the device for motif queries, on host threads for > 64-node queries
//   fdgpu_query_batch      query maps (fd_query_map.hip), ranked scoring and retrieval of the top candidates in one call
//   fdgpu_merge_subindices, fdgpu_hypergeom_enrichment
// The reference is compiled code, so this glue is C++ behind the same C ABI; it contains no f32 arithmetic that decides a hash bit.
#include <array>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <atomic>
#include <thread>
#include <vector>
#include "fdgpu_internal.h"
#include <chrono>

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            char _b[512];                                                                                   \
            snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            (ctx)->err = _b;                                                                                \
            return FDGPU_EHIP;                                                                              \
        }                                                                                                   \
    } while (0)

fd_hash_consts fd_make_consts(const fd_hash_params *p);  // fdgpu_api.hip
fd_hash_consts fd_make_consts_cfg(const fd_hash_params *p, uint32_t k);
uint32_t fd_num_bin_configs(const fd_hash_params *p);
bool fd_multiple_bins_valid(const fd_hash_params *p);
bool fd_hash_type_supported(uint32_t t);
#define CHECK_TYPE(ctx, p)                                                                                                       \
    do {                                                                                                                         \
        if (!fd_hash_type_supported((p)->hash_type)) {                                                                           \
            (ctx)->err = "hash_type: only the encodings over the (d_CA, d_CB, theta, tau1, tau2) descriptor are built (0, 1, 3, 7, 8)"; \
            return FDGPU_EINVAL;                                                                                                 \
        }                                                                                                                        \
    } while (0)

// ------------------------------------------------------------------------------------------ retrieval glue
namespace {
struct Graph {
    std::vector<uint32_t> w;           // node -> residue index (first-appearance order, graph.rs:16-26)
    std::vector<uint32_t> es, et, eh;  // edges in insertion order
    std::unordered_map<uint32_t, uint32_t> at;
    std::vector<int32_t> dense;        // residue -> node for residues below dense.size() (a candidate's residue count is known: no hashing
                                       // for the 6 x 10^4 lookups of a whole-structure query's heaviest candidate)
    uint32_t node_of(uint32_t res) {
        if (res < dense.size()) {
            int32_t &n = dense[res];
            if (n < 0) { n = (int32_t)w.size(); w.push_back(res); }
            return (uint32_t)n;
        }
        auto ins = at.emplace(res, (uint32_t)w.size());
        if (ins.second) w.push_back(res);
        return ins.first->second;
    }
};

// SCCs (Tarjan) and weakly connected components of at least min_nodes nodes, as sorted node lists, sorted and without duplicates (graph.rs:29-50).
// Flat arrays throughout: a 2,500-residue candidate of a whole-structure query has ~2,500 nodes, nearly all of them components of one node — a vector
// per node's neighbours and a map entry + a vector per component were 0.5 ms of such a slot
static std::vector<std::vector<uint32_t>> components(const Graph &g, size_t min_nodes) {
    const uint32_t n = (uint32_t)g.w.size();
    const size_t E = g.es.size();
    std::vector<uint32_t> a_off(n + 1, 0), a_to(E);
    for (size_t e = 0; e < E; ++e) ++a_off[g.es[e] + 1];
    for (uint32_t v = 0; v < n; ++v) a_off[v + 1] += a_off[v];
    {
        std::vector<uint32_t> cur(a_off.begin(), a_off.end() - 1);
        for (size_t e = 0; e < E; ++e) a_to[cur[g.es[e]]++] = g.et[e];       // a node's neighbours in edge order
    }
    std::vector<int> idx(n, -1), low(n, 0), comp(n, -1);
    std::vector<char> on(n, 0);
    std::vector<uint32_t> stk;
    int counter = 0, ncomp = 0;
    struct Fr { uint32_t v; uint32_t it; };
    std::vector<Fr> cs;
    for (uint32_t root = 0; root < n; ++root) {
        if (idx[root] >= 0) continue;
        cs.clear();
        cs.push_back({root, a_off[root]});
        idx[root] = low[root] = counter++; stk.push_back(root); on[root] = 1;
        while (!cs.empty()) {
            Fr &f = cs.back();
            if (f.it < a_off[f.v + 1]) {
                uint32_t x = a_to[f.it++];
                if (idx[x] < 0) { idx[x] = low[x] = counter++; stk.push_back(x); on[x] = 1; cs.push_back({x, a_off[x]}); }
                else if (on[x]) low[f.v] = std::min(low[f.v], idx[x]);
            } else {
                uint32_t v = f.v;
                if (low[v] == idx[v]) {
                    uint32_t x;
                    do { x = stk.back(); stk.pop_back(); on[x] = 0; comp[x] = ncomp; } while (x != v);
                    ++ncomp;
                }
                cs.pop_back();
                if (!cs.empty()) low[cs.back().v] = std::min(low[cs.back().v], low[v]);
            }
        }
    }
    std::vector<uint32_t> uf(n);
    for (uint32_t v = 0; v < n; ++v) uf[v] = v;
    auto find = [&](uint32_t x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    for (size_t e = 0; e < E; ++e) { uint32_t a = find(g.es[e]), b = find(g.et[e]); if (a != b) uf[a] = b; }
    // the sets of at least min_nodes nodes, their nodes ascending (a walk over the nodes in order); the result is the SORTED list of sets without
    // duplicates, so the order they are gathered in does not matter
    std::vector<std::vector<uint32_t>> all;
    auto gather = [&](const std::vector<uint32_t> &label, uint32_t n_labels) {
        std::vector<uint32_t> size(n_labels, 0), at(n_labels, 0xffffffffu);
        for (uint32_t v = 0; v < n; ++v) ++size[label[v]];
        for (uint32_t v = 0; v < n; ++v) {
            const uint32_t l = label[v];
            if (size[l] < min_nodes || size[l] == 0) continue;
            if (at[l] == 0xffffffffu) { at[l] = (uint32_t)all.size(); all.emplace_back(); all.back().reserve(size[l]); }
            all[at[l]].push_back(v);
        }
    };
    std::vector<uint32_t> label(n);
    for (uint32_t v = 0; v < n; ++v) label[v] = (uint32_t)comp[v];
    gather(label, (uint32_t)ncomp);
    for (uint32_t v = 0; v < n; ++v) label[v] = find(v);
    gather(label, n);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    return all;
}
}  // namespace

extern "C" void fdgpu_matches_free(fd_match_rec *m, int32_t *residues) { fdgpu_free(m); fdgpu_free(residues); }

static bool fd_hash_is_sym(uint32_t htype, uint32_t h);
// test-only (include/fdgpu_debug.h): the host glue's graph of found (i, j) residue pairs — nodes in first-appearance order — and its components of at least
// node_count nodes, as the retrieval of > 64-node graphs computes them; and the symmetry flag of hashes.  Pure host code: CPU tests call them without a GPU.
extern "C" int fdgpu_debug_host_components(const uint32_t *edge_i, const uint32_t *edge_j, uint64_t n_edges, uint32_t node_count, uint32_t **residues,
                                           uint64_t **comp_off, uint64_t *n_comps) {
    if ((n_edges && (!edge_i || !edge_j)) || !residues || !comp_off || !n_comps) return FDGPU_EINVAL;
    Graph g;
    for (uint64_t e = 0; e < n_edges; ++e) {
        const uint32_t a = g.node_of(edge_i[e]), b = g.node_of(edge_j[e]);
        g.es.push_back(a); g.et.push_back(b); g.eh.push_back(0u);
    }
    const std::vector<std::vector<uint32_t>> comps = components(g, node_count);
    uint64_t tot = 0;
    for (const auto &cc : comps) tot += cc.size();
    uint32_t *r = (uint32_t *)malloc(std::max<uint64_t>(tot, 1) * 4);
    uint64_t *o = (uint64_t *)malloc((comps.size() + 1) * 8);
    if (!r || !o) { free(r); free(o); return FDGPU_ENOMEM; }
    uint64_t w = 0;
    o[0] = 0;
    for (size_t k = 0; k < comps.size(); ++k) { for (uint32_t v : comps[k]) r[w++] = g.w[v]; o[k + 1] = w; }      // a component's nodes ascending, as residues
    *residues = r; *comp_off = o; *n_comps = comps.size();
    return FDGPU_OK;
}
extern "C" int fdgpu_debug_hash_is_symmetric(uint32_t hash_type, const uint32_t *hashes, uint64_t n, uint8_t *out) {
    if (n && (!hashes || !out)) return FDGPU_EINVAL;
    for (uint64_t k = 0; k < n; ++k) out[k] = fd_hash_is_sym(hash_type, hashes[k]) ? 1 : 0;
    return FDGPU_OK;
}


// coordinates of all candidates in one launch + one copy (a hipMemcpy per candidate costs more than the pair scan)
__global__ __launch_bounds__(256) void k_gather_xyz(const float *__restrict__ ca, const float *__restrict__ cb, const uint64_t *__restrict__ src,
                                                    const uint64_t *__restrict__ dst, uint64_t total, float *__restrict__ out) {
    const uint64_t k = blockIdx.x;
    const uint64_t s3 = 3 * src[k], d3 = 3 * dst[k], n3 = 3 * (dst[k + 1] - dst[k]);
    for (uint64_t t = threadIdx.x; t < n3; t += blockDim.x) {
        out[d3 + t] = ca[s3 + t];
        out[3 * total + d3 + t] = cb[s3 + t];
    }
}

// What a retrieval needs from the query maps alone — no candidates, no scan output: per query the sorted unique hashes + the map entry each one
// resolves to (the FIRST entry holding it, like the reference's hash map), the pair scan's query descriptors, sizes.  A caller that knows the
// maps before it knows the candidates (fdgpu_query_batch: the scoring kernels are still running) builds it ahead of time.
struct fd_rb_prep {
    std::vector<std::vector<uint32_t>> qhs, qkf;
    std::vector<fd_match_query> mqs;
    std::vector<uint32_t> q_sizes;
    uint64_t max_aad = 0, max_nq = 0;
};
static void fd_rb_prepare(uint64_t n_queries, const fd_query_map *const *qms, float ca_distance_cutoff, fd_rb_prep &P) {
    std::vector<std::vector<uint32_t>> &qhs = P.qhs, &qkf = P.qkf;
    std::vector<fd_match_query> &mqs = P.mqs;
    std::vector<uint32_t> &q_sizes = P.q_sizes;
    qhs.assign(n_queries, {}); qkf.assign(n_queries, {});
    mqs.assign(std::max<uint64_t>(n_queries, 1), fd_match_query{});
    q_sizes.assign(std::max<uint64_t>(n_queries, 1), 1);
    // per query: sorted unique hashes + the map entry each one resolves to (the FIRST entry holding it, like the reference's hash map):
    // (hash << 32 | entry) keys through an LSD radix sort — a whole-structure query has 10^5 entries, a hash map cost 8 ms here
    for (uint64_t t = 0; t < n_queries; ++t) {
        const fd_query_map *m = qms[t];
        uint64_t key_small[256];          // a motif query's few dozen entries stay off the heap (two allocations per query, 128 queries per batch)
        std::vector<uint64_t> key_v, tmp_v;
        if (m->n > 256) { key_v.resize(m->n); tmp_v.resize(m->n); }
        uint64_t *key = m->n > 256 ? key_v.data() : key_small, *tmp = tmp_v.data();
        for (uint64_t k = 0; k < m->n; ++k) {
            key[k] = ((uint64_t)m->hash[k] << 32) | (uint32_t)k;
            q_sizes[t] = std::max(q_sizes[t], std::max(m->qi[k], m->qj[k]) + 1);
        }
        if (m->n > 256) {
            for (int pass = 0; pass < 4; ++pass) {           // entries arrive in ascending k: a stable sort by hash keeps the first entry first
                const int sh = 32 + 8 * pass;
                size_t cnt[257] = {0};
                for (uint64_t k = 0; k < m->n; ++k) ++cnt[((key[k] >> sh) & 255u) + 1];
                for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
                for (uint64_t k = 0; k < m->n; ++k) tmp[cnt[(key[k] >> sh) & 255u]++] = key[k];
                std::swap(key, tmp);
            }
        } else std::sort(key, key + m->n);
        qhs[t].reserve(m->n); qkf[t].reserve(m->n);
        for (uint64_t k = 0; k < m->n; ++k)
            if (k == 0 || (key[k] >> 32) != (key[k - 1] >> 32)) { qhs[t].push_back((uint32_t)(key[k] >> 32)); qkf[t].push_back((uint32_t)key[k]); }
        fd_match_query &q = mqs[t];
        q.hashes = qhs[t].data(); q.n_hashes = qhs[t].size();
        q.aad_aa1 = m->aad_aa1; q.aad_aa2 = m->aad_aa2; q.aad_dist = m->aad_dist; q.aad_qi = m->aad_qi; q.n_aad = m->n_aad;
        q.ca_distance_cutoff = ca_distance_cutoff;
        q.use_aa_prefilter = qhs[t].size() <= 200 ? 1 : 0;  // PREFILTER_AA_SKIPPING_SIZE (retrieve.rs:24, 569)
    }
    P.max_aad = 0; P.max_nq = 0;
    for (uint64_t t = 0; t < n_queries; ++t) { P.max_aad = std::max<uint64_t>(P.max_aad, qms[t]->n_aad); P.max_nq = std::max<uint64_t>(P.max_nq, qms[t]->n_indices); }
}

// symmetry flags (geometry/pdb_tr.rs:158-162): aa equal and atan2(sin, cos) of the two torsion fields equal; the other encodings
// compare their residue fields and (Folddisco*) torsion fields (pdb_motif.rs:98, pdb_motif_sincos.rs:105, folddisco_angle.rs:133,
// folddisco_dist.rs:126)
static bool fd_hash_is_sym(uint32_t htype, uint32_t h) {
        const float D2 = 57.2957795130823208767981548141051703f;
        auto c3 = [](uint32_t v, float nb) { float cf = (1.0f - (-1.0f)) / (nb - 1.0f); return (float)v * cf + (-1.0f); };
        if (htype == FD_HASH_TERTIARY) return false;                                   // tertiary_interaction.rs:145-150
        if (htype == FD_HASH_TRROSETTA) {                                              // trrosetta.rs:158-162 (default 3 angle bins)
            const uint32_t pr = (h >> 23) & 0x1ffu;
            float a[5];
            for (int k = 0; k < 5; ++k) a[k] = atan2f(c3((h >> (18 - 4 * k)) & 3u, 3.0f), c3((h >> (16 - 4 * k)) & 3u, 3.0f)) * D2;
            return pr / 20u == pr % 20u && a[1] == a[2] && a[3] == a[4];
        }
        if (htype == FD_HASH_PPF) {                                                    // ppf.rs:124-127
            float a[2];
            for (int k = 0; k < 2; ++k) a[k] = atan2f(c3((h >> (15 - 6 * k)) & 7u, 3.0f), c3((h >> (12 - 6 * k)) & 7u, 3.0f)) * D2;
            return ((h >> 27) & 31u) == ((h >> 22) & 31u) && a[0] == a[1];
        }
        if (htype == FD_HASH_HYBRID) {                                                 // hybrid.rs:185-189 (4 angle bins)
            float a[2];
            for (int k = 0; k < 2; ++k) a[k] = atan2f(c3((h >> (14 - 4 * k)) & 3u, 4.0f), c3((h >> (12 - 4 * k)) & 3u, 4.0f)) * D2;
            return ((h >> 30) & 3u) == ((h >> 28) & 3u) && a[0] == a[1];
        }
        if (htype == FD_HASH_PDBMOTIF) return ((h >> 20) & 31u) == ((h >> 15) & 31u);
        if (htype == FD_HASH_PDBMOTIF_SINCOS) return ((h >> 21) & 31u) == ((h >> 16) & 31u);
        if (htype == FD_HASH_FD_ANGLE || htype == FD_HASH_FD_DIST) {
            const uint32_t pair = (h >> 21) & 0x1ffu;
            const uint32_t p1 = htype == FD_HASH_FD_ANGLE ? (h >> 5) & 31u : (h >> 4) & 15u, p2 = htype == FD_HASH_FD_ANGLE ? h & 31u : h & 15u;
            return pair / 20u == pair % 20u && p1 == p2;
        }
        // the default encoding: the two torsion angles from their (sin, cos) bin pairs — sixteen possible pairs, atan2f of each evaluated once (the same
        // call on the same arguments: same floats); two atan2f per found edge were 0.8 ms of a whole-structure query's plan pass (25,162 edges of the self hit)
        static const std::array<float, 16> ang = [] {
            std::array<float, 16> t{};
            auto cont = [](uint32_t v) { float cf = (1.0f - (-1.0f)) / (4.0f - 1.0f); return (float)v * cf + (-1.0f); };
            const float D = 57.2957795130823208767981548141051703f;
            for (uint32_t v = 0; v < 16; ++v) t[v] = atan2f(cont(v >> 2), cont(v & 3u)) * D;
            return t;
        }();
        return ((h >> 25) & 31u) == ((h >> 20) & 31u) && ang[(h >> 4) & 15u] == ang[h & 15u];
}

// ---- device glue (k_retrieve.hip): the scan output never leaves the GPU — per-slot grouping, graph / components / votes /
// assignment / rescue one wavefront per candidate, superposition and metrics on the problems it wrote, then ONE copy of the match
// records back.  Taken for motif-sized queries (<= 64 query residues, single scan, no --partial-fit); a candidate beyond the
// kernel's limits (64 graph nodes, 1024 found triples) raises a flag and the whole call takes the host path below instead.
// -> FDGPU_OK with *done = true and the outputs set; *done = false: the kernel declined (the caller takes the host path).
static int fd_rb_device_glue(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand, const uint64_t *cand_off,
                             const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct, const fd_hash_params *p, uint32_t node_count,
                             const fd_rb_prep &P, std::chrono::steady_clock::time_point T0, fd_match_rec **matches, uint64_t **match_off, int32_t **residues,
                             uint64_t **res_off, bool *done, fd_rb_dev_out *dev_out = nullptr) {
    *done = false;
    const bool trace = getenv("FDGPU_TRACE") != nullptr;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const uint64_t n_cand = cand_off[n_queries], max_nq = P.max_nq;
    const std::vector<std::vector<uint32_t>> &qhs = P.qhs, &qkf = P.qkf;
    const std::vector<uint32_t> &q_sizes = P.q_sizes;
    const uint32_t htype = p->hash_type;
    auto is_sym = [htype](uint32_t h) { return fd_hash_is_sym(htype, h); };
    int rc = 0;
    uint64_t nf_d = 0, nc_d = 0;
    fd_pair_rec *f_none = nullptr; fd_cand_rec *c_none = nullptr;
    hipStream_t st = c->stream;
    // tables: one packed host block -> one copy.  Built WHILE the pair scan runs (it needs none of the scan's output): the scan's launch
    // is followed by ~0.4 ms of kernel per 128 queries that the host used to wait out before starting on this
    std::vector<rs_query_dev> qt(n_queries);
    std::vector<uint32_t> t_hash, t_kfirst, t_sym, t_qi, t_qj, t_idf, t_idx;
    size_t o_qt = 0, o_h = 0, o_kf = 0, o_sy = 0, o_qi = 0, o_qj = 0, o_idf = 0, o_idx = 0, o_sq = 0, o_cd = 0, o_d0 = 0, o_co = 0, words = 0;
    std::vector<uint32_t> blk_v;
    uint32_t *blk = nullptr;
    const std::function<void()> build_rs_tables = [&]() {
    {
        size_t th = 0, tm = 0, ti = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { th += qhs[t].size(); tm += qms[t]->n; ti += qms[t]->n_indices; }
        t_hash.reserve(th); t_kfirst.reserve(th); t_sym.reserve(th); t_qi.reserve(tm); t_qj.reserve(tm); t_idf.reserve(tm); t_idx.reserve(ti);
    }
    for (uint64_t t = 0; t < n_queries; ++t) {
        const fd_query_map *m = qms[t];
        rs_query_dev &Q = qt[t];
        memset(&Q, 0, sizeof Q);
        Q.qh_off = (uint32_t)t_hash.size(); Q.n_hashes = (uint32_t)qhs[t].size();
        t_hash.insert(t_hash.end(), qhs[t].begin(), qhs[t].end());
        t_kfirst.insert(t_kfirst.end(), qkf[t].begin(), qkf[t].end());
        for (size_t z = 0; z < qhs[t].size(); ++z) t_sym.push_back(is_sym(qhs[t][z]) ? 1u : 0u);
        Q.map_off = (uint32_t)t_qi.size();
        t_qi.insert(t_qi.end(), m->qi, m->qi + m->n);
        t_qj.insert(t_qj.end(), m->qj, m->qj + m->n);
        t_idf.resize(t_idf.size() + m->n);
        if (m->n) memcpy(t_idf.data() + t_idf.size() - m->n, m->idf, m->n * 4);
        Q.idx_off = (uint32_t)t_idx.size(); Q.n_idx = (uint32_t)m->n_indices;
        t_idx.insert(t_idx.end(), m->indices, m->indices + m->n_indices);
        Q.q_size = q_sizes[t];
        Q.q_res0 = (uint32_t)qb->h_res_off[q_struct[t]];
    }
    std::vector<uint32_t> t_slotq(n_cand);
    for (uint64_t t = 0; t < n_queries; ++t) for (uint64_t k = cand_off[t]; k < cand_off[t + 1]; ++k) t_slotq[k] = (uint32_t)t;
    float d0tab[2 * FD_WAVE + 1];
    for (int len = 0; len <= 2 * FD_WAVE; ++len) d0tab[len] = len > 21 ? 1.24f * powf((float)len - 15.0f, 1.0f / 3.0f) - 1.8f : 0.5f;   // metrics.rs:117-123
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t nh = t_hash.size(), nmap = t_qi.size(), nidx = t_idx.size();
    o_qt = 0; o_h = o_qt + up4(n_queries * (sizeof(rs_query_dev) / 4)); o_kf = o_h + up4(nh); o_sy = o_kf + up4(nh); o_qi = o_sy + up4((nh + 3) / 4);
    o_qj = o_qi + up4(nmap); o_idf = o_qj + up4(nmap); o_idx = o_idf + up4(nmap); o_sq = o_idx + up4(nidx); o_cd = o_sq + up4(n_cand);
    o_d0 = o_cd + up4(n_cand); o_co = o_d0 + up4(2 * FD_WAVE + 1); words = o_co + up4(2 * (n_queries + 1)) + 4;
    // packed in a pinned staging buffer of the context (its own: the pair scan's block in slot 0 is being copied while this runs)
    blk = (uint32_t *)c->host_pinned(2, words * 4);
    if (!blk) { blk_v.assign(words, 0); blk = blk_v.data(); }
    memcpy(&blk[o_qt], qt.data(), n_queries * sizeof(rs_query_dev));
    if (nh) { memcpy(&blk[o_h], t_hash.data(), nh * 4); memcpy(&blk[o_kf], t_kfirst.data(), nh * 4); }
    for (size_t k = 0; k < nh; ++k) ((uint8_t *)&blk[o_sy])[k] = (uint8_t)t_sym[k];
    if (nmap) { memcpy(&blk[o_qi], t_qi.data(), nmap * 4); memcpy(&blk[o_qj], t_qj.data(), nmap * 4); memcpy(&blk[o_idf], t_idf.data(), nmap * 4); }
    if (nidx) memcpy(&blk[o_idx], t_idx.data(), nidx * 4);
    memcpy(&blk[o_sq], t_slotq.data(), n_cand * 4);
    memcpy(&blk[o_cd], cand, n_cand * 4);
    memcpy(&blk[o_d0], d0tab, sizeof d0tab);
    memcpy(&blk[o_co], cand_off, (n_queries + 1) * 8);      // (o_co is a multiple of 4 words: 8-byte aligned)
    };
    rc = fd_match_pairs_multi(c, db, resname_std, n_queries, P.mqs.data(), cand, cand_off, p, &f_none, &nf_d, &c_none, &nc_d, 19u, nullptr, nullptr, 0, nullptr, nullptr,
                              nullptr, nullptr, &build_rs_tables);
    if (rc) return rc;
    if (!blk) build_rs_tables();
    auto D1 = t_now();
    const uint64_t cap_m = std::max<uint64_t>(4096, 4 * n_cand), cap_prob = 2 * cap_m, cap_res = cap_m * 2 * max_nq,
                   cap_pts = cap_prob * 2 * max_nq;       // a mapping holds at most one target per query residue: <= 2 max_nq [CA, CB] points
    HIPCHK(c, c->ws[WS_RS_TAB].ensure(words * 4));
    HIPCHK(c, c->ws[WS_RS_SEG].ensure((6 * (n_cand + 1) + nf_d + nc_d + 4) * 4));
    HIPCHK(c, c->ws[WS_RS_OUT].ensure(cap_m * sizeof(rs_match_dev)));
    HIPCHK(c, c->ws[WS_RS_RES].ensure(cap_res * 4));
    HIPCHK(c, c->ws[WS_RS_KX].ensure(cap_pts * 12));
    HIPCHK(c, c->ws[WS_RS_KY].ensure(cap_pts * 12));
    HIPCHK(c, c->ws[WS_RS_GQ].ensure(cap_pts * 4));      // gq | gr: one entry per residue pair (two points) each
    HIPCHK(c, c->ws[WS_RS_KOFF].ensure((cap_prob + 1) * 8 + cap_prob * 4));
    HIPCHK(c, c->ws[WS_RS_SOL].ensure(cap_prob * 18 * 4));
    HIPCHK(c, c->ws[WS_RS_CNT].ensure((2 * RS_CNT_STRIDE + 8) * 8 + (3 * n_cand + 4) * 4));      // the counters, the slots' record counts, the grouping's counts: zeroed by one fill
    // [(records per slot: behind the counters) | slot bases + first residues (2 n_cand + 2) | match_off, res_off (n_queries + 1 each, 8-byte)] for the device-side ordering
    const size_t o_sm = 0, o_scr = o_sm + ((n_cand + 1) & ~(size_t)1), o_mo = (o_scr + 2 * n_cand + 2 + 1) & ~(size_t)1, o_ro = o_mo + 2 * (n_queries + 1),
                 ord_words = o_ro + 2 * (n_queries + 1);
    HIPCHK(c, c->ws[WS_RS_PLAN].ensure(ord_words * 4));
    uint32_t *d_ord = c->ws[WS_RS_PLAN].as<uint32_t>();
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_RS_TAB].p, blk, words * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(c->ws[WS_RS_CNT].p, 0, (2 * RS_CNT_STRIDE + 8) * 8 + (3 * n_cand + 4) * 4, st));
    const uint32_t *dblk = c->ws[WS_RS_TAB].as<uint32_t>();
    uint32_t *sg = c->ws[WS_RS_SEG].as<uint32_t>();
    uint32_t *d_cnt = (uint32_t *)(c->ws[WS_RS_CNT].as<unsigned long long>() + 2 * RS_CNT_STRIDE + 8) + (n_cand + 2), *d_seg = sg + 2 * (n_cand + 1), *d_cur = sg + 4 * (n_cand + 1), *d_pf = sg + 6 * (n_cand + 1), *d_pc = d_pf + nf_d;
    const fd_pair_rec *d_found = c->ws[WS_KEYS_A].as<fd_pair_rec>();
    const fd_cand_rec *d_cands = c->ws[WS_KEYS_B].as<fd_cand_rec>();
    fd_launch_rs_group(d_found, nf_d, d_cands, nc_d, (uint32_t)n_cand, d_cnt, d_seg, d_cur, d_pf, d_pc, st);
    rs_args A;
    memset(&A, 0, sizeof A);
    A.found = d_found; A.cands = d_cands; A.seg_f = d_seg; A.seg_c = d_seg + (n_cand + 1); A.perm_f = d_pf; A.perm_c = d_pc;
    A.cand = dblk + o_cd; A.slot_q = dblk + o_sq;
    A.slot_matches = (uint32_t *)(c->ws[WS_RS_CNT].as<unsigned long long>() + 2 * RS_CNT_STRIDE + 8);
    const bool rs_dbg = getenv("FDGPU_RS_DBG") != nullptr;      // phase clocks of the slots on stderr (measurement aid)
    A.dbg = rs_dbg ? c->ws[WS_RS_CNT].as<unsigned long long>() + 8 : nullptr;
    uint4 *dbg_slot = nullptr;
    if (rs_dbg && hipMalloc((void **)&dbg_slot, std::max<uint64_t>(n_cand, 1) * 16) == hipSuccess) (void)hipMemsetAsync(dbg_slot, 0, n_cand * 16, st);
    struct DbgFree { uint4 *p; ~DbgFree() { if (p) (void)hipFree(p); } } dbg_free{dbg_slot};
    A.dbg_slot = dbg_slot;
    A.order = getenv("FDGPU_RS_ORDER") && getenv("FDGPU_RS_ORDER")[0] == '0' ? nullptr : d_cur;      // 0: slot order (measurement)
    A.db_res_off = db->res_off; A.db_ca = db->ca_xyz; A.db_cb = db->cb_xyz; A.q_ca = qb->ca_xyz; A.q_cb = qb->cb_xyz;
    A.qt = (const rs_query_dev *)(dblk + o_qt); A.hashes = dblk + o_h; A.kfirst = dblk + o_kf; A.sym = (const uint8_t *)(dblk + o_sy);
    A.map_qi = dblk + o_qi; A.map_qj = dblk + o_qj; A.map_idf = (const float *)(dblk + o_idf); A.indices = dblk + o_idx;
    A.d0tab = (const float *)(dblk + o_d0); A.node_count = node_count;
    { const char *nc_env = getenv("FDGPU_RS_NODE_CAP"); const long v = nc_env ? atol(nc_env) : 0; A.node_cap = v > 0 && v < FD_WAVE ? (uint32_t)v : FD_WAVE; }
    A.counters = c->ws[WS_RS_CNT].as<unsigned long long>(); A.flags = (uint32_t *)(A.counters + 4);
    A.matches = c->ws[WS_RS_OUT].as<rs_match_dev>(); A.residues = c->ws[WS_RS_RES].as<int32_t>();
    A.kx = c->ws[WS_RS_KX].as<float>(); A.ky = c->ws[WS_RS_KY].as<float>();
    A.gq = c->ws[WS_RS_GQ].as<uint32_t>(); A.gr = A.gq + cap_pts / 2;
    A.koff = c->ws[WS_RS_KOFF].as<uint64_t>(); A.d0 = (float *)(A.koff + cap_prob + 1);
    A.cap_matches = cap_m; A.cap_res = cap_res; A.cap_prob = cap_prob; A.cap_pts = cap_pts;
    // the split form of the glue (k_rs_setup + a wavefront per component, k_retrieve.hip); FDGPU_RS_SPLIT=0: k_rs_slots alone (read per call: tests compare the two)
    if (!(getenv("FDGPU_RS_SPLIT") && getenv("FDGPU_RS_SPLIT")[0] == '0') && !rs_dbg) {
        auto up16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
        const size_t o_big = 0, o_head = o_big + up16(n_cand * 4), o_nodes = o_head + n_cand * 32, o_comps = o_nodes + n_cand * FD_WAVE * 4,
                     o_work = o_comps + n_cand * 2 * FD_WAVE * 8, o_np = o_work + up16(cap_m * 8), o_gq = o_np + cap_m * 16, o_gr = o_gq + cap_m * 2 * FD_WAVE * 4,
                     o_edges = o_gr + cap_m * 2 * FD_WAVE * 4, sp_bytes = o_edges + (size_t)nf_d * 16 + 16;
        HIPCHK(c, c->ws[WS_RS_SPLIT].ensure(sp_bytes));
        uint8_t *sp = c->ws[WS_RS_SPLIT].as<uint8_t>();
        A.sp_big = (uint32_t *)(sp + o_big); A.sp_head = (uint4 *)(sp + o_head); A.sp_nodes = (uint32_t *)(sp + o_nodes);
        A.sp_comps = (unsigned long long *)(sp + o_comps); A.sp_work = (uint2 *)(sp + o_work); A.sp_edges = (uint4 *)(sp + o_edges);
        A.sp_np = (uint4 *)(sp + o_np); A.sp_gq = (uint32_t *)(sp + o_gq); A.sp_gr = (uint32_t *)(sp + o_gr);
        A.sp_big_n = (uint32_t *)(c->ws[WS_RS_CNT].as<unsigned long long>() + 6);      // zeroed with the counters above (flags at + 4, clocks from + 8)
    }
    {
        StageTimer tm(c, "retrieve_slots", 0);
        fd_launch_rs_slots(A, (uint32_t)n_cand, st);
    }
    HIPCHK(c, hipGetLastError());
    unsigned long long cnt_h[32] = {0};
    {
        std::vector<unsigned long long> cv(2 * RS_CNT_STRIDE + 8);      // the three counters live 4 KB apart (rs_args.counters)
        HIPCHK(c, hipMemcpyAsync(cv.data(), c->ws[WS_RS_CNT].p, cv.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        memcpy(cnt_h, cv.data(), 32 * 8);
        cnt_h[1] = cv[RS_CNT_STRIDE]; cnt_h[2] = cv[2 * RS_CNT_STRIDE];
    }
    if (rs_dbg && dbg_slot) {      // the slots by duration: is the launch its longest slot?
        std::vector<uint4> ds(n_cand);
        if (hipMemcpy(ds.data(), dbg_slot, n_cand * 16, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<uint32_t> idx(n_cand);
            for (uint64_t k = 0; k < n_cand; ++k) idx[k] = (uint32_t)k;
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ds[a].w > ds[b].w; });
            double sum = 0; uint64_t live = 0, big = 0;
            for (uint64_t k = 0; k < n_cand; ++k) { sum += ds[k].w; live += ds[k].w ? 1 : 0; big += ds[k].y > 512 ? 1 : 0; }
            fprintf(stderr, "[rs_slots] %llu slots, %llu with found triples (%llu with more candidate pairs than LDS holds), mean %.1f us; longest (us: found, cands, components):", (unsigned long long)n_cand,
                    (unsigned long long)live, (unsigned long long)big, live ? sum / live / 100.0 : 0.0);
            for (uint64_t k = 0; k < std::min<uint64_t>(n_cand, 24); ++k) fprintf(stderr, " %.0f:%u,%u,%u", ds[idx[k]].w / 100.0, ds[idx[k]].x, ds[idx[k]].y, ds[idx[k]].z);
            fprintf(stderr, "; percentiles 50/90/99: %.0f / %.0f / %.0f us\n", ds[idx[n_cand / 2]].w / 100.0, ds[idx[n_cand / 10]].w / 100.0, ds[idx[n_cand / 100]].w / 100.0);
        }
    }
    if (rs_dbg) {
        const unsigned long long *d = cnt_h + 8;
        const double ns = std::max<double>((double)n_cand, 1.0) * 100.0, nc2 = std::max<double>((double)d[7], 1.0) * 100.0;
        fprintf(stderr, "[rs_slots] per slot (%llu slots): edges + rank %.2f, lookup %.2f, nodes %.2f, components %.2f us; per slot WITH components (%llu): order %.2f, "
                        "components' loop %.2f us = votes %.2f + best / greedy %.2f + rescue votes %.2f + assignment %.2f + output %.2f\n", (unsigned long long)n_cand, d[0] / ns, d[1] / ns,
                d[2] / ns, d[3] / ns, d[7], d[4] / nc2, (d[5] + d[8] + d[9] + d[10] + d[11] + d[12]) / nc2, d[8] / nc2, d[9] / nc2, d[10] / nc2, d[11] / nc2, d[12] / nc2);
    }
    const uint32_t dflags = (uint32_t)cnt_h[4];
    auto D2 = t_now();
    if (dflags == 0) {
        const uint64_t nm = cnt_h[0], nprob = cnt_h[1] >> 40, npts = cnt_h[1] & ((1ull << 40) - 1ull);
        // the 32-byte match headers come back for the ordering; the records themselves (fd_match_rec: 39 words from the solution arrays)
        // and the residue lists are gathered on the device in their final order (k_rs_records) and copied straight into the caller's
        // page-locked arrays — the host loop over 23 k records of a 512-query batch (8 scattered reads + 232 bytes written each) was
        // 1.6-2.3 ms of the call
        float *d_rmsd0 = c->ws[WS_RS_SOL].as<float>();
        const char *ho_env = getenv("FDGPU_RS_HOST_ORDER");       // 1: the records are ordered on the host from their headers (tests compare the two)
        if (!(ho_env && ho_env[0] == '1')) {
            // the records' final places are computed on the device (k_rs_offsets: bases of the slots from their record counts; a record's
            // place inside its slot was fixed when it was written) — no header copy, no host sort, no gather plan: after the counters above
            // nothing but the finished arrays crosses the bus, with one wait
            const uint64_t tot_res = cnt_h[2];
            if (tot_res >= (1ull << 32)) { c->err = "retrieve_batch: residue lists beyond 2^32 entries; split the batch"; return FDGPU_ERANGE; }
            float *d_rot0 = d_rmsd0 + nprob, *d_tran0 = d_rot0 + 9 * nprob, *d_met0 = d_tran0 + 3 * nprob;
            uint64_t *omo = (uint64_t *)malloc((n_queries + 1) * 8), *oro = (uint64_t *)malloc((n_queries + 1) * 8);
            // dev_out: the ordered records stay in the workspaces for the caller (sharded retrieval: gathered device to device); only the offsets return
            fd_match_rec *om = dev_out ? nullptr : (fd_match_rec *)fd_out_alloc(std::max<size_t>(nm, 1) * sizeof(fd_match_rec), true);
            int32_t *orr = dev_out ? nullptr : (int32_t *)fd_out_alloc(std::max<size_t>(tot_res, 1) * sizeof(int32_t), true);
            if (!omo || !oro || (!dev_out && (!om || !orr))) { fdgpu_free(om); fdgpu_free(orr); free(omo); free(oro); return FDGPU_ENOMEM; }
            hipError_t e = c->ws[WS_RS_REC].ensure(std::max<uint64_t>(nm, 1) * sizeof(fd_match_rec));
            if (e == hipSuccess) e = c->ws[WS_RS_RECRES].ensure(std::max<uint64_t>(tot_res, 1) * 4);
            if (e == hipSuccess && nprob) {
                fd_launch_rs_points(A, nprob, npts, st);
                fd_launch_kabsch(A.kx, A.ky, A.koff, nprob, d_rmsd0, d_rot0, d_tran0, st, npts);
                fd_launch_metrics(A.ky, A.kx, A.koff, nprob, d_rot0, d_tran0, A.d0, d_met0, st, npts);
            }
            uint64_t *d_mo = (uint64_t *)(d_ord + o_mo), *d_ro = (uint64_t *)(d_ord + o_ro);
            if (e == hipSuccess) {
                fd_launch_rs_records_dev(A.matches, nm, A.slot_matches, (uint32_t)n_cand, (const uint64_t *)(dblk + o_co), A.slot_q, A.qt, (uint32_t)n_queries, d_ord + o_scr,
                                         d_mo, d_ro, d_rmsd0, d_rot0, d_tran0, d_met0, A.residues, c->ws[WS_RS_REC].p, c->ws[WS_RS_RECRES].as<int32_t>(), st);
                e = hipGetLastError();
            }
            if (e == hipSuccess && nm && !dev_out) e = hipMemcpyAsync(om, c->ws[WS_RS_REC].p, nm * sizeof(fd_match_rec), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && tot_res && !dev_out) e = hipMemcpyAsync(orr, c->ws[WS_RS_RECRES].p, tot_res * 4, hipMemcpyDeviceToHost, st);
            if (dev_out) { dev_out->got = true; dev_out->recs = c->ws[WS_RS_REC].p; dev_out->residues = c->ws[WS_RS_RECRES].as<int32_t>(); dev_out->n_recs = nm; dev_out->n_res = tot_res; }
            // the two offset arrays lie side by side on the device: one copy into the context's page-locked block, parted on the host
            uint64_t *off_land = d_ro == d_mo + (n_queries + 1) ? (uint64_t *)c->host_pinned(1, 2 * (n_queries + 1) * 8) : nullptr;
            if (e == hipSuccess && off_land) e = hipMemcpyAsync(off_land, d_mo, 2 * (n_queries + 1) * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && !off_land) e = hipMemcpyAsync(omo, d_mo, (n_queries + 1) * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && !off_land) e = hipMemcpyAsync(oro, d_ro, (n_queries + 1) * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess && off_land) { memcpy(omo, off_land, (n_queries + 1) * 8); memcpy(oro, off_land + (n_queries + 1), (n_queries + 1) * 8); }
            if (e != hipSuccess) { fdgpu_free(om); fdgpu_free(orr); free(omo); free(oro); c->err = std::string("retrieve_batch records: ") + hipGetErrorString(e); return FDGPU_EHIP; }
            if (trace) fprintf(stderr, "[fdgpu_retrieve] device glue: scan %.3f ms (found %llu, cands %llu), group+slots %.3f, superpose + records ordered on the device + copy (%llu records, %llu problems) %.3f\n",
                               t_ms(T0, D1), (unsigned long long)nf_d, (unsigned long long)nc_d, t_ms(D1, D2), (unsigned long long)nm, (unsigned long long)nprob, t_ms(D2, t_now()));
            *matches = om; *match_off = omo; *residues = orr; *res_off = oro;
            *done = true;
            return FDGPU_OK;
        }
        std::vector<uint8_t> land_v;
        const size_t b_hm = std::max<uint64_t>(nm, 1) * sizeof(rs_match_dev);
        uint8_t *land = (uint8_t *)c->host_pinned(1, b_hm);
        if (!land) { land_v.resize(b_hm); land = land_v.data(); }
        rs_match_dev *hm = (rs_match_dev *)land;
        float *d_rmsd = c->ws[WS_RS_SOL].as<float>(), *d_rot = d_rmsd + nprob, *d_tran = d_rot + 9 * nprob, *d_met = d_tran + 3 * nprob;
        if (nprob) {
            fd_launch_rs_points(A, nprob, npts, st);
            fd_launch_kabsch(A.kx, A.ky, A.koff, nprob, d_rmsd, d_rot, d_tran, st, npts);
            fd_launch_metrics(A.ky, A.kx, A.koff, nprob, d_rot, d_tran, A.d0, d_met, st, npts);
            HIPCHK(c, hipGetLastError());
        }
        if (nm) {
            HIPCHK(c, hipMemcpyAsync(hm, A.matches, nm * sizeof(rs_match_dev), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
        }
        const auto E1 = t_now();
        // the records arrive in append order: into (slot, component) order by a counting sort over the slots and a sort of every slot's few
        // records (a comparison sort of all 45 k records of a 512-query batch was 3 of the stage's 3.7 ms)
        std::vector<uint32_t> order(nm);
        {
            std::vector<uint32_t> at(n_cand + 2, 0);
            bool in_range = true;
            for (uint64_t k = 0; k < nm; ++k) { if (hm[k].slot < n_cand) ++at[hm[k].slot + 1]; else in_range = false; }
            if (in_range) {
                for (uint64_t z = 0; z < n_cand; ++z) at[z + 1] += at[z];
                std::vector<uint32_t> cur(at.begin(), at.end() - 1);
                for (uint64_t k = 0; k < nm; ++k) order[cur[hm[k].slot]++] = (uint32_t)k;
                for (uint64_t z = 0; z < n_cand; ++z)
                    if (at[z + 1] - at[z] > 1)
                        std::sort(order.begin() + at[z], order.begin() + at[z + 1], [&](uint32_t a, uint32_t b) { return hm[a].ci < hm[b].ci; });
            } else {
                for (uint64_t k = 0; k < nm; ++k) order[k] = (uint32_t)k;
                std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hm[a].slot != hm[b].slot ? hm[a].slot < hm[b].slot : hm[a].ci < hm[b].ci; });
            }
        }
        const auto E2 = t_now();
        // the gather plan per output record and the per-query offsets
        uint64_t *omo = (uint64_t *)malloc((n_queries + 1) * 8), *oro = (uint64_t *)malloc((n_queries + 1) * 8);
        if (!omo || !oro) { free(omo); free(oro); return FDGPU_ENOMEM; }
        std::vector<uint32_t> plan_v;
        uint32_t *plan = (uint32_t *)c->host_pinned(0, std::max<uint64_t>(nm, 1) * 16);
        if (!plan) { plan_v.resize(std::max<uint64_t>(nm, 1) * 4); plan = plan_v.data(); }
        uint64_t tq = 0, rpos = 0;
        omo[0] = 0; oro[0] = 0;
        for (uint64_t k = 0; k < nm; ++k) {
            const rs_match_dev &m = hm[order[k]];
            while (m.slot >= cand_off[tq + 1]) { ++tq; omo[tq] = k; oro[tq] = rpos; }
            const uint64_t nq2 = 2 * qms[tq]->n_indices;
            plan[4 * k] = order[k]; plan[4 * k + 1] = (uint32_t)(m.slot - cand_off[tq]); plan[4 * k + 2] = (uint32_t)rpos; plan[4 * k + 3] = (uint32_t)nq2;
            rpos += nq2;
        }
        while (tq < n_queries) { ++tq; omo[tq] = nm; oro[tq] = rpos; }
        const uint64_t tot_res = rpos;
        if (tot_res >= (1ull << 32)) { free(omo); free(oro); c->err = "retrieve_batch: residue lists beyond 2^32 entries; split the batch"; return FDGPU_ERANGE; }
        fd_match_rec *om = (fd_match_rec *)fd_out_alloc(std::max<size_t>(nm, 1) * sizeof(fd_match_rec), true);
        int32_t *orr = (int32_t *)fd_out_alloc(std::max<size_t>(tot_res, 1) * sizeof(int32_t), true);
        if (!om || !orr) { fdgpu_free(om); fdgpu_free(orr); free(omo); free(oro); return FDGPU_ENOMEM; }
        const auto E3 = t_now();
        if (nm) {
            hipError_t e = c->ws[WS_RS_PLAN].ensure(nm * 16);
            if (e == hipSuccess) e = c->ws[WS_RS_REC].ensure(nm * sizeof(fd_match_rec));
            if (e == hipSuccess) e = c->ws[WS_RS_RECRES].ensure(std::max<uint64_t>(tot_res, 1) * 4);
            if (e == hipSuccess) e = hipMemcpyAsync(c->ws[WS_RS_PLAN].p, plan, nm * 16, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                fd_launch_rs_records(A.matches, c->ws[WS_RS_PLAN].p, nm, d_rmsd, d_rot, d_tran, d_met, A.residues, c->ws[WS_RS_REC].p, c->ws[WS_RS_RECRES].as<int32_t>(), st);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(om, c->ws[WS_RS_REC].p, nm * sizeof(fd_match_rec), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && tot_res) e = hipMemcpyAsync(orr, c->ws[WS_RS_RECRES].p, tot_res * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { fdgpu_free(om); fdgpu_free(orr); free(omo); free(oro); c->err = std::string("retrieve_batch records: ") + hipGetErrorString(e); return FDGPU_EHIP; }
        }
        if (trace) fprintf(stderr, "[fdgpu_retrieve] device glue: scan %.3f ms (found %llu, cands %llu), group+slots %.3f, superpose+copy+assemble(%llu) %.3f (superpose + headers %.3f, order %.3f, plan %.3f, records %.3f)\n",
                           t_ms(T0, D1), (unsigned long long)nf_d, (unsigned long long)nc_d, t_ms(D1, D2), (unsigned long long)nprob, t_ms(D2, t_now()), t_ms(D2, E1), t_ms(E1, E2),
                           t_ms(E2, E3), t_ms(E3, t_now()));
        *matches = om; *match_off = omo; *residues = orr; *res_off = oro;
        *done = true;
        return FDGPU_OK;
    }
    if (trace) fprintf(stderr, "[fdgpu_retrieve] device glue declined (flags %u): host path\n", dflags);
    return FDGPU_OK;
}

// retrieval_wrapper for MANY queries: one pair scan, one coordinate gather and one Kabsch launch in total.  Query t = structure
// q_struct[t] of qb with the query map qms[t]; its candidates are cand[cand_off[t] .. cand_off[t+1]).  Matches of query t:
// (*matches)[(*match_off)[t] .. (*match_off)[t+1]) (cand = slot inside the query's own candidate list), residues
// (*residues)[(*res_off)[t] ...], 2 * n_indices(t) per match.
static int fd_retrieve_batch_impl(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand,
                                  const uint64_t *cand_off, const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct,
                                  const fd_hash_params *p, float ca_distance_cutoff, uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches,
                                  uint64_t **match_off, int32_t **residues, uint64_t **res_off, const fd_rb_prep *prep, fd_rb_dev_out *dev_out = nullptr);
int fd_retrieve_batch_dev(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand, const uint64_t *cand_off,
                          const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct, const fd_hash_params *p, float ca_distance_cutoff,
                          uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off, fd_rb_dev_out *dev) { FD_LOCK(c);
    if (dev) *dev = fd_rb_dev_out();
    return fd_retrieve_batch_impl(c, db, resname_std, n_queries, cand, cand_off, qms, qb, q_struct, p, ca_distance_cutoff, node_count, partial_fit, matches, match_off,
                                  residues, res_off, nullptr, dev);
}
extern "C" int fdgpu_retrieve_batch(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand,
                                    const uint64_t *cand_off, const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct,
                                    const fd_hash_params *p, float ca_distance_cutoff, uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches,
                                    uint64_t **match_off, int32_t **residues, uint64_t **res_off) { FD_LOCK(c);
    return fd_retrieve_batch_impl(c, db, resname_std, n_queries, cand, cand_off, qms, qb, q_struct, p, ca_distance_cutoff, node_count, partial_fit, matches, match_off,
                                  residues, res_off, nullptr);
}
// prep: the maps' tables when the caller has built them already (fd_rb_prepare with the same maps and ca_distance_cutoff), else null
static int fd_retrieve_batch_impl(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand,
                                  const uint64_t *cand_off, const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct,
                                  const fd_hash_params *p, float ca_distance_cutoff, uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches,
                                  uint64_t **match_off, int32_t **residues, uint64_t **res_off, const fd_rb_prep *prep, fd_rb_dev_out *dev_out) { FD_LOCK(c);
    if (!c || !db || !qb || !p || !matches || !match_off || !residues || !res_off || !cand_off || (n_queries && (!qms || !q_struct))) return FDGPU_EINVAL;
    *matches = nullptr; *match_off = nullptr; *residues = nullptr; *res_off = nullptr;
    const uint64_t n_cand = cand_off[n_queries];
    for (uint64_t t = 0; t < n_queries; ++t) if (!qms[t] || q_struct[t] >= qb->n_struct) return FDGPU_EINVAL;
    const bool trace = getenv("FDGPU_TRACE") != nullptr;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto T0 = t_now();
    fd_rb_prep prep_own;
    if (!prep) { fd_rb_prepare(n_queries, qms, ca_distance_cutoff, prep_own); prep = &prep_own; }
    const std::vector<std::vector<uint32_t>> &qhs = prep->qhs, &qkf = prep->qkf;
    const std::vector<fd_match_query> &mqs = prep->mqs;
    const std::vector<uint32_t> &q_sizes = prep->q_sizes;
    fd_pair_rec *found = nullptr; fd_cand_rec *cands = nullptr;
    uint64_t nf = 0, nc = 0;
    // Candidate pairs (the rescue's raw material, retrieve.rs:120-121) number ~ pairs x observed-list entries: a whole-structure
    // query makes millions per candidate structure.  Large queries therefore scan twice: found triples only, then — once the
    // components and their residue mappings are known — candidate pairs only for partner residues some component mapped
    // (the rescue counts nothing else, retrieve.rs:498-511).
    uint64_t max_aad = 0;
    for (uint64_t t = 0; t < n_queries; ++t) max_aad = std::max<uint64_t>(max_aad, qms[t]->n_aad);
    const char *tp_env = getenv("FDGPU_TWO_PASS");     // 1 / 0 force the choice (tests)
    const bool two_pass = tp_env ? tp_env[0] == '1' : max_aad > 4096;
    uint32_t *pk_key = nullptr, *pk_val = nullptr;      // packed, device-sorted candidate pairs (see fd_match_pairs_multi, mode bit 3): context-owned pinned buffers
    struct HostBufs {   // the scan outputs live until the slots are processed; every return path below releases them
        fd_pair_rec *&f; fd_cand_rec *&c; uint32_t *&k, *&v;
        ~HostBufs() { free(f); free(c); k = nullptr; v = nullptr; }      // k / v: the context's pinned buffers, not ours to free
    } host_bufs{found, cands, pk_key, pk_val};
    int rc = 0;
    const uint32_t htype = p->hash_type;
    auto is_sym = [htype](uint32_t h) { return fd_hash_is_sym(htype, h); };
    uint64_t max_nq = 0;
    for (uint64_t t = 0; t < n_queries; ++t) max_nq = std::max<uint64_t>(max_nq, qms[t]->n_indices);
    const char *hg_env = getenv("FDGPU_HOST_GLUE");      // 1 forces the host path (tests compare the two)
    const bool dev_glue = !(hg_env && hg_env[0] == '1') && !two_pass && !partial_fit && max_nq <= FD_WAVE && max_nq > 0 && n_cand > 0 && n_cand < (1ull << 20);
    if (dev_glue) {
        bool done = false;
        rc = fd_rb_device_glue(c, db, resname_std, n_queries, cand, cand_off, qms, qb, q_struct, p, node_count, *prep, T0, matches, match_off, residues, res_off, &done, dev_out);
        if (rc || done) return rc;
    }
    if (trace) fprintf(stderr, "[fdgpu_retrieve] query tables %.3f ms\n", t_ms(T0, t_now()));
    // large queries: hash -> map entry through an open-addressing table built once per query (a whole-structure query looks ~10^5 found
    // edges up in ~10^5 hashes per call: 17 bisection steps each were a quarter of the largest candidate's time).  Built by a helper thread
    // while the first pair scan runs on the GPU (1 ms for 10^5 hashes)
    std::vector<std::vector<uint64_t>> e_tab(n_queries);
    std::vector<uint32_t> e_mask(std::max<uint64_t>(n_queries, 1), 0);
    auto build_e_tab = [&]() {
        for (uint64_t t = 0; t < n_queries; ++t) {
            if (qhs[t].size() <= 4096) continue;
            uint32_t cap = 1;
            while (cap < 2 * qhs[t].size()) cap <<= 1;
            e_tab[t].assign(cap, ~0ull);
            e_mask[t] = cap - 1;
            for (size_t z = 0; z < qhs[t].size(); ++z) {
                uint32_t at = (qhs[t][z] * 2654435761u) & e_mask[t];
                while (e_tab[t][at] != ~0ull) at = (at + 1) & e_mask[t];
                e_tab[t][at] = ((uint64_t)qhs[t][z] << 32) | qkf[t][z];
            }
        }
    };
    // the same helper: every query's observed-distance lists in the pair scan's group layout (counting sort by aa_i * 32 + aa_j, entries with
    // residue types below 32, queries concatenated), every group sorted by distance — what the second scan's vote loop searches (k_match.hip)
    std::vector<float> sd_dist;
    std::vector<uint32_t> sd_qi;
    auto build_sorted_lists = [&]() {
        std::vector<std::pair<float, uint32_t>> grp;
        for (uint64_t t = 0; t < n_queries; ++t) {
            const fd_query_map *m = qms[t];
            std::vector<uint32_t> cnt(1025, 0);
            for (uint64_t e = 0; e < m->n_aad; ++e) if (m->aad_aa1[e] < 32 && m->aad_aa2[e] < 32) ++cnt[m->aad_aa1[e] * 32u + m->aad_aa2[e] + 1];
            for (int k = 0; k < 1024; ++k) cnt[k + 1] += cnt[k];
            const size_t base = sd_dist.size();
            sd_dist.resize(base + cnt[1024]); sd_qi.resize(base + cnt[1024]);
            std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
            for (uint64_t e = 0; e < m->n_aad; ++e)
                if (m->aad_aa1[e] < 32 && m->aad_aa2[e] < 32) { const uint32_t k = cur[m->aad_aa1[e] * 32u + m->aad_aa2[e]]++; sd_dist[base + k] = m->aad_dist[e]; sd_qi[base + k] = m->aad_qi[e]; }
            for (int g = 0; g < 1024; ++g) {
                const uint32_t a = cnt[g], b = cnt[g + 1];
                if (b - a < 2) continue;
                grp.clear();
                for (uint32_t k = a; k < b; ++k) grp.emplace_back(sd_dist[base + k], sd_qi[base + k]);
                std::stable_sort(grp.begin(), grp.end(), [](const std::pair<float, uint32_t> &x, const std::pair<float, uint32_t> &y) { return x.first < y.first; });
                for (uint32_t k = a; k < b; ++k) { sd_dist[base + k] = grp[k - a].first; sd_qi[base + k] = grp[k - a].second; }
            }
        }
    };
    bool big_tab = false;
    for (uint64_t t = 0; t < n_queries; ++t) big_tab = big_tab || qhs[t].size() > 4096;
    std::thread e_tab_thread;
    if (big_tab) e_tab_thread = std::thread([&]() { build_e_tab(); if (two_pass) build_sorted_lists(); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } e_tab_join{e_tab_thread};      // every return path below waits for it
    fd_mp_tables mp_tab;      // work items + query tables: built by the first scan, reused by the second
    rc = fd_match_pairs_multi(c, db, resname_std, n_queries, mqs.data(), cand, cand_off, p, &found, &nf, &cands, &nc, two_pass ? 1u : 15u,
                              nullptr, nullptr, 0, &pk_key, &pk_val, nullptr, &mp_tab);
    if (rc) return rc;
    auto T1 = t_now();
    // per candidate: graph -> components -> vote -> rescue; Kabsch problems collected for one GPU batch
    std::vector<fd_match_rec> recs;
    std::vector<int32_t> res;               // 2 * NQ per match: from_hash then processed target residue index (-1 = none)
    std::vector<float> kx, ky;              // Kabsch points (target = moving x, query = fixed y)
    std::vector<uint64_t> koff(1, 0);
    struct Pend { size_t rec; int which; }; // which: 0 from_hash, 1 processed
    std::vector<Pend> pend;
    // host copies of the coordinates needed for the Kabsch point lists (every query structure of qb)
    std::vector<float> qb_ca(std::max<uint64_t>(qb->n_res, 1) * 3), qb_cb(qb_ca.size());
    if (qb->n_res) {
        HIPCHK(c, hipMemcpy(qb_ca.data(), qb->ca_xyz, qb->n_res * 12, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(qb_cb.data(), qb->cb_xyz, qb->n_res * 12, hipMemcpyDeviceToHost));
    }
    std::vector<uint64_t> m_off(n_queries + 1, 0), r_off(n_queries + 1, 0);
    std::vector<uint64_t> g_src(std::max<uint64_t>(n_cand, 1)), g_dst(n_cand + 1, 0);
    for (uint64_t k = 0; k < n_cand; ++k) {
        g_src[k] = db->h_res_off[cand[k]];
        g_dst[k + 1] = g_dst[k] + (db->h_res_off[cand[k] + 1] - db->h_res_off[cand[k]]);
    }
    const uint64_t g_total = g_dst[n_cand];
    std::vector<float> t_all(std::max<uint64_t>(6 * g_total, 1));
    if (g_total && nf) {
        HIPCHK(c, c->ws[WS_MISC0].ensure(n_cand * 8));
        HIPCHK(c, c->ws[WS_MISC1].ensure((n_cand + 1) * 8));
        HIPCHK(c, c->ws[WS_MISC2].ensure(6 * g_total * 4));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, g_src.data(), n_cand * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, g_dst.data(), (n_cand + 1) * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_xyz, dim3((unsigned)n_cand), dim3(256), 0, c->stream, db->ca_xyz, db->cb_xyz, c->ws[WS_MISC0].as<uint64_t>(),
                           c->ws[WS_MISC1].as<uint64_t>(), g_total, c->ws[WS_MISC2].as<float>());
        HIPCHK(c, hipMemcpyAsync(t_all.data(), c->ws[WS_MISC2].p, 6 * g_total * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    auto T2 = t_now();
    std::vector<uint32_t> mask_off(n_cand + 1, 0), cj_mask((g_total + 31) / 32 + 1, 0);
    for (uint64_t k = 0; k < n_cand; ++k) mask_off[k] = (uint32_t)g_dst[k];
    bool any_rescue = false;
    // plan = true: components and mappings only, marking the mapped residues of every component that leaves a query residue
    // unmatched (first pass of a large query); plan = false: the full body
    // two-scan retrievals build the components and from-hash mappings once (plan pass) and reuse them in the full pass
    struct CompPlan { std::vector<uint32_t> q_idx, r_idx; float sub_idf; uint32_t rc_ord = 0; };      // rc_ord: 1-based ordinal among the slot's components that need the rescue
    std::vector<std::vector<CompPlan>> plan_cache(two_pass ? n_cand : 0);
    std::vector<char> plan_have(two_pass ? n_cand : 0, 0);
    // Candidate slots are independent (their own found triples, candidate pairs and outputs): they are processed by a pool of
    // host threads into per-slot outputs that are merged in slot order afterwards, so the result does not depend on the thread count.
    struct SlotOut {
        std::vector<fd_match_rec> recs; std::vector<int32_t> res; std::vector<float> kx, ky; std::vector<uint64_t> klen;
        std::vector<Pend> pend; std::vector<uint32_t> marks, mark_ord;
    };
    // rescue votes counted on the device (second scan, fd_match_pairs_multi mode bit 5): per marked target residue the ordinal of the component
    // that mapped it; rows of (largest count, holders, which) per (slot, ordinal, query residue) come back instead of the candidate pairs
    std::vector<uint8_t> cj_comp(two_pass ? g_total + 1 : 1, 0);
    std::vector<uint32_t> slot_nrc(n_cand + 1, 0);
    std::vector<uint64_t> row_base(n_cand + 1, 0);
    std::vector<fd_vote_row> vote_rows;
    bool vote_conflict = false, have_rows = false;
    std::vector<uint32_t> slot_q(std::max<uint64_t>(n_cand, 1), 0);
    for (uint64_t t = 0; t < n_queries; ++t) for (uint64_t k = cand_off[t]; k < cand_off[t + 1]; ++k) slot_q[k] = (uint32_t)t;
    std::vector<size_t> f_lo(n_cand + 1, 0), c_lo(n_cand + 1, 0);
    auto do_slot = [&](const uint64_t slot, const bool plan, SlotOut &o) {
        const uint64_t tq = slot_q[slot];
        const fd_query_map *qm = qms[tq];
        const std::vector<uint32_t> &e_hash = qhs[tq], &e_first = qkf[tq];
        const uint32_t q_size = q_sizes[tq];
        const uint64_t NQ = qm->n_indices;
        const float *q_ca = qb_ca.data() + 3 * qb->h_res_off[q_struct[tq]], *q_cb = qb_cb.data() + 3 * qb->h_res_off[q_struct[tq]];
        const size_t f0 = f_lo[slot], fpos = f_lo[slot + 1], c0 = c_lo[slot], cpos = c_lo[slot + 1];
        if (fpos == f0) return;
        const bool cached = !plan && two_pass && plan_have[slot];
        Graph g;
        std::vector<std::vector<uint32_t>> comps;
        std::vector<int32_t> edge_k;
        const bool big_trace = trace && fpos - f0 > 4000;
        const auto s_t0 = t_now();
        if (!cached) {
            if (fpos - f0 > 256) g.dense.assign((size_t)(g_dst[slot + 1] - g_dst[slot]), -1);
            g.es.reserve(fpos - f0); g.et.reserve(fpos - f0); g.eh.reserve(fpos - f0);
            for (size_t e = f0; e < fpos; ++e) {
                uint32_t a = g.node_of(found[e].i), b = g.node_of(found[e].j);
                g.es.push_back(a); g.et.push_back(b); g.eh.push_back(found[e].hash);
            }
            if (big_trace) fprintf(stderr, "[slot %llu] %zu edges: graph at %.3f ms\n", (unsigned long long)slot, fpos - f0, t_ms(s_t0, t_now()));
            comps = components(g, node_count);
            if (big_trace) fprintf(stderr, "[slot %llu] %zu components at %.3f ms\n", (unsigned long long)slot, comps.size(), t_ms(s_t0, t_now()));
            // query-map entry of every found edge, looked up once per candidate (a whole-structure query has ~10^5 entries and
            // thousands of edges per candidate; every component walks the edge list)
            edge_k.resize(g.es.size());
            if (!e_tab[tq].empty()) {
                const std::vector<uint64_t> &T = e_tab[tq];
                const uint32_t msk = e_mask[tq];
                for (size_t e = 0; e < g.es.size(); ++e) {
                    if (e + 12 < g.es.size()) __builtin_prefetch(&T[(g.eh[e + 12] * 2654435761u) & msk]);      // the table of a whole-structure query is ~2 MB: a miss per probe otherwise
                    uint32_t at = (g.eh[e] * 2654435761u) & msk;
                    int32_t k = -1;
                    while (T[at] != ~0ull) { if ((uint32_t)(T[at] >> 32) == g.eh[e]) { k = (int32_t)(uint32_t)T[at]; break; } at = (at + 1) & msk; }
                    edge_k[e] = k;
                }
            } else
            for (size_t e = 0; e < g.es.size(); ++e) {
                const auto it = std::lower_bound(e_hash.begin(), e_hash.end(), g.eh[e]);
                edge_k[e] = (it == e_hash.end() || *it != g.eh[e]) ? -1 : (int32_t)e_first[(size_t)(it - e_hash.begin())];
            }
        }
        // the edges of every component, ascending: an edge belongs to a component that holds both its ends (components overlap: a strongly connected set
        // and the weakly connected set around it).  The component loop below walked ALL edges per component — 12 components x 12 k edges for a 2,500-residue
        // candidate of a whole-structure query
        std::vector<std::vector<uint32_t>> comp_edges(comps.size());
        if (!cached && !comps.empty()) {
            const size_t nn = g.w.size();
            std::vector<uint32_t> n_off(nn + 1, 0), n_list;
            for (const auto &cc : comps) for (uint32_t v : cc) ++n_off[v + 1];
            for (size_t v = 0; v < nn; ++v) n_off[v + 1] += n_off[v];
            n_list.resize(n_off[nn]);
            {
                std::vector<uint32_t> cur(n_off.begin(), n_off.end() - 1);
                for (size_t ci = 0; ci < comps.size(); ++ci) for (uint32_t v : comps[ci]) n_list[cur[v]++] = (uint32_t)ci;      // per node: its components, ascending
            }
            for (size_t e = 0; e < g.es.size(); ++e) {
                const uint32_t u = g.es[e], v = g.et[e];
                for (uint32_t x = n_off[u]; x < n_off[u + 1]; ++x)
                    for (uint32_t y = n_off[v]; y < n_off[v + 1]; ++y)
                        if (n_list[x] == n_list[y]) comp_edges[n_list[x]].push_back((uint32_t)e);
            }
        }
        const size_t n_comps = cached ? plan_cache[slot].size() : comps.size();
        if (big_trace) fprintf(stderr, "[slot %llu] edge entries at %.3f ms (%s)\n", (unsigned long long)slot, t_ms(s_t0, t_now()), plan ? "plan" : "full");
        if (n_comps == 0) return;
        const uint32_t s = cand[slot];
        (void)s;
        const float *t_ca = t_all.data() + 3 * g_dst[slot], *t_cb = t_all.data() + 3 * g_total + 3 * g_dst[slot];
        const uint32_t Rt = (uint32_t)(g_dst[slot + 1] - g_dst[slot]);
        // candidate pairs of this structure bucketed by their partner residue j (counting sort): a component's rescue counts, per
        // unmatched query residue, the pairs whose partner the component mapped (retrieve.rs:498-511) — one walk over the
        // buckets of its mapped residues instead of one over every pair per query residue (whole-structure queries: millions)
        const bool have_c = cpos > c0;
        auto c_ok = [&](const fd_cand_rec &r) { return r.qi < q_size && r.i < Rt && r.j < Rt; };
        const bool pk = pk_key != nullptr;
        std::vector<uint32_t> by_cj_off(have_c ? Rt + 2 : 2, 0), by_cj_qi(have_c && !pk ? cpos - c0 : 0), by_cj_i(have_c && !pk ? cpos - c0 : 0);
        if (have_c && pk) {   // already sorted by partner residue on the device: the bucket of j is a slice of the packed arrays
            for (size_t e = c0; e < cpos; ++e) { const uint32_t j = pk_key[e] & 0xffffu; if (j < Rt) ++by_cj_off[j + 1]; }
            for (uint32_t z = 0; z <= Rt; ++z) by_cj_off[z + 1] += by_cj_off[z];
        } else if (have_c) {
            for (size_t e = c0; e < cpos; ++e) if (c_ok(cands[e])) ++by_cj_off[cands[e].j + 1];
            for (uint32_t z = 0; z <= Rt; ++z) by_cj_off[z + 1] += by_cj_off[z];
            std::vector<uint32_t> cur(by_cj_off.begin(), by_cj_off.end() - 1);
            for (size_t e = c0; e < cpos; ++e) if (c_ok(cands[e])) { uint32_t k = cur[cands[e].j]++; by_cj_qi[k] = cands[e].qi; by_cj_i[k] = cands[e].i; }
        }
        std::vector<uint32_t> votes2(have_c ? (size_t)q_size * Rt : 0, 0), v_touched, q_touched;
        std::vector<uint32_t> vote_pos;      // dense (q, r) -> 1 + vote position of the component at hand (see below)
        std::vector<uint32_t> r_mx(q_size, 0), r_nmx(q_size, 0), r_arg(q_size, 0);
        if (plan && two_pass) { plan_cache[slot].resize(n_comps); plan_have[slot] = 1; }
        uint32_t n_rc = 0;
        for (size_t ci = 0; ci < n_comps; ++ci) {
            float sub_idf = 0.0f;
            std::vector<uint32_t> q_idx, r_idx;
            if (cached) { q_idx = plan_cache[slot][ci].q_idx; r_idx = plan_cache[slot][ci].r_idx; sub_idf = plan_cache[slot][ci].sub_idf; }
            else {
            const std::vector<uint32_t> &cc = comps[ci];
            const std::vector<uint32_t> &ce = comp_edges[ci];
            // votes (query residue, target residue) -> saturating u8 count (retrieve.rs:631-666).  Sparse: a component has a
            // handful of edges, the dense q_size x r_size table of the reference is ~1 MB per component for a motif taken
            // from a long chain.  best_c[q] = max count, best_r[q] = smallest target residue holding it (what the
            // reference's running update converges to, counts only grow).
            struct Vote { uint32_t q, r; uint32_t c; };
            std::vector<Vote> votes;
            // (q, r) -> vote: a dense q_size x n_target table when that is small (whole-structure queries against chains of their own
            // size: 10^5 votes per component, a hash map insertion each was most of the component's time), a hash map otherwise (a motif
            // against a long chain: the dense table is ~1 MB per component for a handful of votes)
            const uint32_t n_tr = (uint32_t)(g_dst[slot + 1] - g_dst[slot]);
            const bool dense_votes = (uint64_t)q_size * n_tr <= (1u << 20) && g.es.size() > 4096;
            if (dense_votes) {
                if (vote_pos.size() < (size_t)q_size * n_tr) vote_pos.assign((size_t)q_size * n_tr, 0u);      // 0 = no vote yet, else 1 + position in votes; cleared below
            }
            std::unordered_map<uint64_t, uint32_t> vote_at;     // (q, r) -> position in votes (first-seen order kept in the vector)
            for (size_t z = 0; z < ce.size(); ++z) {
                const size_t e = ce[z];
                if (z + 16 < ce.size() && edge_k[ce[z + 16]] >= 0) {      // the map's arrays of a whole-structure query are ~1 MB: a miss per edge and array otherwise
                    const int32_t k2 = edge_k[ce[z + 16]];
                    __builtin_prefetch(&qm->idf[k2]); __builtin_prefetch(&qm->qi[k2]); __builtin_prefetch(&qm->qj[k2]);
                }
                if (edge_k[e] < 0) continue;
                const uint32_t k = (uint32_t)edge_k[e];
                sub_idf += qm->idf[k];                      // calculate_subgraph_idf (retrieve.rs:705-719)
                uint32_t qi = qm->qi[k], qj = qm->qj[k], ri = g.w[g.es[e]], rj = g.w[g.et[e]];
                uint32_t pq[2], pr[2];
                if (is_sym(g.eh[e])) { pq[0] = std::min(qi, qj); pq[1] = std::max(qi, qj); pr[0] = std::min(ri, rj); pr[1] = std::max(ri, rj); }
                else { pq[0] = qi; pr[0] = ri; pq[1] = qj; pr[1] = rj; }
                for (int z = 0; z < 2; ++z) {
                    if (dense_votes && pq[z] < q_size && pr[z] < n_tr) {
                        uint32_t &slot_v = vote_pos[(size_t)pq[z] * n_tr + pr[z]];
                        if (!slot_v) { votes.push_back({pq[z], pr[z], 1u}); slot_v = (uint32_t)votes.size(); }
                        else if (votes[slot_v - 1].c < 255) ++votes[slot_v - 1].c;
                        continue;
                    }
                    auto ins = vote_at.emplace(((uint64_t)pq[z] << 32) | pr[z], (uint32_t)votes.size());
                    if (ins.second) votes.push_back({pq[z], pr[z], 1u});
                    else if (votes[ins.first->second].c < 255) ++votes[ins.first->second].c;
                }
            }
            if (dense_votes) for (const Vote &v : votes) if (v.q < q_size && v.r < n_tr) vote_pos[(size_t)v.q * n_tr + v.r] = 0u;
            struct Best { uint32_t q, c, r; };
            std::vector<Best> best;
            std::unordered_map<uint32_t, uint32_t> best_at;
            for (auto &v : votes) {
                auto ins = best_at.emplace(v.q, (uint32_t)best.size());
                if (ins.second) { best.push_back({v.q, v.c, v.r}); continue; }
                Best *bq = &best[ins.first->second];
                if (v.c > bq->c || (v.c == bq->c && v.r < bq->r)) { bq->c = v.c; bq->r = v.r; }
            }
            // greedy assignment in (count descending, query residue ascending) order (retrieve.rs:668-690)
            std::sort(best.begin(), best.end(), [](const Best &x, const Best &y) { return x.c != y.c ? x.c > y.c : x.q < y.q; });
            for (auto &x : best) {
                if (q_idx.size() == cc.size()) break;
                if (std::find(r_idx.begin(), r_idx.end(), x.r) != r_idx.end()) continue;
                q_idx.push_back(x.q); r_idx.push_back(x.r);
            }
            }
            if (plan) {
                if (two_pass) { plan_cache[slot][ci].q_idx = q_idx; plan_cache[slot][ci].r_idx = r_idx; plan_cache[slot][ci].sub_idf = sub_idf; }   // a query residue without a target leaves work for the rescue: its votes come from pairs whose partner is mapped
                bool unmatched = false;
                for (uint64_t pos = 0; pos < NQ && !unmatched; ++pos) unmatched = std::find(q_idx.begin(), q_idx.end(), qm->indices[pos]) == q_idx.end();
                if (unmatched) {
                    ++n_rc;
                    if (two_pass) plan_cache[slot][ci].rc_ord = n_rc;
                    for (uint32_t r : r_idx) if (r < Rt) { o.marks.push_back(r); o.mark_ord.push_back(n_rc); }
                }
                continue;
            }
            // rescue votes of this component: (query residue, target residue) -> pairs whose partner is one of its mapped residues
            if (have_c) {
                for (uint32_t r : r_idx) {
                    if (r >= Rt) continue;
                    for (uint32_t z = by_cj_off[r]; z < by_cj_off[r + 1]; ++z) {
                        const uint32_t vq = pk ? pk_val[c0 + z] >> 16 : by_cj_qi[z], vi = pk ? pk_val[c0 + z] & 0xffffu : by_cj_i[z];
                        if (vq >= q_size || vi >= Rt) continue;
                        const uint32_t ix2 = vq * Rt + vi;
                        if (votes2[ix2]++ == 0) v_touched.push_back(ix2);
                    }
                }
                for (uint32_t ix2 : v_touched) {
                    const uint32_t vq = ix2 / Rt, vr = ix2 % Rt, vc = votes2[ix2];
                    if (r_mx[vq] == 0) q_touched.push_back(vq);
                    if (vc > r_mx[vq]) { r_mx[vq] = vc; r_nmx[vq] = 1; r_arg[vq] = vr; }
                    else if (vc == r_mx[vq]) ++r_nmx[vq];
                    votes2[ix2] = 0;
                }
                v_touched.clear();
            }
            if (have_rows && cached && plan_cache[slot][ci].rc_ord) {      // the same three numbers per query residue, counted by the second scan itself
                const fd_vote_row *vr = vote_rows.data() + row_base[slot] + (uint64_t)(plan_cache[slot][ci].rc_ord - 1) * q_size;
                for (uint32_t vq = 0; vq < q_size; ++vq)
                    if (vr[vq].mx) { r_mx[vq] = vr[vq].mx; r_nmx[vq] = vr[vq].nmx; r_arg[vq] = vr[vq].arg; q_touched.push_back(vq); }
            }
            // residue assignment + rescue (retrieve.rs:430-516)
            std::vector<int32_t> from_hash(NQ, -1), processed(NQ, -1);
            std::vector<uint32_t> qs_sc, rs_sc;
            for (uint64_t pos = 0; pos < NQ; ++pos) {
                uint32_t qi = qm->indices[pos];
                int32_t mapped = -1;
                for (size_t k = 0; k < q_idx.size(); ++k) if (q_idx[k] == qi) { mapped = (int32_t)r_idx[k]; break; }
                if (mapped >= 0) {
                    from_hash[pos] = mapped;
                    auto itp = std::find(rs_sc.begin(), rs_sc.end(), (uint32_t)mapped);
                    if (itp == rs_sc.end()) { processed[pos] = mapped; qs_sc.push_back(qi); rs_sc.push_back((uint32_t)mapped); }
                    else {
                        size_t pp = (size_t)(itp - rs_sc.begin());
                        if (pp < NQ) processed[pp] = -1;   // the reference indexes res_vec with the scanned position
                        processed[pos] = mapped;
                        qs_sc.erase(qs_sc.begin() + pp); rs_sc.erase(rs_sc.begin() + pp);
                        qs_sc.push_back(qi); rs_sc.push_back((uint32_t)mapped);
                    }
                } else {
                    // the target residue with the unique highest vote (>= 2) joins, unless it is taken (retrieve.rs:498-511)
                    if (qi < q_size && r_nmx[qi] == 1 && r_mx[qi] >= 2 && std::find(rs_sc.begin(), rs_sc.end(), r_arg[qi]) == rs_sc.end()) {
                        processed[pos] = (int32_t)r_arg[qi]; qs_sc.push_back(qi); rs_sc.push_back(r_arg[qi]);
                    }
                }
            }
            for (uint32_t vq : q_touched) { r_mx[vq] = 0; r_nmx[vq] = 0; }
            q_touched.clear();
            fd_match_rec rec;
            memset(&rec, 0, sizeof rec);
            rec.cand = (uint32_t)(slot - cand_off[tq]); rec.idf = sub_idf;
            rec.same = from_hash == processed ? 1 : 0;
            o.recs.push_back(rec);
            o.res.insert(o.res.end(), from_hash.begin(), from_hash.end());
            o.res.insert(o.res.end(), processed.begin(), processed.end());
            auto add_problem = [&](const std::vector<uint32_t> &qv, const std::vector<uint32_t> &rv, int which) {
                for (size_t k = 0; k < qv.size(); ++k) {   // [CA, CB] interleaved (retrieve.rs:761-767)
                    for (int z = 0; z < 3; ++z) o.ky.push_back(q_ca[3 * qv[k] + z]);
                    for (int z = 0; z < 3; ++z) o.ky.push_back(q_cb[3 * qv[k] + z]);
                    for (int z = 0; z < 3; ++z) o.kx.push_back(t_ca[3 * rv[k] + z]);
                    for (int z = 0; z < 3; ++z) o.kx.push_back(t_cb[3 * rv[k] + z]);
                }
                o.klen.push_back(2 * qv.size());
                o.pend.push_back({o.recs.size() - 1, which});
            };
            add_problem(q_idx, r_idx, 0);
            if (!rec.same) add_problem(qs_sc, rs_sc, 1);
        }
        if (big_trace) fprintf(stderr, "[slot %llu] components done at %.3f ms (%s; the slot began %.3f ms into the stage)\n", (unsigned long long)slot, t_ms(s_t0, t_now()), plan ? "plan" : "full", t_ms(T2, s_t0));
    };
    if (e_tab_thread.joinable()) e_tab_thread.join();
    auto run_slots = [&](const bool plan) {
        // per-slot ranges of the found triples and candidate pairs (both arrive grouped by slot)
        {
            size_t fpos = 0, cpos = 0;
            for (uint64_t slot = 0; slot < n_cand; ++slot) {
                f_lo[slot] = fpos; c_lo[slot] = cpos;
                while (fpos < nf && found[fpos].cand == slot) ++fpos;
                if (pk_key) while (cpos < nc && (pk_key[cpos] >> 16) == slot) ++cpos;
                else while (cpos < nc && cands && cands[cpos].cand == slot) ++cpos;
            }
            f_lo[n_cand] = fpos; c_lo[n_cand] = cpos;
        }
        std::vector<SlotOut> outs(n_cand);
        const char *th_env = getenv("FDGPU_HOST_THREADS");
        unsigned n_thr = th_env ? (unsigned)atoi(th_env) : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        n_thr = (unsigned)std::min<uint64_t>(std::max(1u, n_thr), std::max<uint64_t>(1, (uint64_t)nf / 2048 + 1));   // small jobs stay on the caller's thread
        n_thr = (unsigned)std::min<uint64_t>(n_thr, std::max<uint64_t>(1, n_cand));                                   // a thread per slot at most (spawning one costs ~0.1 ms)
        std::atomic<uint64_t> next(0);
        const std::function<void()> worker = [&]() {
            for (;;) {
                const uint64_t slot = next.fetch_add(1);
                if (slot >= n_cand) break;
                do_slot(slot, plan, outs[slot]);
            }
        };
        const auto rs_t0 = t_now();
        c->host_pool.run(n_thr, worker);      // the context's standing helper threads (a spawn per thread per pass was ~1 ms of the call)
        if (trace) fprintf(stderr, "[fdgpu_retrieve] %s pass: %u threads over %llu slots took %.3f ms\n", plan ? "plan" : "full", n_thr, (unsigned long long)n_cand, t_ms(rs_t0, t_now()));
        // merge in slot order
        uint64_t tq = 0;
        for (uint64_t slot = 0; slot < n_cand; ++slot) {
            while (slot >= cand_off[tq + 1]) { ++tq; m_off[tq] = recs.size(); r_off[tq] = res.size(); }
            SlotOut &o = outs[slot];
            if (plan) {
                for (size_t z = 0; z < o.marks.size(); ++z) {
                    const uint32_t r = o.marks[z], ord = o.mark_ord[z];
                    any_rescue = true;
                    const uint32_t bit = mask_off[slot] + r;
                    cj_mask[bit >> 5] |= 1u << (bit & 31u);
                    if (ord > 255u || (cj_comp[bit] && cj_comp[bit] != ord)) vote_conflict = true;      // two rescued components share a target residue: host counting
                    else cj_comp[bit] = (uint8_t)ord;
                    slot_nrc[slot] = std::max(slot_nrc[slot], ord);
                }
                continue;
            }
            const size_t rec0 = recs.size();
            recs.insert(recs.end(), o.recs.begin(), o.recs.end());
            res.insert(res.end(), o.res.begin(), o.res.end());
            kx.insert(kx.end(), o.kx.begin(), o.kx.end());
            ky.insert(ky.end(), o.ky.begin(), o.ky.end());
            for (uint64_t l : o.klen) koff.push_back(koff.back() + l);
            for (const Pend &pe : o.pend) pend.push_back({rec0 + pe.rec, pe.which});
        }
        while (tq < n_queries) { ++tq; m_off[tq] = recs.size(); r_off[tq] = res.size(); }
    };
    if (two_pass) {
        run_slots(true);
        if (trace) fprintf(stderr, "[fdgpu_retrieve] plan pass %.3f ms\n", t_ms(T2, t_now()));
        free(cands); cands = nullptr; nc = 0;
        // the rescue's vote tables fit the device (a whole-structure query against its top 20: 20 x 300 x 1,700 counters): count there
        uint64_t n_counters = 0, n_rows = 0;
        std::vector<uint64_t> vt_off(n_cand + 1, 0);
        std::vector<uint32_t> vt_qs(n_cand + 1, 0);
        for (uint64_t k = 0; k < n_cand; ++k) {
            vt_off[k] = n_counters; row_base[k] = n_rows; vt_qs[k] = q_sizes[slot_q[k]];
            n_counters += (uint64_t)slot_nrc[k] * vt_qs[k] * (g_dst[k + 1] - g_dst[k]);
            n_rows += (uint64_t)slot_nrc[k] * vt_qs[k];
        }
        const char *dv_env = getenv("FDGPU_DEVICE_VOTES");      // 0: the rescue counts on the host from the copied candidate pairs (tests)
        const bool dev_votes = any_rescue && !vote_conflict && !(dv_env && dv_env[0] == '0') && n_counters <= (1ull << 28) && n_rows <= (1ull << 24) &&
                               g_total < (1ull << 32);
        if (dev_votes) {
            std::vector<uint64_t> row_off(n_rows);
            std::vector<uint32_t> row_len(n_rows);
            for (uint64_t k = 0, z = 0; k < n_cand; ++k) {
                const uint32_t Rk = (uint32_t)(g_dst[k + 1] - g_dst[k]);
                for (uint64_t rr = 0; rr < (uint64_t)slot_nrc[k] * vt_qs[k]; ++rr, ++z) { row_off[z] = vt_off[k] + rr * Rk; row_len[z] = Rk; }
            }
            vote_rows.resize(std::max<uint64_t>(n_rows, 1));
            fd_vote_plan vp;
            vp.cj_comp = cj_comp.data(); vp.n_bits = g_total; vp.vt_off = vt_off.data(); vp.vt_qs = vt_qs.data(); vp.n_counters = n_counters;
            vp.row_off = row_off.data(); vp.row_len = row_len.data(); vp.n_rows = n_rows; vp.rows = vote_rows.data();
            vp.sd_dist = sd_dist.empty() ? nullptr : sd_dist.data(); vp.sd_qi = sd_qi.empty() ? nullptr : sd_qi.data(); vp.n_sd = sd_dist.size();
            fd_pair_rec *f2 = nullptr; uint64_t nf2 = 0;
            rc = fd_match_pairs_multi(c, db, resname_std, n_queries, mqs.data(), cand, cand_off, p, &f2, &nf2, &cands, &nc, 32u, cj_mask.data(),
                                      mask_off.data(), cj_mask.size(), nullptr, nullptr, &vp, &mp_tab);
            free(f2); free(cands); cands = nullptr; nc = 0;
            if (rc) return rc;
            have_rows = true;
            if (trace) fprintf(stderr, "[fdgpu_retrieve] second scan (device votes) done at %.3f ms (%llu rows)\n", t_ms(T2, t_now()), (unsigned long long)n_rows);
        } else if (any_rescue) {
            fd_pair_rec *f2 = nullptr; uint64_t nf2 = 0;
            rc = fd_match_pairs_multi(c, db, resname_std, n_queries, mqs.data(), cand, cand_off, p, &f2, &nf2, &cands, &nc, 14u, cj_mask.data(),
                                      mask_off.data(), cj_mask.size(), &pk_key, &pk_val, nullptr, &mp_tab);
            free(f2);
            if (rc) return rc;
            if (trace) fprintf(stderr, "[fdgpu_retrieve] second scan done at %.3f ms (cands %llu)\n", t_ms(T2, t_now()), (unsigned long long)nc);
        }
    }
    if (nc && !pk_key) {   // unpacked fallback (>= 2^16 candidate slots): atomic-append order -> grouped by candidate slot (counting sort)
        std::vector<uint64_t> so(n_cand + 2, 0);
        for (uint64_t e = 0; e < nc; ++e) ++so[cands[e].cand + 1];
        for (uint64_t k = 0; k < n_cand; ++k) so[k + 1] += so[k];
        fd_cand_rec *g2 = (fd_cand_rec *)malloc(nc * sizeof(fd_cand_rec));
        if (!g2) return FDGPU_ENOMEM;
        for (uint64_t e = 0; e < nc; ++e) g2[so[cands[e].cand]++] = cands[e];
        free(cands); cands = g2;
    }
    run_slots(false);
    if (trace) fprintf(stderr, "[fdgpu_retrieve] slots done at %.3f ms\n", t_ms(T2, t_now()));
    free(found); found = nullptr; free(cands); cands = nullptr; pk_key = nullptr; pk_val = nullptr;
    const uint64_t nprob = pend.size();
    std::vector<float> rmsd(std::max<uint64_t>(nprob, 1)), rot(std::max<uint64_t>(nprob, 1) * 9), tran(std::max<uint64_t>(nprob, 1) * 3);
    auto T3 = t_now();
    if (nprob && (rc = fdgpu_kabsch_batch(c, kx.data(), ky.data(), koff.data(), nprob, rmsd.data(), rot.data(), tran.data()))) return rc;
    if (partial_fit && nprob) {   // --partial-fit: mappings of more than 3 residues take the LMS fit instead (retrieve.rs:733-746)
        std::vector<uint64_t> which, loff(1, 0);
        std::vector<float> lx, ly;
        for (uint64_t k = 0; k < nprob; ++k) {
            const uint64_t a = koff[k], b = koff[k + 1];
            if (b - a <= 6) continue;                       // [CA, CB] per residue: index1.len() <= 3 keeps Kabsch
            which.push_back(k);
            lx.insert(lx.end(), kx.begin() + 3 * a, kx.begin() + 3 * b);
            ly.insert(ly.end(), ky.begin() + 3 * a, ky.begin() + 3 * b);
            loff.push_back(loff.back() + (b - a));
        }
        if (!which.empty()) {
            std::vector<float> lr(which.size()), lrot(which.size() * 9), ltr(which.size() * 3);
            if ((rc = fdgpu_lms_qcp_batch(c, lx.data(), ly.data(), loff.data(), which.size(), lr.data(), lrot.data(), ltr.data(), nullptr, nullptr)))
                return rc;
            for (size_t z = 0; z < which.size(); ++z) {
                rmsd[which[z]] = lr[z];
                memcpy(&rot[9 * which[z]], &lrot[9 * z], 36);
                memcpy(&tran[3 * which[z]], &ltr[3 * z], 12);
            }
        }
    }
    if (trace) fprintf(stderr, "[fdgpu_retrieve] match_pairs %.3f ms (found %llu, cands %llu), gather %.3f, graph/vote %.3f, kabsch(%llu) %.3f\n", t_ms(T0, T1),
                       (unsigned long long)nf, (unsigned long long)nc, t_ms(T1, T2), t_ms(T2, T3), (unsigned long long)nprob, t_ms(T3, t_now()));
    // similarity metrics of every superposition on the device (k_metrics), after a possible --partial-fit override of rot / tran
    std::vector<float> mets(std::max<uint64_t>(nprob, 1) * 5);
    if (nprob && (rc = fdgpu_metrics_batch(c, ky.data(), kx.data(), koff.data(), nprob, rot.data(), tran.data(), mets.data()))) return rc;
    for (uint64_t k = 0; k < nprob; ++k) {
        fd_match_rec &r = recs[pend[k].rec];
        const bool is_out = pend[k].which == 1 || r.same;   // the superposition the match reports
        if (pend[k].which == 0) {
            r.rmsd_from_hash = rmsd[k]; memcpy(r.rot_from_hash, &rot[9 * k], 36); memcpy(r.tran_from_hash, &tran[3 * k], 12);
            memcpy(r.metrics_from_hash, &mets[5 * k], 20);
        }
        if (is_out) {
            r.rmsd = rmsd[k]; memcpy(r.rot, &rot[9 * k], 36); memcpy(r.tran, &tran[3 * k], 12);
            memcpy(r.metrics, &mets[5 * k], 20);
        }
    }
    fd_match_rec *om = (fd_match_rec *)malloc(std::max<size_t>(recs.size(), 1) * sizeof(fd_match_rec));
    int32_t *orr = (int32_t *)malloc(std::max<size_t>(res.size(), 1) * sizeof(int32_t));
    uint64_t *omo = (uint64_t *)malloc((n_queries + 1) * 8), *oro = (uint64_t *)malloc((n_queries + 1) * 8);
    if (!om || !orr || !omo || !oro) { free(om); free(orr); free(omo); free(oro); return FDGPU_ENOMEM; }
    if (!recs.empty()) memcpy(om, recs.data(), recs.size() * sizeof(fd_match_rec));
    if (!res.empty()) memcpy(orr, res.data(), res.size() * sizeof(int32_t));
    memcpy(omo, m_off.data(), (n_queries + 1) * 8);
    memcpy(oro, r_off.data(), (n_queries + 1) * 8);
    *matches = om; *match_off = omo; *residues = orr; *res_off = oro;
    return FDGPU_OK;
}

// The body of query_pdb.rs:376-452 for a batch of queries in ONE call: query maps -> scoring + ranked top_n -> retrieval of every query's first
// match_top candidates.  Same results as fdgpu_make_query_map_batch + fdgpu_count_query_maps_top + fdgpu_retrieve_batch called one after the other
// (tests compare them bit for bit); what the call adds is overlap the three blocking calls cannot have: the retrieval's query tables are built
// while the scoring kernels run, only the candidates' ids (match_top per query) are waited for before the pair scan starts, and the ranked records
// (n_queries x top_n x 20 bytes) cross the bus on a second stream while the retrieval's kernels run.
extern "C" int fdgpu_query_batch(fdgpu_ctx *c, const fdgpu_index *ix, const fdgpu_batch *db, const uint8_t *resname_std, const fdgpu_batch *qb, uint64_t n_queries,
                                 const uint32_t *q_struct, const uint64_t *q_off, const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs,
                                 const float *dist_thr, uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p, float total_structures,
                                 const float *penalty, uint32_t top_n, uint32_t match_top, float ca_distance_cutoff, uint32_t node_count, fd_query_map **maps,
                                 fd_count_rec **recs, uint64_t **rec_off, fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off) { FD_LOCK(c);
    if (!c || !ix || !db || !qb || !p || !maps || !recs || !rec_off || !matches || !match_off || !residues || !res_off || !q_off || (n_queries && !q_struct)) return FDGPU_EINVAL;
    *recs = nullptr; *rec_off = nullptr; *matches = nullptr; *match_off = nullptr; *residues = nullptr; *res_off = nullptr;
    int rc = fdgpu_make_query_map_batch(c, qb, n_queries, q_struct, q_off, q_index, subs, n_subs, dist_thr, n_dist, angle_thr_deg, n_angle, p, ix, total_structures, maps);
    if (rc) return rc;
    auto drop_maps = [&]() { for (uint64_t t = 0; t < n_queries; ++t) { fdgpu_query_map_free(maps[t]); maps[t] = nullptr; } };
    fd_rb_prep prep;
    bool prep_done = false;
    const std::function<void()> build_prep = [&]() { fd_rb_prepare(n_queries, maps, ca_distance_cutoff, prep); prep_done = true; };
    fd_cq_dev_out D;
    D.while_running = &build_prep;
    D.head_n = std::min(match_top, top_n);
    fd_count_rec *rr = nullptr;
    uint64_t *roff = nullptr;
    rc = fd_count_query_maps_top_impl(c, ix, n_queries, maps, penalty, total_structures, top_n, &rr, &roff, &D);
    if (!rc && D.got && D.overflow) {      // more ties at a cut-off than the device selection holds: the compacting path, host records
        fdgpu_free(rr); free(roff); rr = nullptr; roff = nullptr;
        rc = fd_count_query_maps_top_impl(c, ix, n_queries, maps, penalty, total_structures, top_n, &rr, &roff, nullptr, /*allow_dense=*/false);
        D.got = false;
    }
    if (rc) { drop_maps(); return rc; }
    if (!prep_done) build_prep();
    std::vector<uint32_t> cand;
    std::vector<uint64_t> cand_off(n_queries + 1, 0);
    const uint64_t first = ix->first_id;
    bool side_copy = false;
    if (D.got) {
        // the candidates: the first match_top records of every query's ranking (a strided copy of 20 x match_top bytes per query)
        hipStream_t st = c->stream;
        const uint32_t mt = std::min(match_top, top_n);
        std::vector<fd_count_rec> head_v;
        const size_t head_bytes = (size_t)n_queries * std::max<uint32_t>(mt, 1) * sizeof(fd_count_rec);
        const fd_count_rec *head = D.head;       // came with the selection's state (one wait)
        hipError_t e = hipSuccess;
        if (!head) {
            fd_count_rec *h2 = (fd_count_rec *)c->host_pinned(3, head_bytes);
            if (!h2) { head_v.resize((size_t)n_queries * std::max<uint32_t>(mt, 1)); h2 = head_v.data(); }
            if (mt && n_queries)
                e = hipMemcpy2DAsync(h2, (size_t)mt * sizeof(fd_count_rec), D.recs, (size_t)top_n * sizeof(fd_count_rec), (size_t)mt * sizeof(fd_count_rec), n_queries,
                                     hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            head = h2;
        }
        // the full ranking: straight into the caller's (page-locked, pooled) array on the second stream, closed up after the retrieval
        rr = (fd_count_rec *)fd_out_alloc(std::max<uint64_t>((uint64_t)n_queries * top_n, 1) * sizeof(fd_count_rec), true);
        roff = (uint64_t *)calloc(n_queries + 1, 8);
        if (e == hipSuccess && (!rr || !roff)) { fdgpu_free(rr); free(roff); drop_maps(); return FDGPU_ENOMEM; }
        if (e == hipSuccess && !c->side_stream) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
        if (e == hipSuccess && n_queries && top_n) {
            e = hipMemcpyAsync(rr, D.recs, (size_t)n_queries * top_n * sizeof(fd_count_rec), hipMemcpyDeviceToHost, c->side_stream);
            side_copy = e == hipSuccess;
        }
        if (e != hipSuccess) {
            if (side_copy) (void)hipStreamSynchronize(c->side_stream);
            fdgpu_free(rr); free(roff); drop_maps();
            c->err = std::string("query_batch: ") + hipGetErrorString(e);
            return FDGPU_EHIP;
        }
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint32_t m = std::min<uint32_t>(std::min<uint32_t>(D.counts[t], top_n), mt);
            for (uint32_t k = 0; k < m; ++k) cand.push_back((uint32_t)(head[(size_t)t * mt + k].nid - first));
            cand_off[t + 1] = cand.size();
        }
    } else {
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint64_t m = std::min<uint64_t>(roff[t + 1] - roff[t], match_top);
            for (uint64_t k = 0; k < m; ++k) cand.push_back((uint32_t)(rr[roff[t] + k].nid - first));
            cand_off[t + 1] = cand.size();
        }
    }
    rc = fd_retrieve_batch_impl(c, db, resname_std, n_queries, cand.data(), cand_off.data(), maps, qb, q_struct, p, ca_distance_cutoff, node_count, 0, matches, match_off,
                                residues, res_off, &prep);
    if (side_copy) {
        // the side-stream copy read WS_TILE_HO while the retrieval ran: correct only as long as no retrieval stage grows (= frees) or writes that buffer
        if (c->ws[WS_TILE_HO].p != D.recs && !rc) { c->err = "query_batch: a retrieval stage re-allocated the ranking's buffer under its copy"; rc = FDGPU_EHIP; }
        const hipError_t e = hipStreamSynchronize(c->side_stream);
        if (e != hipSuccess && !rc) { c->err = std::string("query_batch records: ") + hipGetErrorString(e); rc = FDGPU_EHIP; }
    }
    if (!rc && D.got) {       // close the fixed-stride ranking up in place (forward moves: the write position never passes the read position)
        uint64_t w = 0;
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint64_t m = std::min<uint32_t>(D.counts[t], top_n);
            roff[t] = w;
            if (m && w != t * top_n) memmove(rr + w, rr + (size_t)t * top_n, (size_t)m * sizeof(fd_count_rec));
            w += m;
        }
        roff[n_queries] = w;
    }
    if (rc) {
        fdgpu_free(rr); free(roff); drop_maps();
        fdgpu_free(*matches); fdgpu_free(*residues); free(*match_off); free(*res_off);
        *matches = nullptr; *match_off = nullptr; *residues = nullptr; *res_off = nullptr;
        return rc;
    }
    *recs = rr; *rec_off = roff;
    return FDGPU_OK;
}

// one query = structure 0 of qb
extern "C" int fdgpu_retrieve(fdgpu_ctx *c, const fdgpu_batch *db, const uint8_t *resname_std, const uint32_t *cand, uint64_t n_cand,
                              const fd_query_map *qm, const fdgpu_batch *qb, const fd_hash_params *p, float ca_distance_cutoff,
                              uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches, uint64_t *n_matches, int32_t **residues) { FD_LOCK(c);
    if (!c || !db || !qm || !qb || !p || !matches || !n_matches || !residues) return FDGPU_EINVAL;
    *matches = nullptr; *n_matches = 0; *residues = nullptr;
    const uint64_t off[2] = {0, n_cand};
    const uint32_t s0 = 0;
    uint64_t *mo = nullptr, *ro = nullptr;
    int rc = fdgpu_retrieve_batch(c, db, resname_std, 1, cand, off, &qm, qb, &s0, p, ca_distance_cutoff, node_count, partial_fit, matches, &mo, residues, &ro);
    if (rc) return rc;
    *n_matches = mo[1];
    free(mo); free(ro);
    return FDGPU_OK;
}


// ------------------------------------------------------------------------------------------ sub-index merge
// Index build shards by structure (one sub-index per GPU / per batch, contiguous ascending id ranges).  The
// reference's single on-disk index is the per-hash concatenation of the shards' posting lists in shard order;
// only the first varint of every continuation (an absolute id in its own sub-index) must be re-encoded as the
// delta from the last id of the list so far (src/index/indextable.rs:171-202 semantics).  Host-side, streaming.
static inline uint64_t fd_decode_last(const uint8_t *b, uint64_t n, uint64_t *first_len, uint64_t *first_val) {
    uint64_t acc = 0, prev = 0, cur = 0;
    unsigned shift = 0;
    bool first = true;
    for (uint64_t k = 0; k < n; ++k) {
        acc |= (uint64_t)(b[k] & 0x7f) << shift;
        if (b[k] & 0x80) { shift += 7; continue; }
        if (first) { cur = acc; *first_len = k + 1; *first_val = acc; first = false; }
        else cur = prev + acc;
        prev = cur; acc = 0; shift = 0;
    }
    return cur;
}
static inline unsigned fd_put_varint(uint64_t v, uint8_t *out) {
    unsigned n = 0;
    do { uint8_t byte = v & 0x7f; v >>= 7; out[n++] = byte | (v ? 0x80 : 0); } while (v);
    return n;
}

extern "C" int fdgpu_merge_subindices(uint64_t n_parts, const uint8_t *const *values, const uint32_t *const *hashes,
                                      const uint64_t *const *offsets, const uint64_t *n_hashes, uint8_t **out_value, uint64_t *out_value_len,
                                      uint32_t **out_hashes, uint64_t **out_offsets, uint64_t *out_n_hashes) {
    if (!values || !hashes || !offsets || !n_hashes || !out_value || !out_value_len || !out_hashes || !out_offsets || !out_n_hashes) return FDGPU_EINVAL;
    uint64_t tot_h = 0, tot_v = 0;
    for (uint64_t p = 0; p < n_parts; ++p) { tot_h += n_hashes[p]; tot_v += offsets[p][n_hashes[p]]; }
    uint8_t *V = (uint8_t *)malloc(std::max<uint64_t>(tot_v + 8, 8));          // re-based first deltas are never longer than the absolute ids
    uint32_t *H = (uint32_t *)malloc(std::max<uint64_t>(tot_h, 1) * 4);
    uint64_t *O = (uint64_t *)malloc((tot_h + 1) * 8);
    if (!V || !H || !O) { free(V); free(H); free(O); return FDGPU_ENOMEM; }
    std::vector<uint64_t> pos(n_parts, 0);
    uint64_t nh = 0, nv = 0;
    O[0] = 0;
    for (;;) {
        bool any = false;
        uint32_t hmin = 0;
        for (uint64_t p = 0; p < n_parts; ++p)
            if (pos[p] < n_hashes[p] && (!any || hashes[p][pos[p]] < hmin)) { hmin = hashes[p][pos[p]]; any = true; }
        if (!any) break;
        bool have_last = false;
        uint64_t last = 0;
        for (uint64_t p = 0; p < n_parts; ++p) {   // shard order = ascending id ranges
            if (pos[p] >= n_hashes[p] || hashes[p][pos[p]] != hmin) continue;
            const uint8_t *b = values[p] + offsets[p][pos[p]];
            uint64_t len = offsets[p][pos[p] + 1] - offsets[p][pos[p]];
            uint64_t flen = 0, fval = 0;
            uint64_t lst = fd_decode_last(b, len, &flen, &fval);
            if (!have_last) { memcpy(V + nv, b, len); nv += len; }
            else {
                if (fval <= last) { free(V); free(H); free(O); return FDGPU_EINVAL; }   // id ranges must ascend across parts
                nv += fd_put_varint(fval - last, V + nv);
                memcpy(V + nv, b + flen, len - flen); nv += len - flen;
            }
            last = lst; have_last = true;
            ++pos[p];
        }
        H[nh] = hmin; O[++nh] = nv;
    }
    *out_value = V; *out_value_len = nv; *out_hashes = H; *out_offsets = O; *out_n_hashes = nh;
    return FDGPU_OK;
}


// ---- analyze: hypergeometric enrichment of encodings (src/controller/summary.rs:543-628) --------------------------------------------
// p[k] = P(X >= x) for X ~ Hypergeometric(N = total_bg + total_query, K = bg[k] + query[k], n = total_query), x = query[k]: the sum of
// the log-space pmf from x to min(n, K) with the reference's log-factorial (exact sum of ln below 20, Stirling's formula above,
// summary.rs:616-628), clamped to 1.  The terms fall monotonically behind the mode, so the walk stops once a term no longer changes
// the f64 sum — the value is the one the full loop gives.  Host threads over the encodings.
namespace {
inline double fd_log_factorial(uint64_t n) {
    if (n <= 1) return 0.0;
    if (n < 20) { double s = 0.0; for (uint64_t i = 2; i <= n; ++i) s += log((double)i); return s; }
    const double x = (double)n;
    return x * log(x) - x + 0.5 * log(2.0 * 3.14159265358979323846 * x);
}
inline double fd_log_binomial(uint64_t n, uint64_t k) {
    if (k > n) return -INFINITY;
    if (k == 0 || k == n) return 0.0;
    return fd_log_factorial(n) - fd_log_factorial(k) - fd_log_factorial(n - k);
}
}
extern "C" int fdgpu_hypergeom_enrichment(const uint64_t *query_count, const uint64_t *bg_count, uint64_t n_enc, uint64_t total_query, uint64_t total_bg,
                                          uint32_t n_threads, double *p_value) {
    if (n_enc && (!query_count || !bg_count || !p_value)) return FDGPU_EINVAL;
    const uint64_t n = total_query, N = total_bg + total_query;
    const double log_den = fd_log_binomial(N, n);
    auto work = [&](uint64_t a, uint64_t b) {
        for (uint64_t e = a; e < b; ++e) {
            const uint64_t x = query_count[e], K = bg_count[e] + query_count[e], max_x = n < K ? n : K;
            const double mode = (double)(n + 1) * (double)(K + 1) / (double)(N + 2);
            double p = 0.0;
            for (uint64_t i = x; i <= max_x; ++i) {
                const double t = exp(fd_log_binomial(K, i) + fd_log_binomial(N - K, n - i) - log_den);
                const double q = p + t;
                if (q == p && (double)i > mode) break;
                p = q;
            }
            p_value[e] = p < 1.0 ? p : 1.0;
        }
    };
    const uint32_t T = std::max<uint32_t>(1, std::min<uint32_t>(n_threads ? n_threads : 1, 256));
    if (T == 1 || n_enc < 64) { work(0, n_enc); return FDGPU_OK; }
    std::vector<std::thread> th;
    const uint64_t per = (n_enc + T - 1) / T;
    for (uint32_t t = 0; t < T; ++t) { const uint64_t a = (uint64_t)t * per, b = std::min<uint64_t>(n_enc, a + per); if (a < b) th.emplace_back(work, a, b); }
    for (auto &x : th) x.join();
    return FDGPU_OK;
}
