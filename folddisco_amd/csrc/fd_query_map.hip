// fd_query_map.hip — make_query_map behind the C ABI (src/controller/query.rs:208-329, helpers :53-206): pair features and hashes on the GPU
// (k_pair_features*, k_hash_features, the device chain k_qm_expand_hash / k_qm_dedupe / posting-length lookup for motif batches and whole-structure
// queries), expansion / substitution / first-insert-wins bookkeeping and the assembly of the fd_query_map blocks on the host.
// The reference is compiled code, so this glue is C++ behind the same C ABI; it contains no f32 arithmetic that decides a hash bit (that all
// happens in the kernels) except the query expansion's feature +- delta adds, which are single IEEE f32 adds exactly like the reference's.
// (Split from fd_host_query.hip, which keeps the retrieval glue.)
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

// ------------------------------------------------------------------------------------------ kernels
// get_single_feature (src/controller/feature.rs:11-24, 84-99) for explicit residue pairs of one structure
__global__ void k_pair_features(fd_batch_view B, uint32_t s, const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj, uint32_t n,
                                float cutoff, uint32_t type, float *__restrict__ feat /*[n][7]*/, uint8_t *__restrict__ valid) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    // s == 0xffffffff: pi / pj are residue indices of the whole batch (pairs of many structures in one launch)
    const uint32_t r0 = s == 0xffffffffu ? 0u : B.res_off[s], r1 = s == 0xffffffffu ? B.res_off[B.n_struct] : B.res_off[s + 1];
    uint32_t i = r0 + pi[k], j = r0 + pj[k];
    bool ok = i < r1 && j < r1 && i != j && B.hash_ok[i] && B.hash_ok[j];
    fd_feature f = {0, 0, 0, 0, 0};
    if (ok) {
        fd_v3 ca1 = fd_load3(B.ca_xyz, i), ca2 = fd_load3(B.ca_xyz, j);
        float d = fd_dist(ca1, ca2);
        if (d > cutoff) ok = false;
        else f = fd_pair_feature(fd_load3(B.n_xyz, i), ca1, fd_load3(B.cb_xyz, i), fd_load3(B.n_xyz, j), ca2, fd_load3(B.cb_xyz, j));
    }
    valid[k] = ok ? 1 : 0;
    float *o = feat + 7ull * k;
    o[0] = ok ? (float)B.aa[i] : 0.f; o[1] = ok ? (float)B.aa[j] : 0.f;
    o[2] = f.ca_dist; o[3] = f.cb_dist; o[4] = type == FD_HASH_PDBMOTIF ? fd_to_degrees(f.angle) : f.angle; o[5] = f.tor1; o[6] = f.tor2;
}
// the same for every encoding, in the query map's own record: [0..9) the feature container of get_single_feature, [9] the CA
// distance and [10], [11] the two residue types (get_list_amino_acids_and_distances, structure/core.rs:462-477 — what the
// observed-distance map holds whatever the encoding puts into the container)
#define FD_QF 12
__global__ void k_pair_features12(fd_batch_view B, const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj, uint32_t n, float cutoff,
                                  uint32_t type, float *__restrict__ feat /*[n][FD_QF]*/, uint8_t *__restrict__ valid) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t R = B.res_off[B.n_struct];
    const uint32_t i = pi[k], j = pj[k];      // residue indices of the whole batch
    float *o = feat + (uint64_t)FD_QF * k;
    for (int z = 0; z < FD_QF; ++z) o[z] = 0.f;
    bool ok = i < R && j < R && i != j && B.aa[i] != 255 && B.aa[j] != 255;
    if (ok) {
        o[9] = fd_dist(fd_load3(B.ca_xyz, i), fd_load3(B.ca_xyz, j)); o[10] = (float)B.aa[i]; o[11] = (float)B.aa[j];
        if (fd_own_descriptor(type)) {
            uint32_t lo = 0, hi = B.n_struct;                  // structure of residue i: res_off[lo] <= i < res_off[lo + 1]
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (B.res_off[mid] <= i) lo = mid; else hi = mid; }
            const uint32_t r0 = B.res_off[lo], r1 = B.res_off[lo + 1];
            float f[FD_NFEAT];
            ok = j >= r0 && j < r1 && fd_feature_other(type, B, r0, r1, i, j, cutoff, f);
            if (ok) for (int z = 0; z < FD_NFEAT; ++z) o[z] = f[z];
        } else {
            ok = B.hash_ok[i] && B.hash_ok[j] && !(o[9] > cutoff);
            if (ok) {
                const fd_feature f = fd_pair_feature(fd_load3(B.n_xyz, i), fd_load3(B.ca_xyz, i), fd_load3(B.cb_xyz, i), fd_load3(B.n_xyz, j),
                                                     fd_load3(B.ca_xyz, j), fd_load3(B.cb_xyz, j));
                o[0] = o[10]; o[1] = o[11]; o[2] = f.ca_dist; o[3] = f.cb_dist; o[4] = type == FD_HASH_PDBMOTIF ? fd_to_degrees(f.angle) : f.angle;
                o[5] = f.tor1; o[6] = f.tor2;
            }
        }
    }
    valid[k] = ok ? 1 : 0;
}
// GeometricHash::perfect_hash (src/geometry/core.rs:213-246 -> pdb_tr.rs:21-75 and the other encodings) on explicit feature vectors
__global__ void k_hash_features(const float *__restrict__ feat, uint64_t n, uint32_t stride, fd_quant q, uint32_t *__restrict__ out) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float *f = feat + (uint64_t)stride * k;
    if (fd_own_descriptor(q.type)) { out[k] = fd_hash_other(q.type, f, q); return; }
    fd_feature ft = {f[2], f[3], f[4], f[5], f[6]};
    out[k] = fd_hash_enc_feat(fd_sat_u32(f[0]), fd_sat_u32(f[1]), ft, q);
}

// Large queries without substitutions (whole-structure queries: ~10^5 pairs, 3.7 candidates each): expand_and_insert (query.rs:179-206) per valid
// pair on the device — the observed container, then near / far per threshold and field in the reference's order, with its f32 restore drift —
// and the hash of every candidate; nothing but the hashes leaves the kernel.  out[v * per_pair + c], c in insertion order.
struct qm_expand_par {
    int di[2], ndi, ai[7], nai;
    float dthr[8], athr[8];
    uint32_t n_dist, n_angle, per_pair;
};
__device__ __forceinline__ uint32_t qm_hash_of(const float *f, const fd_quant &q) {
    if (fd_own_descriptor(q.type)) return fd_hash_other(q.type, f, q);
    fd_feature ft = {f[2], f[3], f[4], f[5], f[6]};
    return fd_hash_enc_feat(fd_sat_u32(f[0]), fd_sat_u32(f[1]), ft, q);
}
__global__ void k_qm_expand_hash(const float *__restrict__ feat, const uint32_t *__restrict__ vp, uint32_t n_valid, qm_expand_par P, fd_quant q,
                                 uint32_t *__restrict__ out) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_valid) return;
    const float *f = feat + (uint64_t)FD_QF * vp[v];
    float near[FD_QF], far[FD_QF];
    for (int z = 0; z < FD_QF; ++z) { near[z] = f[z]; far[z] = f[z]; }
    uint32_t *o = out + (uint64_t)v * P.per_pair;
    *o++ = qm_hash_of(near, q);
    for (int grp = 0; grp < 2; ++grp) {
        const int *idxs = grp ? P.ai : P.di;
        const int n_idx = grp ? P.nai : P.ndi;
        const float *thr = grp ? P.athr : P.dthr;
        const uint32_t n_thr = grp ? P.n_angle : P.n_dist;
        for (uint32_t z2 = 0; z2 < n_thr; ++z2)
            for (int z = 0; z < n_idx; ++z) {
                const int idx = idxs[z];
                // the container field by a run-time index: a select chain over the twelve registers instead of private memory
                float nv = 0.f, fv = 0.f;
#pragma unroll
                for (int w = 0; w < FD_QF; ++w) { nv = w == idx ? near[w] : nv; fv = w == idx ? far[w] : fv; }
                const float n1 = nv - thr[z2], f1 = fv + thr[z2];
#pragma unroll
                for (int w = 0; w < FD_QF; ++w) { near[w] = w == idx ? n1 : near[w]; far[w] = w == idx ? f1 : far[w]; }
                *o++ = qm_hash_of(near, q);
                *o++ = qm_hash_of(far, q);
                const float n2 = n1 + thr[z2], f2 = f1 - thr[z2];      // the reference restores with += / -= (f32, not an exact inverse): keep the drift
#pragma unroll
                for (int w = 0; w < FD_QF; ++w) { near[w] = w == idx ? n2 : near[w]; far[w] = w == idx ? f2 : far[w]; }
            }
    }
}
// first insertion wins: after a stable sort of (hash, insertion position) by hash the first element of every run is the hash's earliest insertion
__global__ void k_qm_first(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, uint64_t n, uint8_t *__restrict__ first) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && (k == 0 || key[k] != key[k - 1])) first[val[k]] = 1;
}
__global__ void k_qm_iota(uint32_t *__restrict__ v, uint64_t n) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) v[k] = (uint32_t)k;
}
// first insertion wins, per query of a motif batch: one workgroup per query, its candidates' hashes through an open-addressing table in LDS
// that keeps the SMALLEST insertion position of every hash (a hash always walks the same probe path and slots never empty, so all
// copies of a hash meet in one slot); a candidate is kept when it holds its hash's slot.  Queries of up to QM_DD_MAX candidates.
#define QM_DD_MAX 2048u
#define QM_DD_SLOTS 4096u
__global__ __launch_bounds__(256) void k_qm_dedupe(const uint32_t *__restrict__ hash, const uint64_t *__restrict__ cand_off, uint8_t *__restrict__ first) {
    __shared__ unsigned long long tab[QM_DD_SLOTS];
    const uint64_t c0 = cand_off[blockIdx.x], n = cand_off[blockIdx.x + 1] - c0;
    for (uint32_t k = threadIdx.x; k < QM_DD_SLOTS; k += 256) tab[k] = ~0ull;
    __syncthreads();
    for (uint32_t pos = threadIdx.x; pos < n; pos += 256) {
        const uint32_t h = hash[c0 + pos];
        const unsigned long long mine = ((unsigned long long)h << 32) | pos;
        uint32_t at = (h * 2654435761u) >> 20;      // 12 bits
        for (;;) {
            const unsigned long long old = atomicCAS(&tab[at], ~0ull, mine);
            if (old == ~0ull) break;
            if ((uint32_t)(old >> 32) == h) { atomicMin(&tab[at], mine); break; }
            at = (at + 1u) & (QM_DD_SLOTS - 1u);
        }
    }
    __syncthreads();
    for (uint32_t pos = threadIdx.x; pos < n; pos += 256) {
        const uint32_t h = hash[c0 + pos];
        uint32_t at = (h * 2654435761u) >> 20;
        while ((uint32_t)(tab[at] >> 32) != h) at = (at + 1u) & (QM_DD_SLOTS - 1u);
        first[c0 + pos] = (uint32_t)tab[at] == pos ? 1 : 0;
    }
}
__global__ void k_qm_keep(const uint8_t *__restrict__ first, const uint64_t *__restrict__ pos, uint64_t n, uint32_t *__restrict__ keep) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && first[k]) keep[pos[k]] = (uint32_t)k;
}

extern "C" int fdgpu_pair_features(fdgpu_ctx *c, const fdgpu_batch *b, uint64_t s, const uint32_t *pi, const uint32_t *pj, uint64_t n,
                                   const fd_hash_params *p, float *features, uint8_t *valid) { FD_LOCK(c);
    if (!c || !b || !p || (s >= b->n_struct && s != 0xffffffffull) || (n && (!pi || !pj || !features || !valid))) return FDGPU_EINVAL;
    CHECK_TYPE(c, p);
    if (fd_own_descriptor(p->hash_type)) { c->err = "pair_features: seven-float records hold the PDBTrRosetta descriptor only"; return FDGPU_EINVAL; }
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    HIPCHK(c, c->ws[WS_MISC0].ensure(n * 4));
    HIPCHK(c, c->ws[WS_MISC1].ensure(n * 4));
    HIPCHK(c, c->ws[WS_MISC2].ensure(n * 28));
    HIPCHK(c, c->ws[WS_MISC3].ensure(n));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, pi, n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC1].p, pj, n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pair_features, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, b->view(), (uint32_t)s, c->ws[WS_MISC0].as<uint32_t>(),
                       c->ws[WS_MISC1].as<uint32_t>(), (uint32_t)n, p->dist_cutoff, p->hash_type, c->ws[WS_MISC2].as<float>(), c->ws[WS_MISC3].as<uint8_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(features, c->ws[WS_MISC2].p, n * 28, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(valid, c->ws[WS_MISC3].p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

static int hash_features_q(fdgpu_ctx *c, const float *features, uint64_t n, fd_quant q, uint32_t *hashes, uint32_t stride = 7);
// the single configuration (nbin_dist, nbin_angle; either count 0 -> the encoding's defaults); multiple_bins is not consulted
extern "C" int fdgpu_hash_features(fdgpu_ctx *c, const float *features, uint64_t n, const fd_hash_params *p, uint32_t *hashes) { FD_LOCK(c);
    if (!c || !p || (n && (!features || !hashes))) return FDGPU_EINVAL;
    CHECK_TYPE(c, p);
    if (fd_own_descriptor(p->hash_type)) { c->err = "hash_features: seven-float records hold the PDBTrRosetta descriptor only"; return FDGPU_EINVAL; }
    return hash_features_q(c, features, n, fd_make_consts(p).q, hashes);
}
// internal form of fdgpu_pair_features: batch-wide residue indices, FD_QF floats per pair, every encoding
// land_out != null: the features and flags stay in the context's page-locked block (*land_out = [n x FD_QF floats | n flags]) when it can be had
// — features / valid are then untouched; else they are copied into features / valid
static int pair_features12(fdgpu_ctx *c, const fdgpu_batch *b, const uint32_t *pi, const uint32_t *pj, uint64_t n, const fd_hash_params *p,
                           float *features, uint8_t *valid, uint8_t **land_out = nullptr) {
    if (land_out) *land_out = nullptr;
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    // one block up ([pi | pj], packed in page-locked memory when it can be had) and one block down ([features | flags]): two copy launches per call, not four
    HIPCHK(c, c->ws[WS_MISC0].ensure(2 * n * 4));
    HIPCHK(c, c->ws[WS_MISC2].ensure(n * 4 * FD_QF + n));
    uint32_t *d_pi = c->ws[WS_MISC0].as<uint32_t>(), *d_pj = d_pi + n;
    uint8_t *d_valid = c->ws[WS_MISC2].as<uint8_t>() + n * 4 * FD_QF;
    uint32_t *up = (uint32_t *)c->host_pinned(5, 2 * n * 4);
    if (up) {
        memcpy(up, pi, n * 4); memcpy(up + n, pj, n * 4);
        HIPCHK(c, hipMemcpyAsync(d_pi, up, 2 * n * 4, hipMemcpyHostToDevice, st));
    } else {
        HIPCHK(c, hipMemcpyAsync(d_pi, pi, n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(d_pj, pj, n * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_pair_features12, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, b->view(), d_pi, d_pj, (uint32_t)n, p->dist_cutoff, p->hash_type,
                       c->ws[WS_MISC2].as<float>(), d_valid);
    HIPCHK(c, hipGetLastError());
    // page-locked landing block (slot 4, read in place by the caller until its next call): the copy does not stage, one wait, no second copy
    uint8_t *land = land_out ? (uint8_t *)c->host_pinned(4, n * 4 * FD_QF + n) : nullptr;
    if (land) {
        HIPCHK(c, hipMemcpyAsync(land, c->ws[WS_MISC2].p, n * 4 * FD_QF + n, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        *land_out = land;
        return FDGPU_OK;
    }
    if (land_out) return FDGPU_OK;      // no page-locked block: the caller asks again with its own arrays
    HIPCHK(c, hipMemcpyAsync(features, c->ws[WS_MISC2].p, n * 4 * FD_QF, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}
static int hash_features_q(fdgpu_ctx *c, const float *features, uint64_t n, fd_quant q, uint32_t *hashes, uint32_t stride) {
    if (!n) return FDGPU_OK;
    hipStream_t st = c->stream;
    fd_hash_consts C;
    C.q = q;
    HIPCHK(c, c->ws[WS_MISC2].ensure(n * 4 * stride));
    HIPCHK(c, c->ws[WS_MISC0].ensure(n * 4));
    HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC2].p, features, n * 4 * stride, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_hash_features, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, c->ws[WS_MISC2].as<float>(), n, stride, C.q, c->ws[WS_MISC0].as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hashes, c->ws[WS_MISC0].p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return FDGPU_OK;
}

// ------------------------------------------------------------------------------------------ make_query_map
template <typename T> static T *dup_vec(const std::vector<T> &v) {
    T *p = (T *)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

extern "C" void fdgpu_query_map_free(fd_query_map *m) {
    if (!m) return;
    if (m->arena_bytes) { free(m); return; }       // the map and its arrays are one block (fdgpu_make_query_map_batch)
    free(m->hash); free(m->qi); free(m->qj); free(m->is_primary); free(m->idf); free(m->indices); free(m->primary_hash);
    free(m->aad_aa1); free(m->aad_aa2); free(m->aad_dist); free(m->aad_qi);
    free(m->post_len); free(m->post_seg); free(m->post_kidx);
    free(m);
}

// make_query_map (src/controller/query.rs:208-329) for MANY queries with three launches in total (pair features, hashes of
// the expanded candidates, posting lengths of the primary hashes): query t is structure q_struct[t] of qb with the residues
// q_index[q_off[t] .. q_off[t+1]); subs / n_subs run parallel to q_index.  out[t] is released with fdgpu_query_map_free.
extern "C" int fdgpu_make_query_map_batch(fdgpu_ctx *c, const fdgpu_batch *qb, uint64_t n_queries, const uint32_t *q_struct, const uint64_t *q_off,
                                          const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs, const float *dist_thr,
                                          uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p,
                                          const fdgpu_index *index, float total_structures, fd_query_map **out) { FD_LOCK(c);
    if (!c || !qb || !p || !out || !q_off || (n_queries && !q_struct) || (q_off[n_queries] && !q_index)) return FDGPU_EINVAL;
    for (uint64_t t = 0; t < n_queries; ++t) { out[t] = nullptr; if (q_struct[t] >= qb->n_struct) return FDGPU_EINVAL; }
    // all ordered pairs of every query's residues, row-major (CombinationIterator, utils/combination.rs:23-44), as residue
    // indices of the whole batch
    const bool qtrace = getenv("FDGPU_TRACE") != nullptr;
    const auto q_t0 = std::chrono::steady_clock::now();
    auto q_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q_t0).count(); };
    std::vector<uint32_t> pi, pj;
    std::vector<uint64_t> pair_off(n_queries + 1, 0);
    {       // (a whole-structure query lists ~10^5 pairs: growing the two arrays by doubling was a third of this stage)
        uint64_t n_pairs_max = 0;
        for (uint64_t t = 0; t < n_queries; ++t) { const uint64_t n_q = q_off[t + 1] - q_off[t]; n_pairs_max += n_q * (n_q ? n_q - 1 : 0); }
        pi.reserve(n_pairs_max); pj.reserve(n_pairs_max);
    }
    for (uint64_t t = 0; t < n_queries; ++t) {
        const uint64_t r0 = qb->h_res_off[q_struct[t]], R = qb->h_res_off[q_struct[t] + 1] - r0;
        const uint32_t *qi = q_index + q_off[t];
        const uint64_t n_q = q_off[t + 1] - q_off[t];
        {       // (written through pointers into the reserved arrays: two push_backs per pair were most of this stage for a whole-structure query's 88 k pairs)
            const size_t at = pi.size();
            pi.resize(at + n_q * (n_q ? n_q - 1 : 0)); pj.resize(pi.size());
            uint32_t *wi = pi.data() + at, *wj = pj.data() + at;
            size_t w = 0;
            for (uint64_t a = 0; a < n_q; ++a) {
                if (!(qi[a] < R)) continue;
                const uint32_t ra = (uint32_t)(r0 + qi[a]);
                for (uint64_t b = 0; b < n_q; ++b)
                    if (a != b && qi[b] < R) { wi[w] = ra; wj[w] = (uint32_t)(r0 + qi[b]); ++w; }
            }
            pi.resize(at + w); pj.resize(at + w);
        }
        pair_off[t + 1] = pi.size();
    }
    const uint64_t np = pi.size();
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] pairs listed at %.3f ms\n", q_ms());
    std::vector<float> feat_v;
    std::vector<uint8_t> valid_v;
    CHECK_TYPE(c, p);
    uint8_t *feat_land = nullptr;
    feat_v.reserve(16); valid_v.reserve(16);
    int rc = 0;
    {
        // try the landing block first; the vectors only when it cannot be had
        rc = pair_features12(c, qb, pi.data(), pj.data(), np, p, nullptr, nullptr, &feat_land);
        if (!rc && np && !feat_land) {
            feat_v.resize(np * FD_QF); valid_v.resize(np);
            rc = pair_features12(c, qb, pi.data(), pj.data(), np, p, feat_v.data(), valid_v.data());
        }
    }
    if (rc) return rc;
    const float *feat = feat_land ? (const float *)feat_land : feat_v.data();
    const uint8_t *valid = feat_land ? feat_land + np * 4 * FD_QF : valid_v.data();
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] %llu pairs: features at %.3f ms\n", (unsigned long long)np, q_ms());
    const float RADS_PER_DEG = 3.14159274101257324f / 180.0f;  // f32::to_radians
    std::vector<float> athr(n_angle);
    for (uint64_t t = 0; t < n_angle; ++t) athr[t] = angle_thr_deg[t] * RADS_PER_DEG;
    // candidate lists in the reference's insertion order; hashed in one GPU call, then first-insert-wins per query
    struct cand_t { uint32_t qi, qj; uint8_t primary; uint32_t pair; };
    std::vector<float> vf;      // FD_QF floats per candidate
    std::vector<cand_t> cands;
    std::vector<uint64_t> cand_off(n_queries + 1, 0);
    {
        uint64_t n_valid = 0;
        for (uint64_t k = 0; k < np; ++k) n_valid += valid[k] ? 1 : 0;
        const uint64_t per_pair = 1 + 2 * (2 * n_dist + 5 * n_angle);       // observed + near / far per threshold and field (upper bound without substitutions)
        vf.reserve(n_valid * per_pair * FD_QF); cands.reserve(n_valid * per_pair);
    }
    struct Aad { std::vector<uint8_t> a1, a2; std::vector<float> ad; std::vector<uint32_t> aq; };
    std::vector<Aad> aads(n_queries);
    auto push = [&](const float *f, uint32_t qi, uint32_t qj, bool primary, uint32_t pair) {
        vf.insert(vf.end(), f, f + FD_QF);
        cands.push_back({qi, qj, (uint8_t)(primary ? 1 : 0), pair});
    };
    // dist_index / angle_index of the encoding (controller/feature.rs:269-291): theta only for the two PDBMotif forms — and
    // PDBMotif shifts its DEGREE-valued theta by the threshold converted to radians, like the reference
    static const int d23[2] = {2, 3}, d2[1] = {2}, d7[1] = {7};
    static const int a456[3] = {4, 5, 6}, a37[5] = {3, 4, 5, 6, 7}, a345[3] = {3, 4, 5}, a06[7] = {0, 1, 2, 3, 4, 5, 6}, a48[5] = {4, 5, 6, 7, 8};
    const int *di = d23, *ai = a456;
    int ndi = 2, nai = 3;
    switch (p->hash_type) {
        case FD_HASH_PDBMOTIF: case FD_HASH_PDBMOTIF_SINCOS: nai = 1; break;
        case FD_HASH_TRROSETTA: di = d2; ndi = 1; ai = a37; nai = 5; break;
        case FD_HASH_PPF: di = d2; ndi = 1; ai = a345; nai = 3; break;
        case FD_HASH_TERTIARY: di = d7; ndi = 1; ai = a06; nai = 7; break;
        case FD_HASH_HYBRID: ai = a48; nai = 5; break;
        default: break;
    }
    // ONE large query without substitutions and with one bin configuration (a whole-structure query: ~10^5 pairs, 3 x 10^5 candidates): expansion,
    // hashes and the first-insertion-wins dedupe run on the device (k_qm_*); only the hashes and the kept positions come back.  cands / vf stay
    // empty: candidate z is (valid pair z / per_pair, insertion z % per_pair).  FDGPU_QM_DEVICE=0: the host form (tests).
    std::vector<uint32_t> vpairs;      // device path: the valid pairs, ascending
    uint64_t dev_pp = 0;
    std::vector<uint32_t> dev_keep;
    bool dev_expand = false, dev_dedupe = false, dev_chain = false;
    const char *qd_env_chain = getenv("FDGPU_QM_DEVICE");      // 2: device expansion without the dedupe / length chain (tests compare the forms)
    std::vector<uint8_t> land_v;
    const uint8_t *ch_first = nullptr;       // the chain's results per candidate (page-locked landing block, valid until this call returns)
    const uint64_t *ch_len = nullptr; const long long *ch_kidx = nullptr; const uint32_t *ch_seg = nullptr;
    {
        bool any_subs = false;
        if (subs && n_subs) for (uint64_t a = 0; a < q_off[n_queries] && !any_subs; ++a) any_subs = subs[a] != nullptr;
        uint64_t n_valid = 0;
        for (uint64_t k = 0; k < np; ++k) n_valid += valid[k] ? 1 : 0;
        const uint64_t per_pair = 1 + 2 * ((uint64_t)ndi * n_dist + (uint64_t)nai * n_angle);
        const char *qd_env = getenv("FDGPU_QM_DEVICE");
        const uint64_t qd_min = qd_env && qd_env[0] == '1' ? 1 : 32768;      // 1: also for small queries (tests)
        // expansion + hashes on the device whenever every candidate's place follows from its pair (no substitutions, one bin configuration): a
        // batch of motif queries too — the host then neither builds nor uploads 48 bytes per candidate; the dedupe moves along for one large query
        dev_expand = !any_subs && p->n_multiple_bins == 0 && n_dist <= 8 && n_angle <= 8 && !(qd_env && qd_env[0] == '0') && n_valid &&
                     n_valid * per_pair < (1ull << 31);
        dev_dedupe = dev_expand && n_queries == 1 && n_valid * per_pair >= qd_min;
        if (dev_expand) { dev_pp = per_pair; vpairs.reserve(n_valid); for (uint64_t k = 0; k < np; ++k) if (valid[k]) vpairs.push_back((uint32_t)k); }
    }
    std::vector<uint32_t> pair_q;       // device expansion: the query of every pair
    if (dev_expand) { pair_q.resize(np); for (uint64_t t = 0; t < n_queries; ++t) for (uint64_t k = pair_off[t]; k < pair_off[t + 1]; ++k) pair_q[k] = (uint32_t)t; }
    uint64_t n_valid_seen = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        const uint64_t r0 = qb->h_res_off[q_struct[t]];
        const uint32_t *qidx = q_index + q_off[t];
        const uint64_t n_q = q_off[t + 1] - q_off[t];
        // substitution map keyed by residue index: a later entry for the same residue overrides (query.rs:236-246)
        std::map<uint32_t, std::pair<const uint8_t *, uint32_t>> sub_of;
        if (subs && n_subs)
            for (uint64_t a = 0; a < n_q; ++a)
                if (subs[q_off[t] + a]) sub_of[qidx[a]] = std::make_pair(subs[q_off[t] + a], n_subs[q_off[t] + a]);
        Aad &A = aads[t];
        { const size_t n_p = (size_t)(pair_off[t + 1] - pair_off[t]); A.a1.reserve(n_p); A.a2.reserve(n_p); A.ad.reserve(n_p); A.aq.reserve(n_p); }
        if (dev_expand && pair_off[t + 1] - pair_off[t] >= 16384) {
            // a whole-structure query on the device path: all that is left here is the observed-distance list, read from the page-locked landing block (a cache
            // miss per pair: the copy engine wrote it) — eight parts on the context's helper threads, joined in pair order
            constexpr unsigned NP = 8;
            Aad part[NP];
            uint64_t n_val[NP] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint64_t k0 = pair_off[t], k1 = pair_off[t + 1];
            auto scan = [&](unsigned z) {
                Aad P;              // (locals: the parts' counters and vector headers share cache lines)
                uint64_t nv = 0;
                const uint64_t ka = k0 + (k1 - k0) * z / NP, kb = k0 + (k1 - k0) * (z + 1) / NP;
                P.a1.reserve(kb - ka); P.a2.reserve(kb - ka); P.ad.reserve(kb - ka); P.aq.reserve(kb - ka);
                for (uint64_t k = ka; k < kb; ++k) {
                    if (!valid[k]) continue;
                    const float *f = &feat[(size_t)FD_QF * k];
                    if (f[9] <= 20.0f) { P.a1.push_back((uint8_t)f[10]); P.a2.push_back((uint8_t)f[11]); P.ad.push_back(f[9]); P.aq.push_back((uint32_t)(pi[k] - r0)); }
                    ++nv;
                }
                part[z] = std::move(P); n_val[z] = nv;
            };
            std::atomic<unsigned> nxt(0);
            const std::function<void()> wk = [&]() { for (;;) { const unsigned z = nxt.fetch_add(1); if (z >= NP) break; scan(z); } };
            c->host_pool.run(c->small_par(NP), wk);
            for (unsigned z = 0; z < NP; ++z) {
                A.a1.insert(A.a1.end(), part[z].a1.begin(), part[z].a1.end()); A.a2.insert(A.a2.end(), part[z].a2.begin(), part[z].a2.end());
                A.ad.insert(A.ad.end(), part[z].ad.begin(), part[z].ad.end()); A.aq.insert(A.aq.end(), part[z].aq.begin(), part[z].aq.end());
                n_valid_seen += n_val[z];
            }
            cand_off[t + 1] = n_valid_seen * dev_pp;
            continue;
        }
        for (uint64_t k = pair_off[t]; k < pair_off[t + 1]; ++k) {
            if (!valid[k]) continue;
            const float *f = &feat[(size_t)FD_QF * k];
            uint32_t qi = (uint32_t)(pi[k] - r0), qj = (uint32_t)(pj[k] - r0);
            // observed (aa_i, aa_j, CA distance) list (structure/core.rs:462-477: distance <= 20.0)
            if (f[9] <= 20.0f) { A.a1.push_back((uint8_t)f[10]); A.a2.push_back((uint8_t)f[11]); A.ad.push_back(f[9]); A.aq.push_back(qi); }
            if (dev_expand) { ++n_valid_seen; continue; }
            push(f, qi, qj, true, (uint32_t)k);
            float near[FD_QF], far[FD_QF];
            memcpy(near, f, sizeof near);
            memcpy(far, f, sizeof far);
            // substitutions touch the residue fields of the container: encodings without them take none (amino_acid_index,
            // controller/feature.rs:260-267)
            const bool has_aa = p->hash_type != FD_HASH_TERTIARY && p->hash_type != FD_HASH_HYBRID;
            auto si = has_aa ? sub_of.find(qi) : sub_of.end(), sj = has_aa ? sub_of.find(qj) : sub_of.end();
            if (si != sub_of.end()) {  // apply_substitutions (query.rs:86-156)
                for (uint32_t a = 0; a < si->second.second; ++a) { float t2[FD_QF]; memcpy(t2, near, sizeof t2); t2[0] = (float)si->second.first[a]; push(t2, qi, qj, false, (uint32_t)k); }
                if (sj != sub_of.end())
                    for (uint32_t a = 0; a < si->second.second; ++a)
                        for (uint32_t b = 0; b < sj->second.second; ++b) {
                            float t2[FD_QF]; memcpy(t2, near, sizeof t2);
                            t2[0] = (float)si->second.first[a]; t2[1] = (float)sj->second.first[b];
                            push(t2, qi, qj, false, (uint32_t)k);
                        }
            } else if (sj != sub_of.end()) {
                for (uint32_t b = 0; b < sj->second.second; ++b) { float t2[FD_QF]; memcpy(t2, near, sizeof t2); t2[1] = (float)sj->second.first[b]; push(t2, qi, qj, false, (uint32_t)k); }
            }
            auto expand = [&](const int *idxs, int n_idx, const float *thr, uint64_t n_thr) {  // expand_and_insert (query.rs:179-206)
                for (uint64_t z2 = 0; z2 < n_thr; ++z2)
                    for (int z = 0; z < n_idx; ++z) {
                        int idx = idxs[z];
                        near[idx] = near[idx] - thr[z2];
                        far[idx] = far[idx] + thr[z2];
                        push(near, qi, qj, false, (uint32_t)k);
                        push(far, qi, qj, false, (uint32_t)k);
                        // the reference restores with += / -= (f32, not an exact inverse): keep the drift
                        near[idx] = near[idx] + thr[z2];
                        far[idx] = far[idx] - thr[z2];
                    }
            };
            expand(di, ndi, dist_thr, n_dist);
            expand(ai, nai, athr.data(), n_angle);
        }
        cand_off[t + 1] = dev_expand ? n_valid_seen * dev_pp : cands.size();
    }
    const uint64_t nc = dev_expand ? vpairs.size() * dev_pp : cands.size();
    // candidate z -> (query residues, observed?, pair): stored by the host expansion, implied by the position on the device path
    auto cand_at = [&](uint64_t z) -> cand_t {
        if (!dev_expand) return cands[z];
        const uint32_t k = vpairs[z / dev_pp];
        const uint64_t r0 = qb->h_res_off[q_struct[pair_q[k]]];
        return cand_t{(uint32_t)(pi[k] - r0), (uint32_t)(pj[k] - r0), (uint8_t)(z % dev_pp == 0 ? 1 : 0), k};
    };
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] %llu candidates expanded at %.3f ms\n", (unsigned long long)nc, q_ms());
    std::vector<uint32_t> hashes(std::max<uint64_t>(nc, 1));
    if (dev_expand && nc) {
        hipStream_t st = c->stream;
        const uint64_t nv = vpairs.size();
        qm_expand_par P;
        memset(&P, 0, sizeof P);
        for (int z = 0; z < ndi; ++z) P.di[z] = di[z];
        for (int z = 0; z < nai; ++z) P.ai[z] = ai[z];
        P.ndi = ndi; P.nai = nai; P.n_dist = (uint32_t)n_dist; P.n_angle = (uint32_t)n_angle; P.per_pair = (uint32_t)dev_pp;
        for (uint64_t z = 0; z < n_dist; ++z) P.dthr[z] = dist_thr[z];
        for (uint64_t z = 0; z < n_angle; ++z) P.athr[z] = athr[z];
        // ws[WS_MISC2] still holds the pairs' containers (pair_features12); the sort takes the build's key / id buffers
        HIPCHK(c, c->ws[WS_MISC0].ensure(nv * 4));
        HIPCHK(c, c->ws[WS_MISC1].ensure(nc * 4));
        if (!dev_dedupe) {
            // a motif batch: expansion + hashes -> per-query first-insertion dedupe (k_qm_dedupe) -> posting lengths, segment counts and list
            // positions of EVERY candidate's hash (the index's length table; 17 k lookups per 128 queries) — three kernels back to back, one
            // landing block, one wait; the host neither dedupes nor makes a second round trip for the lengths.  Without the chain's
            // preconditions: hashes only, the rest on the host as before.
            uint64_t max_ins = 0;
            for (uint64_t t = 0; t < n_queries; ++t) max_ins = std::max(max_ins, cand_off[t + 1] - cand_off[t]);
            static const bool lens_cache = [] { const char *e = getenv("FDGPU_LENS_CACHE"); return !(e && e[0] == '0'); }();
            dev_chain = max_ins <= QM_DD_MAX && !(qd_env_chain && qd_env_chain[0] == '2');
            const bool chain_len = dev_chain && index && lens_cache && index->lens && index->n_hashes && index->n_structures;
            const size_t up_words = nv + 2 * (n_queries + 1) + 2;
            uint32_t *up = (uint32_t *)c->host_pinned(3, up_words * 4);
            std::vector<uint32_t> up_v;
            if (!up) { up_v.resize(up_words); up = up_v.data(); }
            const size_t o_off = (nv + 1) & ~(size_t)1;      // 8-byte aligned
            memcpy(up, vpairs.data(), nv * 4);
            memcpy(up + o_off, cand_off.data(), (n_queries + 1) * 8);
            HIPCHK(c, c->ws[WS_MISC0].ensure(up_words * 4));
            // the chain's outputs in ONE device block laid out like the landing block — [len u64 | kidx i64 | hash u32 | nseg u32 | first u8] x nc —
            // so that they come home in one copy launch (they were five)
            HIPCHK(c, c->ws[WS_MISC1].ensure(nc * 25 + 64));
            uint8_t *d_blk = c->ws[WS_MISC1].as<uint8_t>();
            uint64_t *d_len = (uint64_t *)d_blk;
            long long *d_kidx = (long long *)(d_blk + nc * 8);
            uint32_t *d_hash = (uint32_t *)(d_blk + nc * 16), *d_nseg = (uint32_t *)(d_blk + nc * 20);
            uint8_t *d_first = d_blk + nc * 24;
            HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, up, up_words * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_qm_expand_hash, dim3((unsigned)((nv + 63) / 64)), dim3(64), 0, st, c->ws[WS_MISC2].as<float>(), c->ws[WS_MISC0].as<uint32_t>(), (uint32_t)nv, P,
                               fd_make_consts(p).q, d_hash);
            if (dev_chain)
                hipLaunchKernelGGL(k_qm_dedupe, dim3((unsigned)n_queries), dim3(256), 0, st, d_hash, (const uint64_t *)(c->ws[WS_MISC0].as<uint32_t>() + o_off), d_first);
            if (chain_len) fd_launch_posting_lookup(index->hashes, index->offsets, index->lens, index->n_hashes, d_hash, nc, d_len, d_nseg, d_kidx, st);
            HIPCHK(c, hipGetLastError());
            uint8_t *land = (uint8_t *)c->host_pinned(2, nc * 25 + 64);
            if (!land) { land_v.resize(nc * 25 + 64); land = land_v.data(); }
            if (chain_len) HIPCHK(c, hipMemcpyAsync(land, d_blk, nc * 25, hipMemcpyDeviceToHost, st));
            else HIPCHK(c, hipMemcpyAsync(land + nc * 16, d_blk + nc * 16, dev_chain ? nc * 9 : nc * 4, hipMemcpyDeviceToHost, st));      // hashes (and first-insertion flags; the segment counts between them are not read)
            HIPCHK(c, hipStreamSynchronize(st));
            memcpy(hashes.data(), land + nc * 16, nc * 4);
            if (dev_chain) ch_first = land + nc * 24;
            if (chain_len) { ch_len = (const uint64_t *)land; ch_kidx = (const long long *)(land + nc * 8); ch_seg = (const uint32_t *)(land + nc * 20); }
        } else {
        HIPCHK(c, c->ws[WS_KEYS_A].ensure(nc * 4)); HIPCHK(c, c->ws[WS_KEYS_B].ensure(nc * 4));
        HIPCHK(c, c->ws[WS_IDS_A].ensure(nc * 4)); HIPCHK(c, c->ws[WS_IDS_B].ensure(nc * 4));
        HIPCHK(c, c->ws[WS_GHIST].ensure((size_t)256 * std::max<uint32_t>(fd_rs_num_tiles(nc), 1) * 4));
        HIPCHK(c, c->ws[WS_TOT].ensure((256 + (size_t)(fd_rs_num_tiles(nc) / 128 + 2) * 256) * 8));
        HIPCHK(c, c->ws[WS_MISC4].ensure(nc + 8));
        HIPCHK(c, c->ws[WS_TILE_BO].ensure((nc + 2) * 8));
        HIPCHK(c, c->ws[WS_SCANTMP].ensure(fd_scan_tmp_elems(nc) * 8 + 64));
        HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
        HIPCHK(c, hipMemcpyAsync(c->ws[WS_MISC0].p, vpairs.data(), nv * 4, hipMemcpyHostToDevice, st));
        uint32_t *d_hash = c->ws[WS_MISC1].as<uint32_t>();
        hipLaunchKernelGGL(k_qm_expand_hash, dim3((unsigned)((nv + 63) / 64)), dim3(64), 0, st, c->ws[WS_MISC2].as<float>(), c->ws[WS_MISC0].as<uint32_t>(), (uint32_t)nv, P,
                           fd_make_consts(p).q, d_hash);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hashes.data(), d_hash, nc * 4, hipMemcpyDeviceToHost, st));
        // first insertion wins: stable sort of (hash, position) by hash, first of every run marked, marks compacted in position order
        uint32_t *ka = c->ws[WS_KEYS_A].as<uint32_t>(), *kb = c->ws[WS_KEYS_B].as<uint32_t>(), *va = c->ws[WS_IDS_A].as<uint32_t>(), *vb = c->ws[WS_IDS_B].as<uint32_t>();
        const unsigned gb = (unsigned)((nc + 255) / 256);
        HIPCHK(c, hipMemcpyAsync(ka, d_hash, nc * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_qm_iota, dim3(gb), dim3(256), 0, st, va, nc);
        const int cur = fd_radix_sort_pairs(ka, va, kb, vb, nc, 32, c->ws[WS_GHIST].as<uint32_t>(), c->ws[WS_TOT].as<uint64_t>(), st, nullptr);
        uint8_t *d_first = c->ws[WS_MISC4].as<uint8_t>();
        HIPCHK(c, hipMemsetAsync(d_first, 0, nc, st));
        hipLaunchKernelGGL(k_qm_first, dim3(gb), dim3(256), 0, st, cur ? kb : ka, cur ? vb : va, nc, d_first);
        fd_exclusive_scan<uint8_t>(d_first, nc, c->ws[WS_TILE_BO].as<uint64_t>(), c->ws[WS_SCANTMP].as<uint64_t>(), c->ws[WS_TOTAL].as<uint64_t>(), st);
        uint32_t *d_keep = cur ? ka : kb;      // the sort's other buffer is free again
        hipLaunchKernelGGL(k_qm_keep, dim3(gb), dim3(256), 0, st, d_first, c->ws[WS_TILE_BO].as<uint64_t>(), nc, d_keep);
        HIPCHK(c, hipGetLastError());
        uint64_t nk = 0;
        HIPCHK(c, hipMemcpyAsync(&nk, c->ws[WS_TOTAL].p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        dev_keep.resize(nk);
        if (nk) HIPCHK(c, hipMemcpy(dev_keep.data(), d_keep, nk * 4, hipMemcpyDeviceToHost));
        }
    } else
    if ((rc = hash_features_q(c, vf.data(), nc, fd_make_consts(p).q, hashes.data(), FD_QF))) return rc;
    // --multiple-bins: every candidate is inserted under every bin pair, in list order (insert_binned_hash, query.rs:59-70); the
    // observed hash idf is looked up for stays the single-configuration one (query.rs:283-288)
    const uint32_t n_cfg = p->n_multiple_bins ? p->n_multiple_bins : 0u;
    if (!fd_multiple_bins_valid(p)) { c->err = "multiple_bins: at most 8 (dist, angle) bin pairs, no zero counts"; return FDGPU_EINVAL; }
    std::vector<std::vector<uint32_t>> mh_cfg(n_cfg);
    for (uint32_t k = 0; k < n_cfg; ++k) {
        mh_cfg[k].resize(std::max<uint64_t>(nc, 1));
        if ((rc = hash_features_q(c, vf.data(), nc, fd_make_consts_cfg(p, k).q, mh_cfg[k].data(), FD_QF))) return rc;
    }
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] hashed at %.3f ms\n", q_ms());
    std::vector<uint32_t> pair_primary(std::max<uint64_t>(np, 1), 0u);
    if (dev_expand) { for (uint64_t v = 0; v < vpairs.size(); ++v) pair_primary[vpairs[v]] = hashes[v * dev_pp]; }
    else
    for (uint64_t t = 0; t < nc; ++t)
        if (cands[t].primary) pair_primary[cands[t].pair] = hashes[t];
    // first insertion wins (the reference's hash map keeps the entry a hash was first inserted with).  Few candidates: a hash set;
    // many (whole-structure queries, ~10^5): (hash << 32 | insertion position) keys through an LSD radix sort, first key of every
    // hash kept, survivors back in insertion order — a third of the hash set's time there
    const uint64_t ncfg1 = std::max(n_cfg, 1u);
    std::vector<uint64_t> dd_tab;
    std::vector<std::vector<uint32_t>> keeps(n_queries);      // per query: insertion positions (candidate * n_cfg + bin pair) that enter the map, ascending
    uint64_t n_keep = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        const uint64_t c0 = cand_off[t], c1 = cand_off[t + 1], n_ins = (c1 - c0) * ncfg1;
        auto hash_at = [&](uint64_t pos) { const uint64_t z = c0 + pos / ncfg1; const uint32_t k = (uint32_t)(pos % ncfg1); return n_cfg ? mh_cfg[k][z] : hashes[z]; };
        std::vector<uint32_t> &keep = keeps[t];
        if (dev_dedupe) keep.swap(dev_keep);
        else if (ch_first) { keep.reserve((size_t)n_ins / 2 + 8); for (uint64_t pos = 0; pos < n_ins; ++pos) if (ch_first[c0 + pos]) keep.push_back((uint32_t)pos); }
        else if (n_ins <= 4096) {      // a motif query's few hundred insertions: a small open-addressing table (a node-based set cost 3x this)
            uint32_t cap = 64;
            while (cap < 2 * n_ins) cap <<= 1;
            dd_tab.assign(cap, ~0ull);
            keep.reserve((size_t)n_ins);
            const uint32_t *h1 = n_cfg ? nullptr : hashes.data() + c0;       // one bin configuration: position = candidate (no division per insertion)
            for (uint64_t pos = 0; pos < n_ins; ++pos) {
                const uint32_t h = h1 ? h1[pos] : hash_at(pos);
                uint32_t at = (h * 2654435761u) & (cap - 1);
                while (dd_tab[at] != ~0ull && (uint32_t)dd_tab[at] != h) at = (at + 1) & (cap - 1);
                if (dd_tab[at] == ~0ull) { dd_tab[at] = h; keep.push_back((uint32_t)pos); }
            }
        } else if (n_ins >= (1ull << 32)) {
            std::unordered_set<uint32_t> have;
            have.reserve((size_t)n_ins / 4 + 16);
            for (uint64_t pos = 0; pos < n_ins; ++pos) if (have.insert(hash_at(pos)).second) keep.push_back((uint32_t)pos);
        } else {
            std::vector<uint64_t> key(n_ins), tmp(n_ins);
            for (uint64_t pos = 0; pos < n_ins; ++pos) key[pos] = ((uint64_t)hash_at(pos) << 32) | pos;
            for (int pass = 0; pass < 4; ++pass) {
                const int sh = 32 + 8 * pass;
                size_t cnt[257] = {0};
                for (uint64_t k = 0; k < n_ins; ++k) ++cnt[((key[k] >> sh) & 255u) + 1];
                for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
                for (uint64_t k = 0; k < n_ins; ++k) tmp[cnt[(key[k] >> sh) & 255u]++] = key[k];
                key.swap(tmp);
            }
            std::vector<uint8_t> first(n_ins, 0);      // the stable sort left every hash's earliest insertion first: mark, then walk in order
            for (uint64_t k = 0; k < n_ins; ++k) if (k == 0 || (key[k] >> 32) != (key[k - 1] >> 32)) first[(uint32_t)key[k]] = 1;
            for (uint64_t pos = 0; pos < n_ins; ++pos) if (first[pos]) keep.push_back((uint32_t)pos);
        }
        n_keep += keep.size();
    }
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] first insertions at %.3f ms\n", q_ms());
    // ONE posting-length pass: the observed (primary) hash of every pair — idf = log2(S / len), query.rs:17-32 — and every entry that
    // enters a map; the maps remember the latter (post_len / post_seg) so that scoring them needs no second pass over the same lists
    std::vector<float> pair_idf(std::max<uint64_t>(np, 1), 0.0f);
    std::vector<uint64_t> ent_len;
    std::vector<uint32_t> ent_seg;
    std::vector<long long> ent_kidx;
    const uint64_t *e_len = nullptr; const uint32_t *e_seg = nullptr; const long long *e_kidx = nullptr;      // the three arrays as the loops read them (the vectors, or the landing block)
    if (index) {
        std::vector<uint32_t> ph, pk;
        ph.reserve(n_keep + np);
        for (uint64_t t = 0; t < n_queries; ++t) {
            const uint64_t c0 = cand_off[t];
            if (!n_cfg) { for (uint32_t pos : keeps[t]) ph.push_back(hashes[c0 + pos]); }
            else for (uint32_t pos : keeps[t]) { const uint64_t z = c0 + pos / ncfg1; ph.push_back(mh_cfg[pos % ncfg1][z]); }
        }
        if (dev_expand) { for (uint64_t v = 0; v < vpairs.size(); ++v) { ph.push_back(hashes[v * dev_pp]); pk.push_back(vpairs[v]); } }
        else
        for (uint64_t t = 0; t < nc; ++t)
            if (cands[t].primary) { ph.push_back(hashes[t]); pk.push_back(cands[t].pair); }
        ent_len.assign(std::max<size_t>(ph.size(), 1), 0);
        ent_seg.assign(std::max<size_t>(ph.size(), 1), 0);
        ent_kidx.assign(std::max<size_t>(ph.size(), 1), -1);
        e_len = ent_len.data(); e_seg = ent_seg.data(); e_kidx = ent_kidx.data();
        if (ch_len) {      // the chain looked every candidate up: pick the kept entries and the observed hashes, in ph's order
            size_t w = 0;
            for (uint64_t t = 0; t < n_queries; ++t) {
                const uint64_t c0 = cand_off[t];
                for (uint32_t pos : keeps[t]) { const uint64_t z = c0 + pos; ent_len[w] = ch_len[z]; ent_seg[w] = ch_seg[z]; ent_kidx[w] = ch_kidx[z]; ++w; }
            }
            for (uint64_t v = 0; v < vpairs.size(); ++v, ++w) { const uint64_t z = v * dev_pp; ent_len[w] = ch_len[z]; ent_seg[w] = ch_seg[z]; ent_kidx[w] = ch_kidx[z]; }
        } else
        {
            const uint8_t *pl_land = nullptr;       // large requests stay in the page-locked block (read once, in place, by the loops below)
            if ((rc = fd_posting_lengths_segs(c, index, ph.data(), ph.size(), ent_len.data(), ent_seg.data(), ent_kidx.data(), ph.size() >= 16384 ? &pl_land : nullptr))) return rc;
            if (pl_land) { e_len = (const uint64_t *)pl_land; e_kidx = (const long long *)(pl_land + ph.size() * 8); e_seg = (const uint32_t *)(pl_land + ph.size() * 16); }
        }
        // idf of the observed hash of every pair: log2f is ~8 ns a call — 0.7 ms for the 88 k pairs of a whole-structure query on one thread
        auto idf_range = [&](size_t a, size_t b) {
            for (size_t t = a; t < b; ++t) pair_idf[pk[t]] = e_len[n_keep + t] > 0 ? log2f(total_structures / (float)e_len[n_keep + t]) : 0.0f;
        };
        if (pk.size() < 16384) idf_range(0, pk.size());
        else {
            const unsigned nt = 8;
            std::atomic<unsigned> part(0);
            const std::function<void()> w = [&]() { for (;;) { const unsigned k = part.fetch_add(1); if (k >= nt) break; idf_range(pk.size() * k / nt, pk.size() * (k + 1) / nt); } };
            c->host_pool.run(std::min(nt, std::max(1u, std::thread::hardware_concurrency())), w);
        }
    }
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] posting lengths at %.3f ms\n", q_ms());
    uint64_t keep_at = 0;
    for (uint64_t t = 0; t < n_queries; ++t) {
        const uint64_t c0 = cand_off[t];
        auto hash_at = [&](uint64_t pos) { const uint64_t z = c0 + pos / ncfg1; const uint32_t k = (uint32_t)(pos % ncfg1); return n_cfg ? mh_cfg[k][z] : hashes[z]; };
        const std::vector<uint32_t> &keep = keeps[t];
        const Aad &A = aads[t];
        // the map and all its arrays in ONE block (fd_query_map.arena_bytes != 0: fdgpu_query_map_free releases the block; a map was 15
        // allocations, a batch of 128 queries two thousand)
        const size_t n = keep.size(), n_idx = (size_t)(q_off[t + 1] - q_off[t]), n_aad = A.ad.size();
        const bool with_post = index && n;
        auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t o_hash = up16(sizeof(fd_query_map)), o_qi = o_hash + up16(n * 4), o_qj = o_qi + up16(n * 4), o_idf = o_qj + up16(n * 4), o_ph = o_idf + up16(n * 4),
                     o_pl = o_ph + up16(n * 4), o_pk = o_pl + (with_post ? up16(n * 8) : 0), o_ps = o_pk + (with_post ? up16(n * 8) : 0),
                     o_idx = o_ps + (with_post ? up16(n * 4) : 0), o_ad = o_idx + up16(n_idx * 4), o_aq = o_ad + up16(n_aad * 4), o_prim = o_aq + up16(n_aad * 4),
                     o_a1 = o_prim + up16(n), o_a2 = o_a1 + up16(n_aad), bytes = o_a2 + up16(n_aad) + 16;
        uint8_t *blk = (uint8_t *)malloc(bytes);
        if (!blk) { for (uint64_t u = 0; u < t; ++u) { fdgpu_query_map_free(out[u]); out[u] = nullptr; } return FDGPU_ENOMEM; }
        fd_query_map *m = (fd_query_map *)blk;
        memset(m, 0, sizeof *m);
        m->arena_bytes = bytes;
        m->n = n;
        m->hash = (uint32_t *)(blk + o_hash); m->qi = (uint32_t *)(blk + o_qi); m->qj = (uint32_t *)(blk + o_qj); m->idf = (float *)(blk + o_idf);
        m->primary_hash = (uint32_t *)(blk + o_ph); m->is_primary = blk + o_prim;
        uint32_t *const o_hash_p = m->hash, *const o_qi_p = m->qi, *const o_qj_p = m->qj, *const o_ph_p = m->primary_hash;
        uint8_t *const o_pr_p = m->is_primary;
        float *const o_idf_p = m->idf;
        size_t w = 0;
        if (dev_expand && !n_cfg) {       // kept positions ascend: the pair a position belongs to advances with them (no division per entry)
            const uint64_t r0 = qb->h_res_off[q_struct[t]];
            auto fill = [&](size_t a, size_t b) {       // entries [a, b) of the map
                if (a >= b) return;
                uint64_t v = (c0 + keep[a]) / dev_pp, v_end = (v + 1) * dev_pp;      // candidates of valid pair v: [v * dev_pp, v_end)
                for (size_t e = a; e < b; ++e) {
                    const uint64_t z = c0 + keep[e];
                    while (z >= v_end) { ++v; v_end += dev_pp; }
                    const uint32_t k = vpairs[v];
                    o_hash_p[e] = hashes[z]; o_qi_p[e] = (uint32_t)(pi[k] - r0); o_qj_p[e] = (uint32_t)(pj[k] - r0); o_pr_p[e] = z + dev_pp == v_end ? 1 : 0;
                    o_idf_p[e] = pair_idf[k]; o_ph_p[e] = pair_primary[k];
                }
            };
            if (n < 16384) fill(0, n);
            else {       // a whole-structure query's ~10^5 entries: eight parts on the context's helper threads (0.5 ms of the call on one)
                const unsigned nt = 8;
                std::atomic<unsigned> part(0);
                const std::function<void()> wk = [&]() { for (;;) { const unsigned k = part.fetch_add(1); if (k >= nt) break; fill(n * k / nt, n * (k + 1) / nt); } };
                c->host_pool.run(c->small_par(nt), wk);
            }
            w = n;
        } else
        for (uint32_t pos : keep) {
            const uint64_t z = c0 + pos / ncfg1;
            const cand_t cz = cand_at(z);
            o_hash_p[w] = hash_at(pos); o_qi_p[w] = cz.qi; o_qj_p[w] = cz.qj; o_pr_p[w] = cz.primary;
            o_idf_p[w] = pair_idf[cz.pair]; o_ph_p[w] = pair_primary[cz.pair];
            ++w;
        }
        if (with_post) {
            m->post_len = (uint64_t *)(blk + o_pl); m->post_kidx = (long long *)(blk + o_pk); m->post_seg = (uint32_t *)(blk + o_ps);
            memcpy(m->post_len, e_len + keep_at, n * 8); memcpy(m->post_seg, e_seg + keep_at, n * 4); memcpy(m->post_kidx, e_kidx + keep_at, n * 8);
            m->post_index_uid = index->uid;
        }
        keep_at += n;
        m->n_indices = n_idx; m->indices = (uint32_t *)(blk + o_idx);
        if (n_idx) memcpy(m->indices, q_index + q_off[t], n_idx * 4);
        m->n_aad = n_aad; m->aad_dist = (float *)(blk + o_ad); m->aad_qi = (uint32_t *)(blk + o_aq); m->aad_aa1 = blk + o_a1; m->aad_aa2 = blk + o_a2;
        if (n_aad) { memcpy(m->aad_dist, A.ad.data(), n_aad * 4); memcpy(m->aad_qi, A.aq.data(), n_aad * 4); memcpy(m->aad_aa1, A.a1.data(), n_aad); memcpy(m->aad_aa2, A.a2.data(), n_aad); }
        out[t] = m;
    }
    if (qtrace) fprintf(stderr, "[fdgpu_query_map] maps built at %.3f ms\n", q_ms());
    return FDGPU_OK;
}

// one query = structure 0 of qb
extern "C" int fdgpu_make_query_map(fdgpu_ctx *c, const fdgpu_batch *qb, const uint32_t *q_index, uint64_t n_q, const uint8_t *const *subs,
                                    const uint32_t *n_subs, const float *dist_thr, uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle,
                                    const fd_hash_params *p, const fdgpu_index *index, float total_structures, fd_query_map **out) { FD_LOCK(c);
    if (!c || !qb || !p || !out || qb->n_struct < 1 || (n_q && !q_index)) return FDGPU_EINVAL;
    const uint32_t s0 = 0;
    const uint64_t off[2] = {0, n_q};
    return fdgpu_make_query_map_batch(c, qb, 1, &s0, off, q_index, subs, n_subs, dist_thr, n_dist, angle_thr_deg, n_angle, p, index, total_structures, out);
}
