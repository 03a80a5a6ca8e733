// k_bucket.hip — bucketed index build: the posting stream leaves the pair kernel already partitioned by the CHEAP top 14 bits of
// the PDBTrRosetta hash, so that two 8-bit radix passes inside a bucket finish the sort (four passes over 6-byte elements before).
//
// hash = aa1 << 25 | aa2 << 20 | ca << 16 | cb << 12 | 12 angle bits (pdb_tr.rs:21-75, fields OR-ed unmasked).  The top 14 bits
// (residue types, CA-CA bin, plus the one bit a CB-CB bin >= 16 spills into the CA field) need two distances only — no torsions —
// so a first pass over the residue pairs can histogram them:
//
//   k_bk_count   one workgroup per structure: LDS histogram over the 6,400 buckets (aa1, aa2, ca4)  -> matrix M[structure][bucket]
//   k_bk_col*    exclusive prefix of every bucket column over the structures (in place) + bucket totals
//   k_bk_bases   bucket bases / sort tiles of one bucket group
//   k_bk_emit    one workgroup per structure: the row of M becomes 6,400 LDS cursors; every hash of the structure takes a slot
//                with one LDS atomic and is stored as a 4-byte element  (hash & 0xffff) << 16 | (structure & 0xffff)
//                (+ one byte structure >> 16 for shards beyond 65,536 structures)
//
// Inside a bucket the elements are structure-major (the column prefix orders them by structure, order inside a structure is
// free), so a STABLE sort by the low 16 hash bits (k_sort.hip, segmented passes) leaves ids ascending inside every hash — the
// order in which the reference appends to a posting list (indextable.rs:171-202).
//
// Bucket groups: when the element buffers of the whole shard do not fit, the buckets are split into G groups by
// (aa_i + aa_j) mod G — both orientations of a residue pair land in the same group, so no descriptor is computed twice — and
// the build runs count once, then emit / sort / encode per group; k_bk_assemble interleaves the groups' slices by bucket.
//
// Mapping notes (gfx950): 256-thread workgroups, the four waves take the structure's 64-residue i-tiles round robin; the filter
// loop, the LDS compaction queue and the 64-lane drains are those of k_pair_emit2 (k_hash.hip), wave-private; no workgroup
// barrier inside the pair loop.  XCD remap: an XCD works on a contiguous range of structures, so the partial lines the scattered
// 4-byte stores leave in a bucket are completed by neighbouring structures in the same L2 / the memory-side cache.
#include "fd_device.h"
#include "fd_geom_other.h"

#define BK_NB 6400u          // 20 x 20 residue-type pairs x 16 values of the CA field
#define BK_THREADS 256
#define BK_WAVES (BK_THREADS / FD_WAVE)
#ifndef FD_BK_EMIT_WAVES
#define FD_BK_EMIT_WAVES 3   // workgroups per CU x 4 waves / 4 SIMDs: LDS (cursors 25.6 KB + 4 x 5 KB frame tiles) allows three
#endif

struct fd_frame;   // fd_geom.h

__device__ __forceinline__ uint32_t bk_bucket_of_top(uint32_t top14) {   // top14 = hash >> 16
    return ((top14 >> 9) * 20u + ((top14 >> 4) & 31u)) * 16u + (top14 & 15u);
}
__device__ __forceinline__ float bk_bcast(float v, uint32_t k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)k)); }
__device__ __forceinline__ void bk_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ count
// frames[r] holds CA, CB (first six floats) and the residue type (last word) of residue r
__device__ __forceinline__ void bk_count_drain(const fd_frame *__restrict__ frames, const uint32_t *q, uint32_t n, uint32_t i0, uint32_t r0, float dist_disc,
                                               uint32_t *hist, unsigned long long *wide_flag) {
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < n) {
        const uint32_t e = q[lane];
        const uint32_t i = i0 + (e >> 16), j = r0 + (e & 0xffffu);
        const float4 *pi = reinterpret_cast<const float4 *>(frames + i), *pj = reinterpret_cast<const float4 *>(frames + j);
        const float4 ai = pi[0], bi = pi[1], ei = pi[4], aj = pj[0], bj = pj[1], ej = pj[4];
        const fd_v3 cai = {ai.x, ai.y, ai.z}, cbi = {ai.w, bi.x, bi.y}, caj = {aj.x, aj.y, aj.z}, cbj = {aj.w, bj.x, bj.y};
        const float ca_dist = fd_dist(cai, caj);                     // the operands and order of fd_pair_both_spec (fd_geom.h)
        const fd_v3 v3 = fd_sub(cbj, cbi);
        const float cb_dist = fd_sqrtf(v3.x * v3.x + v3.y * v3.y + v3.z * v3.z);
        const uint32_t mid = fd_q(ca_dist, 2.0f, dist_disc) << 16 | fd_q(cb_dist, 2.0f, dist_disc) << 12;
        const uint32_t aai = __float_as_uint(ei.w), aaj = __float_as_uint(ej.w);
        if ((mid >> 20) != 0u) { if (wide_flag) atomicOr(wide_flag, 1ull); }   // a field spills into the residue-type bits: not a bucketed build
        else {
            const uint32_t c4 = mid >> 16;
            atomicAdd(&hist[(aai * 20u + aaj) * 16u + c4], 1u);
            atomicAdd(&hist[(aaj * 20u + aai) * 16u + c4], 1u);
        }
    }
}

__global__ __launch_bounds__(BK_THREADS) void k_bk_count(const float *__restrict__ ca_xyz, const uint8_t *__restrict__ hash_ok, const uint32_t *__restrict__ res_off,
                                                          uint32_t n_struct, const fd_frame *__restrict__ frames, float d2_max, float dist_disc,
                                                          uint32_t *__restrict__ M, unsigned long long *wide_flag) {
    __shared__ uint32_t hist[BK_NB];
    __shared__ uint32_t qs[BK_WAVES][2 * FD_WAVE];
    const uint32_t s = fd_xcd_remap(blockIdx.x, n_struct);
    if (s >= n_struct) return;
    for (uint32_t k = threadIdx.x; k < BK_NB; k += BK_THREADS) hist[k] = 0;
    __syncthreads();
    const uint32_t r0 = res_off[s], r1 = res_off[s + 1];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t *q = qs[wid];
    for (uint32_t i0 = r0 + wid * FD_WAVE; i0 < r1; i0 += BK_WAVES * FD_WAVE) {
        const uint32_t i = i0 + lane;
        const bool vi = i < r1 && hash_ok[i];
        fd_v3 cai = {0.f, 0.f, 0.f};
        if (vi) cai = fd_load3(ca_xyz, i);
        uint32_t qn = 0;
        for (uint32_t jb = i0; jb < r1; jb += FD_WAVE) {
            const uint32_t jl = jb + lane;
            const bool jin = jl < r1;
            fd_v3 cj = {0.f, 0.f, 0.f};
            if (jin) cj = fd_load3(ca_xyz, jl);
            const uint64_t okm = __ballot(jin && hash_ok[jl]);
            const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
            const bool last_block = jb + FD_WAVE >= r1;
            for (uint32_t k = 0; k < nj; ++k) {
                const bool last = last_block && k + 1 == nj;
                if ((okm >> k) & 1ull) {
                    const fd_v3 caj = {bk_bcast(cj.x, k), bk_bcast(cj.y, k), bk_bcast(cj.z, k)};
                    const float d2 = fd_dist2(cai, caj);
                    const uint32_t j = jb + k;
                    const bool pass = vi && j > i && !(d2 > d2_max);
                    const uint64_t m = __ballot(pass);
                    if (m != 0) {
                        if (pass) q[qn + fd_mbcnt(m)] = (lane << 16) | (j - r0);
                        qn += (uint32_t)__popcll(m);
                    }
                }
                while (qn >= FD_WAVE || (last && qn)) {
                    bk_wave_sync();
                    const uint32_t n = qn < FD_WAVE ? qn : FD_WAVE;
                    qn -= n;
                    bk_count_drain(frames, q + qn, n, i0, r0, dist_disc, hist, wide_flag);
                    bk_wave_sync();
                }
            }
        }
    }
    __syncthreads();
    uint32_t *row = M + (uint64_t)s * BK_NB;
    for (uint32_t k = threadIdx.x; k < BK_NB; k += BK_THREADS) row[k] = hist[k];
}

// ------------------------------------------------------------------ column prefix over the structures (in place)
#define BK_ROWS 128u   // structures per workgroup row block
// part[rb][b] = sum over the row block's structures of M[s][b]
__global__ __launch_bounds__(256) void k_bk_colsum(const uint32_t *__restrict__ M, uint32_t n_struct, uint32_t *__restrict__ part) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x, rb = blockIdx.y;
    if (b >= BK_NB) return;
    const uint32_t s0 = rb * BK_ROWS, s1 = s0 + BK_ROWS < n_struct ? s0 + BK_ROWS : n_struct;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t s = s0; s < s1; ++s) sum += M[(uint64_t)s * BK_NB + b];
    part[(uint64_t)rb * BK_NB + b] = sum;
}
// thread = bucket: exclusive scan of its column of partial sums (positions inside a bucket fit 32 bits), bucket total out
__global__ __launch_bounds__(256) void k_bk_colscan(uint32_t *__restrict__ part, uint32_t n_rb, unsigned long long *__restrict__ btot) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= BK_NB) return;
    unsigned long long run = 0;
#pragma unroll 4
    for (uint32_t rb = 0; rb < n_rb; ++rb) {
        const uint32_t v = part[(uint64_t)rb * BK_NB + b];
        part[(uint64_t)rb * BK_NB + b] = (uint32_t)run;
        run += v;
    }
    btot[b] = run;
}
__global__ __launch_bounds__(256) void k_bk_colapply(uint32_t *__restrict__ M, uint32_t n_struct, const uint32_t *__restrict__ part) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x, rb = blockIdx.y;
    if (b >= BK_NB) return;
    const uint32_t s0 = rb * BK_ROWS, s1 = s0 + BK_ROWS < n_struct ? s0 + BK_ROWS : n_struct;
    uint32_t run = part[(uint64_t)rb * BK_NB + b];
#pragma unroll 8
    for (uint32_t s = s0; s < s1; ++s) {
        const uint32_t v = M[(uint64_t)s * BK_NB + b];
        M[(uint64_t)s * BK_NB + b] = run;
        run += v;
    }
}

// ------------------------------------------------------------------ bucket bases and sort tiles of one group
// group of a bucket = (aa1 + aa2) & (n_groups - 1).  bbase[b] = first element of bucket b in the group's element buffer
// (buckets of other groups are empty), bbase[NB] = group total; tfirst[b] = first sort tile of the bucket, tfirst[NB] = tiles.
__global__ __launch_bounds__(1024) void k_bk_bases(const unsigned long long *__restrict__ btot, uint32_t group, uint32_t n_groups, uint32_t tile,
                                                   unsigned long long *__restrict__ bbase, uint32_t *__restrict__ tfirst) {
    __shared__ unsigned long long se[1024];
    __shared__ uint32_t st[1024];
    // 6400 buckets over 1024 threads: thread t owns buckets [t * 7, t * 7 + 7)
    const uint32_t per = (BK_NB + 1023u) / 1024u, b0 = threadIdx.x * per;
    unsigned long long e = 0;
    uint32_t t = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t b = b0 + k;
        if (b < BK_NB && (((b / 320u) + ((b / 16u) % 20u)) & (n_groups - 1u)) == group) { e += btot[b]; t += (uint32_t)((btot[b] + tile - 1) / tile); }
    }
    se[threadIdx.x] = e; st[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        unsigned long long ve = threadIdx.x >= off ? se[threadIdx.x - off] : 0ull;
        uint32_t vt = threadIdx.x >= off ? st[threadIdx.x - off] : 0u;
        __syncthreads();
        se[threadIdx.x] += ve; st[threadIdx.x] += vt;
        __syncthreads();
    }
    unsigned long long re = se[threadIdx.x] - e;
    uint32_t rt = st[threadIdx.x] - t;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t b = b0 + k;
        if (b >= BK_NB) break;
        bbase[b] = re; tfirst[b] = rt;
        if ((((b / 320u) + ((b / 16u) % 20u)) & (n_groups - 1u)) == group) { re += btot[b]; rt += (uint32_t)((btot[b] + tile - 1) / tile); }
    }
    if (threadIdx.x == 1023) { bbase[BK_NB] = se[1023]; tfirst[BK_NB] = st[1023]; }
}
// tile_bucket[t] = bucket of sort tile t; enc_bucket[u] = bucket of the encoder tile whose first element is u * enc_tile
__global__ __launch_bounds__(256) void k_bk_tilemap(const unsigned long long *__restrict__ bbase, const uint32_t *__restrict__ tfirst, uint32_t enc_tile,
                                                    uint32_t *__restrict__ tile_bucket, uint32_t *__restrict__ enc_bucket) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= BK_NB) return;
    for (uint32_t t = tfirst[b]; t < tfirst[b + 1]; ++t) tile_bucket[t] = b;
    const unsigned long long e0 = bbase[b], e1 = bbase[b + 1];
    for (unsigned long long u = (e0 + enc_tile - 1) / enc_tile; u * enc_tile < e1; ++u) enc_bucket[u] = b;
}

// ------------------------------------------------------------------ emit
__device__ __forceinline__ fd_frame bk_load_frame(const fd_frame *__restrict__ frames, uint32_t r) {
    const float4 *p = reinterpret_cast<const float4 *>(frames + r);
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    fd_frame F;
    F.ca = {a.x, a.y, a.z}; F.cb = {a.w, b.x, b.y}; F.r1 = {b.z, b.w, c.x}; F.t1 = {c.y, c.z, c.w};
    F.s2 = {d.x, d.y, d.z}; F.nv2 = {d.w, e.x, e.y}; F.len = e.z; F.pad = e.w;
    return F;
}
__device__ __attribute__((noinline)) uint2 bk_pair_exact(const fd_frame *__restrict__ frames, uint32_t i, uint32_t j, float dist_disc, float ang_disc,
                                                         const uint32_t *tab) {
    fd_frame Fi = bk_load_frame(frames, i), Fj = bk_load_frame(frames, j);
    fd_quant q;
    q.dist_disc = dist_disc; q.ang_disc = ang_disc; q.ang2_disc = 0.0f; q.type = FD_HASH_PDBTR;
    uint32_t a, b;
    fd_pair_both_tab(Fi, Fj, __float_as_uint(Fi.pad), __float_as_uint(Fj.pad), q, tab, &a, &b);
    return make_uint2(a, b);
}

struct bk_emit_args {
    const float *ca_xyz; const uint8_t *hash_ok; const uint32_t *res_off; uint32_t n_struct;
    const fd_frame *frames;
    float d2_max; fd_quant q; int spec;
    const uint32_t *M;                    // [S][NB] within-bucket prefix of every structure
    const unsigned long long *bbase;      // [NB + 1] of this group
    uint32_t group, n_groups;
    uint32_t *keys; uint8_t *idh;         // element buffers (idh may be null: <= 65,536 structures)
    unsigned long long *spec_miss, *err_flag;
};

template <bool IDH, bool GROUPS>
__device__ __forceinline__ void bk_emit_drain(const bk_emit_args &A, const uint32_t *tab, const uint32_t *q, uint32_t n, uint32_t i0, uint32_t r0, uint32_t s,
                                              const float4 *s_fi, uint32_t *cursor) {
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < n) {
        const uint32_t e = q[lane];
        const uint32_t il = e >> 16, i = i0 + il, j = r0 + (e & 0xffffu);
        const float4 a4 = s_fi[il], b4 = s_fi[64 + il], c4 = s_fi[128 + il], d4 = s_fi[192 + il], e4 = s_fi[256 + il];
        fd_frame Fi;
        Fi.ca = {a4.x, a4.y, a4.z}; Fi.cb = {a4.w, b4.x, b4.y}; Fi.r1 = {b4.z, b4.w, c4.x}; Fi.t1 = {c4.y, c4.z, c4.w};
        Fi.s2 = {d4.x, d4.y, d4.z}; Fi.nv2 = {d4.w, e4.x, e4.y}; Fi.len = e4.z; Fi.pad = e4.w;
        const fd_frame Fj = bk_load_frame(A.frames, j);
        uint32_t h_ij, h_ji;
        if (!A.spec || !fd_pair_both_spec(Fi, Fj, __float_as_uint(Fi.pad), __float_as_uint(Fj.pad), A.q, tab, tab + 32, &h_ij, &h_ji)) {
            const uint2 h = bk_pair_exact(A.frames, i, j, A.q.dist_disc, A.q.ang_disc, tab);
            h_ij = h.x; h_ji = h.y;
            if (A.spec && A.spec_miss) atomicAdd(A.spec_miss, 1ull);
        }
        const uint32_t b1 = bk_bucket_of_top(h_ij >> 16), b2 = bk_bucket_of_top(h_ji >> 16);
        if ((h_ij | h_ji) >> 30 || b1 >= BK_NB || b2 >= BK_NB) { if (A.err_flag) atomicOr(A.err_flag, 1ull); }   // the count pass flags these shards first
        else {
            const uint32_t lo = s & 0xffffu;
            const unsigned long long p1 = A.bbase[b1] + atomicAdd(&cursor[b1], 1u);
            const unsigned long long p2 = A.bbase[b2] + atomicAdd(&cursor[b2], 1u);
            A.keys[p1] = (h_ij << 16) | lo;
            A.keys[p2] = (h_ji << 16) | lo;
            if (IDH) { A.idh[p1] = (uint8_t)(s >> 16); A.idh[p2] = (uint8_t)(s >> 16); }
        }
    }
}

template <bool IDH, bool GROUPS>
__global__ __launch_bounds__(BK_THREADS, FD_BK_EMIT_WAVES) void k_bk_emit(bk_emit_args A) {
    __shared__ uint32_t cursor[BK_NB];
    __shared__ uint32_t qs[BK_WAVES][2 * FD_WAVE];
    __shared__ uint32_t tab[64];
    __shared__ float4 s_fis[BK_WAVES][5 * FD_WAVE];
    const uint32_t s = fd_xcd_remap(blockIdx.x, A.n_struct);
    if (s >= A.n_struct) return;
    {
        const uint32_t *row = A.M + (uint64_t)s * BK_NB;
        for (uint32_t k = threadIdx.x; k < BK_NB; k += BK_THREADS) cursor[k] = row[k];
        if (threadIdx.x == 0) {
            fd_fill_bintab(tab);
            for (int k = 0; k < FD_BINTAB_WORDS; ++k) tab[32 + k] = tab[k];
            for (int m = 0; m < 4; ++m)
                for (int k = 0; k < 4; ++k) { uint32_t v = tab[32 + 7 + 5 * m + k]; tab[32 + 7 + 5 * m + k] = v > 0x7f7fffffu ? 0x7f7fffffu : v; }
        }
    }
    __syncthreads();
    const uint32_t r0 = A.res_off[s], r1 = A.res_off[s + 1];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t *q = qs[wid];
    float4 *s_fi = s_fis[wid];
    for (uint32_t i0 = r0 + wid * FD_WAVE; i0 < r1; i0 += BK_WAVES * FD_WAVE) {
        const uint32_t i = i0 + lane;
        const bool vi = i < r1 && A.hash_ok[i];
        fd_v3 cai = {0.f, 0.f, 0.f};
        if (vi) cai = fd_load3(A.ca_xyz, i);
        uint32_t aai = 0;
        bk_wave_sync();   // the previous tile's drains are done with s_fi
        if (i < r1) {
            const float4 *fp = reinterpret_cast<const float4 *>(A.frames + i);
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float4 v = fp[k]; s_fi[k * FD_WAVE + lane] = v; if (k == 4) aai = __float_as_uint(v.w); }
        }
        bk_wave_sync();
        uint32_t qn = 0;
        for (uint32_t jb = i0; jb < r1; jb += FD_WAVE) {
            const uint32_t jl = jb + lane;
            const bool jin = jl < r1;
            fd_v3 cj = {0.f, 0.f, 0.f};
            uint32_t aaj_l = 0;
            if (jin) { cj = fd_load3(A.ca_xyz, jl); if (GROUPS) aaj_l = __float_as_uint(reinterpret_cast<const float4 *>(A.frames + jl)[4].w); }
            const uint64_t okm = __ballot(jin && A.hash_ok[jl]);
            const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
            const bool last_block = jb + FD_WAVE >= r1;
            for (uint32_t k = 0; k < nj; ++k) {
                const bool last = last_block && k + 1 == nj;
                if ((okm >> k) & 1ull) {
                    const fd_v3 caj = {bk_bcast(cj.x, k), bk_bcast(cj.y, k), bk_bcast(cj.z, k)};
                    const float d2 = fd_dist2(cai, caj);
                    const uint32_t j = jb + k;
                    bool pass = vi && j > i && !(d2 > A.d2_max);
                    if (GROUPS) pass = pass && (((aai + (uint32_t)__builtin_amdgcn_readlane((int)aaj_l, (int)k)) & (A.n_groups - 1u)) == A.group);
                    const uint64_t m = __ballot(pass);
                    if (m != 0) {
                        if (pass) q[qn + fd_mbcnt(m)] = (lane << 16) | (j - r0);
                        qn += (uint32_t)__popcll(m);
                    }
                }
                while (qn >= FD_WAVE || (last && qn)) {
                    bk_wave_sync();
                    const uint32_t n = qn < FD_WAVE ? qn : FD_WAVE;
                    qn -= n;
                    bk_emit_drain<IDH, GROUPS>(A, tab, q + qn, n, i0, r0, s, s_fi, cursor);
                    bk_wave_sync();
                }
            }
        }
    }
}

// ------------------------------------------------------------------ assembly of the groups' index slices (bucket order)
// per bucket b (thread): the slice of its group holds its hashes at [lower_bound(top(b) << 16), lower_bound((top(b) + 1) << 16))
struct bk_slice { const uint32_t *hashes; const uint64_t *offsets; const uint8_t *value; uint64_t H; };
__device__ __forceinline__ uint64_t bk_lower_bound(const uint32_t *__restrict__ h, uint64_t H, uint64_t key) {
    uint64_t lo = 0, hi = H;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)h[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_bk_asm_ranges(const bk_slice *__restrict__ slices, uint32_t n_groups, unsigned long long *__restrict__ src_h0,
                                                       uint32_t *__restrict__ n_h, unsigned long long *__restrict__ n_b) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= BK_NB) return;
    const uint32_t aa1 = b / 320u, aa2 = (b / 16u) % 20u, c4 = b & 15u;
    const bk_slice S = slices[(aa1 + aa2) & (n_groups - 1u)];
    const uint64_t top = (uint64_t)aa1 << 9 | (uint64_t)aa2 << 4 | c4;
    const uint64_t h0 = bk_lower_bound(S.hashes, S.H, top << 16), h1 = bk_lower_bound(S.hashes, S.H, (top + 1) << 16);
    src_h0[b] = h0;
    n_h[b] = (uint32_t)(h1 - h0);
    n_b[b] = S.offsets[h1] - S.offsets[h0];
}
// one workgroup per (bucket, part): hashes and re-based offsets; value bytes staged through LDS so that the 16-byte stores are aligned
#define BK_ASM_CHUNK 16384u   // value bytes per workgroup step
__global__ __launch_bounds__(256) void k_bk_asm_copy(const bk_slice *__restrict__ slices, uint32_t n_groups, const unsigned long long *__restrict__ src_h0,
                                                     const uint32_t *__restrict__ n_h, const uint64_t *__restrict__ dst_h0, const uint64_t *__restrict__ dst_b0,
                                                     uint32_t *__restrict__ out_hashes, uint64_t *__restrict__ out_offsets, uint8_t *__restrict__ out_value,
                                                     uint32_t parts) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[BK_ASM_CHUNK + 64];
    const uint32_t b = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint32_t aa1 = b / 320u, aa2 = (b / 16u) % 20u;
    const bk_slice S = slices[(aa1 + aa2) & (n_groups - 1u)];
    const uint64_t sh0 = src_h0[b], nh = n_h[b], dh0 = dst_h0[b];
    if (nh == 0) return;
    const uint64_t sb0 = S.offsets[sh0], sb1 = S.offsets[sh0 + nh], db0 = dst_b0[b];
    // hashes / offsets: strided over the parts
    for (uint64_t k = (uint64_t)part * 256u + threadIdx.x; k < nh; k += (uint64_t)parts * 256u) {
        out_hashes[dh0 + k] = S.hashes[sh0 + k];
        out_offsets[dh0 + k] = S.offsets[sh0 + k] - sb0 + db0;
    }
    // value bytes: chunks of BK_ASM_CHUNK destination bytes, destination-aligned
    const uint64_t nbytes = sb1 - sb0;
    const uint64_t d_lo = db0 & ~15ull;   // aligned start of the destination window
    for (uint64_t c = (uint64_t)part * BK_ASM_CHUNK; d_lo + c < db0 + nbytes; c += (uint64_t)parts * BK_ASM_CHUNK) {
        // destination bytes [d_lo + c, d_lo + c + CHUNK) intersected with [db0, db0 + nbytes)
        const uint64_t w0 = d_lo + c, w1 = w0 + BK_ASM_CHUNK;
        const uint64_t a = w0 > db0 ? w0 : db0, e = w1 < db0 + nbytes ? w1 : db0 + nbytes;
        __syncthreads();
        for (uint64_t p = a + threadIdx.x; p < e; p += 256u) stage[p - w0] = S.value[sb0 + (p - db0)];
        __syncthreads();
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        for (uint32_t o = threadIdx.x * 16u; o < BK_ASM_CHUNK; o += 256u * 16u) {
            const uint64_t p = w0 + o;
            if (p >= a && p + 16 <= e) *reinterpret_cast<u32x4 *>(out_value + p) = *reinterpret_cast<const u32x4 *>(stage + o);
            else for (uint64_t z = p > a ? p : a; z < p + 16 && z < e; ++z) out_value[z] = stage[z - w0];
        }
    }
}

// ------------------------------------------------------------------ launchers
void fd_bk_count(const fd_batch_view &B, const void *frames, float d2_max, float dist_disc, uint32_t *M, unsigned long long *wide_flag, hipStream_t st) {
    if (!B.n_struct) return;
    const unsigned grid = ((B.n_struct + 7u) / 8u) * 8u;
    hipLaunchKernelGGL(k_bk_count, dim3(grid), dim3(BK_THREADS), 0, st, B.ca_xyz, B.hash_ok, B.res_off, B.n_struct, (const fd_frame *)frames, d2_max, dist_disc, M,
                       wide_flag);
}
uint32_t fd_bk_num_buckets() { return BK_NB; }
uint32_t fd_bk_row_blocks(uint64_t n_struct) { return (uint32_t)((n_struct + BK_ROWS - 1) / BK_ROWS); }
void fd_bk_colscan(uint32_t *M, uint64_t n_struct, uint32_t *part, unsigned long long *btot, hipStream_t st) {
    const uint32_t n_rb = fd_bk_row_blocks(n_struct);
    if (!n_rb) { (void)hipMemsetAsync(btot, 0, BK_NB * 8, st); return; }
    const dim3 g((BK_NB + 255u) / 256u, n_rb);
    hipLaunchKernelGGL(k_bk_colsum, g, dim3(256), 0, st, M, (uint32_t)n_struct, part);
    hipLaunchKernelGGL(k_bk_colscan, dim3((BK_NB + 255u) / 256u), dim3(256), 0, st, part, n_rb, btot);
    hipLaunchKernelGGL(k_bk_colapply, g, dim3(256), 0, st, M, (uint32_t)n_struct, part);
}
void fd_bk_bases(const unsigned long long *btot, uint32_t group, uint32_t n_groups, uint32_t sort_tile, unsigned long long *bbase, uint32_t *tfirst,
                 uint32_t enc_tile, uint32_t *tile_bucket, uint32_t *enc_bucket, hipStream_t st) {
    hipLaunchKernelGGL(k_bk_bases, dim3(1), dim3(1024), 0, st, btot, group, n_groups, sort_tile, bbase, tfirst);
    hipLaunchKernelGGL(k_bk_tilemap, dim3((BK_NB + 255u) / 256u), dim3(256), 0, st, bbase, tfirst, enc_tile, tile_bucket, enc_bucket);
}
void fd_bk_emit(const fd_batch_view &B, const void *frames, const fd_hash_consts &C, const uint32_t *M, const unsigned long long *bbase, uint32_t group,
                uint32_t n_groups, uint32_t *keys, uint8_t *idh, unsigned long long *err_flag, hipStream_t st) {
    if (!B.n_struct) return;
    bk_emit_args A;
    A.ca_xyz = B.ca_xyz; A.hash_ok = B.hash_ok; A.res_off = B.res_off; A.n_struct = B.n_struct; A.frames = (const fd_frame *)frames;
    A.d2_max = C.d2_max; A.q = C.q; A.spec = C.use_tab == 2; A.M = M; A.bbase = bbase; A.group = group; A.n_groups = n_groups;
    A.keys = keys; A.idh = idh; A.spec_miss = C.spec_miss; A.err_flag = err_flag;
    const dim3 g(((B.n_struct + 7u) / 8u) * 8u), b(BK_THREADS);
    if (idh && n_groups > 1) hipLaunchKernelGGL((k_bk_emit<true, true>), g, b, 0, st, A);
    else if (idh) hipLaunchKernelGGL((k_bk_emit<true, false>), g, b, 0, st, A);
    else if (n_groups > 1) hipLaunchKernelGGL((k_bk_emit<false, true>), g, b, 0, st, A);
    else hipLaunchKernelGGL((k_bk_emit<false, false>), g, b, 0, st, A);
}
void fd_bk_asm_ranges(const void *slices, uint32_t n_groups, unsigned long long *src_h0, uint32_t *n_h, unsigned long long *n_b, hipStream_t st) {
    hipLaunchKernelGGL(k_bk_asm_ranges, dim3((BK_NB + 255u) / 256u), dim3(256), 0, st, (const bk_slice *)slices, n_groups, src_h0, n_h, n_b);
}
void fd_bk_asm_copy(const void *slices, uint32_t n_groups, const unsigned long long *src_h0, const uint32_t *n_h, const uint64_t *dst_h0, const uint64_t *dst_b0,
                    uint32_t *out_hashes, uint64_t *out_offsets, uint8_t *out_value, hipStream_t st) {
    const uint32_t parts = 16;
    hipLaunchKernelGGL(k_bk_asm_copy, dim3(BK_NB * parts), dim3(256), 0, st, (const bk_slice *)slices, n_groups, src_h0, n_h, dst_h0, dst_b0, out_hashes,
                       out_offsets, out_value, parts);
}
