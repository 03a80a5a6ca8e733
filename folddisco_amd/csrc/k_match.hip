// k_match.hip — candidate residue-pair scan and batched Kabsch superposition on gfx950.
//
// Replaces the inner loops of HOT LOOP D of the reference:
//   * prefilter_amino_acid + retrieve_with_prefilter (src/controller/retrieve.rs:563-602, 52-156):
//     for a candidate structure, every ordered residue pair (i, j) drawn from the amino-acid
//     prefilter sets (or all pairs when a set is empty / the query is large) is tested for
//     CA distance <= cutoff, then against the query's observed (aa_i, aa_j, CA distance, qi) list
//     with the --ca-distance window; survivors with a valid descriptor are "candidate pairs", and
//     those whose PDBTrRosetta hash is a query hash are "found" triples.
//   * KabschSuperimposer (src/structure/kabsch.rs:157-554, mode 2) for every match.
// The coordinates come from the HBM-resident packed batch instead of re-reading and re-parsing the
// candidate's PDB file (retrieve.rs:375).  Graph components and residue voting stay on the host
// (integer glue between the two GPU stages).
#include "fdgpu_internal.h"


__device__ __forceinline__ bool hash_in_set(const uint32_t *__restrict__ h, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (h[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && h[lo] == x;
}

// descriptor + hash + output of one surviving (i, j) per lane (full-wave drains of the compaction queue: executed
// divergently per survivor this part — ~3000 instructions with the exact libm chain — was 95 % of the kernel time)
template <bool EMIT>
__device__ __forceinline__ void match_drain(const mp_args &A, const uint32_t *q, uint32_t n, uint32_t slot, uint32_t r0, uint32_t i0,
                                            const uint32_t *st_tab, const float *dist_tab, const uint32_t *tab) {
    const uint32_t lane = threadIdx.x;
    const bool on = lane < n;
    const uint32_t e0 = on ? q[lane] : 0u;
    const uint32_t i = i0 + (e0 >> 16), j = r0 + (e0 & 0xffffu);
    uint32_t aai = 255u, aaj = 255u, n_win = 0, h = 0;
    fd_v3 cai = {0.f, 0.f, 0.f}, caj = {0.f, 0.f, 0.f};
    float d = 0.f;
    uint32_t key = 0, e_lo = 0, e_hi = 0;
    bool hit = false;
    if (on) {
        aai = A.B.aa[i]; aaj = A.B.aa[j];
        cai = fd_load3(A.B.ca_xyz, i); caj = fd_load3(A.B.ca_xyz, j);
        d = fd_dist(cai, caj);
        key = (aai & 31u) * 32u + (aaj & 31u);   // queued pairs have aa < 32
        e_lo = st_tab[key]; e_hi = st_tab[key + 1];
        for (uint32_t e = e_lo; e < e_hi; ++e) n_win += (fd_fabsf(d - dist_tab[e]) < A.ca_window) ? 1u : 0u;
        if (A.C.use_tab) {
            // default angle bins: frames + exhaustive tables (fd_geom.h) — same bits as the generic chain, a tenth of the code
            fd_frame Fi = fd_make_frame(fd_load3(A.B.n_xyz, i), cai, fd_load3(A.B.cb_xyz, i));
            fd_frame Fj = fd_make_frame(fd_load3(A.B.n_xyz, j), caj, fd_load3(A.B.cb_xyz, j));
            uint32_t h_ji;
            fd_pair_both_tab(Fi, Fj, aai, aaj, A.C.q, tab, &h, &h_ji);
        } else {
            fd_feature f = fd_pair_feature(fd_load3(A.B.n_xyz, i), cai, fd_load3(A.B.cb_xyz, i), fd_load3(A.B.n_xyz, j), caj, fd_load3(A.B.cb_xyz, j));
            h = fd_hash_pdbtr(aai, aaj, f, A.C.q);
        }
        hit = hash_in_set(A.q_hashes, A.n_hashes, h);
    }
    // one atomic per counter and drain (per-record atomics on two addresses serialise in one L2 channel: that, not the
    // arithmetic, was the kernel time)
    uint32_t incl = n_win;
    for (int off = 1; off < FD_WAVE; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, FD_WAVE);
        if ((int)lane >= off) incl += t;
    }
    const uint32_t tot_win = __shfl(incl, FD_WAVE - 1, FD_WAVE);
    const uint64_t hm = __ballot(hit);
    unsigned long long cbase = 0, fbase = 0;
    if (lane == 0) {
        if (tot_win) cbase = atomicAdd(A.n_cands, (unsigned long long)tot_win);
        if (hm) fbase = atomicAdd(A.n_found, (unsigned long long)__popcll(hm));
    }
    cbase = ((unsigned long long)(uint32_t)__shfl((int)(cbase >> 32), 0, FD_WAVE) << 32) | (uint32_t)__shfl((int)(uint32_t)cbase, 0, FD_WAVE);
    fbase = ((unsigned long long)(uint32_t)__shfl((int)(fbase >> 32), 0, FD_WAVE) << 32) | (uint32_t)__shfl((int)(uint32_t)fbase, 0, FD_WAVE);
    unsigned long long cpos = cbase + (incl - n_win);
    const unsigned long long fpos = fbase + fd_mbcnt(hm);
    // EMIT with capacities: records beyond the caller's buffers are counted but not written (the caller grows and reruns)
    if (EMIT && on && cpos + n_win <= A.cap_cands && (!hit || fpos < A.cap_found)) {
        for (uint32_t e = e_lo; e < e_hi; ++e) {
            if (fd_fabsf(d - dist_tab[e]) < A.ca_window) {
                fd_cand_rec c; c.cand = slot; c.qi = A.aad_qi[e]; c.i = i - r0; c.j = j - r0;
                A.cands[cpos++] = c;
            }
        }
        if (hit) { fd_pair_rec p; p.cand = slot; p.i = i - r0; p.j = j - r0; p.hash = h; A.found[fpos] = p; }
    }
}

#define MP_AAD_LDS 1024
template <bool EMIT>
__global__ __launch_bounds__(FD_WAVE) void k_match_pairs(mp_args A_in) {
    // many queries per launch: this work item's query selects its slice of the concatenated tables (wave-uniform loads)
    mp_args A = A_in;
    if (blockIdx.x < A_in.n_work) {
        const uint32_t tq = A_in.wi_query[blockIdx.x];
        const mp_query_dev Q = A_in.qtab[tq];
        A.q_hashes = A_in.q_hashes + Q.qh_off; A.n_hashes = Q.n_hashes;
        A.aad_start = A_in.aad_start + 1025u * tq; A.aad_dist = A_in.aad_dist + Q.aad_off; A.aad_qi = A_in.aad_qi + Q.aad_off; A.n_aad = Q.n_aad;
        A.aa1_mask = Q.aa1_mask; A.aa2_mask = Q.aa2_mask; A.use_prefilter = Q.use_prefilter; A.ca_window = Q.ca_window;
    }
    __shared__ uint32_t q[2 * FD_WAVE];
    __shared__ uint32_t tab[32];
    __shared__ float s_d_buf[MP_AAD_LDS];
    __shared__ uint32_t s_start[1025];
    const uint32_t w = blockIdx.x;
    if (w >= A.n_work) return;
    // the query's observed (aa_i, aa_j) -> CA distance lists (aa_dist_map, controller/query.rs), grouped by residue-type pair:
    // aad_start[aa_i * 32 + aa_j] .. [+1] indexes the distance / query-residue arrays (host-sorted, stable).  Start table and,
    // for motif-sized queries, the distances live in LDS: per-pair global reads made the scan latency-bound.
    const bool staged = A.n_aad <= MP_AAD_LDS;
    if (threadIdx.x == 0 && A.C.use_tab) fd_fill_bintab(tab);
    for (uint32_t e = threadIdx.x; e < 1025; e += FD_WAVE) s_start[e] = A.aad_start[e];
    if (staged)
        for (uint32_t e = threadIdx.x; e < A.n_aad; e += FD_WAVE) s_d_buf[e] = A.aad_dist[e];
    __syncthreads();
    const float *dist_tab = staged ? s_d_buf : A.aad_dist;
    const uint32_t slot = A.wi_cand[w];
    const uint32_t s = A.cand[slot];
    const uint32_t r0 = A.B.res_off[s], r1 = A.B.res_off[s + 1];
    const uint32_t lane = threadIdx.x;
    // prefilter sets (retrieve.rs:563-602); an empty set on either side switches to the full scan
    bool full = true;
    if (A.use_prefilter) {
        uint64_t any1 = 0, any2 = 0;
        for (uint32_t r = r0 + lane; r < r1 + lane; r += FD_WAVE) {
            bool in = r < r1;
            uint32_t a = in ? A.B.aa[r] : 255u;
            bool stdn = in && a < 20u && (A.resname_std == nullptr || A.resname_std[r]);
            any1 |= __ballot(stdn && ((A.aa1_mask >> a) & 1u));
            any2 |= __ballot(stdn && ((A.aa2_mask >> a) & 1u));
        }
        full = !(any1 && any2);
    }
    const uint32_t i0 = A.wi_i0[w];
    const uint32_t i = i0 + lane;
    const bool in_i = i < r1;
    const uint32_t aai = in_i ? A.B.aa[i] : 255u;
    const bool std_i = aai < 20u && (A.resname_std == nullptr || A.resname_std[i]);
    // get_single_feature (controller/feature.rs:11-24, 84-99) rejects unknown residues / missing CB
    const bool act = in_i && (full || (std_i && ((A.aa1_mask >> aai) & 1u))) && aai != 255u && A.B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (in_i) cai = fd_load3(A.B.ca_xyz, i);
    // partner residue types this lane's residue type has any observation with (aa < 32): one register test per pair
    uint32_t row_mask = 0;
    if (aai < 32u)
        for (uint32_t a2 = 0; a2 < 32u; ++a2) row_mask |= (s_start[aai * 32u + a2 + 1] > s_start[aai * 32u + a2] ? 1u : 0u) << a2;
    uint32_t qn = 0;   // wave-uniform
    // j in blocks of 64: one coalesced load of (aa, CA) per block, then wave-uniform broadcasts (v_readlane)
    for (uint32_t jb = r0; jb < r1; jb += FD_WAVE) {
        const uint32_t jl = jb + lane;
        const bool jin = jl < r1;
        const uint32_t aaj_l = jin ? A.B.aa[jl] : 255u;
        fd_v3 cj = {0.f, 0.f, 0.f};
        if (jin) cj = fd_load3(A.B.ca_xyz, jl);
        bool okj = jin && aaj_l != 255u && A.B.hash_ok[jl];
        if (!full) okj = okj && aaj_l < 20u && (A.resname_std == nullptr || A.resname_std[jl]) && ((A.aa2_mask >> aaj_l) & 1u);
        const uint64_t okm = __ballot(okj);
        const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
        for (uint32_t k = 0; k < nj; ++k) {
            if ((okm >> k) & 1ull) {   // wave-uniform
                const uint32_t j = jb + k;
                const uint32_t aaj = (uint32_t)__builtin_amdgcn_readlane((int)aaj_l, (int)k);
                const fd_v3 caj = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.x), (int)k)),
                                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.y), (int)k)),
                                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.z), (int)k))};
                bool pass = false;
                if (act && i != j && aaj < 32u && ((row_mask >> aaj) & 1u)) {
                    const float d = fd_dist(cai, caj);
                    if (d <= A.cutoff) {
                        // branch-free over the pair's own list: a short-circuit chain costs one LDS round trip per entry
                        const uint32_t e_lo = s_start[aai * 32u + aaj], e_hi = s_start[aai * 32u + aaj + 1];
                        uint32_t any = 0;
                        for (uint32_t e = e_lo; e < e_hi; ++e) any |= (uint32_t)(fd_fabsf(d - dist_tab[e]) < A.ca_window);
                        pass = any != 0;
                    }
                }
                const uint64_t m = __ballot(pass);
                if (m) {
                    if (pass) q[qn + fd_mbcnt(m)] = (lane << 16) | (j - r0);
                    qn += (uint32_t)__popcll(m);
                }
            }
            const bool last = (jb + FD_WAVE >= r1) && (k + 1 == nj);
            while (qn >= FD_WAVE || (last && qn)) {
                __syncthreads();
                uint32_t n = qn < FD_WAVE ? qn : FD_WAVE;
                qn -= n;
                match_drain<EMIT>(A, q + qn, n, slot, r0, i0, s_start, dist_tab, tab);
                __syncthreads();
            }
        }
    }
}

void fd_launch_match_pairs(const mp_args &A, bool emit, hipStream_t st) {
    if (!A.n_work) return;
    if (emit) hipLaunchKernelGGL(k_match_pairs<true>, dim3(A.n_work), dim3(FD_WAVE), 0, st, A);
    else hipLaunchKernelGGL(k_match_pairs<false>, dim3(A.n_work), dim3(FD_WAVE), 0, st, A);
}

// ------------------------------------------------------------------------------------------ Kabsch
// One lane per superposition problem (2..16 points each): the closed-form eigen solve is ~300 f64
// operations, so a batch of thousands of matches is one short launch.  f64 like the reference;
// results are rounded to f32 exactly where the reference rounds (kabsch.rs:537-553).
__global__ __launch_bounds__(64) void k_kabsch(const float *__restrict__ xs, const float *__restrict__ ys, const uint64_t *__restrict__ off,
                                               uint64_t n_prob, float *__restrict__ rmsd_out, float *__restrict__ rot_out,
                                               float *__restrict__ tran_out) {
    uint64_t pidx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= n_prob) return;
    const int IP[9] = {0, 1, 3, 1, 2, 4, 3, 4, 5};
    const int IP2312[4] = {1, 2, 0, 1};
    const double EPSILON = 1.0e-8, TOLERANCE = 0.01, SQRT3 = 1.7320508075688772;
    const uint64_t p0 = off[pidx], p1 = off[pidx + 1];
    const uint64_t n = p1 - p0;
    const float *xf = xs + 3 * p0, *yf = ys + 3 * p0;
    double u[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    float rf = 3.40282347e+38f;
    if (n > 0) {
        double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0}, sz[3] = {0, 0, 0}, xc[3], yc[3], e[3];
        double r[3][3], a[3][3] = {{0}}, b[3][3] = {{0}}, rr[6], ss[6];
        for (uint64_t i = 0; i < n; ++i) {
            double c1[3] = {xf[3 * i], xf[3 * i + 1], xf[3 * i + 2]};
            double c2[3] = {yf[3 * i], yf[3 * i + 1], yf[3 * i + 2]};
            for (int j = 0; j < 3; ++j) { s1[j] += c1[j]; s2[j] += c2[j]; }
            sx[0] += c1[0] * c2[0]; sx[1] += c1[0] * c2[1]; sx[2] += c1[0] * c2[2];
            sy[0] += c1[1] * c2[0]; sy[1] += c1[1] * c2[1]; sy[2] += c1[1] * c2[2];
            sz[0] += c1[2] * c2[0]; sz[1] += c1[2] * c2[1]; sz[2] += c1[2] * c2[2];
        }
        double dn = (double)n;
        for (int j = 0; j < 3; ++j) { xc[j] = s1[j] / dn; yc[j] = s2[j] / dn; }
        for (int j = 0; j < 3; ++j) {
            r[j][0] = sx[j] - s1[0] * s2[j] / dn;
            r[j][1] = sy[j] - s1[1] * s2[j] / dn;
            r[j][2] = sz[j] - s1[2] * s2[j] / dn;
        }
        double det_r = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) +
                       r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
        int m = 0;
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i <= j; ++i) rr[m++] = r[0][i] * r[0][j] + r[1][i] * r[1][j] + r[2][i] * r[2][j];
        double spur = (rr[0] + rr[2] + rr[5]) / 3.0;
        double cof = (((rr[2] * rr[5] - rr[4] * rr[4]) + rr[0] * rr[5] - rr[3] * rr[3]) + rr[0] * rr[2] - rr[1] * rr[1]) / 3.0;
        double det = det_r * det_r;
        e[0] = e[1] = e[2] = spur;
        if (spur > 0.0) {
            double d = spur * spur;
            double h = d - cof;
            double g = (spur * cof - det) / 2.0 - spur * h;
            if (h > 0.0) {
                double sqrth = sqrt(h);
                double disc = h * h * h - g * g;
                if (disc < 0.0) disc = 0.0;
                double sqrt_disc = sqrt(disc);
                double d_ang = fabs(g) > 1e18 ? (g > 0.0 ? 3.14159265358979323846 / 3.0 : 0.0) : atan2(sqrt_disc, -g) / 3.0;
                double cth = sqrth * cos(d_ang);
                double sth = sqrth * SQRT3 * sin(d_ang);
                e[0] = spur + 2.0 * cth;
                e[1] = spur - cth + sth;
                e[2] = spur - cth - sth;
                bool a_failed = false, b_failed = false;
                for (int li = 0; li < 2; ++li) {
                    int l = li == 0 ? 0 : 2;
                    double dl = e[l];
                    ss[0] = (dl - rr[2]) * (dl - rr[5]) - rr[4] * rr[4];
                    ss[1] = (dl - rr[5]) * rr[1] + rr[3] * rr[4];
                    ss[2] = (dl - rr[0]) * (dl - rr[5]) - rr[3] * rr[3];
                    ss[3] = (dl - rr[2]) * rr[3] + rr[1] * rr[4];
                    ss[4] = (dl - rr[0]) * rr[4] + rr[1] * rr[3];
                    ss[5] = (dl - rr[0]) * (dl - rr[2]) - rr[1] * rr[1];
                    for (int k = 0; k < 6; ++k)
                        if (fabs(ss[k]) <= EPSILON) ss[k] = 0.0;
                    double Aa = fabs(ss[0]), Bb = fabs(ss[2]), Cc = fabs(ss[5]);
                    int j = (Aa >= Bb && Aa >= Cc) ? 0 : (Bb >= Cc ? 1 : 2);
                    double dnorm = 0.0;
                    for (int i = 0; i < 3; ++i) { int k = IP[3 * j + i]; a[i][l] = ss[k]; dnorm += ss[k] * ss[k]; }
                    dnorm = dnorm > EPSILON ? 1.0 / sqrt(dnorm) : 0.0;
                    for (int i = 0; i < 3; ++i) a[i][l] *= dnorm;
                }
                double dt = a[0][0] * a[0][2] + a[1][0] * a[1][2] + a[2][0] * a[2][2];
                int m1, mm;
                if (e[0] - e[1] > e[1] - e[2]) { m1 = 2; mm = 0; } else { m1 = 0; mm = 2; }
                double p = 0.0;
                for (int i = 0; i < 3; ++i) { a[i][m1] = a[i][m1] - dt * a[i][mm]; p += a[i][m1] * a[i][m1]; }
                if (p <= TOLERANCE) {
                    int j = 0;
                    p = 1.0;
                    for (int i = 0; i < 3; ++i)
                        if (p < fabs(a[i][mm])) { p = fabs(a[i][mm]); j = i; }
                    int k = IP2312[j], l = IP2312[j + 1];
                    p = sqrt(a[k][mm] * a[k][mm] + a[l][mm] * a[l][mm]);
                    if (p > TOLERANCE) { a[j][m1] = 0.0; a[k][m1] = -a[l][mm] / p; a[l][m1] = a[k][mm] / p; }
                    else a_failed = true;
                } else {
                    p = 1.0 / sqrt(p);
                    for (int i = 0; i < 3; ++i) a[i][m1] *= p;
                }
                if (!a_failed) {
                    a[0][1] = a[1][2] * a[2][0] - a[1][0] * a[2][2];
                    a[1][1] = a[2][2] * a[0][0] - a[2][0] * a[0][2];
                    a[2][1] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
                    for (int l = 0; l < 2; ++l) {
                        double db = 0.0;
                        for (int i = 0; i < 3; ++i) {
                            b[i][l] = r[i][0] * a[0][l] + r[i][1] * a[1][l] + r[i][2] * a[2][l];
                            db += b[i][l] * b[i][l];
                        }
                        db = db > EPSILON ? 1.0 / sqrt(db) : 0.0;
                        for (int i = 0; i < 3; ++i) b[i][l] *= db;
                    }
                    double dot_b = 0.0;
                    for (int i = 0; i < 3; ++i) dot_b += b[i][0] * b[i][1];
                    double pb = 0.0;
                    for (int i = 0; i < 3; ++i) { b[i][1] -= dot_b * b[i][0]; pb += b[i][1] * b[i][1]; }
                    if (pb <= TOLERANCE) {
                        pb = 1.0;
                        int j = 0;
                        for (int i = 0; i < 3; ++i)
                            if (pb < fabs(b[i][0])) { pb = fabs(b[i][0]); j = i; }
                        int k = IP2312[j], l = IP2312[j + 1];
                        pb = sqrt(b[k][0] * b[k][0] + b[l][0] * b[l][0]);
                        if (pb > TOLERANCE) { b[j][1] = 0.0; b[k][1] = -b[l][0] / pb; b[l][1] = b[k][0] / pb; }
                        else b_failed = true;
                    } else {
                        pb = 1.0 / sqrt(pb);
                        for (int i = 0; i < 3; ++i) b[i][1] *= pb;
                    }
                    if (!b_failed) {
                        b[0][2] = b[1][0] * b[2][1] - b[1][1] * b[2][0];
                        b[1][2] = b[2][0] * b[0][1] - b[2][1] * b[0][0];
                        b[2][2] = b[0][0] * b[1][1] - b[0][1] * b[1][0];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j) u[i][j] = b[i][0] * a[j][0] + b[i][1] * a[j][1] + b[i][2] * a[j][2];
                        for (int i = 0; i < 3; ++i) t[i] = yc[i] - (u[i][0] * xc[0] + u[i][1] * xc[1] + u[i][2] * xc[2]);
                    }
                }
            }
        } else {
            for (int i = 0; i < 3; ++i) t[i] = yc[i] - (u[i][0] * xc[0] + u[i][1] * xc[1] + u[i][2] * xc[2]);
        }
        double sum_sq = 0.0;
        for (uint64_t i = 0; i < n; ++i) {
            double X = xf[3 * i], Y = xf[3 * i + 1], Z = xf[3 * i + 2];
            double tr[3] = {u[0][0] * X + u[0][1] * Y + u[0][2] * Z + t[0], u[1][0] * X + u[1][1] * Y + u[1][2] * Z + t[1],
                            u[2][0] * X + u[2][1] * Y + u[2][2] * Z + t[2]};
            for (int j = 0; j < 3; ++j) { double diff = tr[j] - (double)yf[3 * i + j]; sum_sq += diff * diff; }
        }
        rf = (float)sqrt(sum_sq / dn);
        if (rf != rf) rf = 3.40282347e+38f;
    }
    rmsd_out[pidx] = rf;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot_out[9 * pidx + 3 * i + j] = (float)u[i][j];
        tran_out[3 * pidx + i] = (float)t[i];
    }
}

void fd_launch_kabsch(const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot, float *tran, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_kabsch, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, x, y, off, n, rmsd, rot, tran);
}
