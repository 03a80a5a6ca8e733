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

// the slice of the concatenated query tables a work item scans against (its query's hashes, observed-distance lists, prefilter sets)
struct mp_sel {
    const uint32_t *q_hashes; uint32_t n_hashes;
    const uint32_t *aad_start; const float *aad_dist; const uint32_t *aad_qi; uint32_t n_aad;
    uint32_t aa1_mask, aa2_mask; int use_prefilter; float ca_window;
    const uint32_t *iv_start; const float2 *iv;      // large queries: per (aa_i, aa_j) group the merged intervals of distances that pass the window test
    const uint32_t *iv_grp;                          // ... and per group (first interval << 8 | count), the form the work item copies into LDS
    const float *sd_dist; const uint32_t *sd_qi;     // vote mode, optional: the group lists sorted by distance
    const uint32_t *qset; uint32_t qs_mask;          // large queries: the hashes as an open-addressing set (qs_mask = slots - 1; 0: none)
};
// descriptor + hash + output of one surviving (i, j) per lane (full-wave drains of the compaction queue: executed
// divergently per survivor this part — ~3000 instructions with the exact libm chain — was 95 % of the kernel time)
// Nothing is appended here: the pair's hash, (bin pairs whose hash the query holds | window hits << 8) and CA distance go to the chunk's result row,
// the chunk's two totals to cnt — k_mp_offsets turns the totals into record positions, k_mp_emit writes the records.  (One wavefront used to claim its
// record ranges with two returning atomics on the launch's two counters: ~15,000 same-address atomics per launch at ~12 ns each were 150 of the 185 us.)
__device__ __forceinline__ void match_drain(const mp_args &A, const mp_sel &Sx, const uint32_t *q, uint32_t n, uint32_t slot, uint32_t r0, uint32_t r1,
                                            const uint32_t *st_tab, const float *dist_tab, const uint32_t *tab, const uint32_t *qh_lds, unsigned long long v) {
    // is h one of the query's hashes?  From the LDS copy when the work item staged one (a motif query's ~44 hashes: the bisection through global memory
    // was six dependent L2 round trips per drain)
    auto in_query = [&](uint32_t h) -> bool {
        if (qh_lds) {
            uint32_t lo = 0, hi = Sx.n_hashes;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (qh_lds[mid] < h) lo = mid + 1; else hi = mid; }
            return lo < Sx.n_hashes && qh_lds[lo] == h;
        }
        if (Sx.qs_mask) {
            // a whole-structure query holds ~10^5 hashes: the bisection through global memory is 17 dependent round trips per drain (the wavefront waits for its
            // slowest lane); the set is probed in two or three (load factor <= 1/2, linear probing: every lane ends at its hash or at an empty slot)
            uint32_t at = (h * 2654435761u) >> __clz((int)Sx.qs_mask);       // the product's HIGH bits (its low bits repeat the hash's own low fields: long probe runs)
            for (;;) {
                const uint32_t x = Sx.qset[at];
                if (x == h) return true;
                if (x == 0xffffffffu) return false;
                at = (at + 1u) & Sx.qs_mask;
            }
        }
        return hash_in_set(Sx.q_hashes, Sx.n_hashes, h);
    };
    const uint32_t lane = threadIdx.x;
    const bool on = lane < n;
    const uint32_t e0 = on ? q[lane] : 0u;
    const uint32_t i = r0 + (e0 >> 16), j = r0 + (e0 & 0xffffu);      // queue entry: both residues relative to the candidate (a structure holds < 2^16)
    uint32_t aai = 255u, aaj = 255u, n_win = 0, h = 0, hitmask = 0;
    fd_feature feat = {0.f, 0.f, 0.f, 0.f, 0.f};
    fd_v3 cai = {0.f, 0.f, 0.f}, caj = {0.f, 0.f, 0.f};
    float d = 0.f;
    uint32_t key = 0, e_lo = 0, e_hi = 0;
    bool has_feat = true;
    if (on) {
        aai = A.B.aa[i]; aaj = A.B.aa[j];
        cai = fd_load3(A.B.ca_xyz, i); caj = fd_load3(A.B.ca_xyz, j);
        d = fd_dist(cai, caj);
        key = (aai & 31u) * 32u + (aaj & 31u);   // queued pairs have aa < 32
        e_lo = st_tab[key]; e_hi = st_tab[key + 1];
        if (A.mode & 2u)      // the observed-distance window only feeds the candidate pairs: a found-only scan (first pass of a large query) skips it
            for (uint32_t e = e_lo; e < e_hi; ++e) n_win += (fd_fabsf(d - dist_tab[e]) < Sx.ca_window) ? 1u : 0u;
        if (fd_own_descriptor(A.C.q.type)) {
            // encodings with their own descriptor: the pair may still have no feature (CB / point-pair distance, chain ends)
            float f9[FD_NFEAT];
            if (fd_feature_other(A.C.q.type, A.B, r0, r1, i, j, A.cutoff, f9)) {
                h = fd_hash_other(A.C.q.type, f9, A.C.q);
                hitmask = ((A.mode & 1u) && in_query(h)) ? 1u : 0u;
            } else {
                n_win = 0; has_feat = false;
            }
        } else if (!(A.mode & 1u)) {
            // no found triples wanted (second scan of a large query): the PDBTrRosetta descriptor of a pair that passed hash_ok always exists
        } else if (A.C.use_tab && A.n_cfg == 1) {
            // default angle bins: frames + exhaustive tables (fd_geom.h) — same bits as the generic chain, a tenth of the code
            fd_frame Fi = fd_make_frame(fd_load3(A.B.n_xyz, i), cai, fd_load3(A.B.cb_xyz, i));
            fd_frame Fj = fd_make_frame(fd_load3(A.B.n_xyz, j), caj, fd_load3(A.B.cb_xyz, j));
            uint32_t h_ji;
            // the index build's speculative evaluation (fd_pair_both_spec: table lookups on the dot products, exact whenever it answers) with
            // the exact table form behind it for the pairs it declines — the same frames, so the same bits as the build's keys
            if (A.C.use_tab < 2 || !fd_pair_both_spec(Fi, Fj, aai, aaj, A.C.q, tab, tab + 32, &h, &h_ji)) fd_pair_both_tab(Fi, Fj, aai, aaj, A.C.q, tab, &h, &h_ji);
            hitmask = ((A.mode & 1u) && in_query(h)) ? 1u : 0u;
        } else {
            // one descriptor, one hash per bin pair (--multiple-bins: a found triple for every bin pair whose hash the query holds,
            // retrieve.rs:124-131)
            feat = fd_pair_feature(fd_load3(A.B.n_xyz, i), cai, fd_load3(A.B.cb_xyz, i), fd_load3(A.B.n_xyz, j), caj, fd_load3(A.B.cb_xyz, j));
            for (uint32_t k = 0; k < A.n_cfg; ++k) {
                const uint32_t hk = fd_hash_enc(aai, aaj, feat, A.qk[k]);
                if (k == 0) h = hk;
                if ((A.mode & 1u) && in_query(hk)) hitmask |= 1u << k;
            }
        }
        if (!(A.mode & 2u)) n_win = 0;
    }
    if ((A.mode & 32u) && on && has_feat) {
        // rescue votes on the device: the pair's records (one per observed distance in the window) count into the table of the component
        // that mapped the partner residue j
        const uint32_t comp = A.cj_comp[A.mask_off[slot] + (j - r0)], nq = A.vt_qs[slot], nr = r1 - r0;
        if (comp) {
            uint32_t *tabv = A.votes + A.vt_off[slot] + (uint64_t)(comp - 1u) * nq * nr + (i - r0);
            if (Sx.sd_dist) {
                // the group sorted by distance x: (d - x) < window is false ... false, true ... true along it (f32 subtraction is monotone), so the
                // entries inside the window start at the first true and end where |d - x| < window fails again — the same predicate, ~20 of ~200
                uint32_t lo = e_lo, hi = e_hi;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((d - Sx.sd_dist[mid]) < Sx.ca_window) hi = mid; else lo = mid + 1; }
                for (uint32_t e = lo; e < e_hi && fd_fabsf(d - Sx.sd_dist[e]) < Sx.ca_window; ++e) {
                    const uint32_t qi = Sx.sd_qi[e];
                    if (qi < nq) atomicAdd(&tabv[(uint64_t)qi * nr], 1u);
                }
            } else
            for (uint32_t e = e_lo; e < e_hi; ++e)
                if (fd_fabsf(d - dist_tab[e]) < Sx.ca_window) { const uint32_t qi = Sx.aad_qi[e]; if (qi < nq) atomicAdd(&tabv[(uint64_t)qi * nr], 1u); }
        }
    }
    // the chunk's totals: candidate-pair records (window hits) and found triples (one per matching bin pair)
    uint32_t tw = n_win, th = (uint32_t)__builtin_popcount(hitmask);
    for (int off = 32; off > 0; off >>= 1) { tw += __shfl_xor(tw, off, FD_WAVE); th += __shfl_xor(th, off, FD_WAVE); }
    A.res_h[v * FD_WAVE + lane] = h;
    A.res_meta[v * FD_WAVE + lane] = (on ? hitmask : 0u) | (n_win << 8);
    A.res_d[v * FD_WAVE + lane] = d;
    if (lane == 0) A.chunk_cnt[v] = make_uint2(tw, th);
}

#define MP_AAD_LDS 1024
#define MP_SUBQ 64u            /* sub-queues of the chunk queue: one counter each, MP_SUBQ_STRIDE (fdgpu_internal.h) u64 = 128 B apart */
#define MP_SCAN_BLOCKS 64      /* blocks of 64 residues whose activity masks a work item keeps in LDS (longer structures: the two-walk form) */

// many queries per launch: a work item's / chunk's query selects its slice of the concatenated tables (wave-uniform loads).  The selection lives in
// its own few scalars: a modified COPY of the argument block — which holds arrays indexed at run time — is a 480-byte private-memory object per lane,
// written by every wavefront at start (100 MB per launch) and read back field by field.
__device__ __forceinline__ void mp_select(const mp_args &A_in, uint32_t tq, mp_sel &Sx) {
    const mp_query_dev Q = A_in.qtab[tq];
    Sx.q_hashes = A_in.q_hashes + Q.qh_off; Sx.n_hashes = Q.n_hashes;
    Sx.aad_start = A_in.aad_start + 1025u * tq; Sx.aad_dist = A_in.aad_dist + Q.aad_off; Sx.aad_qi = A_in.aad_qi + Q.aad_off; Sx.n_aad = Q.n_aad;
    Sx.aa1_mask = Q.aa1_mask; Sx.aa2_mask = Q.aa2_mask; Sx.use_prefilter = Q.use_prefilter; Sx.ca_window = Q.ca_window;
    Sx.iv_start = A_in.iv_start ? A_in.iv_start + 1025u * tq : nullptr;      // interval offsets are absolute into A.iv
    Sx.iv = A_in.iv;
    Sx.iv_grp = A_in.iv_grp ? A_in.iv_grp + 1024u * tq : nullptr;
    Sx.sd_dist = A_in.sd_dist ? A_in.sd_dist + Q.aad_off : nullptr; Sx.sd_qi = A_in.sd_qi ? A_in.sd_qi + Q.aad_off : nullptr;
    Sx.qset = A_in.qset ? A_in.qset + Q.qs_off : nullptr; Sx.qs_mask = A_in.qset ? Q.qs_mask : 0u;
}
// the hash sets of the launch's large queries: one thread per (query, hash), linear probing from the high bits of hash x 2654435761 (the slots are 0xffffffff on entry;
// a 30-bit hash never is)
__global__ __launch_bounds__(256) void k_mp_qset_build(const mp_query_dev *__restrict__ qtab, const uint32_t *__restrict__ q_hashes, uint32_t *__restrict__ qset) {
    const mp_query_dev Q = qtab[blockIdx.y];
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (!Q.qs_mask || k >= Q.n_hashes) return;
    const uint32_t h = q_hashes[Q.qh_off + k];
    uint32_t *T = qset + Q.qs_off;
    for (uint32_t at = (h * 2654435761u) >> __clz((int)Q.qs_mask);; at = (at + 1u) & Q.qs_mask) {
        const uint32_t old = atomicCAS(&T[at], 0xffffffffu, h);
        if (old == 0xffffffffu || old == h) return;
    }
}
void fd_launch_mp_qset_build(const mp_query_dev *qtab, uint32_t n_queries, uint32_t max_hashes, const uint32_t *q_hashes, uint32_t *qset, hipStream_t st) {
    if (n_queries && max_hashes) hipLaunchKernelGGL(k_mp_qset_build, dim3((max_hashes + 255u) / 256u, n_queries), dim3(256), 0, st, qtab, q_hashes, qset);
}

// The pair scan is two kernels since round 5.  As ONE kernel (scan, and a drain whenever 64 pairs had queued) every wavefront carried the drain's
// ~170 registers — three wavefronts per SIMD — through a scan that is a chain of dependent loads (item -> candidate -> active residues -> tables ->
// partner blocks): the wavefronts waited 73 % of their cycles, the launch took 250 us for 45 us of vector issue (profiles/round5_pmc_query_kernels).
//   k_mp_scan   one wavefront per work item: the window test of every (first residue, partner) pair; survivors leave in CHUNKS of up to 64 packed
//               (i, j) with a header {candidate slot, count | query << 8, first residue, end} — few registers, little LDS: many wavefronts per SIMD.
//   k_mp_drain  one wavefront per chunk: descriptor + hash + records of its pairs (match_drain), the query's hashes and observed distances in LDS.
// COMPACT: every query of the launch observes <= 1,024 distances (motif queries): 16-bit group starts and a 4 KB distance buffer — 7 KB of LDS per
// work item instead of 13.
template <bool COMPACT>
__global__ __launch_bounds__(FD_WAVE) void k_mp_scan(mp_args A_in) {
    const mp_args &A = A_in;
    const uint32_t w = blockIdx.x;
    if (w >= A.n_work) return;
    const uint32_t tq = A_in.wi_query[w];
    mp_sel Sx;
    mp_select(A_in, tq, Sx);
    typedef typename std::conditional<COMPACT, uint16_t, uint32_t>::type start_t;
    __shared__ uint32_t q[2 * FD_WAVE];
    // motif-sized queries: the observed distances (4 KB) and the start table.  Large queries: the first 1,024 merged pass intervals of the query (8 KB;
    // a 300-residue query at a 1 A window has ~700) and, in the start table's place, per (aa_i, aa_j) group (first interval << 8 | count)
    __shared__ __attribute__((aligned(8))) float s_d_buf[COMPACT ? 1024 : 2048];
    __shared__ start_t s_start[1026];
    __shared__ uint32_t s_sel[FD_WAVE];
    const unsigned long long tk0 = A.dbg ? wall_clock64() : 0ull;
    unsigned long long n_vis = 0, n_q = 0;
    const uint32_t slot = A.wi_cand[w];
    const uint32_t lane = threadIdx.x;
    const bool tert = A.C.q.type == FD_HASH_TERTIARY;     // TertiaryInteraction needs no CB (feature.rs:113-160)
    uint32_t r0, r1, n_act = 0, t_sel;
    bool full = true;
    if (A.cinfo) {
        // the candidate's active residues were listed once, by k_mp_items: this item's 64 are one coalesced load
        const uint4 ci = A.cinfo[slot];
        r0 = ci.x; r1 = ci.y; n_act = ci.z & 0x7fffffffu; full = (ci.z >> 31) != 0u;
        t_sel = (A.wi_i0[w] - r0) >> 6;
        if (n_act > 64u * t_sel && 64u * t_sel + lane < n_act) s_sel[lane] = A.act[ci.w + 64u * t_sel + lane];
    } else {
    const uint32_t s = A.cand[slot];
    r0 = A.B.res_off[s]; r1 = A.B.res_off[s + 1];
    // prefilter sets (retrieve.rs:563-602); an empty set on either side switches to the full scan.  ONE walk over the candidate's residue types answers both
    // that and which residues are active: eight blocks of 64 residues are requested together (a block's ballots depend on its loads — one block at a
    // time the walk was six dependent L2 round trips per pass, 30 us per work item, and the work items that only find out that they have nothing to do
    // paid them too), the blocks' ballots go to LDS, and the ranks of the active residues follow from the masks alone.
    unsigned long long *s_mf = reinterpret_cast<unsigned long long *>(s_d_buf), *s_mp = s_mf + MP_SCAN_BLOCKS;      // per block: hashable residues / those also in the
                                                                                                                     // first-residue set (the distance buffer is staged afterwards)
    t_sel = (A.wi_i0[w] - r0) >> 6;
    const uint32_t n_blk = (r1 - r0 + FD_WAVE - 1) / FD_WAVE;
    if (n_blk <= MP_SCAN_BLOCKS) {
        uint64_t any1 = 0, any2 = 0;
        for (uint32_t b0 = 0; b0 < n_blk; b0 += 8) {
            uint32_t aa8[8], ok8[8], sd8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t r = r0 + (b0 + u) * FD_WAVE + lane;
                const bool in = r < r1;
                aa8[u] = in ? A.B.aa[r] : 255u;
                ok8[u] = in ? (tert ? 1u : (uint32_t)A.B.hash_ok[r]) : 0u;
                sd8[u] = in ? (A.resname_std == nullptr ? 1u : (uint32_t)A.resname_std[r]) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (b0 + u < n_blk) {      // wave-uniform
                    const uint32_t a = aa8[u];
                    const bool stdn = a < 20u && sd8[u];
                    const bool in1 = stdn && ((Sx.aa1_mask >> a) & 1u), in2 = stdn && ((Sx.aa2_mask >> a) & 1u), base = a != 255u && ok8[u];
                    any1 |= __ballot(in1); any2 |= __ballot(in2);
                    const uint64_t mf = __ballot(base), mp = __ballot(base && in1);
                    if (lane == 0) { s_mf[b0 + u] = mf; s_mp[b0 + u] = mp; }
                }
            }
        }
        full = !Sx.use_prefilter || !(any1 && any2);
        fd_wave_lds_fence();
        for (uint32_t b = 0; b < n_blk; ++b) {
            const uint64_t m = full ? s_mf[b] : s_mp[b];
            const uint32_t rank = n_act + fd_mbcnt(m);
            if (((m >> lane) & 1ull) && (rank >> 6) == t_sel) s_sel[rank & 63u] = r0 + b * FD_WAVE + lane;
            n_act += (uint32_t)__popcll(m);
            if (n_act >= 64u * (t_sel + 1u)) break;
        }
    } else {      // a structure of more than MP_SCAN_BLOCKS x 64 residues: two walks, a block at a time
        if (Sx.use_prefilter) {
            uint64_t any1 = 0, any2 = 0;
            for (uint32_t r = r0 + lane; r < r1 + lane; r += FD_WAVE) {
                bool in = r < r1;
                uint32_t a = in ? A.B.aa[r] : 255u;
                bool stdn = in && a < 20u && (A.resname_std == nullptr || A.resname_std[r]);
                any1 |= __ballot(stdn && ((Sx.aa1_mask >> a) & 1u));
                any2 |= __ballot(stdn && ((Sx.aa2_mask >> a) & 1u));
            }
            full = !(any1 && any2);
        }
        for (uint32_t rb = r0; rb < r1; rb += FD_WAVE) {
            const uint32_t r = rb + lane;
            const bool in = r < r1;
            const uint32_t a = in ? A.B.aa[r] : 255u;
            const bool stdn = in && a < 20u && (A.resname_std == nullptr || A.resname_std[r]);
            const bool ac = in && (full || (stdn && ((Sx.aa1_mask >> a) & 1u))) && a != 255u && (tert || A.B.hash_ok[r]);
            const uint64_t m = __ballot(ac);
            const uint32_t rank = n_act + fd_mbcnt(m);
            if (ac && (rank >> 6) == t_sel) s_sel[rank & 63u] = r;
            n_act += (uint32_t)__popcll(m);
            if (n_act >= 64u * (t_sel + 1u)) break;
        }
    }
    fd_wave_lds_fence();      // the masks' buffer is the distance buffer
    }
    if (n_act <= 64u * t_sel) {             // (wave-uniform) nothing left for this tile: before any table is staged
        if (A.dbg && threadIdx.x == 0) { atomicAdd(&A.dbg[2], 1ull); atomicAdd(&A.dbg[3], wall_clock64() - tk0); }
        return;
    }
    const uint32_t n_here = n_act - 64u * t_sel < FD_WAVE ? n_act - 64u * t_sel : FD_WAVE;
    // the query's observed (aa_i, aa_j) -> CA distance lists (aa_dist_map, controller/query.rs), grouped by residue-type pair:
    // aad_start[aa_i * 32 + aa_j] .. [+1] indexes the distance / query-residue arrays (host-sorted, stable).  Start table and,
    // for motif-sized queries, the distances live in LDS: per-pair global reads made the scan latency-bound.
    // partner residues in groups of four blocks of 64: one coalesced load of (aa, CA, filters) per block and lane, the group's loads issued together — and
    // the first group's HERE, before the tables are staged — (a block at a time, an item of 512 partners was eight dependent round trips through L2 on top of
    // the item -> candidate -> active residues -> tables -> first residues chain: 94 us per launch for 5 us of arithmetic)
    const uint32_t j_lo = A.j_span ? A.wi_j0[w] : r0;
    const uint32_t j_hi = A.j_span ? (j_lo + A.j_span < r1 ? j_lo + A.j_span : r1) : r1;
    const uint32_t mbit0 = A.cj_mask ? A.mask_off[slot] : 0u;
    uint32_t pa[4], pok[4];
    fd_v3 pc[4];
#define MP_LOAD_GROUP(g0)                                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                                        \
        const uint32_t jl = (g0) + (uint32_t)u * FD_WAVE + lane;                                                                           \
        const bool jin = jl < j_hi;                                                                                                        \
        pa[u] = jin ? (uint32_t)A.B.aa[jl] : 255u;                                                                                         \
        pc[u] = {0.f, 0.f, 0.f};                                                                                                           \
        if (jin) pc[u] = fd_load3(A.B.ca_xyz, jl);                                                                                         \
        uint32_t ok = jin ? (tert ? 1u : (uint32_t)A.B.hash_ok[jl]) : 0u;                                                                  \
        if (!full && jin && A.resname_std) ok &= (uint32_t)A.resname_std[jl] ? 1u : 0u;                                                    \
        if (A.cj_mask && jin) { const uint32_t bit = mbit0 + (jl - r0); ok &= (A.cj_mask[bit >> 5] >> (bit & 31u)) & 1u; }                  \
        pok[u] = ok;                                                                                                                       \
    }
    MP_LOAD_GROUP(j_lo)
    const bool staged = COMPACT || Sx.n_aad <= MP_AAD_LDS;
    const bool big = !COMPACT && !staged && Sx.iv_start != nullptr;      // large query: the only table a work item stages is the dense interval table (8 KB, 16
                                                                          // independent loads per lane)
    if (!big) for (uint32_t e = threadIdx.x; e < 1025; e += FD_WAVE) s_start[e] = (start_t)Sx.aad_start[e];
    if (staged)
        for (uint32_t e = threadIdx.x; e < Sx.n_aad; e += FD_WAVE) s_d_buf[e] = Sx.aad_dist[e];
    uint32_t iv_base = 0;
    if (big) {
        iv_base = Sx.iv_start[0];
        const uint32_t n_iv = Sx.iv_start[1024] - iv_base;
#pragma unroll
        for (uint32_t g = 0; g < 1024; g += FD_WAVE) s_start[g + threadIdx.x] = (start_t)Sx.iv_grp[g + threadIdx.x];
        for (uint32_t e = threadIdx.x; e < n_iv && e < 1024u; e += FD_WAVE) reinterpret_cast<float2 *>(s_d_buf)[e] = Sx.iv[iv_base + e];
    }
    __syncthreads();
    const bool act = lane < n_here;
    const uint32_t i = act ? s_sel[lane] : r0;
    const uint32_t aai = act ? A.B.aa[i] : 255u;
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (act) cai = fd_load3(A.B.ca_xyz, i);
    // partner residue types this lane's residue type has any observation with (aa < 32): one register test per pair
    uint32_t row_mask = 0;
    if (aai < 32u) {
        if (big) for (uint32_t a2 = 0; a2 < 32u; ++a2) row_mask |= ((s_start[aai * 32u + a2] & 255u) ? 1u : 0u) << a2;
        else for (uint32_t a2 = 0; a2 < 32u; ++a2) row_mask |= (s_start[aai * 32u + a2 + 1] > s_start[aai * 32u + a2] ? 1u : 0u) << a2;
    }
    // the loop's two thresholds in VECTOR registers: a uniform value the compiler keeps "in the arguments" is a scalar memory load + wait per partner residue
    float d2_max_v = A.C.d2_max, ca_window_v = Sx.ca_window;
    asm volatile("" : "+v"(d2_max_v), "+v"(ca_window_v));
    // a full queue leaves as one chunk: {candidate slot, pairs | query << 8, first residue, end} + 64 packed (i - r0) << 16 | (j - r0)
    // (64 sub-queues, the item's is w mod 64: chunk slots are claimed with an atomic, and ~8,000 claims on ONE counter took 50 us of the launch)
    auto push = [&](const uint32_t *src, uint32_t n) {
        const uint32_t sq = w & (MP_SUBQ - 1u);
        uint32_t k = 0;
        if (lane == 0) k = (uint32_t)atomicAdd(A.q_cnt + MP_SUBQ_STRIDE * sq, 1ull);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        if (k < A.cap_subq) {      // beyond the sub-queue: counted only (the caller grows the queue and reruns)
            const uint64_t cidx = (uint64_t)sq * A.cap_subq + k;
            if (lane < n) A.chunk_ij[cidx * FD_WAVE + lane] = src[lane];
            if (lane == 0) A.chunk_hdr[cidx] = make_uint4(slot, n | (tq << 8), r0, r1);
        }
        n_q += n;
    };
    uint32_t qn = 0;   // wave-uniform
    // per block: the partners that can pass at all (their own filters: ~1 in 5 for a motif query's residue types) are walked, wave-uniform broadcasts (v_readlane)
    for (uint32_t g0 = j_lo; g0 < j_hi; g0 += 4u * FD_WAVE) {
        if (g0 != j_lo) { MP_LOAD_GROUP(g0) }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t jb = g0 + (uint32_t)u * FD_WAVE;
        if (jb >= j_hi) break;      // (wave-uniform)
        const uint32_t aaj_l = pa[u];
        const fd_v3 cj = pc[u];
        bool okj = aaj_l != 255u && pok[u];
        if (!full) okj = okj && aaj_l < 20u && ((Sx.aa2_mask >> aaj_l) & 1u);
        const uint64_t okm = __ballot(okj);
        uint64_t todo = okm;
        if (A.dbg) n_vis += (unsigned long long)__popcll(okm);
        while (todo) {
            const uint32_t k = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t j = jb + k;
            const uint32_t aaj = (uint32_t)__builtin_amdgcn_readlane((int)aaj_l, (int)k);
            const fd_v3 caj = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.x), (int)k)),
                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.y), (int)k)),
                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj.z), (int)k))};
            bool pass = false;
            if (act && i != j && aaj < 32u && ((row_mask >> aaj) & 1u)) {
                // the cutoff on the SQUARED distance (d2_max = the largest f32 whose square root is <= the cutoff: the same decision as sqrt(d2) <= cutoff
                // without a correctly rounded square root per pair test)
                const float d2 = fd_dist2(cai, caj);
                if (d2 <= d2_max_v) {
                    const float d = fd_sqrtf(d2);
                    // branch-free over the pair's own list: a short-circuit chain costs one LDS round trip per entry
                    uint32_t any = 0;
                    if (big) {
                        // a whole-structure query observes ~100 distances per residue-type pair and keeps them in global memory.  The
                        // window test only asks whether ANY of them is within the window of d: the host merged, per pair of types, the
                        // float intervals [lo_x, hi_x] = {d : |d - x| < window} of all observed x (exact: |d - x| is monotone in d on
                        // either side of x) — usually ONE interval per pair of types — so the test is one offset load and one or two
                        // independent interval loads instead of a walk over the list
                        const uint32_t gw = s_start[aai * 32u + aaj], v0 = gw >> 8, vn = gw & 255u;      // the group's intervals, from LDS
                        if (vn < 255u && v0 + vn <= 1024u) {
                            for (uint32_t e = 0; e < vn; ++e) { const float2 w2 = reinterpret_cast<const float2 *>(s_d_buf)[v0 + e]; any |= (uint32_t)(d >= w2.x && d <= w2.y); }
                        } else {
                            const uint32_t v_lo = Sx.iv_start[aai * 32u + aaj], v_hi = Sx.iv_start[aai * 32u + aaj + 1];
                            for (uint32_t e = v_lo; e < v_hi; ++e) { const float2 w2 = Sx.iv[e]; any |= (uint32_t)(d >= w2.x && d <= w2.y); }
                        }
                    } else {
                        const uint32_t e_lo = s_start[aai * 32u + aaj], e_hi = s_start[aai * 32u + aaj + 1];
                        if (staged) for (uint32_t e = e_lo; e < e_hi; ++e) any |= (uint32_t)(fd_fabsf(d - s_d_buf[e]) < ca_window_v);
                        else for (uint32_t e = e_lo; e < e_hi; ++e) any |= (uint32_t)(fd_fabsf(d - Sx.aad_dist[e]) < ca_window_v);
                    }
                    pass = any != 0;
                }
            }
            const uint64_t m = __ballot(pass);
            if (m) {
                if (pass) q[qn + fd_mbcnt(m)] = ((i - r0) << 16) | (j - r0);
                qn += (uint32_t)__popcll(m);
                if (qn >= FD_WAVE) {      // (wave-scope fences: one wavefront per workgroup, the queue is LDS)
                    fd_wave_lds_fence();
                    qn -= FD_WAVE;
                    push(q + qn, FD_WAVE);
                    fd_wave_lds_fence();
                }
            }
        }
    }
    }
#undef MP_LOAD_GROUP
    if (qn) { fd_wave_lds_fence(); push(q, qn); }
    if (A.dbg && threadIdx.x == 0) {
        atomicAdd(&A.dbg[0], 1ull); atomicAdd(&A.dbg[1], wall_clock64() - tk0); atomicAdd(&A.dbg[6], n_vis); atomicAdd(&A.dbg[7], n_q);
    }
}

// the chunks of the 64 sub-queues in one order: v in [0, total) -> the chunk's slot.  Every lane loads one sub-queue's count; returns the total
__device__ __forceinline__ uint32_t mp_chunk_prefix(const mp_args &A, uint32_t lane, uint32_t &incl) {
    const unsigned long long cn = A.q_cnt[MP_SUBQ_STRIDE * lane];
    uint32_t x = cn > A.cap_subq ? A.cap_subq : (uint32_t)cn;
    for (int off = 1; off < FD_WAVE; off <<= 1) { const uint32_t t = __shfl_up(x, off, FD_WAVE); if ((int)lane >= off) x += t; }
    incl = x;
    return (uint32_t)__shfl((int)x, FD_WAVE - 1, FD_WAVE);
}
__device__ __forceinline__ uint64_t mp_chunk_slot(const mp_args &A, uint32_t incl, uint32_t v) {
    const uint32_t sq = (uint32_t)__popcll(__ballot(incl <= v));      // sub-queues that end at or before v (wave-uniform)
    const uint32_t before = sq ? (uint32_t)__shfl((int)incl, (int)sq - 1, FD_WAVE) : 0u;
    return (uint64_t)sq * A.cap_subq + (v - before);
}

// one wavefront per chunk of queued pairs (grid-stride: the chunk count stays on the device)
__global__ __launch_bounds__(FD_WAVE) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_mp_drain(mp_args A_in) {
    const mp_args &A = A_in;
    __shared__ uint32_t tab[64];      // [0, 27) the exact bin tables (bit patterns), [32, 59) the same with clamped float thresholds for the speculative path
    __shared__ float s_d[MP_AAD_LDS];
    __shared__ uint32_t s_qh[MP_QH_LDS];
    const uint32_t lane = threadIdx.x;
    uint32_t incl;
    const uint32_t nc = mp_chunk_prefix(A, lane, incl);
    if (blockIdx.x >= nc) return;
    tab[lane] = A.bintab[lane];
    for (uint32_t v = blockIdx.x; v < nc; v += gridDim.x) {
        const unsigned long long td = A.dbg ? wall_clock64() : 0ull;
        const uint64_t c = mp_chunk_slot(A, incl, v);
        const uint4 hd = A.chunk_hdr[c];
        const uint32_t slot = hd.x, n = hd.y & 255u, tq = hd.y >> 8, r0 = hd.z, r1 = hd.w;
        mp_sel Sx;
        mp_select(A_in, tq, Sx);
        // a motif query's ~44 hashes and its observed distances in LDS: the membership bisection through global memory was six dependent L2 round trips
        // per drain, every list entry of the window walks one more
        const bool staged = Sx.n_aad <= MP_AAD_LDS, staged_h = Sx.n_hashes <= MP_QH_LDS;
        __syncthreads();      // the previous chunk's readers
        if (staged_h) for (uint32_t e = lane; e < Sx.n_hashes; e += FD_WAVE) s_qh[e] = Sx.q_hashes[e];
        if (staged) for (uint32_t e = lane; e < Sx.n_aad; e += FD_WAVE) s_d[e] = Sx.aad_dist[e];
        __syncthreads();
        match_drain(A, Sx, A.chunk_ij + c * FD_WAVE, n, slot, r0, r1, Sx.aad_start, staged ? s_d : Sx.aad_dist, tab, staged_h ? s_qh : nullptr, v);
        if (A.dbg && lane == 0) { atomicAdd(&A.dbg[4], 1ull); atomicAdd(&A.dbg[5], wall_clock64() - td); }
    }
}

// chunk totals -> first record of every chunk (exclusive sums in chunk order) and the launch's two totals.  One workgroup: a motif batch has ~8,000
// chunks, a whole-structure query's scan a few ten thousand.
__global__ __launch_bounds__(1024) void k_mp_offsets(mp_args A) {
    __shared__ unsigned long long s_w[16], s_h[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t incl;
    const uint32_t nc = mp_chunk_prefix(A, lane, incl);
    // passes of 8,192 chunks: thread t owns eight neighbouring chunks of the pass — their totals are loaded together and stay in registers —, one scan over the
    // 1,024 threads' sums, then the chunks' positions (a tile of 1,024 chunks per round was eight rounds of three barriers for a motif batch: 19 us; ONE pass with
    // `per` chunks per thread walked a whole-structure scan's 40-60 k chunks through a loop of dependent loads: 0.22 ms per launch, two launches per query)
    unsigned long long run_w = 0, run_h = 0;
    for (uint32_t p0 = 0; p0 < nc; p0 += 8192u) {
        const uint32_t v0 = p0 + t * 8u;
        uint2 c8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c8[u] = v0 + u < nc ? A.chunk_cnt[v0 + u] : make_uint2(0u, 0u);
        unsigned long long xw = 0, xh = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { xw += c8[u].x; xh += c8[u].y; }
        const unsigned long long mw = xw, mh = xh;
        for (int off = 1; off < FD_WAVE; off <<= 1) {
            const unsigned long long tw = __shfl_up(xw, off, FD_WAVE), th = __shfl_up(xh, off, FD_WAVE);
            if ((int)lane >= off) { xw += tw; xh += th; }
        }
        __syncthreads();      // (the pass before has read the wave totals)
        if (lane == 63u) { s_w[wv] = xw; s_h[wv] = xh; }
        __syncthreads();
        unsigned long long bw = run_w + xw - mw, bh = run_h + xh - mh;      // exclusive inside the wavefront
        unsigned long long tot_w = 0, tot_h = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) { const unsigned long long a = s_w[k], c = s_h[k]; if (k < wv) { bw += a; bh += c; } tot_w += a; tot_h += c; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v0 + u < nc) { A.chunk_base[v0 + u] = make_ulonglong2(bw, bh); bw += c8[u].x; bh += c8[u].y; }
        run_w += tot_w; run_h += tot_h;
    }
    if (t == 0u) { *A.n_cands = run_w; *A.n_found = run_h; A.n_found[2] = nc; }
}

// the records of one chunk per wavefront: candidate pairs (one per observed distance inside the pair's window) and found triples (one per bin pair
// whose hash the query holds), at the positions k_mp_offsets gave the chunk.  Records beyond the caller's buffers are not written (the totals say so:
// the caller grows the buffers and repeats the launch).
__global__ __launch_bounds__(FD_WAVE) void k_mp_emit(mp_args A_in) {
    const mp_args &A = A_in;
    const uint32_t lane = threadIdx.x;
    uint32_t incl;
    const uint32_t nc = mp_chunk_prefix(A, lane, incl);
    for (uint32_t v = blockIdx.x; v < nc; v += gridDim.x) {
        const uint2 cn = A.chunk_cnt[v];
        if (!(cn.x | cn.y)) continue;      // (wave-uniform)
        const uint64_t c = mp_chunk_slot(A, incl, v);
        const uint4 hd = A.chunk_hdr[c];
        const uint32_t slot = hd.x, n = hd.y & 255u, tq = hd.y >> 8, r0 = hd.z;
        const ulonglong2 base = A.chunk_base[v];
        const bool on = lane < n;
        const uint32_t e0 = on ? A.chunk_ij[c * FD_WAVE + lane] : 0u;
        const uint32_t i = r0 + (e0 >> 16), j = r0 + (e0 & 0xffffu);
        const uint32_t meta = on ? A.res_meta[(uint64_t)v * FD_WAVE + lane] : 0u, h = A.res_h[(uint64_t)v * FD_WAVE + lane];
        const float d = A.res_d[(uint64_t)v * FD_WAVE + lane];
        const uint32_t n_win = meta >> 8, hitmask = meta & 255u, n_hit = (uint32_t)__builtin_popcount(hitmask);
        uint32_t wi = n_win, hi = n_hit;
        for (int off = 1; off < FD_WAVE; off <<= 1) {
            const uint32_t tw = __shfl_up(wi, off, FD_WAVE), th = __shfl_up(hi, off, FD_WAVE);
            if ((int)lane >= off) { wi += tw; hi += th; }
        }
        unsigned long long cpos = base.x + (wi - n_win);
        const unsigned long long fpos = base.y + (hi - n_hit);
        const bool fits = on && cpos + n_win <= A.cap_cands && (!n_hit || fpos + n_hit <= A.cap_found);
        if (fits && n_win) {
            mp_sel Sx;
            mp_select(A_in, tq, Sx);
            const uint32_t key = ((uint32_t)A.B.aa[i] & 31u) * 32u + ((uint32_t)A.B.aa[j] & 31u);
            const uint32_t e_lo = Sx.aad_start[key], e_hi = Sx.aad_start[key + 1];
            for (uint32_t e = e_lo; e < e_hi; ++e)
                if (fd_fabsf(d - Sx.aad_dist[e]) < Sx.ca_window) {
                    fd_cand_rec cr; cr.cand = slot; cr.qi = Sx.aad_qi[e]; cr.i = i - r0; cr.j = j - r0;
                    A.cands[cpos++] = cr;
                }
        }
        if (fits && n_hit) {
            fd_pair_rec p; p.cand = slot; p.i = i - r0; p.j = j - r0;
            if (A.n_cfg == 1 || hitmask == 1u) { p.hash = h; A.found[fpos] = p; }
            else {
                // --multiple-bins: the hash of every bin pair that matched, from the pair's descriptor again (retrieve.rs:124-131)
                const uint32_t aai = A.B.aa[i], aaj = A.B.aa[j];
                const fd_feature feat = fd_pair_feature(fd_load3(A.B.n_xyz, i), fd_load3(A.B.ca_xyz, i), fd_load3(A.B.cb_xyz, i),
                                                        fd_load3(A.B.n_xyz, j), fd_load3(A.B.ca_xyz, j), fd_load3(A.B.cb_xyz, j));
                unsigned long long fp = fpos;
                for (uint32_t k = 0; k < A.n_cfg; ++k)
                    if ((hitmask >> k) & 1u) { p.hash = k == 0 ? h : fd_hash_enc(aai, aaj, feat, A.qk[k]); A.found[fp++] = p; }
            }
        }
    }
}

// candidate pairs -> (slot << 16 | j, qi << 16 | i): 8 bytes instead of 16 for the copy back, and sortable by (slot, partner residue)
__global__ void k_pack_cands(const fd_cand_rec *__restrict__ c, uint64_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const fd_cand_rec r = c[k];
    key[k] = (r.cand << 16) | (r.j & 0xffffu);
    val[k] = (r.qi << 16) | (r.i & 0xffffu);
}
void fd_launch_pack_cands(const fd_cand_rec *c, uint64_t n, uint32_t *key, uint32_t *val, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_pack_cands, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, n, key, val);
}
// one wavefront per row of the rescue-vote table: largest count, how many residues hold it, one of them (the smallest index; the rescue only
// uses it when it is the only one, retrieve.rs:498-511)
__global__ __launch_bounds__(256) void k_vote_rows(const uint32_t *__restrict__ votes, const uint64_t *__restrict__ row_off, const uint32_t *__restrict__ row_len,
                                                   uint64_t n_rows, fd_vote_row *__restrict__ out) {
    const uint64_t row = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t lane = threadIdx.x & 63u, n = row_len[row];
    const uint32_t *v = votes + row_off[row];
    uint32_t mx = 0, nmx = 0, arg = 0xffffffffu;
    for (uint32_t k = lane; k < n; k += FD_WAVE) {
        const uint32_t x = v[k];
        if (x > mx) { mx = x; nmx = 1; arg = k; }
        else if (x == mx && x) { ++nmx; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t omx = __shfl_xor(mx, off, FD_WAVE), onmx = __shfl_xor(nmx, off, FD_WAVE), oarg = __shfl_xor(arg, off, FD_WAVE);
        if (omx > mx) { mx = omx; nmx = onmx; arg = oarg; }
        else if (omx == mx && mx) { nmx += onmx; arg = oarg < arg ? oarg : arg; }
    }
    if (lane == 0) { fd_vote_row r; r.mx = mx; r.nmx = mx ? nmx : 0u; r.arg = mx ? arg : 0u; out[row] = r; }
}
void fd_launch_vote_rows(const uint32_t *votes, const uint64_t *row_off, const uint32_t *row_len, uint64_t n_rows, fd_vote_row *out, hipStream_t st) {
    if (n_rows) hipLaunchKernelGGL(k_vote_rows, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, votes, row_off, row_len, n_rows, out);
}
// found triples into (slot, i, j) order on the device (two stable radix sorts: by i << 16 | j, then by slot), for scans that return ~10^5 of them
__global__ void k_found_key_ij(const fd_pair_rec *__restrict__ f, uint64_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { key[k] = (f[k].i << 16) | (f[k].j & 0xffffu); val[k] = (uint32_t)k; }
}
__global__ void k_found_key_slot(const fd_pair_rec *__restrict__ f, const uint32_t *__restrict__ val, uint64_t n, uint32_t *__restrict__ key) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) key[k] = f[val[k]].cand;
}
__global__ void k_found_gather(const fd_pair_rec *__restrict__ f, const uint32_t *__restrict__ val, uint64_t n, fd_pair_rec *__restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = f[val[k]];
}
void fd_launch_found_key_ij(const fd_pair_rec *f, uint64_t n, uint32_t *key, uint32_t *val, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_found_key_ij, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, n, key, val);
}
void fd_launch_found_key_slot(const fd_pair_rec *f, const uint32_t *val, uint64_t n, uint32_t *key, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_found_key_slot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, val, n, key);
}
void fd_launch_found_gather(const fd_pair_rec *f, const uint32_t *val, uint64_t n, fd_pair_rec *out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_found_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, val, n, out);
}
// The scan's work items, written where they are read: one wavefront per candidate, an item per (64-residue tile i, span of j_span partner
// residues) in the order the host loop made them (tiles outer, spans inner).  j_span = 0: one span = the whole structure.
// Also here, once per CANDIDATE instead of once per work item: which of its residues can be the first residue of a pair at all (prefilter_amino_acid,
// retrieve.rs:563-602: type in the query's first-residue set, standard name — or every hashable residue when a set is empty / the prefilter is off),
// compacted into act[64 * first item ...], and cinfo[k] = {first residue, end, active residues | full << 31, first entry of its list}.  A work item then
// reads its 64 residues with one coalesced load, and the items beyond a candidate's last active tile return after two loads.
__global__ __launch_bounds__(256) void k_mp_items(const uint32_t *__restrict__ db_res_off, const uint32_t *__restrict__ cand, uint32_t n_cand,
                                                  const uint32_t *__restrict__ wbase, const uint32_t *__restrict__ cq, uint32_t j_span, uint32_t *__restrict__ wc,
                                                  uint32_t *__restrict__ wi, uint32_t *__restrict__ wq, uint32_t *__restrict__ wj, const uint8_t *__restrict__ aa,
                                                  const uint8_t *__restrict__ hash_ok, const uint8_t *__restrict__ resname_std, int tert,
                                                  const mp_query_dev *__restrict__ qtab, uint4 *__restrict__ cinfo, uint32_t *__restrict__ act) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= n_cand) return;
    const uint32_t s = cand[k], r0 = db_res_off[s], r1 = db_res_off[s + 1], len = r1 - r0;
    const uint32_t base = wbase[k], n = wbase[k + 1] - base, q = cq[k];
    const uint32_t spans = j_span ? (len + j_span - 1u) / j_span : (len ? 1u : 0u);
    for (uint32_t x = lane; x < n; x += 64u) {
        const uint32_t ti = x / spans, sj = x - ti * spans;
        wc[base + x] = k; wi[base + x] = r0 + ti * FD_WAVE; wq[base + x] = q; wj[base + x] = r0 + sj * j_span;
    }
    if (!cinfo) return;
    const mp_query_dev Q = qtab[q];
    const uint32_t n_blk = (len + FD_WAVE - 1u) / FD_WAVE;
    uint32_t *list = act + 64ull * base;
    uint64_t any1 = 0, any2 = 0;
    uint32_t n_full = 0, n_pref = 0;      // both lists are written (the prefiltered one behind the full one's worst case is not known yet): decide afterwards
    // pass 1: is either prefilter set empty?  (eight blocks of residue types requested together)
    if (Q.use_prefilter)
        for (uint32_t b0 = 0; b0 < n_blk; b0 += 8) {
            uint32_t a8[8], s8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t r = r0 + (b0 + u) * FD_WAVE + lane;
                a8[u] = r < r1 ? aa[r] : 255u;
                s8[u] = r < r1 ? (resname_std ? (uint32_t)resname_std[r] : 1u) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool stdn = a8[u] < 20u && s8[u];
                any1 |= __ballot(stdn && ((Q.aa1_mask >> a8[u]) & 1u));
                any2 |= __ballot(stdn && ((Q.aa2_mask >> a8[u]) & 1u));
            }
        }
    const bool full = !Q.use_prefilter || !(any1 && any2);
    (void)n_full; (void)n_pref;
    uint32_t n_act = 0;
    for (uint32_t b0 = 0; b0 < n_blk; b0 += 8) {
        uint32_t a8[8], o8[8], s8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t r = r0 + (b0 + u) * FD_WAVE + lane;
            const bool in = r < r1;
            a8[u] = in ? aa[r] : 255u;
            o8[u] = in ? (tert ? 1u : (uint32_t)hash_ok[r]) : 0u;
            s8[u] = in ? (resname_std ? (uint32_t)resname_std[r] : 1u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t a = a8[u];
            const bool ac = a != 255u && o8[u] && (full || (a < 20u && s8[u] && ((Q.aa1_mask >> a) & 1u)));
            const uint64_t m = __ballot(ac);
            if (ac) list[n_act + fd_mbcnt(m)] = r0 + (b0 + u) * FD_WAVE + lane;
            n_act += (uint32_t)__popcll(m);
        }
    }
    if (lane == 0) cinfo[k] = make_uint4(r0, r1, n_act | (full ? 0x80000000u : 0u), 64u * base);
}
void fd_launch_mp_items(const uint32_t *db_res_off, const uint32_t *cand, uint32_t n_cand, const uint32_t *wbase, const uint32_t *cq, uint32_t j_span, uint32_t *wc,
                        uint32_t *wi, uint32_t *wq, uint32_t *wj, hipStream_t st, const uint8_t *aa, const uint8_t *hash_ok, const uint8_t *resname_std, int tert,
                        const mp_query_dev *qtab, void *cinfo, uint32_t *act) {
    if (n_cand) hipLaunchKernelGGL(k_mp_items, dim3((n_cand + 3u) / 4u), dim3(256), 0, st, db_res_off, cand, n_cand, wbase, cq, j_span, wc, wi, wq, wj, aa, hash_ok,
                                   resname_std, tert, qtab, (uint4 *)cinfo, act);
}
void fd_launch_match_pairs(const mp_args &A, hipStream_t st) {
    if (!A.n_work) return;
    if (A.compact) hipLaunchKernelGGL(k_mp_scan<true>, dim3(A.n_work), dim3(FD_WAVE), 0, st, A);
    else hipLaunchKernelGGL(k_mp_scan<false>, dim3(A.n_work), dim3(FD_WAVE), 0, st, A);
    const uint64_t cap = (uint64_t)A.cap_subq * MP_SUBQ;
    const uint32_t grid = cap < 8192u ? (cap ? (uint32_t)cap : 1u) : 8192u;
    hipLaunchKernelGGL(k_mp_drain, dim3(grid), dim3(FD_WAVE), 0, st, A);
    hipLaunchKernelGGL(k_mp_offsets, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_mp_emit, dim3(grid), dim3(FD_WAVE), 0, st, A);
}

// ------------------------------------------------------------------------------------------ superposition + similarity metrics
// Optimal rigid superposition of moving points x onto fixed points y (what KabschSuperimposer::run returns: rotation U, translation
// t, rmsd; src/structure/kabsch.rs:47-95) and the similarity metrics of the superposed pair (src/structure/metrics.rs:62-251).
//
// One WAVEFRONT per problem.  The O(n) parts — centroids and the 3x3 cross-covariance, the residual pass, the per-point distances
// of TM-score / GDT — and the O(n^2) nearest-neighbour scans of Chamfer / Hausdorff are strided over the 64 lanes and combined with
// a fixed-order butterfly, so a result does not depend on how many problems share the launch.  The 3x3 algebra in between is tiny
// and runs redundantly in every lane (uniform control flow, no LDS).
//
// The rotation is Kabsch's eigen construction, the same one the reference uses (the eigenvectors of R^T R for the largest and the
// smallest eigenvalue, Gram-Schmidt, b = R a, U = b a^T), because its answers for rank-deficient inputs — two-point matches, collinear
// or duplicated points — are part of the output contract: where the construction gives up (no second direction: |a2|^2 <= 0.01
// twice) the reference reports the identity and a zero translation, and so does this kernel.  f64 throughout, f32 at the end, NaN
// rmsd -> f32::MAX (kabsch.rs:537-553).
struct sp_sym3 { double xx, xy, yy, xz, yz, zz; };   // packed symmetric 3x3 (upper triangle by columns)

template <int G = 64>
__device__ __forceinline__ double sp_wave_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// eigenvalues of a positive semi-definite symmetric 3x3 (trigonometric form of the cubic), descending
__device__ __forceinline__ bool sp_eigenvalues(const sp_sym3 &M, double det_r, double ev[3]) {
    const double mean = (M.xx + M.yy + M.zz) / 3.0;
    ev[0] = ev[1] = ev[2] = mean;
    if (!(mean > 0.0)) return false;
    const double minors = (((M.yy * M.zz - M.yz * M.yz) + M.xx * M.zz - M.xz * M.xz) + M.xx * M.yy - M.xy * M.xy) / 3.0;
    const double h = mean * mean - minors;
    if (!(h > 0.0)) return false;
    const double g = (mean * minors - det_r * det_r) / 2.0 - mean * h;
    double disc = h * h * h - g * g;
    if (disc < 0.0) disc = 0.0;
    const double third = fabs(g) > 1e18 ? (g > 0.0 ? 3.14159265358979323846 / 3.0 : 0.0) : atan2(sqrt(disc), -g) / 3.0;
    const double root = sqrt(h), c = root * cos(third), sn = root * 1.7320508075688772 * sin(third);
    ev[0] = mean + 2.0 * c; ev[1] = mean - c + sn; ev[2] = mean - c - sn;
    return true;
}
// eigenvector of M for eigenvalue l: the largest-diagonal column of adj(M - l I), entries below 1e-8 flushed, normalised (zero vector
// when the column vanishes)
__device__ __forceinline__ void sp_eigenvector(const sp_sym3 &M, double l, double v[3]) {
    double c[6];
    c[0] = (l - M.yy) * (l - M.zz) - M.yz * M.yz;
    c[1] = (l - M.zz) * M.xy + M.xz * M.yz;
    c[2] = (l - M.xx) * (l - M.zz) - M.xz * M.xz;
    c[3] = (l - M.yy) * M.xz + M.xy * M.yz;
    c[4] = (l - M.xx) * M.yz + M.xy * M.xz;
    c[5] = (l - M.xx) * (l - M.yy) - M.xy * M.xy;
#pragma unroll
    for (int k = 0; k < 6; ++k) if (fabs(c[k]) <= 1.0e-8) c[k] = 0.0;
    const double d0 = fabs(c[0]), d1 = fabs(c[2]), d2 = fabs(c[5]);
    if (d0 >= d1 && d0 >= d2) { v[0] = c[0]; v[1] = c[1]; v[2] = c[3]; }
    else if (d1 >= d2) { v[0] = c[1]; v[1] = c[2]; v[2] = c[4]; }
    else { v[0] = c[3]; v[1] = c[4]; v[2] = c[5]; }
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double inv = n2 > 1.0e-8 ? 1.0 / sqrt(n2) : 0.0;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
// w <- the unit vector orthogonal to unit vector u that w suggests; when w has (almost) nothing outside u (|w - (w.u)u|^2 <= 0.01) a
// perpendicular is built from u's two smaller components; false when even that fails
__device__ __forceinline__ bool sp_orthonormal(const double u[3], double w[3]) {
    const double along = u[0] * w[0] + u[1] * w[1] + u[2] * w[2];
    double n2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] -= along * u[k]; n2 += w[k] * w[k]; }
    if (n2 > 0.01) {
        const double inv = 1.0 / sqrt(n2);
        w[0] *= inv; w[1] *= inv; w[2] *= inv;
        return true;
    }
    int big = 0;
    double mx = 1.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (mx < fabs(u[k])) { mx = fabs(u[k]); big = k; }
    const int p = big == 0 ? 1 : (big == 1 ? 2 : 0), q = big == 0 ? 2 : (big == 1 ? 0 : 1);
    const double len = sqrt(u[p] * u[p] + u[q] * u[q]);
    if (!(len > 0.01)) return false;
    w[big] = 0.0; w[p] = -u[q] / len; w[q] = u[p] / len;
    return true;
}

// G = the lanes that own the problem: 64 (the wavefront), or 16 — four problems of at most 16 points side by side.  A point sits in the lane of its
// index either way and the other lanes add +0.0, so the butterfly's steps over 32 and 16 lanes change nothing: the sums of a small problem are the
// same bits in both forms (the 3x3 algebra ran redundantly in every lane already: four problems' worth costs what one did).
template <int G>
__device__ __forceinline__ void sp_superpose(const float *__restrict__ xs, const float *__restrict__ ys, const uint64_t *__restrict__ off, uint64_t prob, bool live, uint32_t lane,
                                             float *__restrict__ rmsd_out, float *__restrict__ rot_out, float *__restrict__ tran_out) {
    const uint64_t p0 = live ? off[prob] : 0, n = live ? off[prob + 1] - p0 : 0;
    const float *x = xs + 3 * p0, *y = ys + 3 * p0;
    double U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, T[3] = {0, 0, 0};
    float rms = 3.40282347e+38f;
    if (n > 0) {
        // centroid sums and the raw second moments  S[a][b] = sum x_a y_b
        double sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0}, S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (uint64_t i = lane; i < n; i += G) {
            const double xv[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]}, yv[3] = {y[3 * i], y[3 * i + 1], y[3 * i + 2]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sx[a] += xv[a]; sy[a] += yv[a];
#pragma unroll
                for (int b = 0; b < 3; ++b) S[a][b] += xv[a] * yv[b];
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            sx[a] = sp_wave_sum<G>(sx[a]); sy[a] = sp_wave_sum<G>(sy[a]);
#pragma unroll
            for (int b = 0; b < 3; ++b) S[a][b] = sp_wave_sum<G>(S[a][b]);
        }
        const double dn = (double)n;
        const double xc[3] = {sx[0] / dn, sx[1] / dn, sx[2] / dn}, yc[3] = {sy[0] / dn, sy[1] / dn, sy[2] / dn};
        // R[b][a] = cov(y_b, x_a): the matrix with y = R x in the least-squares sense
        double R[3][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int a = 0; a < 3; ++a) R[b][a] = S[a][b] - sx[a] * sy[b] / dn;
        const double det_r = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                             R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
        sp_sym3 M;   // R^T R
        M.xx = R[0][0] * R[0][0] + R[1][0] * R[1][0] + R[2][0] * R[2][0];
        M.xy = R[0][0] * R[0][1] + R[1][0] * R[1][1] + R[2][0] * R[2][1];
        M.yy = R[0][1] * R[0][1] + R[1][1] * R[1][1] + R[2][1] * R[2][1];
        M.xz = R[0][0] * R[0][2] + R[1][0] * R[1][2] + R[2][0] * R[2][2];
        M.yz = R[0][1] * R[0][2] + R[1][1] * R[1][2] + R[2][1] * R[2][2];
        M.zz = R[0][2] * R[0][2] + R[1][2] * R[1][2] + R[2][2] * R[2][2];
        double ev[3];
        const bool spread = (M.xx + M.yy + M.zz) / 3.0 > 0.0;
        bool solved = false;
        if (sp_eigenvalues(M, det_r, ev)) {
            // right frame: eigenvectors of the largest (a0) and the smallest (a2) eigenvalue; the better separated one is kept as it is
            double a0[3], a2[3], a1[3];
            sp_eigenvector(M, ev[0], a0);
            sp_eigenvector(M, ev[2], a2);
            const bool keep_first = ev[0] - ev[1] > ev[1] - ev[2];
            const bool ok_a = keep_first ? sp_orthonormal(a0, a2) : sp_orthonormal(a2, a0);
            if (ok_a) {
                a1[0] = a2[1] * a0[2] - a2[2] * a0[1];      // a1 = a2 x a0 (right-handed a0, a1, a2)
                a1[1] = a2[2] * a0[0] - a2[0] * a0[2];
                a1[2] = a2[0] * a0[1] - a2[1] * a0[0];
                // left frame: images of a0 and a1 under R, normalised; b2 completes it
                double b0[3], b1[3], b2[3];
                double n0 = 0.0, n1 = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    b0[k] = R[k][0] * a0[0] + R[k][1] * a0[1] + R[k][2] * a0[2];
                    b1[k] = R[k][0] * a1[0] + R[k][1] * a1[1] + R[k][2] * a1[2];
                    n0 += b0[k] * b0[k]; n1 += b1[k] * b1[k];
                }
                const double i0 = n0 > 1.0e-8 ? 1.0 / sqrt(n0) : 0.0, i1 = n1 > 1.0e-8 ? 1.0 / sqrt(n1) : 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) { b0[k] *= i0; b1[k] *= i1; }
                if (sp_orthonormal(b0, b1)) {
                    b2[0] = b0[1] * b1[2] - b0[2] * b1[1];
                    b2[1] = b0[2] * b1[0] - b0[0] * b1[2];
                    b2[2] = b0[0] * b1[1] - b0[1] * b1[0];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) U[r][cc] = b0[r] * a0[cc] + b1[r] * a1[cc] + b2[r] * a2[cc];
                    solved = true;
                }
            }
        }
        // translation: y-centroid minus the rotated x-centroid; for a moving set without spread (all x equal) the rotation stays the
        // identity; when the frame construction gave up the reference leaves the translation at zero, too
        if (solved || !spread)
#pragma unroll
            for (int r = 0; r < 3; ++r) T[r] = yc[r] - (U[r][0] * xc[0] + U[r][1] * xc[1] + U[r][2] * xc[2]);
        double ss = 0.0;
        for (uint64_t i = lane; i < n; i += G) {
            const double X = x[3 * i], Y = x[3 * i + 1], Z = x[3 * i + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) { const double d = U[r][0] * X + U[r][1] * Y + U[r][2] * Z + T[r] - (double)y[3 * i + r]; ss += d * d; }
        }
        ss = sp_wave_sum<G>(ss);
        rms = (float)sqrt(ss / dn);
        if (rms != rms) rms = 3.40282347e+38f;
    }
    if (lane != 0 || !live) return;
    rmsd_out[prob] = rms;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) rot_out[9 * prob + 3 * r + cc] = (float)U[r][cc];
        tran_out[3 * prob + r] = (float)T[r];
    }
}

__global__ __launch_bounds__(64) void k_superpose(const float *__restrict__ xs, const float *__restrict__ ys, const uint64_t *__restrict__ off, uint64_t n_prob,
                                                  float *__restrict__ rmsd_out, float *__restrict__ rot_out, float *__restrict__ tran_out) {
    if (blockIdx.x < n_prob) sp_superpose<64>(xs, ys, off, blockIdx.x, true, threadIdx.x, rmsd_out, rot_out, tran_out);
}
// four problems per wavefront when all four are small (motif batches: ~10 points per problem, 10-20 k problems per 128 queries — a wavefront each was
// 40 us of a batch's kernels); a group with a larger one runs its four one after the other on the whole wavefront
__global__ __launch_bounds__(64) void k_superpose4(const float *__restrict__ xs, const float *__restrict__ ys, const uint64_t *__restrict__ off, uint64_t n_prob,
                                                   float *__restrict__ rmsd_out, float *__restrict__ rot_out, float *__restrict__ tran_out) {
    const uint32_t lane = threadIdx.x;
    const uint64_t base = 4ull * blockIdx.x, mine = base + (lane >> 4);
    const bool live = mine < n_prob;
    const bool small = !live || off[mine + 1] - off[mine] <= 16u;
    if (__all(small)) { sp_superpose<16>(xs, ys, off, mine, live, lane & 15u, rmsd_out, rot_out, tran_out); return; }
    for (uint64_t k = base; k < base + 4u && k < n_prob; ++k) sp_superpose<64>(xs, ys, off, k, true, lane, rmsd_out, rot_out, tran_out);
}

// Similarity metrics of a superposition (metrics.rs:62-251 on KabschSuperimposer's reference / transformed coordinates,
// kabsch.rs:86-95,145-154): ref = fixed (query) points, mov = moving (target) points, transformed = rot * mov + tran evaluated in f32,
// every distance in f64 then f32.  out = {tm_score, gdt_ts, gdt_ha, chamfer, hausdorff}.  As the reference has it, tm_score and gdt
// compare the DISTANCE (not its square) with d0^2 / cutoff^2 (metrics.rs:141-143, 160-163); d0 comes from the host (powf of glibc).
__device__ __forceinline__ float sp_dist(const float *__restrict__ ref, uint64_t r, float tx, float ty, float tz) {
    const double dx = (double)ref[3 * r] - (double)tx, dy = (double)ref[3 * r + 1] - (double)ty, dz = (double)ref[3 * r + 2] - (double)tz;
    return (float)sqrt(dx * dx + dy * dy + dz * dz);
}
// (G lanes per problem as in sp_superpose: the sums, counts and maxima of a problem of at most 16 points are the same bits in both forms)
template <int G>
__device__ __forceinline__ void sp_metrics(const float *__restrict__ refs, const float *__restrict__ movs, const uint64_t *__restrict__ off, uint64_t prob, bool live, uint32_t lane,
                                           const float *__restrict__ rot, const float *__restrict__ tran, const float *__restrict__ d0s, float *__restrict__ out) {
    if (!live) return;      // (a group's lanes share `live` and n: the shuffles below stay inside the group)
    const uint64_t p0 = off[prob], n = off[prob + 1] - p0;
    float *o = out + 5 * prob;
    if (n == 0) { if (lane == 0) { o[0] = o[1] = o[2] = 0.0f; o[3] = o[4] = __builtin_inff(); } return; }
    const float *ref = refs + 3 * p0, *mov = movs + 3 * p0;
    const float *Rm = rot + 9 * prob, *Tv = tran + 3 * prob;
    const double d0 = (double)d0s[prob], d0_sq = (double)((float)d0 * (float)d0), dn = (double)n;
    double tm = 0.0, ch = 0.0;
    float hd = -1.0f;
    uint32_t c_ts[4] = {0, 0, 0, 0}, c_ha[4] = {0, 0, 0, 0};
    for (uint64_t i = lane; i < n; i += G) {
        const float mx = mov[3 * i], my = mov[3 * i + 1], mz = mov[3 * i + 2];
        const float tx = (Rm[0] * mx + Rm[1] * my + Rm[2] * mz) + Tv[0];
        const float ty = (Rm[3] * mx + Rm[4] * my + Rm[5] * mz) + Tv[1];
        const float tz = (Rm[6] * mx + Rm[7] * my + Rm[8] * mz) + Tv[2];
        const double dii = (double)sp_dist(ref, i, tx, ty, tz);
        tm += 1.0 / (1.0 + dii / d0_sq);
        const double ts[4] = {1.0, 2.0, 4.0, 8.0}, ha[4] = {0.5, 1.0, 2.0, 4.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) { c_ts[k] += dii <= ts[k] * ts[k] ? 1u : 0u; c_ha[k] += dii <= ha[k] * ha[k] ? 1u : 0u; }
        // nearest reference point: (float)sqrt(.) is monotone, so the smallest distance is the image of the smallest SQUARED distance — nine f64 operations per
        // candidate instead of a double-precision square root each (a whole-structure query's problems have ~600 points: 3.5·10^5 roots per problem, 0.75 ms
        // of kernel for its 367 problems); same bits: min of the images = image of the min, fmin drops a NaN on either side like fminf did
        double m2;
        {
            const double dx = (double)ref[0] - (double)tx, dy = (double)ref[1] - (double)ty, dz = (double)ref[2] - (double)tz;
            m2 = dx * dx + dy * dy + dz * dz;
        }
        for (uint64_t j = 1; j < n; ++j) {
            const double dx = (double)ref[3 * j] - (double)tx, dy = (double)ref[3 * j + 1] - (double)ty, dz = (double)ref[3 * j + 2] - (double)tz;
            m2 = fmin(m2, dx * dx + dy * dy + dz * dz);
        }
        const float mn = (float)sqrt(m2);
        ch += (double)mn;
        hd = fmaxf(hd, mn);
    }
    tm = sp_wave_sum<G>(tm); ch = sp_wave_sum<G>(ch);
#pragma unroll
    for (int ofs = G / 2; ofs > 0; ofs >>= 1) {
        hd = fmaxf(hd, __shfl_xor(hd, ofs));
#pragma unroll
        for (int k = 0; k < 4; ++k) { c_ts[k] += __shfl_xor(c_ts[k], ofs); c_ha[k] += __shfl_xor(c_ha[k], ofs); }
    }
    if (lane != 0) return;
    o[0] = (float)(tm / dn);
    double s_ts = 0.0, s_ha = 0.0;
    for (int k = 0; k < 4; ++k) { s_ts += (double)c_ts[k] / dn; s_ha += (double)c_ha[k] / dn; }
    o[1] = (float)(s_ts / 4.0);
    o[2] = (float)(s_ha / 4.0);
    o[3] = (float)(ch / dn);
    o[4] = hd;
}

__global__ __launch_bounds__(64) void k_metrics(const float *__restrict__ refs, const float *__restrict__ movs, const uint64_t *__restrict__ off, uint64_t n_prob,
                                                const float *__restrict__ rot, const float *__restrict__ tran, const float *__restrict__ d0s,
                                                float *__restrict__ out) {
    if (blockIdx.x < n_prob) sp_metrics<64>(refs, movs, off, blockIdx.x, true, threadIdx.x, rot, tran, d0s, out);
}
__global__ __launch_bounds__(64) void k_metrics4(const float *__restrict__ refs, const float *__restrict__ movs, const uint64_t *__restrict__ off, uint64_t n_prob,
                                                 const float *__restrict__ rot, const float *__restrict__ tran, const float *__restrict__ d0s,
                                                 float *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const uint64_t base = 4ull * blockIdx.x, mine = base + (lane >> 4);
    const bool live = mine < n_prob;
    const bool small = !live || off[mine + 1] - off[mine] <= 16u;
    if (__all(small)) { sp_metrics<16>(refs, movs, off, mine, live, lane & 15u, rot, tran, d0s, out); return; }
    for (uint64_t k = base; k < base + 4u && k < n_prob; ++k) sp_metrics<64>(refs, movs, off, k, true, lane, rot, tran, d0s, out);
}

// n_points = the problems' points in all when the caller knows them (0: not known): batches of small problems (16 points a problem or fewer on average)
// take the four-per-wavefront kernels
// (FDGPU_SP_PACK=0: a wavefront per problem always — read per call: tests compare the two forms bit for bit)
static bool sp_pack_on() { const char *e = getenv("FDGPU_SP_PACK"); return !(e && e[0] == '0'); }
void fd_launch_kabsch(const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot, float *tran, hipStream_t st, uint64_t n_points) {
    if (!n) return;
    if (n_points && n_points <= 16 * n && sp_pack_on()) hipLaunchKernelGGL(k_superpose4, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, st, x, y, off, n, rmsd, rot, tran);
    else hipLaunchKernelGGL(k_superpose, dim3((unsigned)n), dim3(64), 0, st, x, y, off, n, rmsd, rot, tran);
}
void fd_launch_metrics(const float *ref, const float *mov, const uint64_t *off, uint64_t n, const float *rot, const float *tran, const float *d0, float *out,
                       hipStream_t st, uint64_t n_points) {
    if (!n) return;
    if (n_points && n_points <= 16 * n && sp_pack_on()) hipLaunchKernelGGL(k_metrics4, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, st, ref, mov, off, n, rot, tran, d0, out);
    else hipLaunchKernelGGL(k_metrics, dim3((unsigned)n), dim3(64), 0, st, ref, mov, off, n, rot, tran, d0, out);
}

// ------------------------------------------------------------------------------------------ LMS-QCP partial fit
// --partial-fit (src/structure/lms_qcp.rs:91-249, default parameters): 500 seeded 3-point trials scored by the median
// squared residual of the other pairs, then forward growth of the core by the closest remaining pair until that pair is
// farther than 2 A and the core holds n/2 pairs.  One wavefront per problem: lane 0 replays the xorshift stream (the number
// of draws per trial depends on the collinearity retries, so the stream is sequential) into LDS, the trials then run one per
// lane, the growth loop scans the pairs 64 at a time.  Arithmetic order follows the reference statement by statement
// (f64 running sums, f32 residuals, no contraction) so the discrete choices (seed, joining order, stop) are the same.
#define FD_LMS_TRIALS 500

struct lms_stats { double n, sx[3], sy[3], sxx, syy, syx[3][3]; };

__device__ __forceinline__ void lms_clear(lms_stats &s) {
    s.n = 0; s.sxx = 0; s.syy = 0;
    for (int a = 0; a < 3; ++a) { s.sx[a] = 0; s.sy[a] = 0; for (int b = 0; b < 3; ++b) s.syx[a][b] = 0; }
}
__device__ __forceinline__ void lms_add(lms_stats &s, const float *xf, const float *yf, uint64_t i) {   // lms_qcp.rs:277-291
    const double x[3] = {xf[3 * i], xf[3 * i + 1], xf[3 * i + 2]}, y[3] = {yf[3 * i], yf[3 * i + 1], yf[3 * i + 2]};
    s.n += 1.0;
    for (int a = 0; a < 3; ++a) { s.sx[a] += x[a]; s.sy[a] += y[a]; }
    s.sxx += x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    s.syy += y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) s.syx[a][b] += y[a] * x[b];
}

// rotation + translation of the current statistics (qcp_from_stats :304-344 with qcp_from_a_e0 :349-462 inlined)
__device__ __noinline__ void lms_solve(const lms_stats &s, float r[9], float t[3]) {
    const double n = s.n, inv = 1.0 / n;
    double mx[3], my[3], A[3][3];
    for (int k = 0; k < 3; ++k) { mx[k] = s.sx[k] * inv; my[k] = s.sy[k] * inv; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = s.syx[i][j] - n * (my[i] * mx[j]);
    const double m2x = mx[0] * mx[0] + mx[1] * mx[1] + mx[2] * mx[2], m2y = my[0] * my[0] + my[1] * my[1] + my[2] * my[2];
    const double e0 = 0.5 * fmax((s.syy - n * m2y) + (s.sxx - n * m2x), 0.0);
    const double sxx = A[0][0], sxy = A[0][1], sxz = A[0][2], syx = A[1][0], syy = A[1][1], syz = A[1][2], szx = A[2][0], szy = A[2][1],
                 szz = A[2][2];
    const double sxx2 = sxx * sxx, syy2 = syy * syy, szz2 = szz * szz, sxy2 = sxy * sxy, syz2 = syz * syz, sxz2 = sxz * sxz, syx2 = syx * syx,
                 szy2 = szy * szy, szx2 = szx * szx;
    const double u = 2.0 * (syz * szy - syy * szz);
    const double v = syy2 + szz2 - sxx2 + syz2 + szy2;
    const double c2 = -2.0 * (sxx2 + syy2 + szz2 + sxy2 + syx2 + sxz2 + szx2 + syz2 + szy2);
    const double c1 = 8.0 * (sxx * syz * szy + syy * szx * sxz + szz * sxy * syx - sxx * syy * szz - syz * szx * sxy - szy * syx * sxz);
    const double xzp = sxz + szx, yzp = syz + szy, xyp = sxy + syx, yzm = syz - szy, xzm = sxz - szx, xym = sxy - syx, xxyyp = sxx + syy,
                 xxyym = sxx - syy;
    const double w = sxy2 + sxz2 - syx2 - szx2;
    const double nxzp = -xzp, nxzm = -xzm, nxym = -xym, tr = xxyyp + szz;
    const double c0 = w * w + (v + u) * (v - u) + (nxzp * yzm + xym * (xxyym - szz)) * (nxzm * yzp + xym * (xxyym + szz)) +
                      (nxzp * yzp - xyp * (xxyyp - szz)) * (nxzm * yzm - xyp * tr) +
                      (xyp * yzp + xzp * (xxyym + szz)) * (nxym * yzm + xzp * tr) +
                      (xyp * yzm + xzm * (xxyym - szz)) * (nxym * yzp + xzm * (xxyyp - szz));
    double lam = fmax(e0, 0.0);
    const double eps = 1e-15;
    for (int it = 0; it < 10; ++it) {
        const double x2 = lam * lam, b = (x2 + c2) * lam, aa = b + c1, f = aa * lam + c0, fp = 2.0 * x2 * lam + b + aa;
        const double nl = fabs(lam - f / (fp + eps));
        const bool done = fabs(nl - lam) < eps * nl;
        lam = nl;
        if (done) break;
    }
    const double a11 = xxyyp + szz - lam, a12 = yzm, a13 = nxzm, a14 = xym, a21 = a12, a22 = xxyym - szz - lam, a23 = xyp, a24 = xzp, a31 = a13,
                 a32 = a23, a33 = syy - sxx - szz - lam, a34 = yzp, a41 = a14, a42 = a24, a43 = a34, a44 = szz - xxyyp - lam;
    const double m3344 = a33 * a44 - a43 * a34, m3244 = a32 * a44 - a42 * a34, m3243 = a32 * a43 - a42 * a33, m3143 = a31 * a43 - a41 * a33,
                 m3144 = a31 * a44 - a41 * a34, m3142 = a31 * a42 - a41 * a32;
    double q1 = a22 * m3344 - a23 * m3244 + a24 * m3243;
    double q2 = -a21 * m3344 + a23 * m3144 - a24 * m3143;
    double q3 = a21 * m3244 - a22 * m3144 + a24 * m3142;
    double q4 = -a21 * m3243 + a22 * m3143 - a23 * m3142;
    double qs = q1 * q1 + q2 * q2 + q3 * q3 + q4 * q4;
    double rot[3][3];
    bool ident = false;
    if (qs < 1e-12) {
        q1 = a12 * m3344 - a13 * m3244 + a14 * m3243;
        q2 = -a11 * m3344 + a13 * m3144 - a14 * m3143;
        q3 = a11 * m3244 - a12 * m3144 + a14 * m3142;
        q4 = -a11 * m3243 + a12 * m3143 - a13 * m3142;
        qs = q1 * q1 + q2 * q2 + q3 * q3 + q4 * q4;
        ident = qs < 1e-12;
    }
    if (ident) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rot[i][j] = i == j ? 1.0 : 0.0;
    } else {
        const double qi = 1.0 / sqrt(qs);
        q1 *= qi; q2 *= qi; q3 *= qi; q4 *= qi;
        const double A2 = q1 * q1, X2 = q2 * q2, Y2 = q3 * q3, Z2 = q4 * q4, xy = q2 * q3, az = q1 * q4, zx = q4 * q2, ay = q1 * q3, yz = q3 * q4,
                     ax = q1 * q2;
        rot[0][0] = A2 + X2 - Y2 - Z2; rot[0][1] = 2.0 * (xy + az);   rot[0][2] = 2.0 * (zx - ay);
        rot[1][0] = 2.0 * (xy - az);   rot[1][1] = A2 - X2 + Y2 - Z2; rot[1][2] = 2.0 * (yz + ax);
        rot[2][0] = 2.0 * (zx + ay);   rot[2][1] = 2.0 * (yz - ax);   rot[2][2] = A2 - X2 - Y2 + Z2;
    }
    for (int i = 0; i < 3; ++i) {
        const double rx = rot[i][0] * mx[0] + rot[i][1] * mx[1] + rot[i][2] * mx[2];
        t[i] = (float)(my[i] - rx);
        for (int j = 0; j < 3; ++j) r[3 * i + j] = (float)rot[i][j];
    }
}

__device__ __forceinline__ float lms_resid2(const float r[9], const float t[3], const float *xf, const float *yf, uint64_t i) {   // :481-507
    const float v0 = xf[3 * i], v1 = xf[3 * i + 1], v2 = xf[3 * i + 2];
    const float p0 = (r[0] * v0 + r[1] * v1 + r[2] * v2) + t[0];
    const float p1 = (r[3] * v0 + r[4] * v1 + r[5] * v2) + t[1];
    const float p2 = (r[6] * v0 + r[7] * v1 + r[8] * v2) + t[2];
    const float dx = p0 - yf[3 * i], dy = p1 - yf[3 * i + 1], dz = p2 - yf[3 * i + 2];
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ uint64_t lms_rng_next(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

// lexicographic (value, index) minimum over the wavefront; value = +inf means "none"
__device__ __forceinline__ void lms_wave_argmin(float &v, uint32_t &ix) {
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const uint32_t oi = __shfl_xor(ix, o);
        if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
    }
}

__global__ __launch_bounds__(64) void k_lms_qcp(const float *__restrict__ xs, const float *__restrict__ ys, const uint64_t *__restrict__ off,
                                                uint64_t n_prob, float *__restrict__ rmsd_out, float *__restrict__ rot_out,
                                                float *__restrict__ tran_out, uint32_t *__restrict__ core_out, uint8_t *__restrict__ in_core,
                                                uint32_t *__restrict__ order) {
    __shared__ uint32_t s_seed[FD_LMS_TRIALS * 3];
    __shared__ uint8_t s_ok[FD_LMS_TRIALS];
    const uint64_t pidx = blockIdx.x;
    if (pidx >= n_prob) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t p0 = off[pidx], n = off[pidx + 1] - p0;
    const float *xf = xs + 3 * p0, *yf = ys + 3 * p0;
    uint8_t *flag = in_core + p0;
    uint32_t *ord = order + p0;
    // ---- the random stream (SmallRng :530-549, sample_three_non_collinear :509-527)
    if (lane == 0) {
        uint64_t z = 0xC0FFEE005EEDull + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        uint64_t st = z ^ (z >> 31);
        for (int tr = 0; tr < FD_LMS_TRIALS; ++tr) {
            bool ok = false;
            for (int tries = 0; tries < 64 && !ok; ++tries) {
                const uint64_t i = lms_rng_next(st) % n;
                uint64_t j = lms_rng_next(st) % n; if (j == i) j = (j + 1) % n;
                uint64_t k = lms_rng_next(st) % n; while (k == i || k == j) k = (k + 1) % n;
                const float a0 = xf[3 * j] - xf[3 * i], a1 = xf[3 * j + 1] - xf[3 * i + 1], a2 = xf[3 * j + 2] - xf[3 * i + 2];
                const float b0 = xf[3 * k] - xf[3 * i], b1 = xf[3 * k + 1] - xf[3 * i + 1], b2 = xf[3 * k + 2] - xf[3 * i + 2];
                const float cx = a1 * b2 - a2 * b1, cy = a2 * b0 - a0 * b2, cz = a0 * b1 - a1 * b0;
                const float area2 = cx * cx + cy * cy + cz * cz;
                if (area2 > 1e-6f) { ok = true; s_seed[3 * tr] = (uint32_t)i; s_seed[3 * tr + 1] = (uint32_t)j; s_seed[3 * tr + 2] = (uint32_t)k; }
            }
            s_ok[tr] = ok;
        }
    }
    for (uint64_t i = lane; i < n; i += 64) flag[i] = 0;
    __syncthreads();
    // ---- one trial per lane: median squared residual of the other pairs (select_quantile_squared :467-477)
    float r[9], t[3];
    float best_q = __builtin_inff();
    uint32_t best_t = 0xffffffffu;
    const uint64_t m = n - 3;
    const uint64_t pos = m > 1 ? (uint64_t)roundf(0.5f * (float)(m - 1)) : 0;
    for (uint32_t tr = lane; tr < FD_LMS_TRIALS; tr += 64) {
        if (!s_ok[tr]) continue;
        const uint32_t a = s_seed[3 * tr], b = s_seed[3 * tr + 1], c = s_seed[3 * tr + 2];
        lms_stats S;
        lms_clear(S);
        lms_add(S, xf, yf, a); lms_add(S, xf, yf, b); lms_add(S, xf, yf, c);
        lms_solve(S, r, t);
        float qv = 0.0f;
        if (m > 0) {   // the element of rank pos = the smallest bit pattern with at least pos + 1 residuals at or below it (residuals are >= +0)
            uint32_t lo = 0, hi = 0x7f800000u;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                uint64_t cnt = 0;
                for (uint64_t i = 0; i < n; ++i) {
                    if (i == a || i == b || i == c) continue;
                    cnt += __float_as_uint(lms_resid2(r, t, xf, yf, i)) <= mid;
                }
                if (cnt >= pos + 1) hi = mid; else lo = mid + 1;
            }
            qv = __uint_as_float(lo);
        }
        if (qv < best_q) { best_q = qv; best_t = tr; }
    }
    lms_wave_argmin(best_q, best_t);
    uint32_t seed[3] = {0, 1, 2};
    if (best_q < __builtin_inff() && best_t != 0xffffffffu) { seed[0] = s_seed[3 * best_t]; seed[1] = s_seed[3 * best_t + 1]; seed[2] = s_seed[3 * best_t + 2]; }
    // ---- forward growth (run() :142-196); lane i mod 64 owns pair i's in-core flag
    uint64_t min_core = n / 2; if (min_core < 3) min_core = 3;
    lms_stats S;
    lms_clear(S);
    uint64_t nc = 0;
    for (int k = 0; k < 3; ++k) {
        lms_add(S, xf, yf, seed[k]);
        if ((seed[k] & 63u) == lane) flag[seed[k]] = 1;
        if (lane == 0) ord[nc] = seed[k];
        ++nc;
    }
    for (;;) {
        lms_solve(S, r, t);
        float b2 = __builtin_inff();
        uint32_t bi = 0xffffffffu;
        for (uint64_t i = lane; i < n; i += 64) {
            if (flag[i]) continue;
            const float d2 = lms_resid2(r, t, xf, yf, i);
            if (d2 < b2) { b2 = d2; bi = (uint32_t)i; }
        }
        lms_wave_argmin(b2, bi);
        if (bi == 0xffffffffu || !(b2 < __builtin_inff())) break;
        if (nc >= min_core && b2 > 4.0f) break;
        lms_add(S, xf, yf, bi);
        if ((bi & 63u) == lane) flag[bi] = 1;
        if (lane == 0) ord[nc] = bi;
        ++nc;
        if (nc == n) break;     // the reference keeps the transform solved before the last pair joined (:186-189)
    }
    if (lane == 0) {            // finish() :226-249: f32 sum in joining order
        float sum = 0.0f;
        for (uint64_t k = 0; k < nc; ++k) sum += lms_resid2(r, t, xf, yf, ord[k]);
        rmsd_out[pidx] = sqrtf(sum / (float)nc);
        for (int k = 0; k < 9; ++k) rot_out[9 * pidx + k] = r[k];
        for (int k = 0; k < 3; ++k) tran_out[3 * pidx + k] = t[k];
        core_out[pidx] = (uint32_t)nc;
    }
}

void fd_launch_lms_qcp(const float *x, const float *y, const uint64_t *off, uint64_t n, float *rmsd, float *rot, float *tran, uint32_t *core_len,
                       uint8_t *flags, uint32_t *order, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_lms_qcp, dim3((unsigned)n), dim3(64), 0, st, x, y, off, n, rmsd, rot, tran, core_len, flags, order);
}
