// k_retrieve.hip — the per-candidate body of retrieval_wrapper on the device (reference: src/controller/retrieve.rs:364-552 and
// 604-702, src/controller/graph.rs:16-50): found triples of one candidate structure -> match graph -> strongly + weakly
// connected components -> residue votes -> greedy assignment -> rescue of unmatched query residues from the candidate
// pairs -> the residue lists and the [CA, CB] point lists of the superposition problems, which k_superpose / k_metrics
// (k_match.hip) then solve without leaving the device.
//
// One wavefront per candidate slot, and the 64 lanes ARE the graph: lane v holds node v (its residue, its adjacency row,
// its reachability row as a 64-bit mask), so
//   * node discovery in first-appearance order (graph.rs:16-26) is one ballot per edge endpoint,
//   * transitive closure is Warshall with one readlane per pivot (64 steps), SCC(v) = reach[v] & reach^T[v], WCC(v) = closure of
//     the symmetrised adjacency; a component is kept by its lowest member, WCCs equal to an SCC are dropped, the rest is ordered
//     like the reference orders its sorted node lists (graph.rs:43-45) by a mask comparison,
//   * the per-query-residue "best target" table and the assignment list live one entry per lane (ballot = lookup).
// Everything sequential in the reference (edge order, vote order, the assignment loop with its erase-and-reinsert quirk) runs as
// wave-uniform loops over LDS, so the result does not depend on scheduling.  Limits (else the slot raises the overflow flag and the
// caller takes the host path for the whole call): 64 graph nodes, 64 query residues, 1024 found triples per candidate.
#include "fdgpu_internal.h"

#define RS_EDGE_CAP 1024u
#define RS_LIST_CAP 2048u
#define RS_CAND_LDS 512u      // candidate pairs of a slot kept in LDS (more: read from global memory where they are needed)

namespace {
__device__ __forceinline__ uint64_t rs_bcast64(uint64_t v, uint32_t src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src, FD_WAVE), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, FD_WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rs_wave_max64(uint64_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, FD_WAVE), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, FD_WAVE);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t rs_wave_max32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, FD_WAVE);
        v = o > v ? o : v;
    }
    return v;
}
// a < b for two different node sets read as ascending node lists (the order Vec<Vec<usize>>::sort gives, graph.rs:44)
__device__ __forceinline__ bool rs_less(uint64_t a, uint64_t b) {
    const uint32_t d = (uint32_t)__builtin_ctzll(a ^ b);
    if ((a >> d) & 1ull) return ((b >> d) >> 1) != 0ull;     // a holds d: smaller unless b ended (b is a prefix of a)
    return ((a >> d) >> 1) == 0ull;                          // b holds d: a is smaller only as a prefix of b
}
}  // namespace

// exclusive scan of one 64-bit value per thread over a workgroup of 1,024 (wave scans by shuffle, the sixteen wave totals through LDS; two barriers)
__device__ __forceinline__ unsigned long long rs_block_excl64(unsigned long long v, uint32_t tid, unsigned long long *s_w, unsigned long long *total) {
    const uint32_t lane = tid & 63u, wv = tid >> 6;
    unsigned long long incl = v;
    for (int off = 1; off < FD_WAVE; off <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, off, FD_WAVE), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off, FD_WAVE);
        if ((int)lane >= off) incl += ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();      // s_w may still be read by the scan before this one
    if (lane == 63u) s_w[wv] = incl;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) { const unsigned long long x = s_w[k]; pre += k < wv ? x : 0ull; tot += x; }
    *total = tot;
    return pre + incl - v;
}
#define RS_SYNC() __syncthreads()
#define RS_OVERFLOW() do { if (threadIdx.x == 0) atomicOr(A.flags, 1u); return; } while (0)

// records of one candidate slot arrive in runs (a drain of the pair scan belongs to one slot): lanes that hold the same counter are
// merged before the global atomic — one add per distinct slot and wavefront instead of one per record
__device__ __forceinline__ uint32_t rs_agg_add(uint32_t *ctr, bool on, uint32_t key) {
    uint32_t ret = 0;
    uint64_t todo = __ballot(on);
    while (todo) {
        const uint32_t lead = (uint32_t)__builtin_ctzll(todo);
        const uint32_t k = (uint32_t)__shfl((int)key, (int)lead, FD_WAVE);
        const uint64_t m = __ballot(on && key == k);
        uint32_t base = 0;
        if (fd_lane() == lead) base = atomicAdd(&ctr[k], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, (int)lead, FD_WAVE);
        if (on && key == k) ret = base + fd_mbcnt(m);
        todo &= ~m;
    }
    return ret;
}
__global__ void k_rs_count(const fd_pair_rec *__restrict__ found, uint64_t nf, const fd_cand_rec *__restrict__ cands, uint64_t nc, uint32_t n_cand,
                           uint32_t *__restrict__ cnt) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = 0;
    bool on = false;
    if (x < nf) { const uint32_t s = found[x].cand; on = s < n_cand; key = s; }
    else if (x < nf + nc) { const uint32_t s = cands[x - nf].cand; on = s < n_cand; key = n_cand + 1 + s; }
    (void)rs_agg_add(cnt, on, key);
}

// exclusive scans of the two count arrays (one block; n_cand is a few thousand at most) -> segment starts + scatter cursors.  Both counts ride
// in one 64-bit word (found triples in the upper half), a thread takes `per` neighbouring slots: ONE scan of the 1,024 threads' sums
__global__ __launch_bounds__(1024) void k_rs_scan(const uint32_t *__restrict__ cnt, uint32_t n_cand, uint32_t *__restrict__ seg, uint32_t *__restrict__ cur) {
    __shared__ unsigned long long s_w[16];
    const uint32_t tid = threadIdx.x, per = (n_cand + 1023u) / 1024u, a = tid * per, b = min(n_cand, a + per);
    const uint32_t *cf = cnt, *cc = cnt + (n_cand + 1);
    unsigned long long mine = 0;
    for (uint32_t k = a; k < b; ++k) mine += ((unsigned long long)cf[k] << 32) | cc[k];
    unsigned long long tot;
    unsigned long long run = rs_block_excl64(mine, tid, s_w, &tot);
    for (uint32_t k = a; k < b; ++k) {
        const uint32_t rf = (uint32_t)(run >> 32), rc = (uint32_t)run;
        seg[k] = rf; cur[k] = rf; seg[n_cand + 1 + k] = rc; cur[n_cand + 1 + k] = rc;
        run += ((unsigned long long)cf[k] << 32) | cc[k];
    }
    if (tid == 0) { seg[n_cand] = (uint32_t)(tot >> 32); seg[2 * n_cand + 1] = (uint32_t)tot; }
}

__global__ void k_rs_scatter(const fd_pair_rec *__restrict__ found, uint64_t nf, const fd_cand_rec *__restrict__ cands, uint64_t nc, uint32_t n_cand,
                             uint32_t *__restrict__ cur, uint32_t *__restrict__ perm_f, uint32_t *__restrict__ perm_c) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = 0;
    bool on = false;
    const bool is_f = x < nf;
    if (is_f) { const uint32_t s = found[x].cand; on = s < n_cand; key = s; }
    else if (x < nf + nc) { const uint32_t s = cands[x - nf].cand; on = s < n_cand; key = n_cand + 1 + s; }
    const uint32_t pos = rs_agg_add(cur, on, key);
    if (on) { if (is_f) perm_f[pos] = (uint32_t)x; else perm_c[pos] = (uint32_t)(x - nf); }
}

// Launch order of the slots: heaviest first.  A slot is one wavefront's serial chain (rank its edges, build the graph, votes and rescue over its
// candidate pairs) and a batch of 128 queries has 4,096 of them for ~770 resident wavefronts: dealt in slot order the launch ends with
// whichever heavy slot happened to start last.  Weight ~ F^2 / 16 + 8 F + C (F found triples, C candidate pairs), bucketed by its top two bits
// (64 buckets, any order inside one: the results do not depend on the schedule, the records are ordered afterwards).
__global__ __launch_bounds__(1024) void k_rs_order(const uint32_t *__restrict__ cnt, uint32_t n_cand, uint32_t *__restrict__ order) {
    __shared__ uint32_t hist[64];
    auto bucket = [&](uint32_t s) -> uint32_t {
        const uint32_t F = cnt[s], C = cnt[n_cand + 1 + s];
        if (!F) return 0u;
        const uint64_t w64 = (uint64_t)F * F / 16u + 8ull * F + C;
        const uint32_t w = w64 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)w64;
        const uint32_t top = 31u - (uint32_t)__clz((int)w);           // w >= 8
        const uint32_t b = 2u * top + ((w >> (top - 1u)) & 1u);
        return b < 63u ? b : 63u;
    };
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < n_cand; s += 1024) atomicAdd(&hist[bucket(s)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int b = 63; b >= 0; --b) { const uint32_t t = hist[b]; hist[b] = run; run += t; } }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < n_cand; s += 1024) order[atomicAdd(&hist[bucket(s)], 1u)] = s;
}

__global__ __launch_bounds__(FD_WAVE) void k_rs_slots(rs_args A) {
    __shared__ uint32_t s_a[RS_LIST_CAP];      // raw i | raw j           -> votes: query residue   -> rescue: partner-residue list
    __shared__ uint32_t s_b[RS_LIST_CAP];      // raw hash | raw position -> votes: target residue  -> rescue: multiplicities
    __shared__ uint32_t s_i[RS_EDGE_CAP], s_j[RS_EDGE_CAP], s_h[RS_EDGE_CAP];   // edges in (i, j, emission) order
    __shared__ int32_t s_k[RS_EDGE_CAP];       // query-map entry of the edge's hash
    __shared__ uint8_t s_es[RS_EDGE_CAP], s_et[RS_EDGE_CAP], s_sym[RS_EDGE_CAP], s_vc[RS_LIST_CAP];
    __shared__ uint64_t s_cm[2 * FD_WAVE], s_cs[2 * FD_WAVE];
    __shared__ uint32_t s_aq[FD_WAVE], s_ar[FD_WAVE], s_qs[FD_WAVE], s_rs[FD_WAVE], s_rmx[FD_WAVE], s_rnm[FD_WAVE], s_rarg[FD_WAVE];
    __shared__ int32_t s_fh[FD_WAVE], s_pr[FD_WAVE];
    __shared__ uint32_t s_misc[2];
    // what the chain below would otherwise fetch again and again with one dependent global load after the other (a slot is ONE wavefront: every
    // load's latency is paid in full): the query's residue indices and the slot's candidate pairs (query residue, i, j), loaded once, in parallel
    __shared__ uint32_t s_idx[FD_WAVE];
    __shared__ uint32_t s_cq[RS_CAND_LDS], s_ci[RS_CAND_LDS], s_cj[RS_CAND_LDS];
    if (A.n_listed && blockIdx.x >= *A.n_listed) return;      // launched over the slots k_rs_setup left to this kernel: order[0 .. *n_listed)
    const uint32_t slot = A.order ? A.order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const uint32_t f0 = A.seg_f[slot], F = A.seg_f[slot + 1] - f0;
    if (F == 0) return;
    if (F > RS_EDGE_CAP) RS_OVERFLOW();
    const rs_query_dev Q = A.qt[A.slot_q[slot]];
    const uint32_t NQ = Q.n_idx;
    if (NQ > FD_WAVE) RS_OVERFLOW();
    const uint32_t st = A.cand[slot];
    const uint32_t r0 = A.db_res_off[st], Rt = A.db_res_off[st + 1] - r0;
    const uint32_t c0 = A.seg_c[slot], c1 = A.seg_c[slot + 1];
    const bool cands_lds = c1 - c0 <= RS_CAND_LDS;
    unsigned long long tstamp = A.dbg ? wall_clock64() : 0ull;
    const unsigned long long t_slot0 = tstamp;
    auto stamp = [&](int k) {      // FDGPU_RS_DBG: phase durations summed over the slots (100 MHz ticks)
        if (A.dbg && lane == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&A.dbg[k], now - tstamp); tstamp = now; }
    };
    // ---- edges in the reference's scan order: (i, j) row-major, several bin pairs of one (i, j) in emission order
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t o = A.perm_f[f0 + x];
        const fd_pair_rec p = A.found[o];
        s_a[x] = p.i; s_a[RS_EDGE_CAP + x] = p.j; s_b[x] = p.hash; s_b[RS_EDGE_CAP + x] = o;
    }
    RS_SYNC();
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t i = s_a[x], j = s_a[RS_EDGE_CAP + x], o = s_b[RS_EDGE_CAP + x];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < F; ++y) {
            const uint32_t yi = s_a[y], yj = s_a[RS_EDGE_CAP + y], yo = s_b[RS_EDGE_CAP + y];
            rank += (yi < i || (yi == i && (yj < j || (yj == j && yo < o)))) ? 1u : 0u;
        }
        s_i[rank] = i; s_j[rank] = j; s_h[rank] = s_b[x];
    }
    RS_SYNC();
    // query-map entry (first one holding the hash, like the reference's hash map) and symmetry flag of every edge
    stamp(0);
    // (the query's sorted hashes through s_a, free between the ranking above and the votes below: the bisection's six dependent loads stay in LDS)
    const bool hs_lds = Q.n_hashes <= RS_LIST_CAP;
    if (hs_lds) { for (uint32_t x = lane; x < Q.n_hashes; x += FD_WAVE) s_a[x] = A.hashes[Q.qh_off + x]; RS_SYNC(); }
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t h = s_h[x];
        const uint32_t *hs = A.hashes + Q.qh_off;
        uint32_t lo = 0, hi = Q.n_hashes;
        if (hs_lds) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_a[mid] < h) lo = mid + 1; else hi = mid; } }
        else { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (hs[mid] < h) lo = mid + 1; else hi = mid; } }
        const bool ok = lo < Q.n_hashes && (hs_lds ? s_a[lo] : hs[lo]) == h;
        s_k[x] = ok ? (int32_t)A.kfirst[Q.qh_off + lo] : -1;
        s_sym[x] = ok ? A.sym[Q.qh_off + lo] : (uint8_t)0;
    }
    stamp(1);
    // ---- nodes in first-appearance order, adjacency rows
    uint32_t node_res = 0xffffffffu, n_nodes = 0;
    uint64_t adj = 0;
    for (uint32_t e = 0; e < F; ++e) {
        uint32_t ids[2];
        for (int z = 0; z < 2; ++z) {
            const uint32_t r = z ? s_j[e] : s_i[e];
            const uint64_t m = __ballot(lane < n_nodes && node_res == r);
            if (m == 0ull) {
                if (n_nodes == A.node_cap) RS_OVERFLOW();
                if (lane == n_nodes) node_res = r;
                ids[z] = n_nodes++;
            } else ids[z] = (uint32_t)__builtin_ctzll(m);
        }
        if (lane == ids[0]) adj |= 1ull << ids[1];
        if (lane == 0) { s_es[e] = (uint8_t)ids[0]; s_et[e] = (uint8_t)ids[1]; }
    }
    RS_SYNC();
    stamp(2);
    // the edges' query-map fields (query residues of the pair, idf) take the place of i / j / hash, which nothing reads any more
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const int32_t k = s_k[x];
        if (k >= 0) { s_h[x] = A.map_qi[Q.map_off + (uint32_t)k]; s_i[x] = A.map_qj[Q.map_off + (uint32_t)k]; s_j[x] = __float_as_uint(A.map_idf[Q.map_off + (uint32_t)k]); }
    }
    RS_SYNC();
    // ---- strongly and weakly connected components (graph.rs:29-50)
    const uint64_t self = lane < n_nodes ? 1ull << lane : 0ull;
    uint64_t reach = adj | self, adjT = 0, reachT = 0;
    for (uint32_t k = 0; k < n_nodes; ++k) { const uint64_t rk = rs_bcast64(reach, k); if ((reach >> k) & 1ull) reach |= rk; }
    for (uint32_t u = 0; u < n_nodes; ++u) {
        const uint64_t au = rs_bcast64(adj, u), ru = rs_bcast64(reach, u);
        if ((au >> lane) & 1ull) adjT |= 1ull << u;
        if ((ru >> lane) & 1ull) reachT |= 1ull << u;
    }
    uint64_t wcc = lane < n_nodes ? (adj | adjT | self) : 0ull;
    for (uint32_t k = 0; k < n_nodes; ++k) { const uint64_t uk = rs_bcast64(wcc, k); if ((wcc >> k) & 1ull) wcc |= uk; }
    const uint64_t scc = reach & reachT;
    const bool scc_rep = lane < n_nodes && (uint32_t)__builtin_ctzll(scc) == lane && (uint32_t)__popcll(scc) >= A.node_count;
    const bool wcc_rep = lane < n_nodes && (uint32_t)__builtin_ctzll(wcc) == lane && (uint32_t)__popcll(wcc) >= A.node_count && wcc != scc;
    const uint64_t ms = __ballot(scc_rep), mw = __ballot(wcc_rep);
    const uint32_t n_scc = (uint32_t)__popcll(ms), n_comp = n_scc + (uint32_t)__popcll(mw);
    stamp(3);
    if (n_comp == 0) {
        if (A.dbg_slot && lane == 0) A.dbg_slot[slot] = make_uint4(F, c1 - c0, 0u, (uint32_t)(wall_clock64() - t_slot0));
        return;
    }
    if (lane < NQ) s_idx[lane] = A.indices[Q.idx_off + lane];
    if (cands_lds)
        for (uint32_t x = lane; x < c1 - c0; x += FD_WAVE) {
            const fd_cand_rec cr = A.cands[A.perm_c[c0 + x]];
            s_cq[x] = cr.qi; s_ci[x] = cr.i; s_cj[x] = cr.j;
        }
    if (A.dbg && lane == 0) atomicAdd(&A.dbg[7], 1ull);
    if (scc_rep) s_cm[fd_mbcnt(ms)] = scc;
    if (wcc_rep) s_cm[n_scc + fd_mbcnt(mw)] = wcc;
    RS_SYNC();
    for (uint32_t x = lane; x < n_comp; x += FD_WAVE) {
        const uint64_t a = s_cm[x];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < n_comp; ++y) { const uint64_t b = s_cm[y]; if (b != a && rs_less(b, a)) ++rank; }
        s_cs[rank] = a;
    }
    RS_SYNC();
    stamp(4);
    // every component writes exactly one record: the slot claims its records and residue ints with ONE returning atomic each, here, long before
    // their values are needed (the per-record claims of 10^4 records on one cache line were two thirds of the output phase)
    unsigned long long mi0 = 0, rp0 = 0;
    if (lane == 0) {
        mi0 = atomicAdd(&A.counters[0], (unsigned long long)n_comp);
        rp0 = atomicAdd(&A.counters[2 * RS_CNT_STRIDE], 2ull * NQ * n_comp);
    }
    uint32_t n_emit = 0;       // records this slot has written (their order among the slot's records: components ascend)
    for (uint32_t ci = 0; ci < n_comp; ++ci) {
        const uint64_t C = s_cs[ci];
        const uint32_t csize = (uint32_t)__popcll(C);
        // ---- votes (query residue, target residue) -> saturating u8 count (retrieve.rs:631-666), subgraph idf (:705-719)
        uint32_t nv = 0;
        float sub_idf = 0.0f;
        for (uint32_t e = 0; e < F; ++e) {
            const uint32_t a = s_es[e], b = s_et[e];
            const int32_t k = s_k[e];
            if (!((C >> a) & 1ull) || !((C >> b) & 1ull) || k < 0) continue;
            sub_idf += __uint_as_float(s_j[e]);
            const uint32_t qi = s_h[e], qj = s_i[e];
            const uint32_t ri = (uint32_t)__shfl((int)node_res, (int)a, FD_WAVE), rj = (uint32_t)__shfl((int)node_res, (int)b, FD_WAVE);
            uint32_t pq[2], pr[2];
            if (s_sym[e]) { pq[0] = min(qi, qj); pq[1] = max(qi, qj); pr[0] = min(ri, rj); pr[1] = max(ri, rj); }
            else { pq[0] = qi; pr[0] = ri; pq[1] = qj; pr[1] = rj; }
            for (int z = 0; z < 2; ++z) {
                int32_t at = -1;
                for (uint32_t base = 0; base < nv && at < 0; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    const uint64_t m = __ballot(x < nv && s_a[x] == pq[z] && s_b[x] == pr[z]);
                    if (m) at = (int32_t)(base + (uint32_t)__builtin_ctzll(m));
                }
                if (at < 0) {
                    if (nv == RS_LIST_CAP) RS_OVERFLOW();
                    if (lane == 0) { s_a[nv] = pq[z]; s_b[nv] = pr[z]; s_vc[nv] = 1; }
                    ++nv;
                } else if (lane == 0 && s_vc[at] < 255) ++s_vc[at];
                RS_SYNC();
            }
        }
        stamp(8);
        // ---- per query residue: highest count, smallest target residue holding it
        uint32_t bq = 0, bc = 0, br = 0, nb = 0;
        for (uint32_t v = 0; v < nv; ++v) {
            const uint32_t q = s_a[v], r = s_b[v], c = s_vc[v];
            const uint64_t m = __ballot(lane < nb && bq == q);
            if (m == 0ull) {
                if (nb == FD_WAVE) RS_OVERFLOW();
                if (lane == nb) { bq = q; bc = c; br = r; }
                ++nb;
            } else if (lane == (uint32_t)__builtin_ctzll(m) && (c > bc || (c == bc && r < br))) { bc = c; br = r; }
        }
        // ---- greedy assignment in (count descending, query residue ascending) order (retrieve.rs:668-690)
        uint32_t my_aq = 0xffffffffu, my_ar = 0xffffffffu, n_asg = 0;
        bool rem = lane < nb;
        for (uint32_t it = 0; it < nb && n_asg < csize; ++it) {
            const uint64_t key = rem ? (((uint64_t)bc << 32) | (0xffffffffu - bq)) : 0ull;
            const uint64_t mx = rs_wave_max64(key);
            const uint32_t w = (uint32_t)__builtin_ctzll(__ballot(rem && key == mx));
            const uint32_t q = (uint32_t)__shfl((int)bq, (int)w, FD_WAVE), r = (uint32_t)__shfl((int)br, (int)w, FD_WAVE);
            if (__ballot(lane < n_asg && my_ar == r) == 0ull) {
                if (lane == n_asg) { my_aq = q; my_ar = r; }
                ++n_asg;
            }
            if (lane == w) rem = false;
        }
        RS_SYNC();           // the vote arrays are free from here on
        if (lane < n_asg) { s_aq[lane] = my_aq; s_ar[lane] = my_ar; }
        RS_SYNC();
        stamp(9);
        // ---- rescue votes (retrieve.rs:498-511): for a query residue without a target, the candidate pairs (query residue, i, j)
        // whose partner j some assignment mapped vote for i; the unique maximum (>= 2) joins
        for (uint32_t pos = 0; pos < NQ; ++pos) {
            const uint32_t qi = s_idx[pos];
            uint32_t mx = 0, nmx = 0, arg = 0;
            if (__ballot(lane < n_asg && my_aq == qi) == 0ull && c1 > c0 && qi < Q.q_size) {
                uint32_t n_t = 0;
                for (uint32_t base = c0; base < c1; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    bool ok = false;
                    uint32_t iv = 0;
                    if (x < c1) {
                        uint32_t c_q, c_i, c_j;
                        if (cands_lds) { c_q = s_cq[x - c0]; c_i = s_ci[x - c0]; c_j = s_cj[x - c0]; }
                        else { const fd_cand_rec cr = A.cands[A.perm_c[x]]; c_q = cr.qi; c_i = cr.i; c_j = cr.j; }
                        iv = c_i;
                        if (c_q == qi && c_i < Rt && c_j < Rt)
                            for (uint32_t k = 0; k < n_asg; ++k) ok |= s_ar[k] == c_j;
                    }
                    const uint64_t m = __ballot(ok);
                    if (ok) { const uint32_t p = n_t + fd_mbcnt(m); if (p < RS_LIST_CAP) s_a[p] = iv; }
                    n_t += (uint32_t)__popcll(m);
                }
                if (n_t > RS_LIST_CAP) RS_OVERFLOW();
                RS_SYNC();
                uint32_t lmx = 0;
                for (uint32_t base = 0; base < n_t; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    if (x < n_t) {
                        const uint32_t v = s_a[x];
                        uint32_t cnt = 0;
                        for (uint32_t y = 0; y < n_t; ++y) cnt += s_a[y] == v ? 1u : 0u;
                        s_b[x] = cnt;
                        lmx = max(lmx, cnt);
                    }
                }
                mx = rs_wave_max32(lmx);
                RS_SYNC();
                uint32_t n_eq = 0;
                for (uint32_t base = 0; base < n_t; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    const uint64_t m = __ballot(x < n_t && s_b[x] == mx);
                    if (m) { n_eq += (uint32_t)__popcll(m); arg = s_a[base + (uint32_t)__builtin_ctzll(m)]; }
                }
                nmx = mx ? n_eq / mx : 0u;
                RS_SYNC();
            }
            if (lane == 0) { s_rmx[pos] = mx; s_rnm[pos] = nmx; s_rarg[pos] = arg; }
        }
        RS_SYNC();
        stamp(10);
        // ---- residue assignment + rescue, sequential as in the reference (retrieve.rs:430-516)
        if (lane < NQ) { s_fh[lane] = -1; s_pr[lane] = -1; }
        RS_SYNC();
        if (lane == 0) {
            uint32_t n_sc = 0;
            for (uint32_t pos = 0; pos < NQ; ++pos) {
                const uint32_t qi = s_idx[pos];
                int32_t mapped = -1;
                for (uint32_t k = 0; k < n_asg; ++k) if (s_aq[k] == qi) { mapped = (int32_t)s_ar[k]; break; }
                if (mapped >= 0) {
                    s_fh[pos] = mapped;
                    uint32_t pp = n_sc;
                    for (uint32_t k = 0; k < n_sc; ++k) if (s_rs[k] == (uint32_t)mapped) { pp = k; break; }
                    if (pp == n_sc) { s_pr[pos] = mapped; s_qs[n_sc] = qi; s_rs[n_sc] = (uint32_t)mapped; ++n_sc; }
                    else {
                        if (pp < NQ) s_pr[pp] = -1;          // the reference indexes its residue vector with the scanned position
                        s_pr[pos] = mapped;
                        for (uint32_t k = pp; k + 1 < n_sc; ++k) { s_qs[k] = s_qs[k + 1]; s_rs[k] = s_rs[k + 1]; }
                        s_qs[n_sc - 1] = qi; s_rs[n_sc - 1] = (uint32_t)mapped;
                    }
                } else if (qi < Q.q_size && s_rnm[pos] == 1 && s_rmx[pos] >= 2) {
                    bool taken = false;
                    for (uint32_t k = 0; k < n_sc; ++k) taken |= s_rs[k] == s_rarg[pos];
                    if (!taken && n_sc < FD_WAVE) { s_pr[pos] = (int32_t)s_rarg[pos]; s_qs[n_sc] = qi; s_rs[n_sc] = s_rarg[pos]; ++n_sc; }
                }
            }
            bool same = true;
            for (uint32_t pos = 0; pos < NQ; ++pos) same &= s_fh[pos] == s_pr[pos];
            s_misc[0] = n_sc; s_misc[1] = same ? 1u : 0u;
        }
        RS_SYNC();
        const uint32_t n_sc = s_misc[0];
        const bool same = s_misc[1] != 0;
        stamp(11);
        // ---- outputs: one record, 2 NQ residues, one or two superposition problems of [CA, CB] points (retrieve.rs:761-767)
        const uint32_t nprob = same ? 1u : 2u, npts = 2u * n_asg + (same ? 0u : 2u * n_sc);
        unsigned long long mi = 0, rp = 0, pk = 0;
        if (lane == 0) {
            mi = mi0 + ci; rp = rp0 + 2ull * NQ * ci;
            pk = atomicAdd(&A.counters[RS_CNT_STRIDE], ((unsigned long long)nprob << 40) | (unsigned long long)npts);
        }
        mi = rs_bcast64(mi, 0); rp = rs_bcast64(rp, 0); pk = rs_bcast64(pk, 0);
        const uint64_t p0 = pk >> 40, pt0 = pk & ((1ull << 40) - 1ull);
        if (mi >= A.cap_matches || rp + 2ull * NQ > A.cap_res || p0 + nprob > A.cap_prob || pt0 + npts > A.cap_pts) {
            if (lane == 0) atomicOr(A.flags, 2u);
            RS_SYNC();
            continue;
        }
        if (lane == 0) {
            rs_match_dev m;
            m.slot = slot; m.ci = ci; m.same = same ? 1u : 0u; m.res_pos = (uint32_t)rp; m.prob0 = (uint32_t)p0; m.prob1 = same ? 0xffffffffu : (uint32_t)(p0 + 1);
            m.idf = sub_idf; m.ord = n_emit;
            A.matches[mi] = m;
            A.koff[p0] = pt0; A.d0[p0] = A.d0tab[2u * n_asg];
            if (!same) { A.koff[p0 + 1] = pt0 + 2ull * n_asg; A.d0[p0 + 1] = A.d0tab[2u * n_sc]; }
        }
        if (lane < NQ) { A.residues[rp + lane] = s_fh[lane]; A.residues[rp + NQ + lane] = s_pr[lane]; }
        for (uint32_t w = 0; w < nprob; ++w) {
            const uint32_t n = w ? n_sc : n_asg;
            const uint64_t base = pt0 + (w ? 2ull * n_asg : 0ull);
            if (lane < n) {
                // which residues the problem's points are: the coordinates themselves are gathered by k_rs_points — 12 random 4-byte loads per pair from the
                // 7 GB coordinate arrays were 29 of a slot's 78 us when this wavefront made them, one dependent round per problem
                const uint32_t q = w ? s_qs[lane] : s_aq[lane], r = w ? s_rs[lane] : s_ar[lane];
                A.gq[(base >> 1) + lane] = Q.q_res0 + q;
                A.gr[(base >> 1) + lane] = r0 + r;
            }
        }
        ++n_emit;
        RS_SYNC();
        stamp(12);
    }
    stamp(5);
    if (A.dbg_slot && lane == 0) A.dbg_slot[slot] = make_uint4(F, c1 - c0, n_comp, (uint32_t)(wall_clock64() - t_slot0));
    if (A.dbg && lane == 0) atomicAdd(&A.dbg[6], 1ull);
    if (lane == 0 && A.slot_matches) A.slot_matches[slot] = n_emit;
}

// ------------------------------------------------------------------ the same glue, a wavefront per COMPONENT
// k_rs_slots runs a slot's components one after the other on one wavefront: a launch lasts as long as its heaviest slot (12 components x ~17 us
// at 128 queries per batch), and its 48 KB of LDS — sized for 1,024 found triples — keep three slots per CU resident.  The two kernels below split
// the slot at the component loop:
//   * k_rs_setup (a wavefront per slot, LDS for 128 found triples: all slots of a batch resident at once): edges in scan order, query-map
//     entries, graph, strongly / weakly connected components in the reference's order — everything up to the loop — and the slot's claim of
//     records; what the components need goes to global memory (an edge's node pair, symmetry flag and query-map fields in 16 bytes; the
//     nodes' residues; the components' node sets), one work item per component lands at the place of its record.  A slot with more found
//     triples is listed for k_rs_slots (launched behind, over that list only);
//   * k_rs_comp (a wavefront per component): votes, greedy assignment, rescue, outputs — the loop body of k_rs_slots, statement for statement,
//     with one change that cannot alter a result: the slot's candidate pairs whose partner residue the assignment mapped are filtered ONCE per
//     component (the test does not depend on the query residue) instead of being read again, through two dependent global loads per 64 pairs,
//     for every unmatched query residue — the chain that made the slots with ~1,000 candidate pairs the launch's tail.
#define RS_S_EDGE 128u        // found triples of a slot the split form takes
#define RS_S_LIST 512u        // votes (two per edge at most) and rescue tallies of one component
#define RS_S_FILT 512u        // candidate pairs with an assigned partner, per component (more: the unfiltered walk of k_rs_slots)
#define RS_WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")      /* one wavefront per workgroup: LDS traffic is in order, nothing to wait for in global memory */

__global__ __launch_bounds__(FD_WAVE) void k_rs_setup(rs_args A) {
    __shared__ uint32_t s_a[2 * RS_S_EDGE];    // raw i | raw j           -> the query's sorted hashes
    __shared__ uint32_t s_b[2 * RS_S_EDGE];    // raw hash | raw position
    __shared__ uint32_t s_i[RS_S_EDGE], s_j[RS_S_EDGE], s_h[RS_S_EDGE];   // edges in (i, j, emission) order
    __shared__ int32_t s_k[RS_S_EDGE];         // query-map entry of the edge's hash
    __shared__ uint8_t s_es[RS_S_EDGE], s_et[RS_S_EDGE], s_sym[RS_S_EDGE];
    __shared__ uint64_t s_cm[2 * FD_WAVE], s_cs[2 * FD_WAVE];
    const uint32_t slot = A.order ? A.order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const uint32_t f0 = A.seg_f[slot], F = A.seg_f[slot + 1] - f0;
    if (F == 0) return;
    if (F > RS_S_EDGE) {       // k_rs_slots takes it
        if (lane == 0) { const uint32_t p = atomicAdd(A.sp_big_n, 1u); A.sp_big[p] = slot; }
        return;
    }
    const rs_query_dev Q = A.qt[A.slot_q[slot]];
    const uint32_t NQ = Q.n_idx;
    if (NQ > FD_WAVE) RS_OVERFLOW();
    // ---- edges in the reference's scan order: (i, j) row-major, several bin pairs of one (i, j) in emission order
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t o = A.perm_f[f0 + x];
        const fd_pair_rec p = A.found[o];
        s_a[x] = p.i; s_a[RS_S_EDGE + x] = p.j; s_b[x] = p.hash; s_b[RS_S_EDGE + x] = o;
    }
    RS_WSYNC();
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t i = s_a[x], j = s_a[RS_S_EDGE + x], o = s_b[RS_S_EDGE + x];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < F; ++y) {
            const uint32_t yi = s_a[y], yj = s_a[RS_S_EDGE + y], yo = s_b[RS_S_EDGE + y];
            rank += (yi < i || (yi == i && (yj < j || (yj == j && yo < o)))) ? 1u : 0u;
        }
        s_i[rank] = i; s_j[rank] = j; s_h[rank] = s_b[x];
    }
    RS_WSYNC();
    // query-map entry (first one holding the hash, like the reference's hash map) and symmetry flag of every edge
    const bool hs_lds = Q.n_hashes <= 2 * RS_S_EDGE;
    if (hs_lds) { for (uint32_t x = lane; x < Q.n_hashes; x += FD_WAVE) s_a[x] = A.hashes[Q.qh_off + x]; RS_WSYNC(); }
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const uint32_t h = s_h[x];
        const uint32_t *hs = A.hashes + Q.qh_off;
        uint32_t lo = 0, hi = Q.n_hashes;
        if (hs_lds) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_a[mid] < h) lo = mid + 1; else hi = mid; } }
        else { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (hs[mid] < h) lo = mid + 1; else hi = mid; } }
        const bool ok = lo < Q.n_hashes && (hs_lds ? s_a[lo] : hs[lo]) == h;
        s_k[x] = ok ? (int32_t)A.kfirst[Q.qh_off + lo] : -1;
        s_sym[x] = ok ? A.sym[Q.qh_off + lo] : (uint8_t)0;
    }
    // ---- nodes in first-appearance order, adjacency rows
    uint32_t node_res = 0xffffffffu, n_nodes = 0;
    uint64_t adj = 0;
    for (uint32_t e = 0; e < F; ++e) {
        uint32_t ids[2];
        for (int z = 0; z < 2; ++z) {
            const uint32_t r = z ? s_j[e] : s_i[e];
            const uint64_t m = __ballot(lane < n_nodes && node_res == r);
            if (m == 0ull) {
                if (n_nodes == A.node_cap) RS_OVERFLOW();
                if (lane == n_nodes) node_res = r;
                ids[z] = n_nodes++;
            } else ids[z] = (uint32_t)__builtin_ctzll(m);
        }
        if (lane == ids[0]) adj |= 1ull << ids[1];
        if (lane == 0) { s_es[e] = (uint8_t)ids[0]; s_et[e] = (uint8_t)ids[1]; }
    }
    RS_WSYNC();
    // ---- strongly and weakly connected components (graph.rs:29-50)
    const uint64_t self = lane < n_nodes ? 1ull << lane : 0ull;
    uint64_t reach = adj | self, adjT = 0, reachT = 0;
    for (uint32_t k = 0; k < n_nodes; ++k) { const uint64_t rk = rs_bcast64(reach, k); if ((reach >> k) & 1ull) reach |= rk; }
    for (uint32_t u = 0; u < n_nodes; ++u) {
        const uint64_t au = rs_bcast64(adj, u), ru = rs_bcast64(reach, u);
        if ((au >> lane) & 1ull) adjT |= 1ull << u;
        if ((ru >> lane) & 1ull) reachT |= 1ull << u;
    }
    uint64_t wcc = lane < n_nodes ? (adj | adjT | self) : 0ull;
    for (uint32_t k = 0; k < n_nodes; ++k) { const uint64_t uk = rs_bcast64(wcc, k); if ((wcc >> k) & 1ull) wcc |= uk; }
    const uint64_t scc = reach & reachT;
    const bool scc_rep = lane < n_nodes && (uint32_t)__builtin_ctzll(scc) == lane && (uint32_t)__popcll(scc) >= A.node_count;
    const bool wcc_rep = lane < n_nodes && (uint32_t)__builtin_ctzll(wcc) == lane && (uint32_t)__popcll(wcc) >= A.node_count && wcc != scc;
    const uint64_t ms = __ballot(scc_rep), mw = __ballot(wcc_rep);
    const uint32_t n_scc = (uint32_t)__popcll(ms), n_comp = n_scc + (uint32_t)__popcll(mw);
    if (n_comp == 0) return;
    if (scc_rep) s_cm[fd_mbcnt(ms)] = scc;
    if (wcc_rep) s_cm[n_scc + fd_mbcnt(mw)] = wcc;
    RS_WSYNC();
    for (uint32_t x = lane; x < n_comp; x += FD_WAVE) {
        const uint64_t a = s_cm[x];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < n_comp; ++y) { const uint64_t b = s_cm[y]; if (b != a && rs_less(b, a)) ++rank; }
        s_cs[rank] = a;
    }
    RS_WSYNC();
    // ---- what the components need: per edge {source node | target node << 8 | symmetric << 16 | in the query map << 24, the entry's two query
    // residues, its idf}, the nodes' residues, the components in order.  No claim of records here: k_rs_bases scans the slots' component counts
    // (a returning atomic on ONE address completes every ~20 ns whoever issues it: the 13,000 claims of a 128-query batch — records and
    // residue ints per slot, problems | points per component — WERE the 226 us of k_rs_slots)
    for (uint32_t x = lane; x < F; x += FD_WAVE) {
        const int32_t k = s_k[x];
        uint4 ed = make_uint4((uint32_t)s_es[x] | ((uint32_t)s_et[x] << 8) | ((uint32_t)s_sym[x] << 16) | (k >= 0 ? 1u << 24 : 0u), 0u, 0u, 0u);
        if (k >= 0) { ed.y = A.map_qi[Q.map_off + (uint32_t)k]; ed.z = A.map_qj[Q.map_off + (uint32_t)k]; ed.w = __float_as_uint(A.map_idf[Q.map_off + (uint32_t)k]); }
        A.sp_edges[f0 + x] = ed;
    }
    A.sp_nodes[(uint64_t)slot * FD_WAVE + lane] = node_res;
    for (uint32_t x = lane; x < n_comp; x += FD_WAVE) A.sp_comps[(uint64_t)slot * (2 * FD_WAVE) + x] = s_cs[x];
    if (lane == 0) {
        A.sp_head[2ull * slot] = make_uint4(F, n_comp, n_nodes, NQ);
        A.slot_matches[slot] = n_comp;
    }
}
// first record and first residue int of every slot: exclusive scans of the slots' component counts (x 2 NQ for the residue ints) in slot order,
// one work item (slot, component) per record, the totals where k_rs_slots' claims continue (one workgroup)
__global__ __launch_bounds__(1024) void k_rs_bases(rs_args A, uint32_t n_cand) {
    __shared__ unsigned long long s_w[16];
    const uint32_t tid = threadIdx.x;
    unsigned long long run_m = 0, run_r = 0;
    // passes of 8 x 1,024 slots (slot = pass + 1,024 u + thread: neighbouring threads read neighbouring slots), the pass's loads issued together — a scan
    // per 1,024 slots that waits for its own loads pays a global round trip per scan
    for (uint32_t s0 = 0; s0 < n_cand; s0 += 8192u) {
        uint32_t nc[8], nq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t slot = s0 + 1024u * (uint32_t)u + tid; nc[u] = slot < n_cand ? A.slot_matches[slot] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t slot = s0 + 1024u * (uint32_t)u + tid; nq[u] = nc[u] ? A.sp_head[2ull * slot].w : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + 1024u * (uint32_t)u >= n_cand) break;
            const uint32_t slot = s0 + 1024u * (uint32_t)u + tid;
            unsigned long long tot_m, tot_r;
            const unsigned long long mi0 = run_m + rs_block_excl64((unsigned long long)nc[u], tid, s_w, &tot_m);
            const unsigned long long rp0 = run_r + rs_block_excl64(2ull * nq[u] * nc[u], tid, s_w, &tot_r);
            if (nc[u]) {
                A.sp_head[2ull * slot + 1] = make_uint4((uint32_t)mi0, (uint32_t)(mi0 >> 32), (uint32_t)rp0, (uint32_t)(rp0 >> 32));
                for (uint32_t x = 0; x < nc[u]; ++x) if (mi0 + x < A.cap_matches) A.sp_work[mi0 + x] = make_uint2(slot, x);
            }
            run_m += tot_m; run_r += tot_r;
        }
    }
    if (tid == 0) {
        A.counters[0] = run_m; A.counters[2 * RS_CNT_STRIDE] = run_r;
        if (run_m > A.cap_matches || run_r > A.cap_res) atomicOr(A.flags, 2u);
    }
}
// problems and points of the records k_rs_comp wrote: exclusive scans of (problems, points) over the records, then every record's places — its
// problems' first points and d0, its residue pairs side by side in gq / gr (k_rs_points gathers the coordinates) — and the totals (one workgroup)
__global__ __launch_bounds__(1024) void k_rs_pack(rs_args A) {
    __shared__ unsigned long long s_w[16];
    __shared__ float s_d0[2 * FD_WAVE + 1];
    const uint32_t tid = threadIdx.x;
    const unsigned long long n_all = A.counters[0];
    const uint64_t n_rec = n_all < A.cap_matches ? n_all : A.cap_matches;
    if (tid <= 2 * FD_WAVE) s_d0[tid] = A.d0tab[tid];
    __syncthreads();
    unsigned long long run = 0;       // problems << 40 | points
    for (uint64_t k0 = 0; k0 < n_rec; k0 += 8192u) {       // passes of 8 x 1,024 records, the pass's loads issued together (as k_rs_bases)
        uint4 np[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t k = k0 + 1024u * (uint32_t)u + tid;
            np[u] = k < n_rec ? A.sp_np[k] : make_uint4(0u, 0u, 0u, 0u);        // {problems, points, assigned, rescued list}
            if (np[u].x > 2u || np[u].y > 4u * FD_WAVE) np[u] = make_uint4(0u, 0u, 0u, 0u);      // (a component that bailed out — the call's overflow flag is up — left its entry unwritten)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (k0 + 1024u * (uint32_t)u >= n_rec) break;
            const uint64_t k = k0 + 1024u * (uint32_t)u + tid;
            unsigned long long tot;
            const unsigned long long pk = run + rs_block_excl64(((unsigned long long)np[u].x << 40) | np[u].y, tid, s_w, &tot);
            run += tot;
            if (k >= n_rec) continue;
            const uint64_t p0 = pk >> 40, pt0 = pk & ((1ull << 40) - 1ull);
            if (!np[u].x) { A.sp_np[k].y = 0u; continue; }
            if (p0 + np[u].x > A.cap_prob || pt0 + np[u].y > A.cap_pts) { atomicOr(A.flags, 2u); A.sp_np[k].y = 0u; continue; }
            A.matches[k].prob0 = (uint32_t)p0; A.matches[k].prob1 = np[u].x == 2u ? (uint32_t)(p0 + 1) : 0xffffffffu;
            A.koff[p0] = pt0; A.d0[p0] = s_d0[2u * np[u].z];
            if (np[u].x == 2u) { A.koff[p0 + 1] = pt0 + 2ull * np[u].z; A.d0[p0 + 1] = s_d0[2u * np[u].w]; }
            A.sp_np[k].w = (uint32_t)(pt0 >> 1);        // first residue pair of the record in gq / gr (k_rs_pairs)
        }
    }
    if (tid == 0) A.counters[RS_CNT_STRIDE] = run;
}
// a record's residue pairs from its own place to the problems' (a wavefront per record)
__global__ __launch_bounds__(FD_WAVE) void k_rs_pairs(rs_args A) {
    const unsigned long long n_all = A.counters[0];
    const uint64_t n_rec = n_all < A.cap_matches ? n_all : A.cap_matches;
    for (uint64_t k = blockIdx.x; k < n_rec; k += gridDim.x) {
        const uint4 np = A.sp_np[k];
        const uint32_t n_pairs = np.y >> 1;
        for (uint32_t x = threadIdx.x; x < n_pairs && x < 2 * FD_WAVE; x += FD_WAVE) {
            A.gq[(uint64_t)np.w + x] = A.sp_gq[k * (2 * FD_WAVE) + x];
            A.gr[(uint64_t)np.w + x] = A.sp_gr[k * (2 * FD_WAVE) + x];
        }
    }
}

__global__ __launch_bounds__(FD_WAVE) void k_rs_comp(rs_args A) {
    __shared__ uint32_t s_a[RS_S_LIST];        // votes: query residue   -> rescue: partner-residue list
    __shared__ uint32_t s_b[RS_S_LIST];        // votes: target residue  -> rescue: multiplicities
    __shared__ uint8_t s_vc[RS_S_LIST];
    __shared__ uint32_t s_i[RS_S_EDGE], s_j[RS_S_EDGE], s_h[RS_S_EDGE];   // an edge's query-map fields: second query residue, idf, first query residue
    __shared__ uint8_t s_es[RS_S_EDGE], s_et[RS_S_EDGE], s_sym[RS_S_EDGE], s_kv[RS_S_EDGE];
    __shared__ uint32_t s_aq[FD_WAVE], s_ar[FD_WAVE], s_qs[FD_WAVE], s_rs[FD_WAVE], s_rmx[FD_WAVE], s_rnm[FD_WAVE], s_rarg[FD_WAVE];
    __shared__ int32_t s_fh[FD_WAVE], s_pr[FD_WAVE];
    __shared__ uint32_t s_misc[2];
    __shared__ uint32_t s_idx[FD_WAVE];
    __shared__ uint32_t s_fq[RS_S_FILT], s_fi[RS_S_FILT];     // candidate pairs (query residue, i) whose partner j the assignment mapped
    const uint32_t lane = threadIdx.x;
    const unsigned long long n_all = A.counters[0];
    const uint64_t n_work = n_all < A.cap_matches ? n_all : A.cap_matches;
    for (uint64_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint2 wk = A.sp_work[wi];
        const uint32_t slot = wk.x, ci = wk.y;
        const uint4 h0 = A.sp_head[2ull * slot], h1 = A.sp_head[2ull * slot + 1];
        const uint32_t F = h0.x, n_nodes = h0.z;
        const unsigned long long mi0 = (unsigned long long)h1.x | ((unsigned long long)h1.y << 32), rp0 = (unsigned long long)h1.z | ((unsigned long long)h1.w << 32);
        const uint32_t f0 = A.seg_f[slot];
        const rs_query_dev Q = A.qt[A.slot_q[slot]];
        const uint32_t NQ = Q.n_idx;
        const uint32_t st = A.cand[slot];
        const uint32_t r0 = A.db_res_off[st], Rt = A.db_res_off[st + 1] - r0;
        const uint32_t c0 = A.seg_c[slot], c1 = A.seg_c[slot + 1];
        const uint64_t C = A.sp_comps[(uint64_t)slot * (2 * FD_WAVE) + ci];
        const uint32_t csize = (uint32_t)__popcll(C);
        const uint32_t node_res = lane < n_nodes ? A.sp_nodes[(uint64_t)slot * FD_WAVE + lane] : 0xffffffffu;
        for (uint32_t x = lane; x < F; x += FD_WAVE) {
            const uint4 ed = A.sp_edges[f0 + x];
            s_es[x] = (uint8_t)ed.x; s_et[x] = (uint8_t)(ed.x >> 8); s_sym[x] = (uint8_t)(ed.x >> 16); s_kv[x] = (uint8_t)(ed.x >> 24);
            s_h[x] = ed.y; s_i[x] = ed.z; s_j[x] = ed.w;
        }
        if (lane < NQ) s_idx[lane] = A.indices[Q.idx_off + lane];
        RS_WSYNC();
        // ---- votes (query residue, target residue) -> saturating u8 count (retrieve.rs:631-666), subgraph idf (:705-719)
        uint32_t nv = 0;
        float sub_idf = 0.0f;
        for (uint32_t e = 0; e < F; ++e) {
            const uint32_t a = s_es[e], b = s_et[e];
            if (!((C >> a) & 1ull) || !((C >> b) & 1ull) || !s_kv[e]) continue;
            sub_idf += __uint_as_float(s_j[e]);
            const uint32_t qi = s_h[e], qj = s_i[e];
            const uint32_t ri = (uint32_t)__shfl((int)node_res, (int)a, FD_WAVE), rj = (uint32_t)__shfl((int)node_res, (int)b, FD_WAVE);
            uint32_t pq[2], pr[2];
            if (s_sym[e]) { pq[0] = min(qi, qj); pq[1] = max(qi, qj); pr[0] = min(ri, rj); pr[1] = max(ri, rj); }
            else { pq[0] = qi; pr[0] = ri; pq[1] = qj; pr[1] = rj; }
            for (int z = 0; z < 2; ++z) {
                int32_t at = -1;
                for (uint32_t base = 0; base < nv && at < 0; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    const uint64_t m = __ballot(x < nv && s_a[x] == pq[z] && s_b[x] == pr[z]);
                    if (m) at = (int32_t)(base + (uint32_t)__builtin_ctzll(m));
                }
                if (at < 0) {
                    if (nv == RS_S_LIST) RS_OVERFLOW();
                    if (lane == 0) { s_a[nv] = pq[z]; s_b[nv] = pr[z]; s_vc[nv] = 1; }
                    ++nv;
                } else if (lane == 0 && s_vc[at] < 255) ++s_vc[at];
                RS_WSYNC();
            }
        }
        // ---- per query residue: highest count, smallest target residue holding it
        uint32_t bq = 0, bc = 0, br = 0, nb = 0;
        for (uint32_t v = 0; v < nv; ++v) {
            const uint32_t q = s_a[v], r = s_b[v], c = s_vc[v];
            const uint64_t m = __ballot(lane < nb && bq == q);
            if (m == 0ull) {
                if (nb == FD_WAVE) RS_OVERFLOW();
                if (lane == nb) { bq = q; bc = c; br = r; }
                ++nb;
            } else if (lane == (uint32_t)__builtin_ctzll(m) && (c > bc || (c == bc && r < br))) { bc = c; br = r; }
        }
        // ---- greedy assignment in (count descending, query residue ascending) order (retrieve.rs:668-690)
        uint32_t my_aq = 0xffffffffu, my_ar = 0xffffffffu, n_asg = 0;
        bool rem = lane < nb;
        for (uint32_t it = 0; it < nb && n_asg < csize; ++it) {
            const uint64_t key = rem ? (((uint64_t)bc << 32) | (0xffffffffu - bq)) : 0ull;
            const uint64_t mx = rs_wave_max64(key);
            const uint32_t w = (uint32_t)__builtin_ctzll(__ballot(rem && key == mx));
            const uint32_t q = (uint32_t)__shfl((int)bq, (int)w, FD_WAVE), r = (uint32_t)__shfl((int)br, (int)w, FD_WAVE);
            if (__ballot(lane < n_asg && my_ar == r) == 0ull) {
                if (lane == n_asg) { my_aq = q; my_ar = r; }
                ++n_asg;
            }
            if (lane == w) rem = false;
        }
        RS_WSYNC();           // the vote arrays are free from here on
        if (lane < n_asg) { s_aq[lane] = my_aq; s_ar[lane] = my_ar; }
        RS_WSYNC();
        // ---- the slot's candidate pairs (query residue, i, j) whose partner j some assignment mapped, in their order: what can vote in the rescue
        // (eight chunks of 64 pairs at a time: their positions, then their records, requested together — a chunk per step paid two dependent
        // global round trips per 64 pairs, ~40 us for a slot with a thousand pairs: the tail of the launch)
        uint32_t n_f = 0;
        for (uint32_t base0 = c0; base0 < c1; base0 += 8u * FD_WAVE) {
            uint32_t pc[8];
            fd_cand_rec cr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t x = base0 + (uint32_t)u * FD_WAVE + lane; pc[u] = x < c1 ? A.perm_c[x] : 0xffffffffu; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { cr[u].cand = 0; cr[u].qi = 0; cr[u].i = 0xffffffffu; cr[u].j = 0xffffffffu; if (pc[u] != 0xffffffffu) cr[u] = A.cands[pc[u]]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (base0 + (uint32_t)u * FD_WAVE >= c1) break;
                bool ok = false;
                if (pc[u] != 0xffffffffu && cr[u].i < Rt && cr[u].j < Rt)
                    for (uint32_t k = 0; k < n_asg; ++k) ok |= s_ar[k] == cr[u].j;
                const uint64_t m = __ballot(ok);
                if (ok) { const uint32_t p = n_f + fd_mbcnt(m); if (p < RS_S_FILT) { s_fq[p] = cr[u].qi; s_fi[p] = cr[u].i; } }
                n_f += (uint32_t)__popcll(m);
            }
        }
        const bool filt = n_f <= RS_S_FILT;
        RS_WSYNC();
        // ---- rescue votes (retrieve.rs:498-511): for a query residue without a target, those pairs of it vote for i; the unique maximum (>= 2) joins
        for (uint32_t pos = 0; pos < NQ; ++pos) {
            const uint32_t qi = s_idx[pos];
            uint32_t mx = 0, nmx = 0, arg = 0;
            if (__ballot(lane < n_asg && my_aq == qi) == 0ull && c1 > c0 && qi < Q.q_size) {
                uint32_t n_t = 0;
                const uint32_t lo = filt ? 0u : c0, hi = filt ? n_f : c1;
                for (uint32_t base = lo; base < hi; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    bool ok = false;
                    uint32_t iv = 0;
                    if (x < hi) {
                        if (filt) { iv = s_fi[x]; ok = s_fq[x] == qi; }
                        else {
                            const fd_cand_rec cr = A.cands[A.perm_c[x]];
                            iv = cr.i;
                            if (cr.qi == qi && cr.i < Rt && cr.j < Rt)
                                for (uint32_t k = 0; k < n_asg; ++k) ok |= s_ar[k] == cr.j;
                        }
                    }
                    const uint64_t m = __ballot(ok);
                    if (ok) { const uint32_t p = n_t + fd_mbcnt(m); if (p < RS_S_LIST) s_a[p] = iv; }
                    n_t += (uint32_t)__popcll(m);
                }
                if (n_t > RS_S_LIST) RS_OVERFLOW();
                RS_WSYNC();
                uint32_t lmx = 0;
                for (uint32_t base = 0; base < n_t; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    if (x < n_t) {
                        const uint32_t v = s_a[x];
                        uint32_t cnt = 0;
                        for (uint32_t y = 0; y < n_t; ++y) cnt += s_a[y] == v ? 1u : 0u;
                        s_b[x] = cnt;
                        lmx = max(lmx, cnt);
                    }
                }
                mx = rs_wave_max32(lmx);
                RS_WSYNC();
                uint32_t n_eq = 0;
                for (uint32_t base = 0; base < n_t; base += FD_WAVE) {
                    const uint32_t x = base + lane;
                    const uint64_t m = __ballot(x < n_t && s_b[x] == mx);
                    if (m) { n_eq += (uint32_t)__popcll(m); arg = s_a[base + (uint32_t)__builtin_ctzll(m)]; }
                }
                nmx = mx ? n_eq / mx : 0u;
                RS_WSYNC();
            }
            if (lane == 0) { s_rmx[pos] = mx; s_rnm[pos] = nmx; s_rarg[pos] = arg; }
        }
        RS_WSYNC();
        // ---- residue assignment + rescue, sequential as in the reference (retrieve.rs:430-516)
        if (lane < NQ) { s_fh[lane] = -1; s_pr[lane] = -1; }
        RS_WSYNC();
        if (lane == 0) {
            uint32_t n_sc = 0;
            for (uint32_t pos = 0; pos < NQ; ++pos) {
                const uint32_t qi = s_idx[pos];
                int32_t mapped = -1;
                for (uint32_t k = 0; k < n_asg; ++k) if (s_aq[k] == qi) { mapped = (int32_t)s_ar[k]; break; }
                if (mapped >= 0) {
                    s_fh[pos] = mapped;
                    uint32_t pp = n_sc;
                    for (uint32_t k = 0; k < n_sc; ++k) if (s_rs[k] == (uint32_t)mapped) { pp = k; break; }
                    if (pp == n_sc) { s_pr[pos] = mapped; s_qs[n_sc] = qi; s_rs[n_sc] = (uint32_t)mapped; ++n_sc; }
                    else {
                        if (pp < NQ) s_pr[pp] = -1;          // the reference indexes its residue vector with the scanned position
                        s_pr[pos] = mapped;
                        for (uint32_t k = pp; k + 1 < n_sc; ++k) { s_qs[k] = s_qs[k + 1]; s_rs[k] = s_rs[k + 1]; }
                        s_qs[n_sc - 1] = qi; s_rs[n_sc - 1] = (uint32_t)mapped;
                    }
                } else if (qi < Q.q_size && s_rnm[pos] == 1 && s_rmx[pos] >= 2) {
                    bool taken = false;
                    for (uint32_t k = 0; k < n_sc; ++k) taken |= s_rs[k] == s_rarg[pos];
                    if (!taken && n_sc < FD_WAVE) { s_pr[pos] = (int32_t)s_rarg[pos]; s_qs[n_sc] = qi; s_rs[n_sc] = s_rarg[pos]; ++n_sc; }
                }
            }
            bool same = true;
            for (uint32_t pos = 0; pos < NQ; ++pos) same &= s_fh[pos] == s_pr[pos];
            s_misc[0] = n_sc; s_misc[1] = same ? 1u : 0u;
        }
        RS_WSYNC();
        const uint32_t n_sc = s_misc[0];
        const bool same = s_misc[1] != 0;
        // ---- outputs: one record, 2 NQ residues, one or two superposition problems of [CA, CB] points (retrieve.rs:761-767)
        const uint32_t nprob = same ? 1u : 2u, npts = 2u * n_asg + (same ? 0u : 2u * n_sc);
        // (the problems' places follow from a scan over the records: k_rs_pack; here the record, its residues, and its residue pairs at the record's own place)
        const unsigned long long mi = mi0 + ci, rp = rp0 + 2ull * NQ * ci;
        if (rp + 2ull * NQ > A.cap_res) {
            if (lane == 0) { atomicOr(A.flags, 2u); A.sp_np[mi] = make_uint4(0u, 0u, 0u, 0u); }
            RS_WSYNC();
            continue;
        }
        if (lane == 0) {
            rs_match_dev m;
            m.slot = slot; m.ci = ci; m.same = same ? 1u : 0u; m.res_pos = (uint32_t)rp; m.prob0 = 0u; m.prob1 = 0xffffffffu;
            m.idf = sub_idf; m.ord = ci;
            A.matches[mi] = m;
            A.sp_np[mi] = make_uint4(nprob, npts, n_asg, n_sc);
        }
        if (lane < NQ) { A.residues[rp + lane] = s_fh[lane]; A.residues[rp + NQ + lane] = s_pr[lane]; }
        for (uint32_t w = 0; w < nprob; ++w) {
            const uint32_t n = w ? n_sc : n_asg, base = w ? n_asg : 0u;
            if (lane < n) {
                const uint32_t q = w ? s_qs[lane] : s_aq[lane], r = w ? s_rs[lane] : s_ar[lane];
                A.sp_gq[mi * (2 * FD_WAVE) + base + lane] = Q.q_res0 + q;
                A.sp_gr[mi * (2 * FD_WAVE) + base + lane] = r0 + r;
            }
        }
        RS_WSYNC();
    }
}

void fd_launch_rs_group(const fd_pair_rec *found, uint64_t nf, const fd_cand_rec *cands, uint64_t nc, uint32_t n_cand, uint32_t *cnt, uint32_t *seg, uint32_t *cur,
                        uint32_t *perm_f, uint32_t *perm_c, hipStream_t st) {
    const uint64_t n = nf + nc;      // (cnt[2 (n_cand + 1)] arrives zeroed: the caller's one fill covers it together with the glue's counters)
    if (n) hipLaunchKernelGGL(k_rs_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, found, nf, cands, nc, n_cand, cnt);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, cnt, n_cand, seg, cur);
    if (n) hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, found, nf, cands, nc, n_cand, cur, perm_f, perm_c);
    // the cursors are spent: their array takes the slots' launch order (rs_args.order = cur)
    if (n_cand) hipLaunchKernelGGL(k_rs_order, dim3(1), dim3(1024), 0, st, cnt, n_cand, cur);
}

// one thread per residue pair of the superposition problems: its two points [CA, CB] of the target (kx) and of the query (ky)
// (and the end of the last problem's points, koff[n_prob] — the host knows both totals from the counters it waited for: written here, not by an 8-byte copy)
__global__ __launch_bounds__(256) void k_rs_points(const uint32_t *__restrict__ gq, const uint32_t *__restrict__ gr, uint64_t n_pairs, const float *__restrict__ db_ca,
                                                   const float *__restrict__ db_cb, const float *__restrict__ q_ca, const float *__restrict__ q_cb, float *__restrict__ kx,
                                                   float *__restrict__ ky, uint64_t *__restrict__ koff_end, uint64_t n_points) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k == 0) *koff_end = n_points;
    if (k >= n_pairs) return;
    const uint64_t q = gq[k], r = gr[k];
    float a[6], b[6];
#pragma unroll
    for (int z = 0; z < 3; ++z) { a[z] = db_ca[3 * r + z]; a[3 + z] = db_cb[3 * r + z]; b[z] = q_ca[3 * q + z]; b[3 + z] = q_cb[3 * q + z]; }
#pragma unroll
    for (int z = 0; z < 6; ++z) { kx[6 * k + z] = a[z]; ky[6 * k + z] = b[z]; }
}
void fd_launch_rs_points(const rs_args &A, uint64_t n_prob, uint64_t n_points, hipStream_t st) {
    const uint64_t n_pairs = n_points / 2;
    hipLaunchKernelGGL(k_rs_points, dim3((unsigned)std::max<uint64_t>((n_pairs + 255) / 256, 1)), dim3(256), 0, st, A.gq, A.gr, n_pairs, A.db_ca, A.db_cb, A.q_ca, A.q_cb, A.kx,
                       A.ky, A.koff + n_prob, n_points);
}

void fd_launch_rs_slots(const rs_args &A, uint32_t n_cand, hipStream_t st) {
    if (!n_cand) return;
    if (!A.sp_work) { hipLaunchKernelGGL(k_rs_slots, dim3(n_cand), dim3(FD_WAVE), 0, st, A); return; }
    // split form: set-up per slot, a wavefront per component, then k_rs_slots over the slots with more found triples than the split form takes
    hipLaunchKernelGGL(k_rs_setup, dim3(n_cand), dim3(FD_WAVE), 0, st, A);
    hipLaunchKernelGGL(k_rs_bases, dim3(1), dim3(1024), 0, st, A, n_cand);
    const uint64_t grid = A.cap_matches < 16384 ? A.cap_matches : 16384;
    hipLaunchKernelGGL(k_rs_comp, dim3((unsigned)grid), dim3(FD_WAVE), 0, st, A);
    hipLaunchKernelGGL(k_rs_pack, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_rs_pairs, dim3((unsigned)grid), dim3(FD_WAVE), 0, st, A);
    rs_args B = A;
    B.order = A.sp_big; B.n_listed = A.sp_big_n;
    hipLaunchKernelGGL(k_rs_slots, dim3(n_cand), dim3(FD_WAVE), 0, st, B);
}

// The caller's match records (fd_match_rec, 39 words) and residue lists in their final order, gathered on the device: one wavefront per
// record; plan[k] = {source record, candidate slot inside its query, first output residue, residue ints}.  The host only orders the
// records (a counting sort over 32-byte headers); the 8 scattered reads per record it did in the solution arrays are lanes here.
__global__ __launch_bounds__(FD_WAVE) void k_rs_records(const rs_match_dev *__restrict__ m, const uint4 *__restrict__ plan, const float *__restrict__ rmsd,
                                                        const float *__restrict__ rot, const float *__restrict__ tran, const float *__restrict__ met,
                                                        const int32_t *__restrict__ residues, uint32_t *__restrict__ out, int32_t *__restrict__ out_res) {
    const uint64_t k = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint4 pl = plan[k];
    const rs_match_dev r = m[pl.x];
    const uint32_t pf = r.prob0, po = r.same ? r.prob0 : r.prob1;
    uint32_t v = 0;
    if (lane == 0) v = pl.y;
    else if (lane == 1) v = r.same;
    else if (lane == 2) v = __float_as_uint(r.idf);
    else if (lane == 3) v = __float_as_uint(rmsd[po]);
    else if (lane == 4) v = __float_as_uint(rmsd[pf]);
    else if (lane < 14) v = __float_as_uint(rot[9ull * po + (lane - 5)]);
    else if (lane < 17) v = __float_as_uint(tran[3ull * po + (lane - 14)]);
    else if (lane < 22) v = __float_as_uint(met[5ull * po + (lane - 17)]);
    else if (lane < 31) v = __float_as_uint(rot[9ull * pf + (lane - 22)]);
    else if (lane < 34) v = __float_as_uint(tran[3ull * pf + (lane - 31)]);
    else if (lane < 39) v = __float_as_uint(met[5ull * pf + (lane - 34)]);
    if (lane < 39) out[39ull * k + lane] = v;
    for (uint32_t z = lane; z < pl.w; z += FD_WAVE) out_res[(uint64_t)pl.z + z] = residues[(uint64_t)r.res_pos + z];
}
// Final places of the records without the host: exclusive scan of the slots' record counts (base of every slot), the per-query offsets the
// caller gets (match_off[t] = base of query t's first slot, res_off = running 2 * n_idx * records) and every slot's first output residue.
__global__ __launch_bounds__(1024) void k_rs_offsets(const uint32_t *__restrict__ slot_matches, uint32_t n_cand, const uint64_t *__restrict__ cand_off,
                                                     const uint32_t *__restrict__ slot_q, const rs_query_dev *__restrict__ qt, uint32_t n_queries,
                                                     uint32_t *__restrict__ mbase, uint32_t *__restrict__ rbase, uint64_t *__restrict__ match_off, uint64_t *__restrict__ res_off) {
    __shared__ unsigned long long part[16];
    const uint32_t tid = threadIdx.x;
    {   // slots
        const uint32_t per = (n_cand + 1023u) / 1024u, a = min(n_cand, tid * per), b = min(n_cand, a + per);
        unsigned long long s = 0;
        for (uint32_t k = a; k < b; ++k) s += slot_matches[k];
        unsigned long long tot;
        uint32_t run = (uint32_t)rs_block_excl64(s, tid, part, &tot);      // (was: thread 0 walking 1,024 LDS words twice — 30 us per batch for a kernel that moves 40 KB)
        if (tid == 0) mbase[n_cand] = (uint32_t)tot;
        for (uint32_t k = a; k < b; ++k) { mbase[k] = run; run += slot_matches[k]; }
        __syncthreads();
    }
    __threadfence_block();
    for (uint32_t t = tid; t <= n_queries; t += 1024) match_off[t] = mbase[cand_off[t]];
    __syncthreads();
    {   // queries: residue ints before each
        const uint32_t per = (n_queries + 1023u) / 1024u, a = min(n_queries, tid * per), b = min(n_queries, a + per);
        auto ints = [&](uint32_t t) { return (match_off[t + 1] - match_off[t]) * 2ull * qt[t].n_idx; };
        unsigned long long s = 0;
        for (uint32_t t = a; t < b; ++t) s += ints(t);
        unsigned long long tot;
        unsigned long long run = rs_block_excl64(s, tid, part, &tot);
        if (tid == 0) res_off[n_queries] = tot;
        for (uint32_t t = a; t < b; ++t) { res_off[t] = run; run += ints(t); }
        __syncthreads();
    }
    for (uint32_t k = tid; k < n_cand; k += 1024) {
        const uint32_t q = slot_q[k];
        rbase[k] = (uint32_t)(res_off[q] + (unsigned long long)(mbase[k] - (uint32_t)match_off[q]) * 2ull * qt[q].n_idx);
    }
}
// one wavefront per SOURCE record: its place is base of its slot + its order inside the slot
__global__ __launch_bounds__(FD_WAVE) void k_rs_records_dev(const rs_match_dev *__restrict__ m, const uint32_t *__restrict__ mbase, const uint32_t *__restrict__ rbase,
                                                            const uint64_t *__restrict__ cand_off, const uint32_t *__restrict__ slot_q, const rs_query_dev *__restrict__ qt,
                                                            const float *__restrict__ rmsd, const float *__restrict__ rot, const float *__restrict__ tran,
                                                            const float *__restrict__ met, const int32_t *__restrict__ residues, uint32_t *__restrict__ out,
                                                            int32_t *__restrict__ out_res) {
    const uint32_t lane = threadIdx.x;
    const rs_match_dev r = m[blockIdx.x];
    const uint32_t q = slot_q[r.slot], nq2 = 2u * qt[q].n_idx;
    const uint64_t k = (uint64_t)mbase[r.slot] + r.ord;
    const uint64_t rpos = (uint64_t)rbase[r.slot] + (uint64_t)r.ord * nq2;
    const uint32_t pf = r.prob0, po = r.same ? r.prob0 : r.prob1;
    uint32_t v = 0;
    if (lane == 0) v = r.slot - (uint32_t)cand_off[q];
    else if (lane == 1) v = r.same;
    else if (lane == 2) v = __float_as_uint(r.idf);
    else if (lane == 3) v = __float_as_uint(rmsd[po]);
    else if (lane == 4) v = __float_as_uint(rmsd[pf]);
    else if (lane < 14) v = __float_as_uint(rot[9ull * po + (lane - 5)]);
    else if (lane < 17) v = __float_as_uint(tran[3ull * po + (lane - 14)]);
    else if (lane < 22) v = __float_as_uint(met[5ull * po + (lane - 17)]);
    else if (lane < 31) v = __float_as_uint(rot[9ull * pf + (lane - 22)]);
    else if (lane < 34) v = __float_as_uint(tran[3ull * pf + (lane - 31)]);
    else if (lane < 39) v = __float_as_uint(met[5ull * pf + (lane - 34)]);
    if (lane < 39) out[39ull * k + lane] = v;
    for (uint32_t z = lane; z < nq2; z += FD_WAVE) out_res[rpos + z] = residues[(uint64_t)r.res_pos + z];
}
void fd_launch_rs_records_dev(const void *matches, uint64_t n, const uint32_t *slot_matches, uint32_t n_cand, const uint64_t *cand_off, const uint32_t *slot_q,
                              const rs_query_dev *qt, uint32_t n_queries, uint32_t *scratch, uint64_t *match_off, uint64_t *res_off, const float *rmsd,
                              const float *rot, const float *tran, const float *met, const int32_t *residues, void *out, int32_t *out_res, hipStream_t st) {
    uint32_t *mbase = scratch, *rbase = scratch + n_cand + 1;
    hipLaunchKernelGGL(k_rs_offsets, dim3(1), dim3(1024), 0, st, slot_matches, n_cand, cand_off, slot_q, qt, n_queries, mbase, rbase, match_off, res_off);
    if (n) hipLaunchKernelGGL(k_rs_records_dev, dim3((unsigned)n), dim3(FD_WAVE), 0, st, (const rs_match_dev *)matches, mbase, rbase, cand_off, slot_q, qt, rmsd, rot, tran, met,
                              residues, (uint32_t *)out, out_res);
}
void fd_launch_rs_records(const void *matches, const void *plan, uint64_t n, const float *rmsd, const float *rot, const float *tran, const float *met,
                          const int32_t *residues, void *out, int32_t *out_res, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_rs_records, dim3((unsigned)n), dim3(FD_WAVE), 0, st, (const rs_match_dev *)matches, (const uint4 *)plan, rmsd, rot, tran, met, residues,
                              (uint32_t *)out, out_res);
}
