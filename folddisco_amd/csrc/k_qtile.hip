// k_qtile.hip — motif-batch scoring without the occupancy matrix: posting lists entered per STRUCTURE TILE, scores kept in LDS.
//
// Replaces, for batches of motif queries with a top-N selection, the k_cq_seg / k_cq_rows_keys / k_topn_*_dense chain of k_query.hip
// (count_query, src/controller/count_query.rs:82-220, followed by the candidate selection of src/cli/workflows/query_pdb.rs:404-411).
// That chain wrote one occupancy row per query hash (rows x S/8 bytes), read it twice and wrote / re-read one ranking key per structure:
// 5-7x the bytes SURVEY §8(d) counts for a query.  Here a workgroup owns (query, tile of 2^14 structures):
//   * the index carries CHECKPOINTS (k_ck_count / k_ck_fill, built once per index like the posting lengths): for every list long enough, the
//     byte position and the preceding id at every 2^j-th boundary of 2,048 ids (QT_CELL_LOG2) — a delta stream can be entered there.  The
//     on-disk format is untouched (the table is derived, device-only);
//   * pass A, two forms.  k_qt_plan + k_qt_score<false> (this file; any batch): (row, cell) -> byte range; the tile's ranges decoded 16 bytes per
//     lane (lane-local varint decode with a 4-byte look-back, one wave scan per 1 KB) into count << 46 | idf in 128 KB of LDS accumulators with one
//     returning ds_add_u64 per posting — a count of zero before the add lists the structure (first touch).  k_qt_layout + k_qt_score32
//     (k_qscore32.hip; batches whose idf sums fit 32 bits — the default for motif queries): planned slot stream, u32 sums, no first-touch list,
//     the tile's own cut.  Either way pass A leaves: (structure, ranking key) lists per (query, tile), a 2,048-bin key histogram per query, and
//     the DECODED STREAM — every 16-byte slot of the tile's ranges as sixteen 16-bit tile-local ids + the slot's row;
//   * k_qt_thr finds the key threshold of the top N (bins of 1.5 % relative width; a second level over the lists only when the threshold
//     bin holds more than the selection's slack);
//   * pass B = k_qt_rows: survivors (key >= threshold) as a bitmap of the tile + ranks, one thread per stream slot tests its ids against the
//     bitmap — no second decode — and sets (row, survivor) bits in LDS; match / edge / node counts and the exact idf sum follow from the rows
//     in (node, partner) order exactly as k_topn_emit_dense derived them.  (k_qt_score<true> = pass B by a second decode: calls without the
//     rows' posting lengths, FDGPU_QT_STREAM=0);
//   * k_qt_sort ranks the survivors;
//   * one query of ~10^5 rows (whole-structure mode): k_qt_score<.., BIG> per (tile, row slice) + k_qd_*.
// Arithmetic (fixed-point idf sums, the ranking key, the record) is the arithmetic of k_query.hip: same bits.
#include <algorithm>
#include "fdgpu_internal.h"
#include "k_qtile.h"

// ------------------------------------------------------------------ checkpoints of an index
// entries of list k: stride 2^j cells (one cell = 2^QT_CELL_LOG2 structure ids), n_e = ceil(NC / 2^j) chunks, the boundaries 1..n_e-1
// stored as (byte offset of the first varint whose id falls at or behind the boundary, id of the posting before it).  j is the
// smallest stride that leaves >= 48 bytes of postings per chunk on average: the table stays below a sixth of the posting bytes.
__device__ __forceinline__ uint32_t qt_stride(uint64_t bytes, uint32_t NC, uint32_t *n_e) {
    uint32_t j = 0;
    for (;; ++j) {
        const uint32_t ne = (uint32_t)(((uint64_t)NC + (1ull << j) - 1ull) >> j);
        if (ne <= 1u || bytes >= 48ull * ne) { *n_e = ne ? ne : 1u; return j; }
    }
}
__global__ void k_ck_count(const uint64_t *__restrict__ offsets, uint64_t H, uint32_t NC, uint32_t *__restrict__ cnt) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H) return;
    uint32_t ne;
    (void)qt_stride(offsets[k + 1] - offsets[k], NC, &ne);
    cnt[k] = ne - 1u;
}
__global__ __launch_bounds__(256) void k_ck_fill(const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ value, uint64_t H, uint32_t NC, uint32_t S,
                                                 uint32_t first_id, const uint64_t *__restrict__ ent_off, unsigned long long *__restrict__ meta,
                                                 uint2 *__restrict__ ent) {
    const uint64_t k = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= H) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = offsets[k], b1 = offsets[k + 1];
    uint32_t n_e;
    const uint32_t j = qt_stride(b1 - b0, NC, &n_e);
    const uint64_t eoff = ent_off[k];
    if (lane == 0) meta[k] = eoff | ((unsigned long long)j << 56);
    if (n_e <= 1u) return;
    uint2 *e = ent + eoff;         // e[b - 1] = boundary b
    auto entry_of = [&](uint32_t id) -> uint32_t {
        if (id < first_id) return 0u;
        const uint32_t rel = id - first_id;
        return rel >= S ? n_e : ((rel >> QT_CELL_LOG2) >> j);
    };
    uint32_t run_id = 0, carry_val = 0, carry_shift = 0, e_last = 0;
    uint64_t pend_start = b0;      // where the varint that straddles into the next block began
    for (uint64_t base = b0; base < b1; base += FD_WAVE) {
        const uint64_t p = base + lane;
        const bool in = p < b1;
        const uint32_t byte = in ? value[p] : 0x80u;
        const bool term = in && !(byte & 0x80u);
        const uint64_t tm = __ballot(term);
        const uint64_t below = tm & ((1ull << lane) - 1ull);
        const int prev_t = below ? 63 - __clzll(below) : -1;
        const uint32_t len_here = lane - (uint32_t)(prev_t + 1) + 1;
        uint32_t v = 0;
        const uint32_t pay = byte & 0x7fu;
#pragma unroll
        for (int back = 4; back >= 0; --back) {
            const uint32_t pb = __shfl(pay, (int)lane - back, FD_WAVE);
            if ((uint32_t)back < len_here) v |= pb << (7u * (len_here - 1u - (uint32_t)back));
        }
        if (term && prev_t < 0) v = carry_val | (v << carry_shift);
        uint32_t s2 = term ? v : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(s2, off, FD_WAVE);
            if ((int)lane >= off) s2 += t;
        }
        const uint32_t id = run_id + s2;
        const uint32_t ec = entry_of(id);
        // the posting before this one: the terminator lane below, or the last posting of the blocks before
        const uint32_t id_prev_l = (uint32_t)__shfl((int)id, prev_t < 0 ? 0 : prev_t, FD_WAVE);
        const uint32_t ec_prev_l = (uint32_t)__shfl((int)ec, prev_t < 0 ? 0 : prev_t, FD_WAVE);
        const uint32_t id_prev = prev_t < 0 ? run_id : id_prev_l;
        const uint32_t ec_prev = prev_t < 0 ? e_last : ec_prev_l;
        if (term && ec > ec_prev) {
            const uint64_t start = prev_t < 0 ? pend_start : base + (uint64_t)prev_t + 1ull;
            const uint32_t hi = ec < n_e ? ec : n_e - 1u;
            for (uint32_t b = ec_prev + 1u; b <= hi; ++b) e[b - 1u] = make_uint2((uint32_t)(start - b0), id_prev);
        }
        if (tm) {
            const int last_t = 63 - __clzll(tm);
            run_id = (uint32_t)__shfl((int)id, last_t, FD_WAVE);
            e_last = (uint32_t)__shfl((int)ec, last_t, FD_WAVE);
            pend_start = base + (uint64_t)last_t + 1ull;
            const uint32_t tail = 63u - (uint32_t)last_t;
            uint32_t pv = 0;
            for (uint32_t t2 = 0; t2 < tail && t2 < 5; ++t2) pv |= (uint32_t)__shfl((int)pay, last_t + 1 + (int)t2, FD_WAVE) << (7u * t2);
            carry_val = pv; carry_shift = 7u * tail;
        }
    }
    // boundaries behind the last posting: empty chunks at the list's end
    for (uint32_t b = e_last + 1u + lane; b <= n_e - 1u; b += FD_WAVE) e[b - 1u] = make_uint2((uint32_t)(b1 - b0), run_id);
}
void fd_launch_ck_count(const uint64_t *offsets, uint64_t H, uint32_t NC, uint32_t *cnt, hipStream_t st) {
    if (H) hipLaunchKernelGGL(k_ck_count, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, offsets, H, NC, cnt);
}
void fd_launch_ck_fill(const uint64_t *offsets, const uint8_t *value, uint64_t H, uint32_t NC, uint32_t S, uint32_t first_id, const uint64_t *ent_off,
                       unsigned long long *meta, void *ent, hipStream_t st) {
    if (H) hipLaunchKernelGGL(k_ck_fill, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, st, offsets, value, H, NC, S, first_id, ent_off, meta, (uint2 *)ent);
}

// ------------------------------------------------------------------ (row, granule) -> byte range
// ranges[g * nq + r] = {first byte (absolute, 64 bits), bytes, id before the first posting}: the piece of row r's list that the checkpoints
// delimit around granule g = 2^plan_log2 structure ids (motif batches: one checkpoint cell, so that no piece is longer than a wavefront
// step or two even for the densest lists; a query of 10^5 rows: a whole tile — 8x fewer, 8x longer pieces).  A list whose entries lie
// further apart than a granule gives the same piece for all granules of an entry: it is handed to the FIRST of them inside the tile, the
// others get none — a tile decodes every piece once.
__global__ void k_qt_plan(qt_args A) {
    // a wavefront = 8 rows x 8 granules (lane = row + 8 * granule), wavefronts walk a row group's granules first.  Threads in granule-major order
    // (lane = row) made every lane chase its own list's offsets, checkpoint metadata and entries — three scattered lines per thread, 39 us per
    // 128 queries, the longest kernel of the prefilter after the scoring itself; row-major order (a wavefront = 64 granules of one row) read one
    // address per wavefront but scattered the 16-byte range stores over 64 lines (25 us, and PMC counted every store as a 32-byte write).  This
    // shape reads 8 lists per wavefront (each address shared by the 8 lanes of a row) and stores 8 full 128-byte lines (the rows of a granule
    // side by side: the table keeps its granule-major layout, a scoring workgroup reads its query's rows side by side).
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cpg_log2 = A.plan_log2 - QT_CELL_LOG2, n_gran = (A.NC + (1u << cpg_log2) - 1u) >> cpg_log2;
    const uint32_t n_gg = (n_gran + 7u) >> 3;
    const uint64_t wave = t >> 6;
    const uint32_t lane = (uint32_t)t & 63u;
    const uint64_t rg = wave / n_gg;
    const uint32_t gg = (uint32_t)(wave - rg * n_gg);
    const uint64_t r64 = rg * 8u + (lane & 7u);
    const uint32_t gran = gg * 8u + (lane >> 3);
    if (r64 >= A.nq || gran >= n_gran) return;
    const uint32_t r = (uint32_t)r64;
    const uint64_t g = (uint64_t)gran * A.nq + r;
    const uint4 out = qt_piece_range(A, r, gran, cpg_log2);
    A.ranges[g] = out;
}

// ------------------------------------------------------------------ tile scoring
// RICH = false: scores of every structure of the tile (pass A).  RICH = true: the records of the survivors (pass B).
// TL2: log2 structures per tile; NTHR threads; RB rows planned per batch; RBW words of row bits (pass B)
// BIG = false: a batch of motif queries, workgroup = (query, tile), all rows of the query.  BIG = true: ONE query of ~10^5 rows (a whole
// structure as the query), workgroup = (tile, slice of the rows): pass A leaves the tile's sums of its slice in A.partial (k_qd_reduce adds the
// slices up and ranks), pass B takes the survivors' bitmap from k_qd_surv and sets their (row, structure) bits in a global matrix.
template <bool RICH, int TL2, int NTHR, int RB, int RBW, bool BIG = false>
__global__ __launch_bounds__(NTHR) void k_qt_score(qt_args A) {
    constexpr uint32_t TILE = 1u << TL2;
    __shared__ unsigned long long s_acc[RICH ? 1 : TILE + FD_WAVE];       // A: count << 46 | idf sum per structure of the tile (+ a slot per lane for adds of nothing)
    __shared__ uint32_t s_hist[RICH || BIG ? 1 : QT_BINS / 2];
    __shared__ uint32_t s_bm[RICH ? TILE / 32 : 1], s_rank[RICH ? TILE / 32 : 1], s_rowbits[RICH && !BIG ? RBW : 1];      // B: survivors, their ranks, their row bits
    __shared__ unsigned long long s_meta[RICH && !BIG ? QT_MAX_ROWS : 1];
    __shared__ uint32_t s_eend[RICH && !BIG ? QT_MAX_ROWS / 32 : 1], s_nend[RICH && !BIG ? QT_MAX_ROWS / 32 : 1];      // B: rows that end an edge / a node
    __shared__ unsigned long long s_byte0[RB], s_add[RB];
    __shared__ uint32_t s_P[RB + 1], s_nbytes[RB], s_prev[RB];
    __shared__ uint16_t s_units[RB], s_row[RB];
    __shared__ uint32_t s_sbase;                  // A: first record of this batch in the decoded stream
    __shared__ uint32_t s_w[NTHR / 64];
    __shared__ uint32_t s_mark[NTHR / 4];         // 64 bytes per wavefront: first lanes of the rows that begin inside a step
    __shared__ uint32_t s_nheavy, s_nlight, s_ucur, s_n, s_cnt, s_base;
    __shared__ uint32_t s_dbg[8];
    // (query, tile) of the workgroup, tiles of a query on consecutive workgroups = spread over the XCDs (all tiles of a query on ONE XCD
    // measured 25 % slower: the queries' weights differ and the XCDs finish apart)
    const uint32_t wg = blockIdx.x;
    if (wg >= A.NT * (BIG ? A.n_slices : A.n_queries)) return;
    const uint32_t t = wg % A.NT, q = BIG ? 0u : wg / A.NT, slice = BIG ? wg / A.NT : 0u, tid = threadIdx.x, lane = tid & 63u;
    unsigned long long tstamp = A.dbg ? wall_clock64() : 0ull;
    auto stamp = [&](int k) {      // FDGPU_QT_DBG: phase durations of the workgroup's first thread, summed over the launch (100 MHz ticks)
        if (A.dbg && tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&A.dbg[(RICH ? 8 : 0) + k], now - tstamp); tstamp = now; }
    };
    if (A.dbg && tid < 8) s_dbg[tid] = 0;
    const uint64_t r0 = BIG ? A.slices[slice] : A.q_rows[q];
    const uint32_t nrows = (uint32_t)((BIG ? A.slices[slice + 1] : A.q_rows[q + 1]) - r0);
    const uint32_t tile_lo = t << TL2;
    const uint32_t tile_lim = A.S - tile_lo < TILE ? A.S - tile_lo : TILE;
    const uint32_t tile_id0 = A.first_id + tile_lo;
    const uint64_t cbase = ((uint64_t)q * A.NT + t) << TL2;
    // work entries of the tile: (granule of the tile, row), granule-major — entry e = granule e / nrows, row e % nrows (BIG: one granule = the tile)
    constexpr uint32_t CPT = BIG ? 1u : 1u << (TL2 - QT_CELL_LOG2);
    const uint32_t cell0 = t * CPT, n_gran = BIG ? A.NT : A.NC, ncell = n_gran - cell0 < CPT ? n_gran - cell0 : CPT;
    const uint32_t n_ent = nrows * ncell;
    auto range_of = [&](uint32_t e) -> uint4 { return A.ranges[(uint64_t)(cell0 + e / nrows) * A.nq + r0 + e % nrows]; };
    // the first batch's ranges are requested before anything else: their latency hides behind the set-up below
    uint4 rg0 = make_uint4(0u, 0u, 0u, 0u);
    unsigned long long rm0 = 0ull;
    if (tid < n_ent && tid < RB) { rg0 = range_of(tid); if (!RICH) rm0 = A.row_meta[r0 + tid % nrows]; }
    uint32_t n_surv = 0, wpr = 1, per_round = 1, n_rounds = 1, out_base = 0;
    if (!RICH) {
        for (uint32_t k = tid; k < TILE; k += NTHR) s_acc[k] = 0ull;
        if (!BIG) for (uint32_t k = tid; k < QT_BINS / 2; k += NTHR) s_hist[k] = 0u;
        if (tid == 0) s_cnt = 0;
    } else if (BIG) {
        if (A.g_tcount[t] == 0u) return;
        for (uint32_t k = tid; k < TILE / 32; k += NTHR) { s_bm[k] = A.g_bm[(uint64_t)t * (TILE / 32) + k]; s_rank[k] = A.g_rank[(uint64_t)t * (TILE / 32) + k]; }
        n_surv = 1; wpr = A.g_wpr; per_round = 0xffffffffu;
    } else {
        // the tile's (structure, key) list: its first entries are requested before its length is known (the buffer holds a full tile)
        constexpr int SPEC = (TILE / NTHR) < 8 ? (TILE / NTHR) : 8;
        uint32_t x[SPEC];        // ranking keys; a survivor's structure comes from the other array
#pragma unroll
        for (int u = 0; u < SPEC; ++u) x[u] = A.c_key[cbase + u * NTHR + tid];
        const uint32_t n = A.ccount[(uint64_t)q * A.NT + t];
        const uint32_t thr = A.state[q].thr_key;
        for (uint32_t k = tid; k < TILE / 32; k += NTHR) s_bm[k] = 0u;
        for (uint32_t k = tid; k < QT_MAX_ROWS / 32; k += NTHR) { s_eend[k] = 0u; s_nend[k] = 0u; }
        __syncthreads();
        for (uint32_t k = tid; k < nrows; k += NTHR) {
            const unsigned long long m = A.row_meta[r0 + k];
            s_meta[k] = m;
            if (m & 1ull) atomicOr(&s_eend[k >> 5], 1u << (k & 31u));
            if (m & 2ull) atomicOr(&s_nend[k >> 5], 1u << (k & 31u));
        }
#pragma unroll
        for (int u = 0; u < SPEC; ++u)
            if ((uint32_t)u * NTHR + tid < n && x[u] >= thr) { const uint32_t i = A.c_nid[cbase + u * NTHR + tid] - tile_lo; atomicOr(&s_bm[i >> 5], 1u << (i & 31u)); }
        for (uint32_t e = SPEC * NTHR + tid; e < n; e += NTHR)
            if (A.c_key[cbase + e] >= thr) { const uint32_t i = A.c_nid[cbase + e] - tile_lo; atomicOr(&s_bm[i >> 5], 1u << (i & 31u)); }
        __syncthreads();
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < TILE / 32; w0 += NTHR) {
            uint32_t tot;
            const uint32_t pc = w0 + tid < TILE / 32 ? (uint32_t)__popc(s_bm[w0 + tid]) : 0u;
            const uint32_t ex = qt_block_excl<NTHR>(pc, tid, s_w, &tot);
            if (w0 + tid < TILE / 32) s_rank[w0 + tid] = run + ex;
            run += tot;
        }
        n_surv = run;
        if (!n_surv) return;
        wpr = (nrows + 31u) >> 5;
        per_round = (uint32_t)RBW / wpr;            // >= 1: the launcher keeps nrows <= QT_MAX_ROWS
        n_rounds = (n_surv + per_round - 1u) / per_round;
        if (tid == 0) out_base = atomicAdd(&A.state[q].count, n_surv);       // the answer is needed when the records are written
    }
    __syncthreads();
    stamp(0);
    for (uint32_t round = 0; round < n_rounds; ++round) {
        const uint32_t s_lo = round * per_round;
        if (RICH && !BIG) {
            const uint32_t nw = (n_surv - s_lo < per_round ? n_surv - s_lo : per_round) * wpr;
            for (uint32_t k = tid; k < nw; k += NTHR) s_rowbits[k] = 0u;
            __syncthreads();
        }
        for (uint32_t ra = 0; ra < n_ent; ra += RB) {
            // ---- the batch's ranges: empty ones dropped, slot starts by one scan (slots in the low 22 bits, entries above)
            const uint32_t nb = n_ent - ra < RB ? n_ent - ra : RB;
            uint4 rg = rg0;
            unsigned long long rm = rm0;
            if (ra || round) {
                rg = make_uint4(0u, 0u, 0u, 0u);
                if (tid < nb) { rg = range_of(ra + tid); if (!RICH) rm = A.row_meta[r0 + (ra + tid) % nrows]; }
            }
            const uint32_t ns = tid < nb ? (rg.z + 15u) >> 4 : 0u;
            uint32_t tot;
            const uint32_t ex = qt_block_excl<NTHR>(ns ? (ns | (1u << 22)) : 0u, tid, s_w, &tot);
            if (ns) {
                const uint32_t c = ex >> 22;
                s_P[c] = ex & 0x3fffffu;
                s_byte0[c] = (unsigned long long)rg.x | ((unsigned long long)rg.y << 32);
                s_nbytes[c] = rg.z; s_prev[c] = rg.w;
                s_add[c] = RICH ? (unsigned long long)((ra + tid) % nrows) + (BIG ? r0 : 0ull) : ((1ull << QT_CNT_SHIFT) | (rm >> 2));
                if (!RICH && !BIG) s_row[c] = (uint16_t)((ra + tid) % nrows);
            }
            if (tid == 0) {
                s_n = tot >> 22; s_P[tot >> 22] = tot & 0x3fffffu; s_nheavy = 0; s_nlight = 0; s_ucur = 0;
                if (!RICH && !BIG && A.stream_ids) {       // this batch's records of the decoded stream: one claim, remembered for k_qt_rows
                    const uint32_t ns_b = tot & 0x3fffffu, sb = ns_b ? atomicAdd(A.stream_used, ns_b) : 0u;
                    s_sbase = sb;
                    A.stream_tab[((uint64_t)q * A.NT + t) * QT_MAXB + ra / RB] = make_uint2(sb, sb + ns_b <= A.stream_cap ? ns_b : 0xffffffffu);
                }
            }
            __syncthreads();
            stamp(1);
            const uint32_t n = s_n;
            // ---- units: the rows that START in one 64-slot window, processed by one wavefront (a row longer than the window stays whole);
            // units that open with a multi-step row are handed out first (from the front of s_units, the others from its back)
            if (tid < n && (tid == 0 || (s_P[tid - 1] >> 6) != (s_P[tid] >> 6))) {
                if (s_P[tid + 1] - s_P[tid] > FD_WAVE) s_units[atomicAdd(&s_nheavy, 1u)] = (uint16_t)tid;
                else s_units[RB - 1u - atomicAdd(&s_nlight, 1u)] = (uint16_t)tid;
            }
            __syncthreads();
            stamp(2);
            const uint32_t n_heavy = s_nheavy, n_units = n_heavy + s_nlight;
            const unsigned long long t_loop = A.dbg ? wall_clock64() : 0ull;
            uint32_t my_units = 0, my_steps = 0;
            for (;;) {
                uint32_t ui = 0;
                if (lane == 0) ui = atomicAdd(&s_ucur, 1u);
                ui = (uint32_t)__builtin_amdgcn_readfirstlane((int)ui);
                if (ui >= n_units) break;
                ++my_units;
                const uint32_t c0 = s_units[ui < n_heavy ? ui : RB - 1u - (ui - n_heavy)], u = s_P[c0] >> 6;
                const uint32_t pc = c0 + 1u + lane;
                const uint64_t outm = __ballot(pc >= n || (s_P[pc < n ? pc : n] >> 6) != u);
                const uint32_t c1 = c0 + 1u + (uint32_t)__builtin_ctzll(outm);
                const uint32_t s_beg = s_P[c0], s_end = s_P[c1];
                // a step = 64 slots of 16 bytes; the next step's bytes are requested before this step's are decoded.  The row of a
                // slot: the rows that start inside the step's window mark their first lane (bytes in the wavefront's scratch), a ballot
                // of the marks + popcount below the lane counts the rows begun so far
                // (volatile: the lanes talk to each other through these bytes — without it the compiler forwards a lane's own 0 to its read)
                volatile uint8_t *mark = reinterpret_cast<volatile uint8_t *>(s_mark) + (tid >> 6) * FD_WAVE;
                uint32_t c_next = c0;          // first row that has not begun yet
                auto prep = [&](uint32_t base, qt_step &X) {
                    mark[lane] = 0;
                    if (c_next + lane < c1) { const uint32_t pos = s_P[c_next + lane] - base; if (pos < FD_WAVE) mark[pos] = 1; }
                    const uint64_t begun = __ballot(mark[lane] != 0);
                    const uint32_t c = c_next - 1u + (uint32_t)__popcll(begun & ((2ull << lane) - 1ull));
                    c_next += (uint32_t)__popcll(begun);
                    const bool active = base + lane < s_end;
                    const uint32_t s = active ? base + lane : s_end - 1u;
                    X.c = c; X.pstart = s_P[c]; X.rel = s - X.pstart;
                    const uint32_t nbytes = s_nbytes[c];
                    X.nby = active ? (nbytes - 16u * X.rel < 16u ? nbytes - 16u * X.rel : 16u) : 0u;
                    __builtin_memcpy(&X.w, A.value + s_byte0[c] + 16ull * X.rel, 16);
                };
                // (the look-ahead only where units have many steps — a large query's tile-sized pieces; in a motif batch pieces end after a step
                // or two and the second step's registers would be spilled)
                constexpr bool AHEAD = BIG;
                qt_step cur, nxt;
                prep(s_beg, cur);
                uint32_t carry = 0, prev_last = 0;
                for (uint32_t base = s_beg; base < s_end; base += FD_WAVE) {
                    const bool more = base + FD_WAVE < s_end;
                    ++my_steps;
                    if (AHEAD && more) prep(base + FD_WAVE, nxt);
                    // ---- lane-local decode: the varints that END in these 16 bytes; the leading bytes of the first are the tail of the
                    // slot before — the lane below's last four bytes (lane 0: lane 63 of the step before)
                    uint32_t lb = (uint32_t)__shfl_up((int)cur.w[3], 1, FD_WAVE);
                    if (lane == 0) lb = prev_last;
                    prev_last = (uint32_t)__builtin_amdgcn_readlane((int)cur.w[3], 63);
                    uint32_t cv = 0, sh = 0;
                    if (cur.rel) {
                        const uint32_t tb = ~lb & 0x80808080u;
                        const uint32_t kc = tb ? (uint32_t)__clz((int)tb) >> 3 : 4u;         // continuation bytes at the end of the look-back
                        if (kc) {
                            const uint32_t x = (lb >> (8u * (4u - kc))) & 0x7f7f7f7fu;
                            cv = (x & 0x7fu) | ((x >> 1) & 0x3f80u) | ((x >> 2) & 0x1fc000u) | ((x >> 3) & 0xfe00000u);
                            sh = 7u * kc;
                        }
                    }
                    uint32_t v[16], T = 0, D = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t b = (cur.w[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        cv |= (b & 0x7fu) << (sh & 31u);
                        const bool term = (uint32_t)i < cur.nby && !(b & 0x80u);
                        v[i] = term ? cv : 0u;
                        T |= term ? (1u << i) : 0u;
                        D += v[i];
                        sh = term ? 0u : sh + 7u;
                        cv = term ? 0u : cv;
                    }
                    // ---- ids: prefix of the lane sums inside the row, from the row's checkpoint id (or the step before)
                    const uint32_t incl = qt_wave_incl(D, lane);
                    const uint32_t fl = cur.pstart > base ? cur.pstart - base : 0u;            // the row's first lane in this step
                    const uint32_t pre = (uint32_t)__shfl((int)(incl - D), (int)fl, FD_WAVE);
                    const uint32_t row_base = cur.pstart < base ? carry : s_prev[cur.c];
                    const uint32_t id_first = row_base + (incl - D) - pre;
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)(id_first + D), 63);
                    const unsigned long long add = s_add[cur.c];
                    uint32_t id = id_first;
                    if (!RICH && BIG) {
                        // every structure is touched by a query of 10^5 rows: no list of touched structures, plain adds
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            id += v[i];
                            const uint32_t x = id - tile_id0;
                            if (((T >> i) & 1u) && x < tile_lim) atomicAdd(&s_acc[x], add);
                        }
                    } else if (!RICH) {
                        // one returning 64-bit LDS add per posting, eight in flight (lanes without a posting add 0 to a slot of their own);
                        // a count of 0 before the add = the structure's first posting of this query
                        uint32_t first = 0;
                        // the slot's postings also leave as 16-bit structure ids inside the tile (0xffff: none) for the decoded stream pass B reads
                        // instead of decoding the lists again; four adds in flight, a quarter of the record stored as soon as it is complete
                        // (eight in flight + the whole record in registers spilled 24 VGPRs)
                        const bool to_stream = A.stream_ids && base + lane < s_end && (uint64_t)s_sbase + base + lane < A.stream_cap;
                        uint2 *sdst = reinterpret_cast<uint2 *>(A.stream_ids) + 4ull * ((uint64_t)s_sbase + base + lane);
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            unsigned long long old[4];
                            uint32_t okm = 0, s2[2] = {0u, 0u};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                id += v[h * 4 + i];
                                const uint32_t x = id - tile_id0;
                                const bool ok = ((T >> (h * 4 + i)) & 1u) && x < tile_lim;
                                okm |= ok ? (1u << i) : 0u;
                                old[i] = atomicAdd(&s_acc[ok ? x : TILE + lane], ok ? add : 0ull);
                                s2[i >> 1] |= (ok ? x : 0xffffu) << ((i & 1) * 16);
                            }
                            if (to_stream) sdst[h] = make_uint2(s2[0], s2[1]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (((okm >> i) & 1u) && (old[i] >> QT_CNT_SHIFT) == 0ull) first |= 1u << (h * 4 + i);
                        }
                        if (to_stream) A.stream_row[(uint64_t)s_sbase + base + lane] = s_row[cur.c];
                        // the touched structures are listed as they are met: the slots of a step's first hits by one LDS atomic per wavefront
                        const uint32_t nf = (uint32_t)__popc(first), fi = qt_wave_incl(nf, lane);
                        const uint32_t ftot = (uint32_t)__builtin_amdgcn_readlane((int)fi, 63);
                        if (ftot) {
                            uint32_t fb = 0;
                            if (lane == 0) fb = atomicAdd(&s_cnt, ftot);
                            uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)fb) + fi - nf;
                            id = id_first;
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                id += v[i];
                                if ((first >> i) & 1u) A.c_nid[cbase + pos++] = id - A.first_id;
                            }
                        }
                    } else {
                        // survivors are rare: the bitmap words of all sixteen postings first, the row bits only where one is set
                        uint32_t hit = 0, hx = 0;       // hx: the LAST hit's place in the tile
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            id += v[i];
                            const uint32_t x = id - tile_id0;
                            const bool ok = ((T >> i) & 1u) && x < tile_lim;
                            const uint32_t wd = s_bm[ok ? x >> 5 : 0u];
                            const bool h1 = ok && ((wd >> (x & 31u)) & 1u);
                            hit |= h1 ? (1u << i) : 0u;
                            hx = h1 ? x : hx;
                        }
                        // a wavefront step of a whole-structure query meets ~1.6 survivors among its ~450 postings: nearly every step has a hit somewhere, nearly never
                        // two in one lane — one hit per lane is handled from hx; the sixteen-step walk below only when some lane holds two
                        const bool multi = (hit & (hit - 1u)) != 0u;
                        if (!__ballot(multi)) {
                            if (hit) {
                                const uint32_t wd = s_bm[hx >> 5];
                                const uint32_t rk = s_rank[hx >> 5] + (uint32_t)__popc(wd & ((1u << (hx & 31u)) - 1u)) - s_lo;
                                if (BIG) { if (rk < A.cap) atomicOr(&A.g_rowbits[(uint64_t)rk * wpr + ((uint32_t)add >> 5)], 1u << ((uint32_t)add & 31u)); }
                                else if (rk < per_round) atomicOr(&s_rowbits[rk * wpr + ((uint32_t)add >> 5)], 1u << ((uint32_t)add & 31u));
                            }
                        } else if (hit) {
                            id = id_first;
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                id += v[i];
                                if ((hit >> i) & 1u) {
                                    const uint32_t x = id - tile_id0, wd = s_bm[x >> 5];
                                    const uint32_t rk = s_rank[x >> 5] + (uint32_t)__popc(wd & ((1u << (x & 31u)) - 1u)) - s_lo;
                                    if (BIG) { if (rk < A.cap) atomicOr(&A.g_rowbits[(uint64_t)rk * wpr + ((uint32_t)add >> 5)], 1u << ((uint32_t)add & 31u)); }
                                    else if (rk < per_round) atomicOr(&s_rowbits[rk * wpr + ((uint32_t)add >> 5)], 1u << ((uint32_t)add & 31u));
                                }
                            }
                        }
                    }
                    if (more) { if (AHEAD) cur = nxt; else prep(base + FD_WAVE, cur); }
                }
            }
            if (A.dbg && lane == 0) {       // per wavefront: units, steps, time in the loop (LDS; the workgroup's first thread reports)
                const uint32_t dt = (uint32_t)(wall_clock64() - t_loop);
                atomicAdd(&s_dbg[0], my_units); atomicAdd(&s_dbg[1], my_steps); atomicMax(&s_dbg[2], dt); atomicAdd(&s_dbg[3], dt);
                if (dt == 0xffffffffu) s_dbg[4] = 0;
            }
            stamp(3);
            __syncthreads();
            stamp(4);
            if (A.dbg && tid == 0) {
                unsigned long long *d = A.dbg + (RICH ? 24 : 16);
                atomicAdd(&d[0], (unsigned long long)s_dbg[0]); atomicAdd(&d[1], (unsigned long long)s_dbg[1]); atomicAdd(&d[2], (unsigned long long)s_dbg[2]);
                atomicAdd(&d[3], (unsigned long long)s_dbg[3]);
                s_dbg[0] = 0; s_dbg[1] = 0; s_dbg[2] = 0; s_dbg[3] = 0;
            }
        }
        if (RICH && !BIG) {
            // ---- the survivors' records, one survivor per thread: match count = set rows, edge / node counts = row groups with a set
            // row (k_topn_emit_dense walked the rows in (node, partner) order for the same numbers), idf sum over the set rows
            const uint32_t n_here = n_surv - s_lo < per_round ? n_surv - s_lo : per_round;
            if (round == 0) { if (tid == 0) s_base = out_base; __syncthreads(); }
            for (uint32_t k = tid; k < n_here; k += NTHR) {
                const uint32_t rk = s_lo + k;
                uint32_t lo = 0, hi = TILE / 32;         // the bitmap word that holds the survivor of rank rk: the last word with rank <= rk
                while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_rank[mid] <= rk) lo = mid; else hi = mid; }
                uint32_t bits = s_bm[lo];
                for (uint32_t z = rk - s_rank[lo]; z; --z) bits &= bits - 1u;
                const uint32_t i = lo * 32u + (uint32_t)__builtin_ctz(bits);
                const float pen = A.penalty[tile_lo + i];
                const uint32_t *rb = s_rowbits + k * wpr;
                uint32_t cnt = 0, edges = 0, nodes = 0, ce = 0, cn = 0;
                unsigned long long sum = 0;
                for (uint32_t rw = 0; rw < wpr; ++rw) {
                    uint32_t m = rb[rw];
                    const uint32_t left = nrows - rw * 32u, valid = left >= 32u ? 0xffffffffu : (1u << left) - 1u;
                    cnt += (uint32_t)__popc(m);
                    edges += qt_groups_hit(m, s_eend[rw], valid, ce);
                    nodes += qt_groups_hit(m, s_nend[rw], valid, cn);
                    for (; m; m &= m - 1u) sum += s_meta[rw * 32u + (uint32_t)__builtin_ctz(m)] >> 2;
                }
                const uint32_t pos = s_base + rk;
                if (pos < A.cap) {
                    qt_rec rec;
                    rec.nid = tile_lo + i + A.first_id; rec.total_match_count = cnt; rec.node_count = nodes; rec.edge_count = edges;
                    rec.idf = (float)((double)sum * (1.0 / QT_IDF_SCALE)) * pen;      // count_query.rs:200 idf_sum *= nres^(-lp)
                    ((qt_rec *)A.out)[(uint64_t)q * A.cap + pos] = rec;
                }
            }
            __syncthreads();
            stamp(5);
        }
    }
    if (RICH) return;
    if (BIG) {      // the tile's sums of this slice of the rows
        unsigned long long *dst = A.partial + (((uint64_t)slice * A.NT + t) << TL2);
        for (uint32_t k = tid; k < TILE; k += NTHR) dst[k] = s_acc[k];
        return;
    }
    // ---- pass A: ranking keys of the touched structures (listed in the order they were met), first histogram level
    const uint32_t n_t = s_cnt;
    for (uint32_t e0 = 0; e0 < n_t; e0 += 4 * NTHR) {       // four structures per thread in flight
        uint32_t sid[4]; float pen[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t e = e0 + u * NTHR + tid; sid[u] = e < n_t ? A.c_nid[cbase + e] : tile_lo; }
#pragma unroll
        for (int u = 0; u < 4; ++u) pen[u] = A.penalty[sid[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = e0 + u * NTHR + tid;
            if (e < n_t) {
                const unsigned long long a = s_acc[sid[u] - tile_lo];
                const uint32_t key = qt_order_key((float)((double)(a & QT_SUM_MASK) * (1.0 / QT_IDF_SCALE)) * pen[u]);
                const uint32_t bin = qt_bin(key);
                atomicAdd(&s_hist[bin >> 1], 1u << ((bin & 1u) * 16u));
                A.c_key[cbase + e] = key;
            }
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < QT_BINS; k += NTHR) {
        const uint32_t cn = (s_hist[k >> 1] >> ((k & 1u) * 16u)) & 0xffffu;
        if (cn) atomicAdd(&A.ghist[(uint64_t)q * QT_BINS + k], cn);
    }
    if (tid == 0) A.ccount[(uint64_t)q * A.NT + t] = n_t;
    stamp(5);
}

// ------------------------------------------------------------------ pass B from the decoded stream
// The survivors' records without a second decode: pass A left every 16-byte slot of the tile's posting ranges as sixteen 16-bit structure
// ids + the slot's row (34 bytes per slot, A.stream_*).  A workgroup = (query, tile): survivors' bitmap and ranks as in k_qt_score<RICH>, then
// one thread per slot tests its ids against the bitmap — no varint decode, no scan, no plan — and the records follow from the row bits.
// MR: rows per query the workgroup's tables hold (a batch of motif queries has ~44: the 1,024-row tables and row bits for 3,072 survivors per round were
// 32 of the workgroup's 37 KB of LDS — four workgroups per CU for a kernel that is a chain of five dependent loads and seven barriers)
template <int TL2, int NTHR, int RBW, int RBA, int MR = QT_MAX_ROWS>
__global__ __launch_bounds__(NTHR) void k_qt_rows(qt_args A) {
    constexpr uint32_t TILE = 1u << TL2;
    __shared__ uint32_t s_bm[TILE / 32], s_rank[TILE / 32], s_rowbits[RBW];
    __shared__ unsigned long long s_meta[MR];
    __shared__ uint32_t s_eend[MR / 32], s_nend[MR / 32];
    __shared__ uint32_t s_w[NTHR / 64];
    __shared__ uint32_t s_base;
    const uint32_t wg = blockIdx.x;
    const uint32_t t = wg % A.NT, q = wg / A.NT, tid = threadIdx.x;
    const uint64_t r0 = A.q_rows[q];
    const uint32_t nrows = (uint32_t)(A.q_rows[q + 1] - r0);
    const uint32_t tile_lo = t << TL2;
    const uint64_t cbase = ((uint64_t)q * A.NT + t) << TL2;
    // (keys requested before the list's length is known: the 32-bit pass A lists ~top_n keys per tile, the 64-bit one every touched structure)
    constexpr int SPEC0 = MR <= 128 ? 3 : 8, SPEC = (TILE / NTHR) < SPEC0 ? (TILE / NTHR) : SPEC0;
    uint32_t x[SPEC];        // ranking keys; a survivor's structure comes from the other array
#pragma unroll
    for (int u = 0; u < SPEC; ++u) x[u] = A.c_key[cbase + u * NTHR + tid];
    const uint32_t n = A.ccount[(uint64_t)q * A.NT + t];
    const uint32_t thr = A.state[q].thr_key;
    for (uint32_t k = tid; k < TILE / 32; k += NTHR) s_bm[k] = 0u;
    for (uint32_t k = tid; k < MR / 32; k += NTHR) { s_eend[k] = 0u; s_nend[k] = 0u; }
    __syncthreads();
    for (uint32_t k = tid; k < nrows; k += NTHR) {
        const unsigned long long m = A.row_meta[r0 + k];
        s_meta[k] = m;
        if (m & 1ull) atomicOr(&s_eend[k >> 5], 1u << (k & 31u));
        if (m & 2ull) atomicOr(&s_nend[k >> 5], 1u << (k & 31u));
    }
#pragma unroll
    for (int u = 0; u < SPEC; ++u)
        if ((uint32_t)u * NTHR + tid < n && x[u] >= thr) { const uint32_t i = A.c_nid[cbase + u * NTHR + tid] - tile_lo; atomicOr(&s_bm[i >> 5], 1u << (i & 31u)); }
    for (uint32_t e = SPEC * NTHR + tid; e < n; e += NTHR)
        if (A.c_key[cbase + e] >= thr) { const uint32_t i = A.c_nid[cbase + e] - tile_lo; atomicOr(&s_bm[i >> 5], 1u << (i & 31u)); }
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t w0 = 0; w0 < TILE / 32; w0 += NTHR) {
        uint32_t tot;
        const uint32_t pc = w0 + tid < TILE / 32 ? (uint32_t)__popc(s_bm[w0 + tid]) : 0u;
        const uint32_t ex = qt_block_excl<NTHR>(pc, tid, s_w, &tot);
        if (w0 + tid < TILE / 32) s_rank[w0 + tid] = run + ex;
        run += tot;
    }
    const uint32_t n_surv = run;
    // (a tile k_qt_layout / k_qt_bases could not place — window table or decoded stream too small — was not scored at all by k_qt_score32: it has no
    // survivors to trip the check in the loop below, so the query is flagged here)
    if (A.heads && A.stream_tab[((uint64_t)q * A.NT + t) * QT_MAXB].y == 0xffffffffu) { if (tid == 0) atomicOr(&A.state[q].count, 0x80000000u); return; }
    if (!n_surv) return;
    const uint32_t wpr = (nrows + 31u) >> 5, per_round = (uint32_t)RBW / wpr, n_rounds = (n_surv + per_round - 1u) / per_round;
    if (tid == 0) s_base = atomicAdd(&A.state[q].count, n_surv);
    constexpr uint32_t CPT = 1u << (TL2 - QT_CELL_LOG2);
    // (k_qt_score32 leaves one run of records per (query, tile): first record + slot)
    const uint32_t cell0 = t * CPT, ncell = A.NC - cell0 < CPT ? A.NC - cell0 : CPT, n_batches = A.heads ? 1u : (nrows * ncell + RBA - 1u) / RBA;
    const qt_u32x4 *ids = reinterpret_cast<const qt_u32x4 *>(A.stream_ids);
    for (uint32_t round = 0; round < n_rounds; ++round) {
        const uint32_t s_lo = round * per_round;
        const uint32_t n_here = n_surv - s_lo < per_round ? n_surv - s_lo : per_round;
        for (uint32_t k = tid; k < n_here * wpr; k += NTHR) s_rowbits[k] = 0u;
        __syncthreads();
        for (uint32_t b = 0; b < n_batches; ++b) {
            const uint2 rec = A.stream_tab[((uint64_t)q * A.NT + t) * QT_MAXB + b];
            if (rec.y == 0xffffffffu) {       // the stream did not hold this batch (its bound is the host's): the call reports an overflowing selection
                if (tid == 0) atomicOr(&A.state[q].count, 0x80000000u);
                return;
            }
            for (uint32_t sl = tid; sl < rec.y; sl += NTHR) {
                const qt_u32x4 a0 = ids[2ull * (rec.x + sl)], a1 = ids[2ull * (rec.x + sl) + 1];
                const uint32_t row = A.stream_row[(uint64_t)rec.x + sl];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t w2 = i < 8 ? a0[i >> 1] : a1[(i - 8) >> 1];
                    const uint32_t xx = (i & 1) ? w2 >> 16 : w2 & 0xffffu;
                    if (xx != 0xffffu) {
                        const uint32_t wd = s_bm[xx >> 5];
                        if ((wd >> (xx & 31u)) & 1u) {
                            const uint32_t rk = s_rank[xx >> 5] + (uint32_t)__popc(wd & ((1u << (xx & 31u)) - 1u)) - s_lo;
                            if (rk < per_round) atomicOr(&s_rowbits[rk * wpr + (row >> 5)], 1u << (row & 31u));
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t k = tid; k < n_here; k += NTHR) {
            const uint32_t rk = s_lo + k;
            uint32_t lo = 0, hi = TILE / 32;
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_rank[mid] <= rk) lo = mid; else hi = mid; }
            uint32_t bits = s_bm[lo];
            for (uint32_t z = rk - s_rank[lo]; z; --z) bits &= bits - 1u;
            const uint32_t i = lo * 32u + (uint32_t)__builtin_ctz(bits);
            const float pen = A.penalty[tile_lo + i];
            const uint32_t *rb = s_rowbits + k * wpr;
            uint32_t cnt = 0, edges = 0, nodes = 0, ce = 0, cn = 0;
            unsigned long long sum = 0;
            for (uint32_t rw = 0; rw < wpr; ++rw) {
                uint32_t m = rb[rw];
                const uint32_t left = nrows - rw * 32u, valid = left >= 32u ? 0xffffffffu : (1u << left) - 1u;
                cnt += (uint32_t)__popc(m);
                edges += qt_groups_hit(m, s_eend[rw], valid, ce);
                nodes += qt_groups_hit(m, s_nend[rw], valid, cn);
                for (; m; m &= m - 1u) sum += s_meta[rw * 32u + (uint32_t)__builtin_ctz(m)] >> 2;
            }
            const uint32_t pos = s_base + rk;
            if (pos < A.cap) {
                qt_rec r;
                r.nid = tile_lo + i + A.first_id; r.total_match_count = cnt; r.node_count = nodes; r.edge_count = edges;
                r.idf = (float)((double)sum * (1.0 / QT_IDF_SCALE)) * pen;      // count_query.rs:200 idf_sum *= nres^(-lp)
                ((qt_rec *)A.out)[(uint64_t)q * A.cap + pos] = r;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ one query of ~10^5 rows: reduce, survivors, records
// k_qd_reduce: the structures' sums over the row slices -> ranking keys of the touched structures + first histogram level (what pass A's own
// finalize does for a motif query).  One workgroup per 1,024 structures; a tile's list is filled through its counter (zero on entry).
template <int TL2>
__global__ __launch_bounds__(1024) void k_qd_reduce(qt_args A) {
    __shared__ uint32_t s_hist[QT_BINS];
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t i = blockIdx.x * 1024u + tid, t = i >> TL2;        // structure (relative id), its tile
    for (uint32_t k = tid; k < QT_BINS; k += 1024) s_hist[k] = 0u;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    unsigned long long a = 0;
    if (i < A.S) for (uint32_t sl = 0; sl < A.n_slices; ++sl) a += A.partial[(((uint64_t)sl * A.NT) << TL2) + i];
    const bool touched = (a >> QT_CNT_SHIFT) != 0ull;
    uint32_t key = 0, pos = 0;
    if (touched) {
        key = qt_order_key((float)((double)(a & QT_SUM_MASK) * (1.0 / QT_IDF_SCALE)) * A.penalty[i]);
        atomicAdd(&s_hist[qt_bin(key)], 1u);
    }
    const uint64_t m = __ballot(touched);
    if (m) {
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
        pos = (uint32_t)__shfl((int)base, __builtin_ctzll(m), FD_WAVE) + fd_mbcnt(m);
    }
    __syncthreads();
    if (tid == 0 && s_cnt) s_base = atomicAdd(&A.ccount[t], s_cnt);      // 1,024 divides the tile: the workgroup's structures share one tile
    __syncthreads();
    if (touched) { A.c_nid[((uint64_t)t << TL2) + s_base + pos] = i; A.c_key[((uint64_t)t << TL2) + s_base + pos] = key; }
    for (uint32_t k = tid; k < QT_BINS; k += 1024) if (s_hist[k]) atomicAdd(&A.ghist[k], s_hist[k]);
}
// k_qd_surv: the survivors of a tile (key >= threshold) as a bitmap + the number of survivors in the words before (local ranks) + the count
template <int TL2>
__global__ __launch_bounds__(512) void k_qd_surv(qt_args A) {
    constexpr uint32_t TILE = 1u << TL2, W = TILE / 32;
    __shared__ uint32_t s_bm[W];
    __shared__ uint32_t s_w[8];
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    for (uint32_t k = tid; k < W; k += 512) s_bm[k] = 0u;
    __syncthreads();
    const uint32_t n = A.ccount[t], thr = A.state[0].thr_key, tile_lo = t << TL2;
    const uint64_t cb = (uint64_t)t << TL2;
    for (uint32_t i = tid; i < n; i += 512) if (A.c_key[cb + i] >= thr) { const uint32_t z = A.c_nid[cb + i] - tile_lo; atomicOr(&s_bm[z >> 5], 1u << (z & 31u)); }
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t w0 = 0; w0 < W; w0 += 512) {
        uint32_t tot;
        const uint32_t pc = (uint32_t)__popc(s_bm[w0 + tid]);
        const uint32_t ex = qt_block_excl<512>(pc, tid, s_w, &tot);
        A.g_bm[(uint64_t)t * W + w0 + tid] = s_bm[w0 + tid];
        A.g_rank[(uint64_t)t * W + w0 + tid] = run + ex;
        run += tot;
    }
    if (tid == 0) A.g_tcount[t] = run;
}
// ... global slots: a tile's ranks start behind the survivors of the tiles before it; the survivors' structure ids by slot; the total
template <int TL2>
__global__ __launch_bounds__(512) void k_qd_surv_slots(qt_args A) {
    constexpr uint32_t W = (1u << TL2) / 32;
    __shared__ uint32_t s_base, s_tot;
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        uint32_t b = 0, tot = 0;
        for (uint32_t k = 0; k < A.NT; ++k) { const uint32_t c = A.g_tcount[k]; if (k < t) b += c; tot += c; }
        s_base = b; s_tot = tot;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t w = tid; w < W; w += 512) {
        uint32_t rk = A.g_rank[(uint64_t)t * W + w] + base;
        A.g_rank[(uint64_t)t * W + w] = rk;
        for (uint32_t bits = A.g_bm[(uint64_t)t * W + w]; bits; bits &= bits - 1u, ++rk)
            if (rk < A.cap) A.g_nid[rk] = (t << TL2) + w * 32u + (uint32_t)__builtin_ctz(bits);
    }
    if (t == 0 && tid == 0) A.state[0].count = s_tot;
}
// k_qd_records: a survivor's record from its row bits, one wavefront per survivor: match count = set rows, edge / node counts = row groups
// with a set row (qt_groups_hit word by word; a lane walks its share of the ~3,200 words once with and once without an incoming carry,
// lane 0 chains the 64 shares), idf sum over the set rows
__global__ __launch_bounds__(FD_WAVE) void k_qd_records(qt_args A) {
    __shared__ uint32_t s_h[4][FD_WAVE];      // hits of a lane's share: edges / nodes x carry-in 0 / 1
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = A.state[0].count < A.cap ? A.state[0].count : A.cap;
    if (slot >= n) return;
    const uint32_t *rb = A.g_rowbits + (uint64_t)slot * A.g_wpr;
    const uint32_t per = (A.g_wpr + FD_WAVE - 1) / FD_WAVE, w0 = lane * per, w1 = w0 + per < A.g_wpr ? w0 + per : A.g_wpr;
    uint32_t cnt = 0, he[2] = {0, 0}, hn[2] = {0, 0}, oe[2] = {0, 1}, on[2] = {0, 1};
    unsigned long long sum = 0;
    for (int cin = 0; cin < 2; ++cin) {
        uint32_t ce = (uint32_t)cin, cn = (uint32_t)cin;
        for (uint32_t rw = w0; rw < w1; ++rw) {
            uint32_t m = rb[rw];
            const uint32_t left = A.nq - rw * 32u, valid = left >= 32u ? 0xffffffffu : (1u << left) - 1u;
            he[cin] += qt_groups_hit(m, A.g_eend[rw], valid, ce);
            hn[cin] += qt_groups_hit(m, A.g_nend[rw], valid, cn);
            if (cin == 0) {
                cnt += (uint32_t)__popc(m);
                for (; m; m &= m - 1u) sum += A.row_meta[rw * 32u + (uint32_t)__builtin_ctz(m)] >> 2;
            }
        }
        oe[cin] = ce; on[cin] = cn;       // an empty share hands the carry through
    }
    s_h[0][lane] = he[0]; s_h[1][lane] = he[1]; s_h[2][lane] = hn[0]; s_h[3][lane] = hn[1];
    const uint64_t oe0 = __ballot(oe[0] != 0u), oe1 = __ballot(oe[1] != 0u), on0 = __ballot(on[0] != 0u), on1 = __ballot(on[1] != 0u);
    for (int off = 32; off > 0; off >>= 1) {
        cnt += (uint32_t)__shfl_down((int)cnt, off, FD_WAVE);
        const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)sum, off, FD_WAVE), hi = (uint32_t)__shfl_down((int)(uint32_t)(sum >> 32), off, FD_WAVE);
        sum += ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t edges = 0, nodes = 0, ce = 0, cn = 0;
        for (uint32_t l = 0; l < FD_WAVE; ++l) {
            edges += s_h[ce][l]; ce = (uint32_t)(((ce ? oe1 : oe0) >> l) & 1ull);
            nodes += s_h[2 + cn][l]; cn = (uint32_t)(((cn ? on1 : on0) >> l) & 1ull);
        }
        const uint32_t i = A.g_nid[slot];
        qt_rec rec;
        rec.nid = i + A.first_id; rec.total_match_count = cnt; rec.node_count = nodes; rec.edge_count = edges;
        rec.idf = (float)((double)sum * (1.0 / QT_IDF_SCALE)) * A.penalty[i];
        ((qt_rec *)A.out)[slot] = rec;
    }
}

// ------------------------------------------------------------------ threshold
// The key threshold of a query's top N from its first-level histogram: the highest bin b with (keys above b) + hist[b] >= top_n.  When
// that many survivors fit the selection's slots the bin's lower edge is the threshold; else (rare: the cut falls into a crowd of nearly
// equal keys) the workgroup splits the bin once more over the query's (structure, key) lists.  One workgroup per query; the table is left zero.
__global__ __launch_bounds__(1024) void k_qt_thr(qt_args A, uint32_t top_n) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[QT_BINS];
    __shared__ uint32_t s_bin, s_above, s_at;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t *h = A.ghist + (uint64_t)q * QT_BINS;
    for (uint32_t k = tid; k < QT_BINS; k += 1024) { hist[k] = h[k]; h[k] = 0u; }
    __syncthreads();
    // -> s_bin = the highest bin b with above0 + (keys in the bins above b) + hist[b] >= top_n (bin 0 when there are fewer keys), s_above = the keys above it,
    // s_at = hist[b].  One wavefront: lane l sums bins 32 l .. 32 l + 31, a wave scan finds the lane where the count from the top crosses top_n, its 32 bins are
    // split over the lanes once more (thread 0 walking 256 partial sums and then 8 bins through dependent LDS reads was 3-4 of the kernel's 9-12 us)
    auto search = [&](uint32_t above0) {
        if (tid < FD_WAVE) {
            const uint32_t lane = tid;
            uint32_t tot = 0;
#pragma unroll
            for (uint32_t u = 0; u < 32; u += 4) {
                const qt_u32x4 hw = *reinterpret_cast<const qt_u32x4 *>(&hist[32u * lane + u]);
                tot += hw[0] + hw[1] + hw[2] + hw[3];
            }
            const uint32_t incl = qt_wave_incl(tot, lane), all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t above_l = above0 + (all - incl);                   // keys in the lanes above this one
            const uint64_t mk = __ballot(above_l + tot >= top_n);
            if (!mk) { if (lane == 0) { s_bin = 0u; s_above = above0 + all - hist[0]; s_at = hist[0]; } }
            else {
                const uint32_t L = 63u - (uint32_t)__clzll((long long)mk);
                const uint32_t accL = (uint32_t)__builtin_amdgcn_readlane((int)above_l, (int)L);
                const uint32_t h = lane < 32u ? hist[32u * L + lane] : 0u;
                const uint32_t i2 = qt_wave_incl(h, lane), all2 = (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
                const uint32_t above_j = accL + (all2 - i2);
                const uint64_t mk2 = __ballot(lane < 32u && above_j + h >= top_n);      // never empty: lane 0 sees the whole of lane L's count
                const uint32_t j = 63u - (uint32_t)__clzll((long long)mk2);
                if (lane == j) { s_bin = 32u * L + j; s_above = above_j; s_at = h; }
            }
        }
        __syncthreads();
    };
    search(0u);
    const uint32_t b1 = s_bin, above = s_above, edge = qt_edge(b1), sh2 = qt_shift2(b1);
    uint32_t thr_key = edge;
    if (above + s_at > A.cap) {       // block-uniform
        __syncthreads();
        for (uint32_t k = tid; k < QT_BINS; k += 1024) hist[k] = 0u;
        __syncthreads();
        for (uint32_t t = 0; t < A.NT; ++t) {
            const uint32_t n = A.ccount[(uint64_t)q * A.NT + t];
            const uint32_t *e = A.c_key + (((uint64_t)q * A.NT + t) << A.tile_log2);
            for (uint32_t i = tid; i < n; i += 1024) {
                const uint32_t key = e[i];
                if (qt_bin(key) == b1) { const uint32_t s = (key - edge) >> sh2; atomicAdd(&hist[s < QT_BINS - 1u ? s : QT_BINS - 1u], 1u); }
            }
        }
        __syncthreads();
        search(above);
        thr_key = edge + (s_bin << sh2);
    }
    if (tid == 0) { qt_state s; s.thr_bin = b1; s.above = above; s.thr_key = thr_key; s.count = 0; A.state[q] = s; }
}

// ------------------------------------------------------------------ ranking of the survivors
// idf descending, ties by ascending structure id (query_pdb.rs:404-411), cut to top_n.  The survivors' keys are spread over the selection's own
// 2,048 bins (1.5 % wide): a survivor's rank = the survivors in higher bins + its rank among the few of its own bin.  Counting sort by bin (LDS
// atomics, one block scan), then every survivor counts the members of its bin that precede it — five barriers and a short loop, where the bitonic
// network of rounds 4-5 needed 66 compare-exchange steps and 14-28 barriers (33 us per launch at any batch size: a third of `cq_topn`).  Same total
// order (~key << 32 | nid ascending), so the same records in the same places.  A query whose selection overflowed its slots (count > cap) is left to the host.
#define QT_SORT_T 1024
__global__ __launch_bounds__(QT_SORT_T) void k_qt_sort(const qt_rec *__restrict__ sel, uint32_t cap, const qt_state *__restrict__ st, uint32_t top_n,
                                                        qt_rec *__restrict__ out) {
    __shared__ uint32_t s_cur[QT_BINS], s_start[QT_BINS];
    __shared__ unsigned long long s_key[4096];
    __shared__ uint16_t s_src[4096];
    __shared__ uint32_t s_w[QT_SORT_T / 64];
    const uint32_t q = blockIdx.x, cnt = st[q].count, tid = threadIdx.x;
    if (cnt > cap || cnt == 0 || cnt > 4096u) return;
    const qt_rec *r = sel + (uint64_t)q * cap;
    for (uint32_t k = tid; k < QT_BINS; k += QT_SORT_T) s_cur[k] = 0u;
    __syncthreads();
    unsigned long long key[4];
    uint32_t bin[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t i = e * 1024u + tid;
        key[e] = ~0ull; bin[e] = 0u;
        if (i < cnt) {
            const uint32_t ok = qt_order_key(r[i].idf);
            key[e] = ((unsigned long long)(~ok) << 32) | r[i].nid;
            bin[e] = qt_bin(ok);
            atomicAdd(&s_cur[bin[e]], 1u);
        }
    }
    __syncthreads();
    // first place of every bin's members, highest bin first: cnt - (members of the bins below) - (its own)
    {
        const uint32_t c0 = s_cur[2u * tid], c1 = s_cur[2u * tid + 1u];
        uint32_t tot;
        const uint32_t below = qt_block_excl<QT_SORT_T>(c0 + c1, tid, s_w, &tot);
        const uint32_t st0 = cnt - below - c0, st1 = cnt - below - c0 - c1;
        s_start[2u * tid] = st0; s_start[2u * tid + 1u] = st1;
        s_cur[2u * tid] = st0; s_cur[2u * tid + 1u] = st1;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t i = e * 1024u + tid;
        if (i < cnt) { const uint32_t p = atomicAdd(&s_cur[bin[e]], 1u); s_key[p] = key[e]; s_src[p] = (uint16_t)i; }
    }
    __syncthreads();
    const uint32_t m = cnt < top_n ? cnt : top_n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t p = e * 1024u + tid;
        if (p >= cnt) break;
        const unsigned long long k = s_key[p];
        const uint32_t b = qt_bin(~(uint32_t)(k >> 32)), a0 = s_start[b], a1 = s_cur[b];
        uint32_t rank = a0;
        for (uint32_t x = a0; x < a1; ++x) rank += s_key[x] < k ? 1u : 0u;
        if (rank < m) out[(uint64_t)q * top_n + rank] = r[s_src[p]];
    }
}

void fd_launch_qt_plan(const qt_args &A, hipStream_t st) {
    const uint32_t cpg_log2 = A.plan_log2 - QT_CELL_LOG2;
    const uint64_t n_gran = (A.NC + (1u << cpg_log2) - 1u) >> cpg_log2;
    const uint64_t n = (((uint64_t)A.nq + 7u) >> 3) * ((n_gran + 7u) >> 3) * 64u;      // wavefronts of 8 rows x 8 granules
    if (A.nq && n_gran) hipLaunchKernelGGL(k_qt_plan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A);
}
// one query of ~10^5 rows (a whole structure as the query): scores per (tile, slice of the rows) -> per-tile reduction + keys + histogram;
// then threshold -> survivors -> their row bits (second decode) -> records -> ranking.  A.tile_log2 = 14, A.plan_log2 = 14.
void fd_launch_qt_big_score(const qt_args &A, hipStream_t st) {
    if (!A.S) return;
    hipLaunchKernelGGL((k_qt_score<false, 14, 1024, 512, 1, true>), dim3(A.NT * A.n_slices), dim3(1024), 0, st, A);
    (void)hipMemsetAsync(A.ccount, 0, (size_t)A.NT * 4, st);
    hipLaunchKernelGGL(k_qd_reduce<14>, dim3((A.S + 1023u) / 1024u), dim3(1024), 0, st, A);
}
void fd_launch_qt_big_select(const qt_args &A, uint32_t top_n, void *sorted, hipStream_t st) {
    if (!A.S) return;
    hipLaunchKernelGGL(k_qt_thr, dim3(1), dim3(1024), 0, st, A, top_n);
    hipLaunchKernelGGL(k_qd_surv<14>, dim3(A.NT), dim3(512), 0, st, A);
    hipLaunchKernelGGL(k_qd_surv_slots<14>, dim3(A.NT), dim3(512), 0, st, A);
    (void)hipMemsetAsync(A.g_rowbits, 0, (size_t)A.cap * A.g_wpr * 4, st);
    hipLaunchKernelGGL((k_qt_score<true, 14, 1024, 512, 1, true>), dim3(A.NT * A.n_slices), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_qd_records, dim3(A.cap), dim3(FD_WAVE), 0, st, A);
    hipLaunchKernelGGL(k_qt_sort, dim3(1), dim3(QT_SORT_T), 0, st, (const qt_rec *)A.out, A.cap, A.state, top_n, (qt_rec *)sorted);
}
// pass A: scores of every (query, tile) in LDS -> (structure, key) of the touched structures + first histogram level
void fd_launch_qt_score(const qt_args &A, hipStream_t st) {
    if (!A.n_queries || !A.S) return;
    const dim3 g(A.NT * A.n_queries);
    if (A.tile_log2 == 14) hipLaunchKernelGGL((k_qt_score<false, 14, 1024, 512, 1>), g, dim3(1024), 0, st, A);
    else hipLaunchKernelGGL((k_qt_score<false, 13, 512, 256, 1>), g, dim3(512), 0, st, A);
}
// threshold -> pass B: survivors' records in A.out[query][cap] (any order; A.state[query].count of them, > cap: overflow) -> ranked
// top_n records per query in sorted[query][top_n]
void fd_launch_qt_select(const qt_args &A, uint32_t top_n, void *sorted, hipStream_t st) {
    if (!A.n_queries || !A.S) return;
    const dim3 g(A.NT * A.n_queries);
    hipLaunchKernelGGL(k_qt_thr, dim3(A.n_queries), dim3(1024), 0, st, A, top_n);
    if (A.stream_ids && A.tile_log2 == 15) hipLaunchKernelGGL((k_qt_rows<15, 512, 6144, 512>), g, dim3(512), 0, st, A);
    else if (A.stream_ids && A.tile_log2 == 14 && A.max_rows && A.max_rows <= 128u) {
        hipLaunchKernelGGL((k_qt_rows<14, 512, 1024, 512, 128>), g, dim3(512), 0, st, A);       // (256 / 128 threads per workgroup: 75 / 97 us against 75)
    }
    else if (A.stream_ids && A.tile_log2 == 14) hipLaunchKernelGGL((k_qt_rows<14, 512, 6144, 512>), g, dim3(512), 0, st, A);
    else if (A.stream_ids && A.heads && A.max_rows && A.max_rows <= 128u) hipLaunchKernelGGL((k_qt_rows<13, 256, 1024, 256, 128>), g, dim3(256), 0, st, A);
    else if (A.stream_ids) hipLaunchKernelGGL((k_qt_rows<13, 512, 6144, 256>), g, dim3(512), 0, st, A);
    else if (A.tile_log2 == 14) hipLaunchKernelGGL((k_qt_score<true, 14, 1024, 512, 6144>), g, dim3(1024), 0, st, A);
    else hipLaunchKernelGGL((k_qt_score<true, 13, 512, 256, 6144>), g, dim3(512), 0, st, A);
    hipLaunchKernelGGL(k_qt_sort, dim3(A.n_queries), dim3(QT_SORT_T), 0, st, (const qt_rec *)A.out, A.cap, A.state, top_n, (qt_rec *)sorted);
}
