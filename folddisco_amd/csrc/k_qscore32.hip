// k_qscore32.hip — pass A of the tiled motif scoring with 32-bit accumulators and a planned slot stream.
//
// Same job as k_qt_score<false> of k_qtile.hip (idf sums of count_query, src/controller/count_query.rs:82-220, for every structure of a tile
// in LDS; ranking keys and their histogram for the candidate selection of src/cli/workflows/query_pdb.rs:404-411), for the batches it serves:
// every row's idf is at least one unit of 2^-22 and a query's idf units sum below 2^32 — then a structure was touched exactly when its u32 sum is
// non-zero, the sum is the exact sum, and the key (float(sum) * 2^-22 * penalty) has the bits the 64-bit kernel computes.  Other batches keep
// the 64-bit kernel.  What is different:
//   * k_qt_layout (one wavefront per (query, tile)) does what every scoring workgroup did for itself between two barriers: the tile's non-empty
//     (row, cell) pieces become a stream of 16-byte slots cut into WINDOWS of 64 — a piece of at most 64 slots never straddles a window (it
//     starts the next one instead), a longer piece starts one — with a table of the windows' first pieces;
//   * k_qt_score32: a wavefront takes whole windows (window w of wavefront w mod W: no claims, no units, no barrier between the set-up and the
//     end of the decode), fetches the <= 64 piece descriptors of its window into registers (fields by ds_bpermute), decodes as before and adds
//     with NON-returning ds_add_u32; 16,384 accumulators are 64 KB: two workgroups share a CU and cover each other's latencies;
//   * no first-touch list: the finalize sweeps the accumulators (keys back into the same LDS words, 2,048-bin histogram), finds the tile's OWN
//     cut — the bin of its top_n-th key; a structure of the global top N is in its tile's top N — and lists only the keys from that bin up
//     (~top_n of the tile's ~4,000 touched structures), and only those bins go to the query's histogram (k_qt_thr never looks below the global
//     cut's bin, which is at or above every tile's);
//   * the decoded stream k_qt_rows reads is indexed by slot: record = first record of (query, tile) + slot, padding slots read as empty.
#include "fdgpu_internal.h"
#include "k_qtile.h"

// ------------------------------------------------------------------ layout of a (query, tile)
// One wavefront per (query, tile).  A round = 64 >> cpt_log2 rows x the tile's cells (lane = row + rows x cell: the lanes of a row share its list's
// offsets and checkpoint lines); the byte ranges of QL_RG rounds are worked out TOGETHER, stage by stage (list position -> offsets + checkpoint
// metadata -> the two checkpoint entries): three dependent round trips per QL_RG rounds instead of per round (a wavefront's rounds one after the
// other: 69 us per 128 queries, as long as a third of the scoring).  Loads of lanes without a list go to a harmless address instead of being
// branched around, so that the compiler can issue a stage's loads back to back.  Same ranges as qt_piece_range(row, cell) — the 64-bit path's.
#define QL_RG 6
__global__ __launch_bounds__(FD_WAVE) void k_qt_layout(qt_args A) {
    const uint32_t wg = blockIdx.x, t = wg % A.NT, q = wg / A.NT, lane = threadIdx.x;
    const uint64_t r0 = A.q_rows[q];
    const uint32_t nrows = (uint32_t)(A.q_rows[q + 1] - r0);
    const uint32_t cpt_log2 = A.tile_log2 - QT_CELL_LOG2, CPT = 1u << cpt_log2;
    const uint32_t cell0 = t << cpt_log2, ncell = A.NC - cell0 < CPT ? A.NC - cell0 : CPT;
    const uint64_t rbase = r0 * A.NT + (uint64_t)nrows * t;
    const uint64_t pbase = rbase << cpt_log2, wbase = rbase * A.win_per_row + 2ull * ((uint64_t)q * A.NT + t);
    const uint32_t wcap = nrows * A.win_per_row + 2u;
    const uint32_t rpr_log2 = 6u - cpt_log2, rpr = 1u << rpr_log2, rsub = lane & (rpr - 1u), cell = lane >> rpr_log2;
    const uint32_t c0 = cell0 + cell;                          // the lane's checkpoint cell
    const uint64_t *const dummy = A.q_rows;                   // two readable 64-bit words for the loads of lanes without a list
    uint32_t np = 0, S_tot = 0, E_tot = 0, base = 0, n_open = 0;      // pieces, weight and extra windows so far, the open window's first weight, windows opened
    bool bad = false, open = false;
    for (uint32_t row00 = 0; row00 < nrows; row00 += QL_RG * rpr) {
        long long k[QL_RG];
        uint64_t b0[QL_RG], b1[QL_RG];
        unsigned long long m[QL_RG];
        uint4 rg[QL_RG];
#pragma unroll
        for (int r = 0; r < QL_RG; ++r) {
            const uint32_t row = row00 + (uint32_t)r * rpr + rsub;
            k[r] = (row < nrows && cell < ncell) ? A.kidx[r0 + row] : -1ll;
        }
#pragma unroll
        for (int r = 0; r < QL_RG; ++r) {
            const uint64_t *po = k[r] >= 0 ? A.offsets + k[r] : dummy;
            const unsigned long long *pm = k[r] >= 0 ? A.ck_meta + k[r] : (const unsigned long long *)dummy;
            b0[r] = po[0]; b1[r] = po[1]; m[r] = *pm;
        }
        uint2 x0[QL_RG]; uint32_t x1[QL_RG];
        bool take[QL_RG], has0[QL_RG], has1[QL_RG];
#pragma unroll
        for (int r = 0; r < QL_RG; ++r) {
            const uint32_t j = (uint32_t)(m[r] >> 56);
            const uint2 *e = A.ck_ent + (m[r] & ((1ull << 56) - 1ull));
            const uint32_t n_e = (uint32_t)(((uint64_t)A.NC + (1ull << j) - 1ull) >> j);
            const uint32_t e0 = c0 >> j, e1 = e0 + 1u;
            const uint32_t first_c = (e0 << j) > cell0 ? (e0 << j) : cell0;       // the entry's first cell inside this tile
            take[r] = k[r] >= 0 && (j == 0u || c0 == first_c);
            has0[r] = take[r] && e0 && n_e > 1u;
            has1[r] = take[r] && !(e1 >= n_e || n_e <= 1u);
            const uint2 *p0 = has0[r] ? e + (e0 - 1u) : (const uint2 *)dummy;
            const uint2 *p1 = has1[r] ? e + (e1 - 1u) : (const uint2 *)dummy;
            x0[r] = *p0; x1[r] = p1->x;
        }
#pragma unroll
        for (int r = 0; r < QL_RG; ++r) {
            const uint32_t sb = has0[r] ? x0[r].x : 0u, prev = has0[r] ? x0[r].y : 0u;
            const uint64_t eb = has1[r] ? (uint64_t)x1[r] : b1[r] - b0[r];
            const uint64_t p = b0[r] + sb;
            rg[r] = take[r] ? make_uint4((uint32_t)p, (uint32_t)(p >> 32), (uint32_t)(eb - sb), prev) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < QL_RG; ++r) {
            // ---- next-fit of the round's pieces into windows of 64 slots, window by window (a piece per scalar step cost ~20 scalar
            // instructions x 260 pieces: the CU's one scalar unit, shared by its ~17 wavefronts, was the kernel's bound).  A piece weighs its
            // slots, one of more than 64 weighs a whole window (and takes whole windows); S = running weight; a window opened at weight `base`
            // holds the pieces that end at or below base + 64: one ballot finds the first that does not — the next window's first piece.
            const uint32_t row = row00 + (uint32_t)r * rpr + rsub;
            const uint32_t ns = (rg[r].z + 15u) >> 4, wt = ns < 64u ? ns : 64u, ex = ns > 64u ? (ns - 1u) >> 6 : 0u;
            const uint32_t s_in = qt_wave_incl(wt, lane), e_in = qt_wave_incl(ex, lane);
            const uint32_t S_ex = S_tot + s_in - wt, S_in = S_tot + s_in, E_ex = E_tot + e_in - ex;
            const uint64_t mm0 = __ballot(ns != 0u);
            uint64_t starts = 0, todo = mm0;         // pieces of this round not yet known to fit the open window
            const uint32_t k_before = n_open;
            const uint32_t base_in = base;
            for (;;) {
                const uint64_t nf = __ballot(ns != 0u && (!open || S_in - base > 64u)) & todo;
                if (!nf) break;
                const int l = __builtin_ctzll(nf);
                starts |= 1ull << l;
                base = (uint32_t)__builtin_amdgcn_readlane((int)S_ex, l);
                open = true; ++n_open;
                todo = nf & (nf - 1ull);             // the pieces behind it that did not fit the window before: against the new one
            }
            S_tot = (uint32_t)__builtin_amdgcn_readlane((int)S_in, 63);
            E_tot = (uint32_t)__builtin_amdgcn_readlane((int)(E_ex + ex), 63);
            // a piece's window: the last one opened at or below its lane (its first weight from that lane), or the one open since an earlier round
            const uint64_t m_le = starts & ((2ull << lane) - 1ull);
            const uint32_t sl = m_le ? 63u - (uint32_t)__clzll((long long)m_le) : 0u;
            const uint32_t b_s = (uint32_t)__shfl((int)S_ex, (int)sl, FD_WAVE);
            if (ns) {
                const bool mine = (starts >> lane) & 1ull;                      // this piece opens a window
                const uint32_t wbase_p = m_le ? b_s : base_in;
                const uint32_t win0 = k_before + (uint32_t)__popcll(m_le) - 1u + E_ex;      // the piece's (first) window
                const uint32_t P = (win0 << 6) + (S_ex - wbase_p);
                const uint32_t pidx = np + fd_mbcnt(mm0);
                A.pieces[pbase + pidx] = make_uint4(rg[r].x, (rg[r].y & 0xffffu) | (row << 16), rg[r].z, rg[r].w);
                A.piece_p[pbase + pidx] = P;
                if (mine) {
                    for (uint32_t w = win0; w <= win0 + ex; ++w) {
                        if (w < wcap) A.win[wbase + w] = pidx | (w > win0 ? QT_WIN_CONT : 0u);
                        else bad = true;
                    }
                }
            }
            np += (uint32_t)__popcll(mm0);
        }
    }
    const uint32_t nwin = n_open + E_tot;
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) A.heads[(uint64_t)q * A.NT + t] = make_uint4(np, nwin, 0u, any_bad ? 1u : 0u);
}
// where a (query, tile)'s records of the decoded stream begin: an exclusive scan of the tiles' windows (one workgroup; a claim per tile from one
// counter — 4,352 returning atomics on one address — was 50 of k_qt_layout's 67 us per 128 queries)
__global__ __launch_bounds__(1024) void k_qt_bases(qt_args A) {
    __shared__ uint32_t s_w[16];
    // passes of 8 x 1,024 entries (entry = pass + 1,024 u + thread: neighbouring threads read neighbouring entries), the pass's loads issued together
    const uint32_t n = A.n_queries * A.NT, tid = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 8192u) {
        uint4 hd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + 1024u * (uint32_t)u + tid; hd[u] = i < n ? A.heads[i] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 1024u * (uint32_t)u >= n) break;
            const uint32_t i = i0 + 1024u * (uint32_t)u + tid, v = hd[u].y << 6;
            uint32_t tot;
            const uint32_t sb = run + qt_block_excl<1024>(v, tid, s_w, &tot);
            run += tot;
            if (i < n) {
                const bool fits = (uint64_t)sb + v <= A.stream_cap;
                A.heads[i] = make_uint4(hd[u].x, hd[u].y, sb, hd[u].w | (fits ? 0u : 2u));
                A.stream_tab[(uint64_t)i * QT_MAXB] = make_uint2(sb, fits && !hd[u].w ? v : 0xffffffffu);
            }
        }
    }
}

// ------------------------------------------------------------------ scores of a (query, tile)
// workgroup barrier that waits for the wavefront's LDS traffic only: __syncthreads() is a release / acquire fence over global memory as well and on
// gfx9 one counter (vmcnt) covers loads AND stores — every barrier would wait for the decoded stream's stores and the prefetched penalties
__device__ __forceinline__ void q32_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct q32_win { uint32_t w, c0, nxt; bool cont, valid; };                      // wave-uniform: a window, its first piece, the table entry of the window behind it
struct q32_pieces { uint4 d; uint32_t Pp; };                                   // lane l: piece c0 + l of the window
struct q32_slot { qt_u32x4 x4; uint32_t add, nby, rel, pstart, prev, row; };   // the lane's slot: its 16 bytes (in flight), its piece

template <int TL2, int NTHR>
__global__ __launch_bounds__(NTHR) void k_qt_score32(qt_args A) {
    constexpr uint32_t TILE = 1u << TL2, NW = NTHR / 64;
    __shared__ __attribute__((aligned(16))) uint32_t s_acc[TILE];               // idf sum (2^-22 units) per structure of the tile
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[QT_BINS / 2];       // 16-bit counts, two bins per word
    __shared__ uint32_t s_mark[NTHR / 4];          // 64 bytes per wavefront: first lanes of the pieces that begin inside a window
    __shared__ uint32_t s_cnt;                     // (structure, key) pairs listed so far
    const uint32_t wg = blockIdx.x;
    const uint32_t t = wg % A.NT, q = wg / A.NT, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    unsigned long long tstamp = A.dbg ? wall_clock64() : 0ull;
    auto stamp = [&](int k) {      // FDGPU_QT_DBG: phase durations of the workgroup's first thread, summed over the launch (100 MHz ticks)
        if (A.dbg && tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&A.dbg[k], now - tstamp); tstamp = now; }
    };
    const uint4 hd = A.heads[(uint64_t)q * A.NT + t];
    const uint32_t n_pieces = hd.x, n_win = hd.w ? 0u : hd.y, sbase = hd.z;
    const uint64_t r0 = A.q_rows[q];
    const uint32_t nrows = (uint32_t)(A.q_rows[q + 1] - r0);
    const uint32_t tile_lo = t << TL2;
    const uint32_t tile_lim = A.S - tile_lo < TILE ? A.S - tile_lo : TILE;
    const uint32_t tile_id0 = A.first_id + tile_lo;
    const uint64_t cbase = ((uint64_t)q * A.NT + t) << TL2;
    const uint64_t rbase = r0 * A.NT + (uint64_t)nrows * t;
    const uint64_t pbase = rbase << (TL2 - QT_CELL_LOG2), wbase = rbase * A.win_per_row + 2ull * ((uint64_t)q * A.NT + t);
    // the wavefront's windows (w = wv + NW * i; lane i holds window i's table entry and the entry behind it), requested before the accumulators are cleared
    uint32_t we = QT_WIN_CONT, wn = 0;
    { const uint32_t wi = wv + NW * lane; if (wi < n_win) { we = A.win[wbase + wi]; if (wi + 1u < n_win) wn = A.win[wbase + wi + 1u]; } }
    for (uint32_t k = tid * 4u; k < TILE; k += NTHR * 4u) *reinterpret_cast<qt_u32x4 *>(&s_acc[k]) = qt_u32x4{0u, 0u, 0u, 0u};
    for (uint32_t k = tid; k < QT_BINS / 2; k += NTHR) s_hist[k] = 0u;
    if (tid == 0) s_cnt = 0u;
    q32_barrier_lds();
    stamp(0);
    // ---- decode: three stages in flight per wavefront — the piece descriptors of the window after next are requested while the posting bytes of
    // the next window travel and the current window is decoded (a window costs two dependent global round trips; a wavefront has ~2 windows)
    uint32_t my_steps = 0;
    uint32_t i0 = 0, it = 0xffffffffu;       // chunk of 64 of the wavefront's windows, position inside it
    auto next_win = [&](const q32_win &X) -> q32_win {
        q32_win Y;
        if (X.valid && (X.nxt & QT_WIN_CONT)) {        // the window behind continues a piece of more than 64 slots: same wavefront, carried id
            Y.w = X.w + 1u; Y.c0 = X.nxt & ~QT_WIN_CONT; Y.cont = true; Y.valid = true;
            Y.nxt = Y.w + 1u < n_win ? A.win[wbase + Y.w + 1u] : 0u;
            return Y;
        }
        Y.w = 0; Y.c0 = 0; Y.nxt = 0; Y.cont = false; Y.valid = false;
        for (;;) {
            ++it;
            if (it >= FD_WAVE) {       // the next 64 of the wavefront's windows
                i0 += FD_WAVE; it = 0;
                if (wv + NW * i0 >= n_win) return Y;
                we = QT_WIN_CONT; wn = 0;
                const uint32_t wi = wv + NW * (i0 + lane);
                if (wi < n_win) { we = A.win[wbase + wi]; if (wi + 1u < n_win) wn = A.win[wbase + wi + 1u]; }
            }
            const uint32_t w = wv + NW * (i0 + it);
            if (w >= n_win) return Y;
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)we, (int)it);
            if (e & QT_WIN_CONT) continue;        // inside a long piece: the wavefront that began the piece decodes it
            Y.w = w; Y.c0 = e; Y.nxt = (uint32_t)__builtin_amdgcn_readlane((int)wn, (int)it); Y.valid = true;
            return Y;
        }
    };
    auto load_pieces = [&](const q32_win &X) -> q32_pieces {
        q32_pieces R; R.d = make_uint4(0u, 0u, 0u, 0u); R.Pp = 0xffffffffu;
        const uint32_t ci = X.c0 + lane;
        if (X.valid && ci < n_pieces) { R.d = A.pieces[pbase + ci]; R.Pp = A.piece_p[pbase + ci]; }
        return R;
    };
    // the pieces that BEGIN inside the window mark their first lane (bytes in the wavefront's scratch), a ballot of the marks + popcount below the
    // lane = the lane's piece; its fields come from the lane that fetched it (volatile: the lanes talk to each other through these bytes — without
    // it the compiler forwards a lane's own 0 to its read)
    auto prep = [&](const q32_win &X, const q32_pieces &R) -> q32_slot {
        q32_slot S;
        const uint32_t w0 = X.w << 6;
        volatile uint8_t *mark = reinterpret_cast<volatile uint8_t *>(s_mark) + wv * FD_WAVE;
        mark[lane] = 0;
        if (R.Pp - w0 < FD_WAVE) mark[R.Pp - w0] = 1;
        const uint64_t begun = __ballot(mark[lane] != 0);
        const uint32_t kk = (uint32_t)__popcll(begun & ((2ull << lane) - 1ull));
        const bool has = X.cont || kk != 0u;
        const int k = (int)(X.cont ? kk : (kk ? kk - 1u : 0u));
        const uint32_t bx = (uint32_t)__shfl((int)R.d.x, k, FD_WAVE), by = (uint32_t)__shfl((int)R.d.y, k, FD_WAVE);
        const uint32_t nbytes = (uint32_t)__shfl((int)R.d.z, k, FD_WAVE);
        S.prev = (uint32_t)__shfl((int)R.d.w, k, FD_WAVE);
        S.pstart = (uint32_t)__shfl((int)R.Pp, k, FD_WAVE);
        S.rel = w0 + lane - S.pstart;
        const bool active = has && S.rel < ((nbytes + 15u) >> 4);
        S.nby = active ? (nbytes - 16u * S.rel < 16u ? nbytes - 16u * S.rel : 16u) : 0u;
        S.row = by >> 16;
        __builtin_memcpy(&S.x4, A.value + ((((uint64_t)(by & 0xffffu)) << 32) | bx) + (active ? 16ull * S.rel : 0ull), 16);
        S.add = active ? (uint32_t)(A.row_meta[r0 + S.row] >> 2) : 0u;
        return S;
    };
    uint32_t carry = 0, prev_last = 0;
    auto decode = [&](const q32_win &X, const q32_slot &S) {
        ++my_steps;
        const uint32_t w0 = X.w << 6;
        // ---- lane-local decode: the varints that END in these 16 bytes; the leading bytes of the first are the tail of the slot before —
        // the lane below's last four bytes (lane 0 of a continued piece: lane 63 of the window before)
        uint32_t lb = (uint32_t)__shfl_up((int)S.x4[3], 1, FD_WAVE);
        if (lane == 0) lb = prev_last;
        prev_last = (uint32_t)__builtin_amdgcn_readlane((int)S.x4[3], 63);
        uint32_t cv = 0, sh = 0;
        if (S.nby && S.rel) {
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
            const uint32_t b = (S.x4[i >> 2] >> (8 * (i & 3))) & 0xffu;
            cv |= (b & 0x7fu) << (sh & 31u);
            const bool term = (uint32_t)i < S.nby && !(b & 0x80u);
            v[i] = term ? cv : 0u;
            T |= term ? (1u << i) : 0u;
            D += v[i];
            sh = term ? 0u : sh + 7u;
            cv = term ? 0u : cv;
        }
        // ---- ids: prefix of the lane sums inside the piece, from the piece's checkpoint id (or the window before)
        const uint32_t incl = qt_wave_incl(D, lane);
        const uint32_t fl = S.pstart > w0 ? S.pstart - w0 : 0u;            // the piece's first lane in this window
        const uint32_t pre = (uint32_t)__shfl((int)(incl - D), (int)fl, FD_WAVE);
        const uint32_t row_base = S.pstart < w0 ? carry : S.prev;
        const uint32_t id_first = row_base + (incl - D) - pre;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)(id_first + D), 63);
        // ---- adds (nothing comes back) and the slot's record of the decoded stream: sixteen 16-bit tile-local ids, 0xffff = none
        uint32_t id = id_first, sw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            id += v[i];
            const uint32_t x = id - tile_id0;
            const bool ok = ((T >> i) & 1u) && x < tile_lim;
            if (ok) atomicAdd(&s_acc[x], S.add);
            sw[i >> 1] |= (ok ? x : 0xffffu) << ((i & 1) * 16);
        }
        const uint64_t rec = (uint64_t)sbase + w0 + lane;
        qt_u32x4 *dst = reinterpret_cast<qt_u32x4 *>(A.stream_ids) + 2ull * rec;
        dst[0] = qt_u32x4{sw[0], sw[1], sw[2], sw[3]};
        dst[1] = qt_u32x4{sw[4], sw[5], sw[6], sw[7]};
        A.stream_row[rec] = (uint16_t)S.row;
    };
    {
        q32_win none; none.w = 0; none.c0 = 0; none.nxt = 0; none.cont = false; none.valid = false;
        q32_win W0 = next_win(none);
        if (W0.valid) {
            q32_pieces P0 = load_pieces(W0);
            q32_win W1 = next_win(W0);
            q32_pieces P1 = load_pieces(W1);
            q32_slot S0 = prep(W0, P0);
            for (;;) {
                const q32_win W2 = W1.valid ? next_win(W1) : none;
                const q32_pieces P2 = load_pieces(W2);
                q32_slot S1 = S0;
                if (W1.valid) S1 = prep(W1, P1);
                decode(W0, S0);
                if (!W1.valid) break;
                W0 = W1; S0 = S1; W1 = W2; P1 = P2;
            }
        }
    }
    stamp(1);
    // ---- finalize.  A wavefront owns BLK = 2,048 neighbouring structures (lane: 8 groups of four).  Their penalties travel while the slower
    // wavefronts finish; the barrier waits for LDS only (__syncthreads would drain the stream's stores and these loads first)
    constexpr uint32_t BLK = TILE / NW, J = BLK / 256u;
    static_assert(BLK == 2048u, "a wavefront's block of accumulators doubles as its list of 1,024 touched structures");
    const uint32_t kb = wv * BLK + lane * 4u;             // the lane's first structure of group j: kb + 256 j
    float pen[J][4];
#pragma unroll
    for (uint32_t j = 0; j < J; ++j) {
        const uint32_t k = kb + 256u * j;
#pragma unroll
        for (int u = 0; u < 4; ++u) pen[j][u] = k + u < tile_lim ? A.penalty[tile_lo + k + u] : 0.0f;
    }
    q32_barrier_lds();
    stamp(2);
    if (A.dbg && lane == 0) atomicAdd(&A.dbg[16 + 1], (unsigned long long)my_steps);
    // the sums leave LDS for registers (the wavefront's block is then free), idf * penalty for every one of them (three instructions; the dozen
    // that follow — order key, bin, histogram — only for the touched structures, a quarter of the tile: compacted first)
    uint32_t kf[J][4];
    uint32_t n_t = 0;
#pragma unroll
    for (uint32_t j = 0; j < J; ++j) {
        const qt_u32x4 a = *reinterpret_cast<const qt_u32x4 *>(&s_acc[kb + 256u * j]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (float)((double)sum * 2^-22) of the 64-bit kernel: one rounding of the integer, the power of two is exact; all ones = untouched
            kf[j][u] = a[u] ? __float_as_uint(((float)a[u] * (float)(1.0 / QT_IDF_SCALE)) * pen[j][u]) : 0xffffffffu;
            n_t += (uint32_t)__popcll(__ballot(a[u] != 0u));
        }
    }
    const bool dense = n_t <= BLK / 2u;              // (key, structure) pairs of 8 bytes in the block's 8 KB
    uint32_t *const lst = &s_acc[wv * BLK];          // entry e = words 2 e (key), 2 e + 1 (structure of the tile)
    if (dense) {
        uint32_t at = 0;
#pragma unroll
        for (uint32_t j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t m = __ballot(kf[j][u] != 0xffffffffu);
                if (kf[j][u] != 0xffffffffu) *reinterpret_cast<uint2 *>(&lst[2u * (at + fd_mbcnt(m))]) = make_uint2(kf[j][u], kb + 256u * j + (uint32_t)u);
                at += (uint32_t)__popcll(m);
            }
        for (uint32_t e = lane; e < n_t; e += FD_WAVE) {
            const uint32_t key = qt_order_key(__uint_as_float(lst[2u * e])), bin = qt_bin(key);
            lst[2u * e] = key;
            atomicAdd(&s_hist[bin >> 1], 1u << ((bin & 1u) * 16u));
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kf[j][u] = kf[j][u] != 0xffffffffu ? qt_order_key(__uint_as_float(kf[j][u])) : 0u;       // 0 = untouched (a key is never 0)
                if (kf[j][u]) { const uint32_t bin = qt_bin(kf[j][u]); atomicAdd(&s_hist[bin >> 1], 1u << ((bin & 1u) * 16u)); }
            }
    }
    q32_barrier_lds();
    stamp(3);
    // ---- the tile's own cut: the highest bin b with (keys in the bins above b) + hist[b] >= top_n (bin 0 when the tile holds fewer) — every
    // wavefront works it out for itself, no hand-over through LDS: lane l sums bins 32 l .. 32 l + 31, the lane where the count from the top
    // crosses top_n is split over the lanes once more
    uint32_t bloc = 0;
    {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t u = 0; u < 16; u += 4) {
            const qt_u32x4 hw = *reinterpret_cast<const qt_u32x4 *>(&s_hist[16u * lane + u]);
#pragma unroll
            for (int z = 0; z < 4; ++z) tot += (hw[z] & 0xffffu) + (hw[z] >> 16);
        }
        const uint32_t incl = qt_wave_incl(tot, lane);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t suffix = all - (incl - tot);         // keys in this lane's bins and above
        const uint64_t mk = __ballot(suffix >= A.top_n);
        if (mk) {
            const int L = 63 - __clzll((long long)mk);
            const uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)(suffix - tot), L);
            const uint32_t wd = s_hist[16u * (uint32_t)L + ((lane & 31u) >> 1)];
            const uint32_t h = lane < 32u ? ((lane & 1u) ? wd >> 16 : wd & 0xffffu) : 0u;
            const uint32_t i2 = qt_wave_incl(h, lane);
            const uint32_t all2 = (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
            const uint64_t mk2 = __ballot(lane < 32u && above + all2 - (i2 - h) >= A.top_n);       // never empty: lane 0 sees the whole of lane L's count
            bloc = 32u * (uint32_t)L + (uint32_t)(63 - __clzll((long long)mk2));
        }
    }
    stamp(4);
    // ---- (structure, key) of the keys from that bin up, a wavefront's picks side by side (slots claimed from the workgroup's counter, any order);
    // those bins into the query's histogram
    const uint32_t thr = qt_edge(bloc);
    if (dense) {
        for (uint32_t e0 = 0; e0 < n_t; e0 += FD_WAVE) {
            const uint2 x = e0 + lane < n_t ? *reinterpret_cast<const uint2 *>(&lst[2u * (e0 + lane)]) : make_uint2(0u, 0u);
            const bool sel = x.x != 0u && x.x >= thr;
            const uint64_t m = __ballot(sel);
            if (!m) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (sel) { const uint64_t at = cbase + base + fd_mbcnt(m); A.c_nid[at] = tile_lo + x.y; A.c_key[at] = x.x; }
        }
    } else {
        uint32_t mycnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) mycnt += (uint32_t)__popcll(__ballot(kf[j][u] != 0u && kf[j][u] >= thr));
        uint32_t pos = 0;
        if (lane == 0 && mycnt) pos = atomicAdd(&s_cnt, mycnt);
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        if (mycnt) {
#pragma unroll
            for (uint32_t j = 0; j < J; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool sel = kf[j][u] != 0u && kf[j][u] >= thr;
                    const uint64_t m = __ballot(sel);
                    if (sel) { const uint64_t at = cbase + pos + fd_mbcnt(m); A.c_nid[at] = tile_lo + kb + 256u * j + (uint32_t)u; A.c_key[at] = kf[j][u]; }
                    pos += (uint32_t)__popcll(m);
                }
        }
    }
    for (uint32_t b = bloc + tid; b < QT_BINS; b += NTHR) {
        const uint32_t cn = (s_hist[b >> 1] >> ((b & 1u) * 16u)) & 0xffffu;
        if (cn) atomicAdd(&A.ghist[(uint64_t)q * QT_BINS + b], cn);
    }
    q32_barrier_lds();
    if (tid == 0) A.ccount[(uint64_t)q * A.NT + t] = s_cnt;
    stamp(5);
}

void fd_launch_qt_layout(const qt_args &A, hipStream_t st) {
    if (!A.n_queries || !A.S) return;
    hipLaunchKernelGGL(k_qt_layout, dim3(A.NT * A.n_queries), dim3(FD_WAVE), 0, st, A);
    hipLaunchKernelGGL(k_qt_bases, dim3(1), dim3(1024), 0, st, A);
}
void fd_launch_qt_score32(const qt_args &A, hipStream_t st) {
    if (!A.n_queries || !A.S) return;
    const dim3 g(A.NT * A.n_queries);
    if (A.tile_log2 == 15) hipLaunchKernelGGL((k_qt_score32<15, 1024>), g, dim3(1024), 0, st, A);
    else if (A.tile_log2 == 13) hipLaunchKernelGGL((k_qt_score32<13, 256>), g, dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_qt_score32<14, 512>), g, dim3(512), 0, st, A);
}
