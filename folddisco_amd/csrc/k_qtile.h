// k_qtile.h — arithmetic and wavefront helpers shared by the tiled scoring kernels (k_qtile.hip, k_qscore32.hip): the ranking key and its
// histogram bins (candidate selection of src/cli/workflows/query_pdb.rs:404-411 over the idf sums of src/controller/count_query.rs:82-220),
// DPP wave scan, block scan, the row-group hit counter.
#pragma once
#include "fdgpu_internal.h"

#define QT_IDF_SCALE 4194304.0 /* 2^22 */
#define QT_CNT_SHIFT 46
#define QT_SUM_MASK ((1ull << QT_CNT_SHIFT) - 1ull)
struct qt_rec { uint32_t nid, total_match_count, node_count, edge_count; float idf; };      // = fd_count_rec

__device__ __forceinline__ uint32_t qt_order_key(float v) {
    uint32_t b = __float_as_uint(v + 0.0f);   // -0 -> +0
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// first selection level: 2,048 bins over the order-preserving key; bins 1..2046 are 2^17 keys wide (64 per binade: 1.5 % relative
// width) from 2^-16 up, bin 0 holds everything below (idf 0, negative penalties), bin 2047 everything from ~2^16 up
#define QT_K0 0xB7800000u
__device__ __forceinline__ uint32_t qt_bin(uint32_t key) {
    if (key < QT_K0) return 0u;
    const uint32_t b = ((key - QT_K0) >> 17) + 1u;
    return b < (QT_BINS - 1u) ? b : (QT_BINS - 1u);
}
__device__ __forceinline__ uint32_t qt_edge(uint32_t bin) { return bin ? QT_K0 + ((bin - 1u) << 17) : 0u; }
__device__ __forceinline__ uint32_t qt_shift2(uint32_t bin) { return bin == 0u ? 21u : bin == QT_BINS - 1u ? 20u : 6u; }

// inclusive prefix sum over the wavefront in six DPP adds (row_shr 1/2/4/8 inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31
// carry the row totals across) — __shfl_up costs a ds_bpermute round trip per step
__device__ __forceinline__ uint32_t qt_wave_incl(uint32_t v, uint32_t /*lane*/) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// exclusive scan over the workgroup's NTHR values (two barriers); s_w: NTHR / 64 words of LDS
template <int NTHR>
__device__ __forceinline__ uint32_t qt_block_excl(uint32_t v, uint32_t tid, uint32_t *s_w, uint32_t *total) {
    const uint32_t lane = tid & 63u, wv = tid >> 6;
    const uint32_t incl = qt_wave_incl(v, lane);
    if (lane == 63u) s_w[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < NTHR / 64; ++k) { const uint32_t x = s_w[k]; pre += k < wv ? x : 0u; tot += x; }
    __syncthreads();
    *total = tot;
    return pre + incl - v;
}
typedef unsigned int qt_u32x4 __attribute__((ext_vector_type(4)));
struct qt_step { qt_u32x4 w; uint32_t c, pstart, nby, rel; };

// groups (runs of rows that end at a set bit of `ends`) holding at least one set bit of `m`, one 32-row word of a longer row list:
// adding the non-end hits to the non-end positions lets a hit's carry run up to its group's end bit; carry = the group straddles the word
__device__ __forceinline__ uint32_t qt_groups_hit(uint32_t m, uint32_t ends, uint32_t valid, uint32_t &carry) {
    const unsigned long long sum = (unsigned long long)(m & ~ends) + (unsigned long long)(~ends & valid) + carry;
    carry = (uint32_t)(sum >> 32);
    return (uint32_t)__popc((((uint32_t)sum) & ends) | (m & ends));
}


// byte range of row r's posting list around granule `gran` (2^cpg_log2 checkpoint cells) as the checkpoints delimit it: {first byte lo / hi
// (absolute), bytes, id before the first posting}; bytes = 0: nothing to decode here (absent hash, or a list whose entries lie further apart than
// a granule — the piece is handed to the FIRST granule of the entry inside the tile, a tile decodes every piece once)
__device__ __forceinline__ uint4 qt_piece_range(const qt_args &A, uint32_t r, uint32_t gran, uint32_t cpg_log2) {
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    const long long k = A.kidx[r];
    if (k >= 0) {
        const uint64_t b0 = A.offsets[k], len = A.offsets[k + 1] - b0;
        const unsigned long long m = A.ck_meta[k];
        const uint32_t j = (uint32_t)(m >> 56);
        const uint2 *e = A.ck_ent + (m & ((1ull << 56) - 1ull));
        const uint32_t n_e = (uint32_t)(((uint64_t)A.NC + (1ull << j) - 1ull) >> j);
        const uint32_t cpt_log2 = A.tile_log2 - QT_CELL_LOG2;
        const uint32_t c0 = gran << cpg_log2, c1 = c0 + (1u << cpg_log2) < A.NC ? c0 + (1u << cpg_log2) : A.NC;
        const uint32_t tile_c0 = (c0 >> cpt_log2) << cpt_log2;
        const uint32_t e0 = c0 >> j, e1 = ((c1 - 1u) >> j) + 1u;
        const uint32_t first_c = (e0 << j) > tile_c0 ? (e0 << j) : tile_c0;       // the entry's first cell inside this tile
        if (j <= cpg_log2 || c0 == first_c) {
            uint32_t sb = 0, prev = 0;
            if (e0 && n_e > 1u) { const uint2 x = e[e0 - 1u]; sb = x.x; prev = x.y; }
            const uint64_t eb = (e1 >= n_e || n_e <= 1u) ? len : (uint64_t)e[e1 - 1u].x;
            const uint64_t p = b0 + sb;
            out = make_uint4((uint32_t)p, (uint32_t)(p >> 32), (uint32_t)(eb - sb), prev);
        }
    }
    return out;
}
