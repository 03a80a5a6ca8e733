// k_sort.hip — device-wide primitives for the posting build: exclusive scans and a stable
// LSD radix sort of (u32 hash key, u32 structure id) pairs.
//
// Replaces HOT LOOP B of the reference (src/controller/mod.rs:349-358 +
// src/index/indextable.rs:88-105,171-202: random read-modify-write into dense 2^30-entry tables)
// with sort-based grouping: keys arrive id-major (segments per structure), a *stable* sort by the
// 30-bit hash therefore yields, for every hash, its structure ids in ascending order — exactly the
// order in which the reference appends to a posting list.
//
// HBM-bound integer work; wavefront idioms used: ballot-based multi-split for stable in-wave ranks
// (wave64: one 64-bit ballot per digit bit), LDS-staged reorder so that global writes go out as
// per-digit runs instead of 64 scattered dwords.
#include "fdgpu_internal.h"

// ------------------------------------------------------------------------ generic exclusive scan
// out[k] = sum_{t<k} in[t], out[n] = total.  Three launches: chunk sums -> scan of sums -> apply.
#define SCAN_THREADS 256
#define SCAN_ITEMS 16
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t t = __shfl_up(v, off, FD_WAVE);
        if ((int)lane >= off) v += t;
    }
    return v;
}
// block-wide exclusive scan of one value per thread (blockDim.x <= 1024); returns exclusive prefix,
// *total = block sum.  sm must hold 16 u64.
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *sm, uint64_t *total) {
    uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint64_t inc = wave_incl_scan_u64(v);
    if (lane == 63) sm[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        uint64_t t = lane < nw ? sm[lane] : 0;
        uint64_t ti = wave_incl_scan_u64(t);
        if (lane < nw) sm[lane] = ti - t;
        if (lane == nw - 1) sm[16] = ti;
    }
    __syncthreads();
    uint64_t r = inc - v + sm[wid];
    *total = sm[16];
    __syncthreads();
    return r;
}

template <typename TIn>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const TIn *__restrict__ in, uint64_t n, uint64_t *__restrict__ chunk_sum) {
    __shared__ uint64_t sm[17];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t idx = base + (uint64_t)k * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += (uint64_t)in[idx];
    }
    uint64_t tot;
    block_excl_scan_u64(s, sm, &tot);
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = tot;
}
// single block: exclusive scan of chunk sums in place; writes grand total to total_out[0]
__global__ __launch_bounds__(1024) void k_scan_sums(uint64_t *__restrict__ chunk_sum, uint64_t n_chunks, uint64_t *__restrict__ total_out) {
    __shared__ uint64_t sm[17];
    uint64_t carry = 0;
    for (uint64_t b = 0; b < n_chunks; b += 1024) {
        uint64_t idx = b + threadIdx.x;
        uint64_t v = idx < n_chunks ? chunk_sum[idx] : 0;
        uint64_t tot;
        uint64_t ex = block_excl_scan_u64(v, sm, &tot);
        if (idx < n_chunks) chunk_sum[idx] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) total_out[0] = carry;
}
template <typename TIn>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const TIn *__restrict__ in, uint64_t n, const uint64_t *__restrict__ chunk_sum,
                                                             const uint64_t *__restrict__ total, uint64_t *__restrict__ out) {
    __shared__ uint64_t sm[17];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t idx = base + k;
        v[k] = idx < n ? (uint64_t)in[idx] : 0;
        s += v[k];
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(s, sm, &tot) + chunk_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t idx = base + k;
        if (idx < n) out[idx] = ex;
        ex += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total[0];
}

// NOTE: reduce uses a strided item->thread map, apply a blocked one; both cover the same chunk.
template <typename TIn>
void fd_exclusive_scan(const TIn *in, uint64_t n, uint64_t *out /*[n+1]*/, uint64_t *chunk_tmp, uint64_t *total_dev, hipStream_t st) {
    uint64_t nc = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nc == 0) nc = 1;
    hipLaunchKernelGGL(k_scan_reduce<TIn>, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, in, n, chunk_tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, chunk_tmp, nc, total_dev);
    hipLaunchKernelGGL(k_scan_apply<TIn>, dim3((unsigned)nc), dim3(SCAN_THREADS), 0, st, in, n, chunk_tmp, total_dev, out);
}
template void fd_exclusive_scan<uint32_t>(const uint32_t *, uint64_t, uint64_t *, uint64_t *, uint64_t *, hipStream_t);
template void fd_exclusive_scan<uint8_t>(const uint8_t *, uint64_t, uint64_t *, uint64_t *, uint64_t *, hipStream_t);
uint64_t fd_scan_tmp_elems(uint64_t n) { return (n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1; }

// ------------------------------------------------------------------------ radix sort
#define RS_BINS 256

// tile histogram -> ghist[tile * 256 + digit]  (tile-major: a tile's 256 counts are one 1 KiB line group — the hist kernel
// writes them and the scatter kernel reads them with ONE coalesced access instead of 256 strided 4-byte ones)
template <int THREADS, int ITEMS, bool XCD>
__global__ __launch_bounds__(THREADS) void k_rs_hist(const uint32_t *__restrict__ keys, uint64_t n, uint32_t shift, uint32_t mask,
                                                     uint32_t *__restrict__ ghist, uint32_t nb) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t h[RS_BINS];
    const uint32_t tile = XCD ? fd_xcd_remap(blockIdx.x, nb) : blockIdx.x;
    if (tile >= nb) return;
    for (int k = threadIdx.x; k < RS_BINS; k += THREADS) h[k] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)tile * TILE;
    if (base + TILE <= n) {
        // full tile: 16-byte loads (4 consecutive keys per lane), 1 KiB per wave instruction
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; ++k) {
            u32x4 v = __builtin_nontemporal_load(&k4[k * THREADS + threadIdx.x]);
            atomicAdd(&h[(v.x >> shift) & mask], 1u);
            atomicAdd(&h[(v.y >> shift) & mask], 1u);
            atomicAdd(&h[(v.z >> shift) & mask], 1u);
            atomicAdd(&h[(v.w >> shift) & mask], 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            uint64_t idx = base + (uint64_t)k * THREADS + threadIdx.x;
            if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < RS_BINS; k += THREADS) ghist[(uint64_t)tile * RS_BINS + k] = h[k];
}

// Exclusive scan over tiles of every digit's count, in place on the tile-major table (thread = digit, so every access is a
// coalesced 1 KiB row): chunk sums -> scan of the chunk sums per digit (+ digit totals) -> apply.
#define RS_SCAN_CHUNK 128   // tiles per workgroup
__global__ __launch_bounds__(RS_BINS) void k_rs_scan_csum(const uint32_t *__restrict__ ghist, uint32_t nb, uint64_t *__restrict__ csum) {
    const uint32_t t0 = blockIdx.x * RS_SCAN_CHUNK, t1 = t0 + RS_SCAN_CHUNK < nb ? t0 + RS_SCAN_CHUNK : nb;
    uint64_t s = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) s += ghist[(uint64_t)t * RS_BINS + threadIdx.x];
    csum[(uint64_t)blockIdx.x * RS_BINS + threadIdx.x] = s;
}
// 16 workgroups of 16 digits x 64 parts: thread (digit d, part p) scans 1/64 of the chunk sums of digit d; the 64 partial totals are
// combined in LDS
__global__ __launch_bounds__(1024) void k_rs_scan_chunks(uint64_t *__restrict__ csum, uint32_t n_chunks, uint64_t *__restrict__ tot) {
    __shared__ uint64_t part[64][16];
    const uint32_t dl = threadIdx.x & 15u, p = threadIdx.x >> 4, d = blockIdx.x * 16u + dl;
    const uint32_t per = (n_chunks + 63) / 64, c0 = p * per < n_chunks ? p * per : n_chunks, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    uint64_t s = 0;
#pragma unroll 4
    for (uint32_t c = c0; c < c1; ++c) s += csum[(uint64_t)c * RS_BINS + d];
    part[p][dl] = s;
    __syncthreads();
    uint64_t run = 0;
    for (uint32_t q = 0; q < p; ++q) run += part[q][dl];
    if (p == 63) tot[d] = run + s;
    for (uint32_t c = c0; c < c1; ++c) { uint64_t v = csum[(uint64_t)c * RS_BINS + d]; csum[(uint64_t)c * RS_BINS + d] = run; run += v; }
}
__global__ __launch_bounds__(RS_BINS) void k_rs_scan_apply(uint32_t *__restrict__ ghist, uint32_t nb, const uint64_t *__restrict__ csum) {
    const uint32_t t0 = blockIdx.x * RS_SCAN_CHUNK, t1 = t0 + RS_SCAN_CHUNK < nb ? t0 + RS_SCAN_CHUNK : nb;
    // positions inside one digit's run of a bucket, 32 bits: the unsegmented sort takes n < 2^32 keys; the segmented sort (calls of up to 2^35
    // keys) checks every (bucket, digit) total against 2^32 in k_rs_scan_tot_seg and flags the call (FDGPU_ERANGE) instead of wrapping here
    uint32_t run = (uint32_t)csum[(uint64_t)blockIdx.x * RS_BINS + threadIdx.x];
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) {
        uint32_t v = ghist[(uint64_t)t * RS_BINS + threadIdx.x];
        ghist[(uint64_t)t * RS_BINS + threadIdx.x] = run;
        run += v;
    }
}
__global__ __launch_bounds__(RS_BINS) void k_rs_scan_tot(uint64_t *__restrict__ tot) {
    __shared__ uint64_t sm[17];
    uint64_t v = tot[threadIdx.x], t;
    uint64_t ex = block_excl_scan_u64(v, sm, &t);
    tot[threadIdx.x] = ex;
}


// Stable scatter of one tile.  Wave w owns a contiguous sub-tile and walks it in 64-key chunks (chunk c, lane l -> element
// c*64 + l) so that "earlier element" == "earlier chunk or lower lane".
//   1. rank first, no LDS atomics: a ballot multi-split gives every key its running rank among the wave's keys of the same
//      digit (the leader lane keeps the per-wave digit counter); the per-bit peer mask update is ONE v_bitop3_b32 per 32-bit
//      half (peers & ~(ballot ^ bitmask), truth table 0x90), the ballot one v_cmp on the extracted bit;
//   2. cross-wave exclusive scan of the digit counts, global offsets from the tile histogram pass;
//   3. keys / payloads are placed digit-contiguously in LDS and leave as runs (neighbouring tiles of one XCD complete each
//      other's partial lines in L2: PMC traffic 1.03x algorithmic).
// Full tiles run without any bounds predicate.  The kernel sits at 5.9 ms per pass against 4.8 ms for its own memory
// skeleton (k_rs_copy_floor).
template <int THREADS, int ITEMS, typename V, bool FULL, bool PACK = false>
__device__ __forceinline__ void rs_scatter4_body(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                 uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n, uint32_t shift,
                                                 uint32_t mask, const uint32_t *__restrict__ ghist_row, const uint64_t *__restrict__ dbase,
                                                 uint64_t tile_base, uint32_t n_tile_in, uint32_t *s_keys, V *s_vals, uint32_t (*s_cnt)[RS_BINS], long long *s_gofs,
                                                 uint64_t *sm) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
    const uint32_t n_tile = FULL ? TILE : n_tile_in;
    const uint32_t n_wave = FULL ? 64 * ITEMS : (n_tile > wid * 64 * ITEMS ? n_tile - wid * 64 * ITEMS : 0u);   // valid keys of this wave

    for (int k = tid; k < WAVES * RS_BINS; k += THREADS) (&s_cnt[0][0])[k] = 0;
    uint32_t key[ITEMS];
    V val[ITEMS];
    const uint32_t *kp = keys_in + wave_base + lane;
    const V *vp = vals_in + wave_base + lane;
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        bool ok = FULL || (uint32_t)(c * 64) + lane < n_wave;
        key[c] = ok ? kp[c * 64] : 0xffffffffu;
        val[c] = ok ? vp[c * 64] : (V)0;
    }
    long long gbase = 0;
    if (tid < RS_BINS) gbase = (long long)(dbase[tid] + ghist_row[tid]);
    __syncthreads();
    uint32_t rnk[ITEMS];
    uint32_t *cnt = s_cnt[wid];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        const uint32_t d = (key[c] >> shift) & mask;
        uint32_t plo = ~0u, phi = ~0u;
        bool ok = true;
        if (!FULL) {
            ok = (uint32_t)(c * 64) + lane < n_wave;
            uint64_t okm = __ballot(ok);
            plo = (uint32_t)okm; phi = (uint32_t)(okm >> 32);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int32_t m = __builtin_amdgcn_sbfe((int32_t)d, b, 1);      // 0 or ~0
            const uint64_t bal = __builtin_amdgcn_ballot_w64(m < 0);   // sign test of the extracted bit: one v_cmp on m itself
            plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, (uint32_t)m, 0x90);
            phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t pcount = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
        uint32_t base = 0;
        if (ok) base = cnt[d];
        if (ok && rank == pcount - 1) cnt[d] = base + pcount;   // in-order LDS within the wave: reads above precede this write
        rnk[c] = base + rank;
    }
    __syncthreads();
    uint32_t dstart;
    {
        uint32_t my_total = 0;
        if (tid < RS_BINS) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) my_total += s_cnt[w][tid];
        }
        uint64_t tot;
        dstart = (uint32_t)block_excl_scan_u64(tid < RS_BINS ? (uint64_t)my_total : 0ull, sm, &tot);
        if (tid < RS_BINS) {
            uint32_t run = dstart;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
            s_gofs[tid] = gbase - (long long)dstart;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        if (FULL || (uint32_t)(c * 64) + lane < n_wave) {
            uint32_t pos = cnt[(key[c] >> shift) & mask] + rnk[c];
            if (PACK) reinterpret_cast<uint2 *>(s_keys)[pos] = make_uint2(key[c], (uint32_t)val[c]);
            else { s_keys[pos] = key[c]; s_vals[pos] = val[c]; }
        }
    }
    __syncthreads();
    if (FULL) {
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            uint32_t k = c * THREADS + tid;
            uint32_t kk, vv;
            if (PACK) { uint2 e = reinterpret_cast<const uint2 *>(s_keys)[k]; kk = e.x; vv = e.y; }
            else { kk = s_keys[k]; vv = (uint32_t)s_vals[k]; }
            long long g = (long long)k + s_gofs[(kk >> shift) & mask];
            keys_out[g] = kk;
            vals_out[g] = (V)vv;
        }
    } else {
        for (uint32_t k = tid; k < n_tile; k += THREADS) {
            uint32_t kk, vv;
            if (PACK) { uint2 e = reinterpret_cast<const uint2 *>(s_keys)[k]; kk = e.x; vv = e.y; }
            else { kk = s_keys[k]; vv = (uint32_t)s_vals[k]; }
            long long g = (long long)k + s_gofs[(kk >> shift) & mask];
            keys_out[g] = kk;
            vals_out[g] = (V)vv;
        }
    }
}

// measurement aid (FDGPU_SORT=classic30): the scatter's memory skeleton without ranking — load the tile like k_rs_scatter4, stage
// through LDS, store the tile back in place order.  NOT a sort; used only to read off the kernel's memory floor.
template <int THREADS, int ITEMS, typename V>
__global__ __launch_bounds__(THREADS) void k_rs_copy_floor(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                           uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n, uint32_t nb) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    const uint32_t tile = fd_xcd_remap(blockIdx.x, nb);
    if (tile >= nb || (uint64_t)(tile + 1) * TILE > n) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    const uint32_t *kp = keys_in + tile_base + (uint64_t)wid * (64 * ITEMS) + lane;
    const V *vp = vals_in + tile_base + (uint64_t)wid * (64 * ITEMS) + lane;
    uint32_t key[ITEMS];
    V val[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) { key[c] = kp[c * 64]; val[c] = vp[c * 64]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) { uint32_t pos = wid * 64 * ITEMS + c * 64 + lane; s_keys[pos] = key[c]; s_vals[pos] = val[c]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint32_t k = c * THREADS + tid;
        keys_out[tile_base + k] = s_keys[k];
        vals_out[tile_base + k] = s_vals[k];
    }
}

template <int THREADS, int ITEMS, typename V, bool PACK = false>
__global__ __launch_bounds__(THREADS) void k_rs_scatter4(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n,
                                                         uint32_t shift, uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb,
                                                         const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    __shared__ __attribute__((aligned(8))) uint32_t s_keys[PACK ? 2 * TILE : TILE];
    __shared__ V s_vals[PACK ? 1 : TILE];
    __shared__ uint32_t s_cnt[WAVES][RS_BINS];
    __shared__ long long s_gofs[RS_BINS];
    __shared__ uint64_t sm[17];
    const uint32_t tile = fd_xcd_remap(blockIdx.x, nb);
    if (tile >= nb) return;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    if (tile_base + TILE <= n)
        rs_scatter4_body<THREADS, ITEMS, V, true, PACK>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, ghist + (uint64_t)tile * RS_BINS, dbase, tile_base, TILE, s_keys,
                                                        s_vals, s_cnt, s_gofs, sm);
    else
        rs_scatter4_body<THREADS, ITEMS, V, false, PACK>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, ghist + (uint64_t)tile * RS_BINS, dbase, tile_base,
                                                         (uint32_t)(n - tile_base), s_keys, s_vals, s_cnt, s_gofs, sm);
}

// ------------------------------------------------------------------------ segmented form (MSD index build)
// The keys arrive partitioned into n_seg buckets (k_pair_emit2<.., MSD>: bucket = top six hash bits) and every bucket is sorted on its own
// by the remaining 24 hash bits: three passes instead of four.  Tiles stay aligned to multiples of TILE in memory (16-byte loads, full
// lines); a tile that a bucket boundary cuts becomes two partial "virtual tiles".  Virtual tiles are numbered bucket by bucket, every
// bucket's first one at a multiple of RS_SCAN_CHUNK, so that a scan chunk never spans two buckets (the padding tiles are empty).
// rs_seg_tab lives in device memory: bstart[b] = first key of bucket b, vt0[b] = its first virtual tile.
#define RS_MAX_SEG 64
struct rs_seg_tab { uint64_t bstart[RS_MAX_SEG + 1]; uint32_t vt0[RS_MAX_SEG + 1]; uint32_t n_seg, pad; };
__global__ void k_rs_seg_tiles(const uint64_t *__restrict__ seg_off, uint64_t stride, uint32_t n_seg, uint32_t tile, rs_seg_tab *__restrict__ T) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t v = 0;
    for (uint32_t b = 0; b <= n_seg; ++b) T->bstart[b] = seg_off[(uint64_t)b * stride];
    for (uint32_t b = 0; b < n_seg; ++b) {
        T->vt0[b] = v;
        const uint64_t lo = T->bstart[b], hi = T->bstart[b + 1];
        const uint32_t nvt = hi > lo ? (uint32_t)((hi + tile - 1) / tile - lo / tile) : 0u;
        v += (nvt + RS_SCAN_CHUNK - 1) / RS_SCAN_CHUNK * RS_SCAN_CHUNK;
    }
    T->vt0[n_seg] = v;
    T->n_seg = n_seg;
}
// virtual tile v -> its bucket and key range (cnt = 0: a padding tile), written once per sort into a 16-byte descriptor per tile: the hist
// and scatter kernels of the three passes read ONE descriptor instead of walking the bucket table (forty dependent scalar loads per
// workgroup cost 10 % of a pass)
struct rs_vtile { uint64_t lo; uint32_t cnt, seg; };
__global__ __launch_bounds__(256) void k_rs_seg_desc(const rs_seg_tab *__restrict__ T, uint32_t tile, uint32_t nbv, rs_vtile *__restrict__ D) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nbv) return;
    const uint32_t n_seg = T->n_seg;
    rs_vtile d = {0, 0, 0};
    if (v < T->vt0[n_seg]) {
        uint32_t lo = 0, hi = n_seg;                      // largest b with vt0[b] <= v
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (T->vt0[mid] <= v) lo = mid; else hi = mid; }
        const uint32_t b = lo;
        const uint64_t s0 = T->bstart[b], s1 = T->bstart[b + 1];
        const uint64_t t = s0 / tile + (v - T->vt0[b]);
        const uint64_t a = t * tile > s0 ? t * tile : s0, e = (t + 1) * tile < s1 ? (t + 1) * tile : s1;
        d.lo = a; d.seg = b; d.cnt = e > a ? (uint32_t)(e - a) : 0u;
    }
    D[v] = d;
}
__device__ __forceinline__ void rs_seg_lookup(const rs_vtile *__restrict__ D, uint32_t v, uint32_t *seg, uint64_t *lo, uint32_t *cnt) {
    const rs_vtile d = D[v];
    *seg = d.seg; *lo = d.lo; *cnt = d.cnt;
}
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void k_rs_hist_seg(const uint32_t *__restrict__ keys, const rs_vtile *__restrict__ D, uint32_t shift, uint32_t mask,
                                                         uint32_t *__restrict__ ghist, uint32_t nbv) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t h[RS_BINS];
    const uint32_t v = fd_xcd_remap(blockIdx.x, nbv);
    if (v >= nbv) return;
    uint32_t seg, cnt;
    uint64_t base;
    rs_seg_lookup(D, v, &seg, &base, &cnt);
    for (int k = threadIdx.x; k < RS_BINS; k += THREADS) h[k] = 0;
    __syncthreads();
    if (cnt == TILE) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; ++k) {
            u32x4 x = __builtin_nontemporal_load(&k4[k * THREADS + threadIdx.x]);
            atomicAdd(&h[(x.x >> shift) & mask], 1u);
            atomicAdd(&h[(x.y >> shift) & mask], 1u);
            atomicAdd(&h[(x.z >> shift) & mask], 1u);
            atomicAdd(&h[(x.w >> shift) & mask], 1u);
        }
    } else {
        for (uint32_t k = threadIdx.x; k < cnt; k += THREADS) atomicAdd(&h[(keys[base + k] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < RS_BINS; k += THREADS) ghist[(uint64_t)v * RS_BINS + k] = h[k];
}
// per bucket: exclusive scan over its chunks of every digit's chunk sums + the bucket's digit totals.  grid (16, n_seg)
__global__ __launch_bounds__(1024) void k_rs_scan_chunks_seg(uint64_t *__restrict__ csum, const rs_seg_tab *__restrict__ T, uint64_t *__restrict__ tot) {
    __shared__ uint64_t part[64][16];
    const uint32_t b = blockIdx.y;
    const uint32_t cb0 = T->vt0[b] / RS_SCAN_CHUNK, n_chunks = T->vt0[b + 1] / RS_SCAN_CHUNK - cb0;
    const uint32_t dl = threadIdx.x & 15u, p = threadIdx.x >> 4, d = blockIdx.x * 16u + dl;
    const uint32_t per = (n_chunks + 63) / 64, c0 = p * per < n_chunks ? p * per : n_chunks, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    uint64_t s = 0;
#pragma unroll 4
    for (uint32_t c = c0; c < c1; ++c) s += csum[(uint64_t)(cb0 + c) * RS_BINS + d];
    part[p][dl] = s;
    __syncthreads();
    uint64_t run = 0;
    for (uint32_t q = 0; q < p; ++q) run += part[q][dl];
    if (p == 63) tot[(uint64_t)b * RS_BINS + d] = run + s;
    for (uint32_t c = c0; c < c1; ++c) { uint64_t x = csum[(uint64_t)(cb0 + c) * RS_BINS + d]; csum[(uint64_t)(cb0 + c) * RS_BINS + d] = run; run += x; }
}
// per bucket: digit totals -> absolute position of every digit's first key (bucket start + exclusive scan over the digits)
__global__ __launch_bounds__(RS_BINS) void k_rs_scan_tot_seg(uint64_t *__restrict__ tot, const rs_seg_tab *__restrict__ T, unsigned long long *__restrict__ overflow) {
    __shared__ uint64_t sm[17];
    const uint32_t b = blockIdx.x;
    uint64_t x = tot[(uint64_t)b * RS_BINS + threadIdx.x], t;
    if (overflow && x >= (1ull << 32)) *overflow = 1ull;       // k_rs_scan_apply keeps a digit's positions in 32 bits
    uint64_t ex = block_excl_scan_u64(x, sm, &t);
    tot[(uint64_t)b * RS_BINS + threadIdx.x] = ex + T->bstart[b];
}
template <int THREADS, int ITEMS, typename V>
__global__ __launch_bounds__(THREADS) void k_rs_scatter4_seg(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
                                                             V *__restrict__ vals_out, const rs_vtile *__restrict__ D, uint32_t shift, uint32_t mask,
                                                             const uint32_t *__restrict__ ghist, uint32_t nbv, const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    __shared__ __attribute__((aligned(8))) uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    __shared__ uint32_t s_cnt[WAVES][RS_BINS];
    __shared__ long long s_gofs[RS_BINS];
    __shared__ uint64_t sm[17];
    const uint32_t v = fd_xcd_remap(blockIdx.x, nbv);
    if (v >= nbv) return;
    uint32_t seg, cnt;
    uint64_t base;
    rs_seg_lookup(D, v, &seg, &base, &cnt);
    if (!cnt) return;
    if (cnt == TILE)
        rs_scatter4_body<THREADS, ITEMS, V, true, false>(keys_in, vals_in, keys_out, vals_out, 0, shift, mask, ghist + (uint64_t)v * RS_BINS, dbase + (uint64_t)seg * RS_BINS, base,
                                                         TILE, s_keys, s_vals, s_cnt, s_gofs, sm);
    else
        rs_scatter4_body<THREADS, ITEMS, V, false, false>(keys_in, vals_in, keys_out, vals_out, 0, shift, mask, ghist + (uint64_t)v * RS_BINS, dbase + (uint64_t)seg * RS_BINS, base,
                                                          cnt, s_keys, s_vals, s_cnt, s_gofs, sm);
}

// LSD radix sort, 8-bit digits.  FDGPU_SORT=classicN selects measured alternatives: 18 = 512x16-key tiles (default, fastest),
// 19 = 256x16, 20 = 512x8, 21 = 512x16 with packed 8-byte LDS staging, 30 = memory skeleton only (NOT a sort, see
// k_rs_copy_floor).  Earlier generations (LDS-atomic ranking, persistent software-pipelined scatter, 16-bit counters for a third
// workgroup per CU, decoupled-look-back onesweep, 3 x 10-bit digits) were slower and are documented in DESIGN.md §4 / §9.
static int g_rs_variant = 18;
void fd_rs_set_variant(int v) { g_rs_variant = v; }
uint32_t fd_rs_num_tiles(uint64_t n) { return (uint32_t)((n + 2048 - 1) / 2048); }  // upper bound over variants (workspace sizing)

template <int THREADS, int ITEMS, typename V, bool PACK = false>
static void rs_pass4(uint32_t *ki, V *vi, uint32_t *ko, V *vo, uint64_t n, uint32_t shift, uint32_t mask, uint32_t *ghist, uint64_t *tot,
                     hipStream_t st, fdgpu_ctx *tc) {
    uint32_t nb = (uint32_t)((n + THREADS * ITEMS - 1) / (THREADS * ITEMS));
    uint32_t grid = ((nb + 7u) / 8u) * 8u;
    {
        StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nb * RS_BINS * 4);
        hipLaunchKernelGGL((k_rs_hist<THREADS, ITEMS, true>), dim3(grid), dim3(THREADS), 0, st, ki, n, shift, mask, ghist, nb);
    }
    {
        StageTimer t(tc, "rs_scan", (uint64_t)nb * RS_BINS * 8);
        const uint32_t n_chunks = (nb + RS_SCAN_CHUNK - 1) / RS_SCAN_CHUNK;
        uint64_t *csum = tot + RS_BINS;   // [n_chunks][256] behind the 256 digit totals
        hipLaunchKernelGGL(k_rs_scan_csum, dim3(n_chunks), dim3(RS_BINS), 0, st, ghist, nb, csum);
        hipLaunchKernelGGL(k_rs_scan_chunks, dim3(RS_BINS / 16), dim3(1024), 0, st, csum, n_chunks, tot);
        hipLaunchKernelGGL(k_rs_scan_apply, dim3(n_chunks), dim3(RS_BINS), 0, st, ghist, nb, csum);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        hipLaunchKernelGGL((k_rs_scatter4<THREADS, ITEMS, V, PACK>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
    }
}

template <typename V>
static int radix_sort_pairs_t(uint32_t *keys_a, V *vals_a, uint32_t *keys_b, V *vals_b, uint64_t n, int key_bits, uint32_t *ghist,
                              uint64_t *tot, hipStream_t st, fdgpu_ctx *tc) {
    if (n == 0) return 0;
    int cur = 0;
    for (int shift = 0; shift < key_bits; shift += 8) {
        int bits = key_bits - shift < 8 ? key_bits - shift : 8;
        uint32_t mask = (uint32_t)((1ull << bits) - 1ull);
        uint32_t *ki = cur ? keys_b : keys_a, *ko = cur ? keys_a : keys_b;
        V *vi = cur ? vals_b : vals_a, *vo = cur ? vals_a : vals_b;
        switch (g_rs_variant) {
            case 19: rs_pass4<256, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 20: rs_pass4<512, 8, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 21: rs_pass4<512, 16, V, true>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 30: {   // memory floor of the scatter (see k_rs_copy_floor); the result is NOT sorted
                uint32_t nb = (uint32_t)((n + 8191) / 8192), grid = ((nb + 7u) / 8u) * 8u;
                StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
                hipLaunchKernelGGL((k_rs_copy_floor<512, 16, V>), dim3(grid), dim3(512), 0, st, ki, vi, ko, vo, n, nb);
                break;
            }
            default: rs_pass4<512, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
        }
        cur ^= 1;
    }
    return cur;
}
// workspace of the segmented sort for n keys in n_seg buckets: virtual tiles (rows of the histogram table), u64 words behind `tot`
uint32_t fd_rs_seg_num_tiles(uint64_t n, uint32_t n_seg) { return (uint32_t)(n / 8192 + 2) + n_seg * (RS_SCAN_CHUNK + 1); }
uint64_t fd_rs_seg_tot_words(uint64_t n, uint32_t n_seg) { return (uint64_t)n_seg * RS_BINS + (uint64_t)(fd_rs_seg_num_tiles(n, n_seg) / RS_SCAN_CHUNK + 2) * RS_BINS; }
size_t fd_rs_seg_tab_bytes(uint64_t n, uint32_t n_seg) { return ((sizeof(rs_seg_tab) + 15) & ~(size_t)15) + (size_t)fd_rs_seg_num_tiles(n, n_seg) * sizeof(rs_vtile); }
// Stable sort of every bucket [seg_off[b * stride], seg_off[(b + 1) * stride]) by key bits [shift0, shift0 + 8 * passes): 6-byte elements.
// seg_off / seg_tab are device memory; nothing is synchronised.  Returns the buffer (0 = a, 1 = b) that holds the result.
int fd_radix_sort_pairs16_seg(uint32_t *keys_a, uint16_t *vals_a, uint32_t *keys_b, uint16_t *vals_b, uint64_t n, const uint64_t *seg_off, uint64_t stride,
                              uint32_t n_seg, int shift0, int passes, uint32_t *ghist, uint64_t *tot, void *seg_tab, hipStream_t st, fdgpu_ctx *tc,
                              unsigned long long *overflow) {
    if (n == 0 || n_seg == 0 || n_seg > RS_MAX_SEG) return 0;
    constexpr int THREADS = 512, ITEMS = 16;
    rs_seg_tab *T = (rs_seg_tab *)seg_tab;
    const uint32_t nbv = fd_rs_seg_num_tiles(n, n_seg), grid = ((nbv + 7u) / 8u) * 8u, n_chunks = (nbv + RS_SCAN_CHUNK - 1) / RS_SCAN_CHUNK;
    uint64_t *csum = tot + (uint64_t)n_seg * RS_BINS;
    rs_vtile *D = (rs_vtile *)((uint8_t *)seg_tab + ((sizeof(rs_seg_tab) + 15) & ~(size_t)15));
    hipLaunchKernelGGL(k_rs_seg_tiles, dim3(1), dim3(64), 0, st, seg_off, stride, n_seg, (uint32_t)(THREADS * ITEMS), T);
    hipLaunchKernelGGL(k_rs_seg_desc, dim3((nbv + 255) / 256), dim3(256), 0, st, T, (uint32_t)(THREADS * ITEMS), nbv, D);
    int cur = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const uint32_t shift = (uint32_t)(shift0 + 8 * pass), mask = 255u;
        uint32_t *ki = cur ? keys_b : keys_a, *ko = cur ? keys_a : keys_b;
        uint16_t *vi = cur ? vals_b : vals_a, *vo = cur ? vals_a : vals_b;
        {
            StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nbv * RS_BINS * 4);
            hipLaunchKernelGGL((k_rs_hist_seg<THREADS, ITEMS>), dim3(grid), dim3(THREADS), 0, st, ki, D, shift, mask, ghist, nbv);
        }
        {
            StageTimer t(tc, "rs_scan", (uint64_t)nbv * RS_BINS * 8);
            hipLaunchKernelGGL(k_rs_scan_csum, dim3(n_chunks), dim3(RS_BINS), 0, st, ghist, nbv, csum);
            hipLaunchKernelGGL(k_rs_scan_chunks_seg, dim3(RS_BINS / 16, n_seg), dim3(1024), 0, st, csum, T, tot);
            hipLaunchKernelGGL(k_rs_scan_apply, dim3(n_chunks), dim3(RS_BINS), 0, st, ghist, nbv, csum);
            hipLaunchKernelGGL(k_rs_scan_tot_seg, dim3(n_seg), dim3(RS_BINS), 0, st, tot, T, overflow);
        }
        {
            StageTimer t(tc, "rs_scatter", n * 12);
            hipLaunchKernelGGL((k_rs_scatter4_seg<THREADS, ITEMS, uint16_t>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, D, shift, mask, ghist, nbv, tot);
        }
        cur ^= 1;
    }
    return cur;
}
int fd_radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint64_t n, int key_bits,
                        uint32_t *ghist, uint64_t *tot, hipStream_t st, fdgpu_ctx *tc) {
    return radix_sort_pairs_t<uint32_t>(keys_a, vals_a, keys_b, vals_b, n, key_bits, ghist, tot, st, tc);
}
// 6-byte elements: 32-bit keys with 16-bit payload (index build with <= 2^18 structures per shard)
int fd_radix_sort_pairs16(uint32_t *keys_a, uint16_t *vals_a, uint32_t *keys_b, uint16_t *vals_b, uint64_t n, int key_bits,
                          uint32_t *ghist, uint64_t *tot, hipStream_t st, fdgpu_ctx *tc) {
    return radix_sort_pairs_t<uint16_t>(keys_a, vals_a, keys_b, vals_b, n, key_bits, ghist, tot, st, tc);
}
