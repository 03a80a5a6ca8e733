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

// tile histogram -> ghist[digit * nb + tile]
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
    for (int k = threadIdx.x; k < RS_BINS; k += THREADS) ghist[(uint64_t)k * nb + tile] = h[k];
}

// one workgroup per digit: exclusive scan of its row of nb tile counts (in place), row total -> tot[d];
// 8 consecutive counts per thread and iteration
__global__ __launch_bounds__(1024) void k_rs_scan_rows(uint32_t *__restrict__ ghist, uint32_t nb, uint64_t *__restrict__ tot) {
    __shared__ uint64_t sm[17];
    uint32_t *row = ghist + (uint64_t)blockIdx.x * nb;
    uint64_t carry = 0;
    for (uint32_t b = 0; b < nb; b += 1024 * 8) {
        uint32_t i0 = b + threadIdx.x * 8;
        uint32_t v[8];
        uint64_t s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k) < nb ? row[i0 + k] : 0u; s += v[k]; }
        uint64_t t;
        uint64_t ex = carry + block_excl_scan_u64(s, sm, &t);
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i0 + k < nb) row[i0 + k] = (uint32_t)ex; ex += v[k]; }
        carry += t;
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}
__global__ __launch_bounds__(RS_BINS) void k_rs_scan_tot(uint64_t *__restrict__ tot) {
    __shared__ uint64_t sm[17];
    uint64_t v = tot[threadIdx.x], t;
    uint64_t ex = block_excl_scan_u64(v, sm, &t);
    tot[threadIdx.x] = ex;
}

// stable scatter of one tile. Wave w owns a contiguous sub-tile and walks it in 64-key chunks
// (chunk c, lane l -> element c*64 + l) so that "earlier element" == "earlier chunk or lower lane".
// NT: streaming (non-temporal) loads of the input so that the once-read input does not evict the dirty partial
// output lines from the XCD's L2 before the neighbouring tile completes them
template <int THREADS, int ITEMS, bool XCD, typename V, bool NT = false>
__global__ __launch_bounds__(THREADS) void k_rs_scatter(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                        uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n,
                                                        uint32_t shift, uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb,
                                                        const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    __shared__ uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    __shared__ uint32_t s_cnt[WAVES][RS_BINS];  // per-wave digit counts, then running local positions
    __shared__ long long s_gofs[RS_BINS];       // global position = local position + s_gofs[digit]
    __shared__ uint64_t sm[17];

    const uint32_t tile = XCD ? fd_xcd_remap(blockIdx.x, nb) : blockIdx.x;
    if (tile >= nb) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
    const uint32_t n_tile = (uint32_t)((n - tile_base) < TILE ? (n - tile_base) : TILE);

    for (int k = tid; k < WAVES * RS_BINS; k += THREADS) (&s_cnt[0][0])[k] = 0;
    __syncthreads();

    uint32_t key[ITEMS];
    V val[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        if (NT) {
            key[c] = ok ? __builtin_nontemporal_load(&keys_in[idx]) : 0xffffffffu;
            val[c] = ok ? __builtin_nontemporal_load(&vals_in[idx]) : (V)0;
        } else {
            key[c] = ok ? keys_in[idx] : 0xffffffffu;
            val[c] = ok ? vals_in[idx] : (V)0;
        }
        if (ok) atomicAdd(&s_cnt[wid][(key[c] >> shift) & mask], 1u);
    }
    __syncthreads();
    {
        uint32_t my_total = 0;
        if (tid < RS_BINS) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) my_total += s_cnt[w][tid];
        }
        uint64_t tot;
        uint32_t dstart = (uint32_t)block_excl_scan_u64(tid < RS_BINS ? (uint64_t)my_total : 0ull, sm, &tot);
        if (tid < RS_BINS) {
            uint32_t run = dstart;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
            s_gofs[tid] = (long long)(dbase[tid] + ghist[(uint64_t)tid * nb + tile]) - (long long)dstart;
        }
    }
    __syncthreads();
    // in-wave stable ranks by ballot multi-split, running positions in s_cnt[wid][*]
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        uint32_t d = (key[c] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = fd_mbcnt(peers);               // peers in lower lanes
        uint32_t pcount = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (ok) pos = s_cnt[wid][d] + rank;
        // the wave's LDS ops execute in order: every read above precedes the update below
        if (ok && rank == pcount - 1) s_cnt[wid][d] = pos + 1;
        if (ok) { s_keys[pos] = key[c]; s_vals[pos] = val[c]; }
    }
    __syncthreads();
    // digit-contiguous in LDS -> runs in global memory
    for (uint32_t k = tid; k < n_tile; k += THREADS) {
        uint32_t kk = s_keys[k];
        long long g = (long long)k + s_gofs[(kk >> shift) & mask];
        keys_out[g] = kk;
        vals_out[g] = s_vals[k];
    }
}

// Variant without LDS atomics: the ballot multi-split runs FIRST and yields each element's running rank among the
// wave's keys of the same digit (leader lane keeps the per-wave digit counter), the cross-wave digit starts are
// scanned afterwards and added when the element is placed.
template <int THREADS, int ITEMS, bool XCD, typename V>
__global__ __launch_bounds__(THREADS) void k_rs_scatter2(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n,
                                                         uint32_t shift, uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb,
                                                         const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    __shared__ uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    __shared__ uint32_t s_cnt[WAVES][RS_BINS];
    __shared__ long long s_gofs[RS_BINS];
    __shared__ uint64_t sm[17];

    const uint32_t tile = XCD ? fd_xcd_remap(blockIdx.x, nb) : blockIdx.x;
    if (tile >= nb) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
    const uint32_t n_tile = (uint32_t)((n - tile_base) < TILE ? (n - tile_base) : TILE);

    for (int k = tid; k < WAVES * RS_BINS; k += THREADS) (&s_cnt[0][0])[k] = 0;
    uint32_t key[ITEMS];
    V val[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        key[c] = ok ? keys_in[idx] : 0xffffffffu;
        val[c] = ok ? vals_in[idx] : (V)0;
    }
    __syncthreads();
    uint16_t rnk[ITEMS];
    uint32_t *cnt = s_cnt[wid];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        uint32_t d = (key[c] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = fd_mbcnt(peers);
        uint32_t pcount = (uint32_t)__popcll(peers);
        uint32_t base = ok ? cnt[d] : 0u;
        // in-order LDS within the wave: all reads above precede the leader's update
        if (ok && rank == pcount - 1) cnt[d] = base + pcount;
        rnk[c] = (uint16_t)(base + rank);
    }
    __syncthreads();
    {
        uint32_t my_total = 0;
        if (tid < RS_BINS) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) my_total += s_cnt[w][tid];
        }
        uint64_t tot;
        uint32_t dstart = (uint32_t)block_excl_scan_u64(tid < RS_BINS ? (uint64_t)my_total : 0ull, sm, &tot);
        if (tid < RS_BINS) {
            uint32_t run = dstart;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
            s_gofs[tid] = (long long)(dbase[tid] + ghist[(uint64_t)tid * nb + tile]) - (long long)dstart;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        if (idx < n) {
            uint32_t pos = cnt[(key[c] >> shift) & mask] + rnk[c];
            s_keys[pos] = key[c];
            s_vals[pos] = val[c];
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < n_tile; k += THREADS) {
        uint32_t kk = s_keys[k];
        long long g = (long long)k + s_gofs[(kk >> shift) & mask];
        keys_out[g] = kk;
        vals_out[g] = s_vals[k];
    }
}

// As k_rs_scatter2 with 16-bit per-wave counters and the global digit offsets aliased onto the counter array once the
// placement is done: 52.1 KiB of LDS per 512x16 tile -> three workgroups per CU instead of two.
template <int THREADS, int ITEMS, typename V>
__global__ __launch_bounds__(THREADS) void k_rs_scatter3(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n,
                                                         uint32_t shift, uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb,
                                                         const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    static_assert(TILE <= 65535 && WAVES * RS_BINS * 2 >= RS_BINS * 8, "u16 counters / gofs alias");
    __shared__ uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    __shared__ __attribute__((aligned(8))) uint16_t s_cnt[WAVES][RS_BINS];
    __shared__ uint64_t sm[17];
    long long *s_gofs = reinterpret_cast<long long *>(&s_cnt[0][0]);

    const uint32_t tile = fd_xcd_remap(blockIdx.x, nb);
    if (tile >= nb) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
    const uint32_t n_tile = (uint32_t)((n - tile_base) < TILE ? (n - tile_base) : TILE);

    for (int k = tid; k < WAVES * RS_BINS / 2; k += THREADS) reinterpret_cast<uint32_t *>(&s_cnt[0][0])[k] = 0;
    uint32_t key[ITEMS];
    V val[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        key[c] = ok ? keys_in[idx] : 0xffffffffu;
        val[c] = ok ? vals_in[idx] : (V)0;
    }
    long long gbase = 0;
    if (tid < RS_BINS) gbase = (long long)(dbase[tid] + ghist[(uint64_t)tid * nb + tile]);
    __syncthreads();
    uint16_t rnk[ITEMS];
    uint16_t *cnt = s_cnt[wid];
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        uint32_t d = (key[c] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = fd_mbcnt(peers);
        uint32_t pcount = (uint32_t)__popcll(peers);
        uint32_t base = ok ? cnt[d] : 0u;
        if (ok && rank == pcount - 1) cnt[d] = (uint16_t)(base + pcount);
        rnk[c] = (uint16_t)(base + rank);
    }
    __syncthreads();
    uint32_t dstart = 0;
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
            for (int w = 0; w < WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = (uint16_t)run; run += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        if (idx < n) {
            uint32_t pos = (uint32_t)cnt[(key[c] >> shift) & mask] + rnk[c];
            s_keys[pos] = key[c];
            s_vals[pos] = val[c];
        }
    }
    __syncthreads();
    if (tid < RS_BINS) s_gofs[tid] = gbase - (long long)dstart;
    __syncthreads();
    for (uint32_t k = tid; k < n_tile; k += THREADS) {
        uint32_t kk = s_keys[k];
        long long g = (long long)k + s_gofs[(kk >> shift) & mask];
        keys_out[g] = kk;
        vals_out[g] = s_vals[k];
    }
}

// VALU-lean form of k_rs_scatter2 (the scatter is VALU-issue bound: ~130 VALU per 64-key chunk at 4 cycles each):
// the per-bit peer mask update is ONE v_bitop3_b32 per 32-bit half (peers & ~(ballot ^ bitmask), truth table 0x90),
// full tiles run without any bounds predicate, ranks stay in 32-bit registers.
template <int THREADS, int ITEMS, typename V, bool FULL, bool PACK = false>
__device__ __forceinline__ void rs_scatter4_body(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                 uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n, uint32_t shift,
                                                 uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb, const uint64_t *__restrict__ dbase,
                                                 uint32_t tile, uint32_t *s_keys, V *s_vals, uint32_t (*s_cnt)[RS_BINS], long long *s_gofs,
                                                 uint64_t *sm) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)tile * TILE;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
    const uint32_t n_tile = FULL ? TILE : (uint32_t)(n - tile_base);
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
    if (tid < RS_BINS) gbase = (long long)(dbase[tid] + ghist[(uint64_t)tid * nb + tile]);
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
    if ((uint64_t)(tile + 1) * TILE <= n)
        rs_scatter4_body<THREADS, ITEMS, V, true, PACK>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, ghist, nb, dbase, tile, s_keys, s_vals, s_cnt, s_gofs, sm);
    else
        rs_scatter4_body<THREADS, ITEMS, V, false, PACK>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, ghist, nb, dbase, tile, s_keys, s_vals, s_cnt, s_gofs, sm);
}

// ------------------------------------------------------------------------ onesweep variant
// One kernel per digit: the global digit histograms of all passes come from ONE upfront read of the
// keys (the key multiset does not change between passes), and the per-tile digit offsets are
// resolved inside the scatter kernel by decoupled look-back over per-(tile, digit) descriptors
// instead of a histogram pass + scan per digit.  Inter-workgroup visibility on gfx950 (per-XCD L2s are
// not coherent): a descriptor is ONE naturally aligned 8-byte word {value:62, state:2} written and
// read with relaxed agent-scope atomics (sc1 store / sc1 load), the "granule" form of
// MI355X_MICROARCH.md — no separate flag, so no ordering between two words is needed.  Tiles take
// their index from an atomic ticket so that every predecessor a tile waits for is already running.
#define RS_WAVES_UNUSED 0
#define OS_THREADS 512
#define OS_ITEMS 16
#define OS_TILE (OS_THREADS * OS_ITEMS)   // 8192 keys
#define OS_WAVES (OS_THREADS / 64)
#define OS_AGG 1ull
#define OS_INC 2ull

__global__ __launch_bounds__(256) void k_os_hist(const uint32_t *__restrict__ keys, uint64_t n, int key_bits,
                                                 unsigned long long *__restrict__ ghist /*[4][256]*/) {
    __shared__ uint32_t h[4][RS_BINS];
    for (int k = threadIdx.x; k < 4 * RS_BINS; k += blockDim.x) (&h[0][0])[k] = 0;
    __syncthreads();
    const int npass = (key_bits + 7) / 8;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        uint32_t k = keys[idx];
        for (int p = 0; p < npass; ++p) {
            int bits = key_bits - 8 * p < 8 ? key_bits - 8 * p : 8;
            atomicAdd(&h[p][(k >> (8 * p)) & ((1u << bits) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 4 * RS_BINS; k += blockDim.x) {
        uint32_t v = (&h[0][0])[k];
        if (v) atomicAdd(&ghist[k], (unsigned long long)v);
    }
}
// exclusive scan of each pass's 256 global counts (in place)
__global__ __launch_bounds__(RS_BINS) void k_os_scan(unsigned long long *__restrict__ ghist) {
    __shared__ uint64_t sm[17];
    unsigned long long *row = ghist + (uint64_t)blockIdx.x * RS_BINS;
    uint64_t v = row[threadIdx.x], t;
    uint64_t ex = block_excl_scan_u64(v, sm, &t);
    row[threadIdx.x] = ex;
}

__global__ __launch_bounds__(OS_THREADS) void k_os_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                           uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint64_t n,
                                                           uint32_t shift, uint32_t mask, const unsigned long long *__restrict__ dbase /*[256]*/,
                                                           unsigned long long *__restrict__ desc /*[tiles][256]*/, uint32_t *__restrict__ ticket) {
    __shared__ uint32_t s_keys[OS_TILE];
    __shared__ uint32_t s_vals[OS_TILE];
    __shared__ uint32_t s_cnt[OS_WAVES][RS_BINS];
    __shared__ long long s_gofs[RS_BINS];
    __shared__ uint64_t sm[17];
    __shared__ uint32_t s_tile;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int k = tid; k < OS_WAVES * RS_BINS; k += OS_THREADS) (&s_cnt[0][0])[k] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t tile_base = (uint64_t)tile * OS_TILE;
    const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * OS_ITEMS);
    const uint32_t n_tile = (uint32_t)((n - tile_base) < OS_TILE ? (n - tile_base) : OS_TILE);

    uint32_t key[OS_ITEMS], val[OS_ITEMS];
#pragma unroll
    for (int c = 0; c < OS_ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        key[c] = ok ? keys_in[idx] : 0xffffffffu;
        val[c] = ok ? vals_in[idx] : 0u;
        if (ok) atomicAdd(&s_cnt[wid][(key[c] >> shift) & mask], 1u);
    }
    __syncthreads();
    uint32_t my_total = 0, dstart = 0;
    if (tid < RS_BINS) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < OS_WAVES; ++w) run += s_cnt[w][tid];
        my_total = run;
        // publish the tile aggregate (tile 0: already inclusive)
        unsigned long long d0 = ((unsigned long long)my_total << 2) | (tile == 0 ? OS_INC : OS_AGG);
        __hip_atomic_store(&desc[(uint64_t)tile * RS_BINS + tid], d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        uint64_t tot;
        uint64_t ex = block_excl_scan_u64(tid < RS_BINS ? (uint64_t)my_total : 0ull, sm, &tot);
        dstart = (uint32_t)ex;
    }
    if (tid < RS_BINS) {
        // per-wave local bases (running positions during ranking)
        uint32_t run = dstart;
#pragma unroll
        for (int w = 0; w < OS_WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
        // decoupled look-back for digit `tid`
        unsigned long long excl = 0;
        if (tile > 0) {
            int64_t t = (int64_t)tile - 1;
            while (true) {
                unsigned long long d = __hip_atomic_load(&desc[(uint64_t)t * RS_BINS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long st = d & 3ull;
                if (st == 0ull) { __builtin_amdgcn_s_sleep(1); continue; }
                excl += d >> 2;
                if (st == OS_INC) break;
                --t;  // aggregate only: keep looking back (t cannot underflow: tile 0 is always inclusive)
            }
            unsigned long long d1 = ((excl + my_total) << 2) | OS_INC;
            __hip_atomic_store(&desc[(uint64_t)tile * RS_BINS + tid], d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gofs[tid] = (long long)(dbase[tid] + excl) - (long long)dstart;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < OS_ITEMS; ++c) {
        uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
        bool ok = idx < n;
        uint32_t d = (key[c] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = fd_mbcnt(peers);
        uint32_t pcount = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (ok) pos = s_cnt[wid][d] + rank;
        if (ok && rank == pcount - 1) s_cnt[wid][d] = pos + 1;
        if (ok) { s_keys[pos] = key[c]; s_vals[pos] = val[c]; }
    }
    __syncthreads();
    for (uint32_t k = tid; k < n_tile; k += OS_THREADS) {
        uint32_t kk = s_keys[k];
        long long g = (long long)k + s_gofs[(kk >> shift) & mask];
        keys_out[g] = kk;
        vals_out[g] = s_vals[k];
    }
}

uint32_t fd_os_num_tiles(uint64_t n) { return (uint32_t)((n + OS_TILE - 1) / OS_TILE); }
// workspace: desc u64[tiles*256], ghist u64[4*256], ticket u32[4]
int fd_onesweep_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint64_t n, int key_bits,
                           unsigned long long *desc, unsigned long long *ghist, uint32_t *ticket, hipStream_t st, fdgpu_ctx *tc) {
    if (n == 0) return 0;
    uint32_t nb = fd_os_num_tiles(n);
    (void)hipMemsetAsync(ghist, 0, 4 * RS_BINS * 8, st);
    (void)hipMemsetAsync(ticket, 0, 16, st);
    {
        StageTimer t(tc, "os_hist", n * 4);
        hipLaunchKernelGGL(k_os_hist, dim3(2048), dim3(256), 0, st, keys_a, n, key_bits, ghist);
        hipLaunchKernelGGL(k_os_scan, dim3(4), dim3(RS_BINS), 0, st, ghist);
    }
    int cur = 0, pass = 0;
    for (int shift = 0; shift < key_bits; shift += 8, ++pass) {
        int bits = key_bits - shift < 8 ? key_bits - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        uint32_t *ki = cur ? keys_b : keys_a, *vi = cur ? vals_b : vals_a;
        uint32_t *ko = cur ? keys_a : keys_b, *vo = cur ? vals_a : vals_b;
        StageTimer t(tc, "os_scatter", n * 16);
        (void)hipMemsetAsync(desc, 0, (size_t)nb * RS_BINS * 8, st);
        hipLaunchKernelGGL(k_os_scatter, dim3(nb), dim3(OS_THREADS), 0, st, ki, vi, ko, vo, n, (uint32_t)shift, mask,
                           ghist + (size_t)pass * RS_BINS, desc, ticket + pass);
        cur ^= 1;
    }
    return cur;
}

// persistent, software-pipelined scatter: a fixed grid (PERSIST_BLOCKS_PER_CU x 256 CUs) walks the tiles; the keys
// and values of the NEXT tile are requested (global loads in flight) before the current tile is ranked and
// written, so every CU always has reads outstanding instead of alternating load / LDS / store phases.
// Tile order is XCD-aware: the blocks of one XCD (b % 8) sweep one contiguous eighth of the tiles together, so
// the partial-line digit runs of neighbouring tiles merge in that XCD's L2.
#define PERSIST_BLOCKS_PER_CU 4
template <int THREADS, int ITEMS, typename V>
__global__ __launch_bounds__(THREADS) void k_rs_scatter_p(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                          uint32_t *__restrict__ keys_out, V *__restrict__ vals_out, uint64_t n,
                                                          uint32_t shift, uint32_t mask, const uint32_t *__restrict__ ghist, uint32_t nb,
                                                          const uint64_t *__restrict__ dbase) {
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    __shared__ uint32_t s_keys[TILE];
    __shared__ V s_vals[TILE];
    __shared__ uint32_t s_cnt[WAVES][RS_BINS];
    __shared__ long long s_gofs[RS_BINS];
    __shared__ uint64_t sm[17];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
    const uint32_t per = (nb + 7u) / 8u;
    const uint32_t t_lo = xcd * per, t_hi = (t_lo + per < nb) ? t_lo + per : nb;
    uint32_t tile = t_lo + j;
    if (tile >= t_hi) return;

    uint32_t key[ITEMS], nkey[ITEMS];
    V val[ITEMS], nval[ITEMS];
    auto load_tile = [&](uint32_t t, uint32_t *k, V *v) {
        const uint64_t wb = (uint64_t)t * TILE + (uint64_t)wid * (64 * ITEMS);
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            uint64_t idx = wb + (uint64_t)c * 64 + lane;
            bool ok = idx < n;
            k[c] = ok ? keys_in[idx] : 0xffffffffu;
            v[c] = ok ? vals_in[idx] : (V)0;
        }
    };
    load_tile(tile, key, val);
    for (;;) {
        const uint32_t next = tile + per_xcd_blocks;
        const bool has_next = next < t_hi;
        if (has_next) load_tile(next, nkey, nval);   // in flight while this tile is processed

        const uint64_t tile_base = (uint64_t)tile * TILE;
        const uint64_t wave_base = tile_base + (uint64_t)wid * (64 * ITEMS);
        const uint32_t n_tile = (uint32_t)((n - tile_base) < TILE ? (n - tile_base) : TILE);
        for (int k = tid; k < WAVES * RS_BINS; k += THREADS) (&s_cnt[0][0])[k] = 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
            if (idx < n) atomicAdd(&s_cnt[wid][(key[c] >> shift) & mask], 1u);
        }
        __syncthreads();
        {
            uint32_t my_total = 0;
            if (tid < RS_BINS) {
#pragma unroll
                for (int w = 0; w < WAVES; ++w) my_total += s_cnt[w][tid];
            }
            uint64_t tot;
            uint32_t dstart = (uint32_t)block_excl_scan_u64(tid < RS_BINS ? (uint64_t)my_total : 0ull, sm, &tot);
            if (tid < RS_BINS) {
                uint32_t run = dstart;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
                s_gofs[tid] = (long long)(dbase[tid] + ghist[(uint64_t)tid * nb + tile]) - (long long)dstart;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            uint64_t idx = wave_base + (uint64_t)c * 64 + lane;
            bool ok = idx < n;
            uint32_t d = (key[c] >> shift) & mask;
            uint64_t peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                uint64_t bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            uint32_t rank = fd_mbcnt(peers);
            uint32_t pcount = (uint32_t)__popcll(peers);
            uint32_t pos = 0;
            if (ok) pos = s_cnt[wid][d] + rank;
            if (ok && rank == pcount - 1) s_cnt[wid][d] = pos + 1;
            if (ok) { s_keys[pos] = key[c]; s_vals[pos] = val[c]; }
        }
        __syncthreads();
        for (uint32_t k = tid; k < n_tile; k += THREADS) {
            uint32_t kk = s_keys[k];
            long long g = (long long)k + s_gofs[(kk >> shift) & mask];
            keys_out[g] = kk;
            vals_out[g] = s_vals[k];
        }
        if (!has_next) break;
        __syncthreads();   // LDS is reused by the next tile
        tile = next;
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) { key[c] = nkey[c]; val[c] = nval[c]; }
    }
}

// LSD variants: 0 = 256x16 tiles, 1 = 256x16 + XCD-aware tile order, 2 = 512x16, 3 = 512x16 + XCD-aware,
// 4 = 256x16 persistent software-pipelined scatter (XCD-aware)
static int g_rs_variant = 18;
void fd_rs_set_variant(int v) { g_rs_variant = v; }
static inline uint32_t rs_tile(int v) { return (v >= 2 ? 512u : 256u) * 16u; }
uint32_t fd_rs_num_tiles(uint64_t n) { return (uint32_t)((n + 2048 - 1) / 2048); }  // upper bound over variants (workspace sizing)

template <int THREADS, int ITEMS, bool XCD, typename V, bool NT = false>
static void rs_pass(uint32_t *ki, V *vi, uint32_t *ko, V *vo, uint64_t n, uint32_t shift, uint32_t mask, uint32_t *ghist,
                    uint64_t *tot, hipStream_t st, fdgpu_ctx *tc) {
    uint32_t nb = (uint32_t)((n + THREADS * ITEMS - 1) / (THREADS * ITEMS));
    uint32_t grid = XCD ? ((nb + 7u) / 8u) * 8u : nb;
    {
        StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nb * RS_BINS * 4);
        hipLaunchKernelGGL((k_rs_hist<THREADS, ITEMS, XCD>), dim3(grid), dim3(THREADS), 0, st, ki, n, shift, mask, ghist, nb);
    }
    {
        StageTimer t(tc, "rs_scan", (uint64_t)nb * RS_BINS * 8);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, st, ghist, nb, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        hipLaunchKernelGGL((k_rs_scatter<THREADS, ITEMS, XCD, V, NT>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
    }
}

template <int THREADS, int ITEMS, typename V>
static void rs_pass2(uint32_t *ki, V *vi, uint32_t *ko, V *vo, uint64_t n, uint32_t shift, uint32_t mask, uint32_t *ghist, uint64_t *tot,
                     hipStream_t st, fdgpu_ctx *tc) {
    uint32_t nb = (uint32_t)((n + THREADS * ITEMS - 1) / (THREADS * ITEMS));
    uint32_t grid = ((nb + 7u) / 8u) * 8u;
    {
        StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nb * RS_BINS * 4);
        hipLaunchKernelGGL((k_rs_hist<THREADS, ITEMS, true>), dim3(grid), dim3(THREADS), 0, st, ki, n, shift, mask, ghist, nb);
    }
    {
        StageTimer t(tc, "rs_scan", (uint64_t)nb * RS_BINS * 8);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, st, ghist, nb, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        hipLaunchKernelGGL((k_rs_scatter2<THREADS, ITEMS, true, V>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
    }
}

template <int THREADS, int ITEMS, typename V>
static void rs_pass3(uint32_t *ki, V *vi, uint32_t *ko, V *vo, uint64_t n, uint32_t shift, uint32_t mask, uint32_t *ghist, uint64_t *tot,
                     hipStream_t st, fdgpu_ctx *tc) {
    uint32_t nb = (uint32_t)((n + THREADS * ITEMS - 1) / (THREADS * ITEMS));
    uint32_t grid = ((nb + 7u) / 8u) * 8u;
    {
        StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nb * RS_BINS * 4);
        hipLaunchKernelGGL((k_rs_hist<THREADS, ITEMS, true>), dim3(grid), dim3(THREADS), 0, st, ki, n, shift, mask, ghist, nb);
    }
    {
        StageTimer t(tc, "rs_scan", (uint64_t)nb * RS_BINS * 8);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, st, ghist, nb, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        hipLaunchKernelGGL((k_rs_scatter3<THREADS, ITEMS, V>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
    }
}

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
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, st, ghist, nb, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        hipLaunchKernelGGL((k_rs_scatter4<THREADS, ITEMS, V, PACK>), dim3(grid), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
    }
}

template <int THREADS, int ITEMS, typename V>
static void rs_pass_p(uint32_t *ki, V *vi, uint32_t *ko, V *vo, uint64_t n, uint32_t shift, uint32_t mask, uint32_t *ghist, uint64_t *tot,
                      hipStream_t st, fdgpu_ctx *tc) {
    uint32_t nb = (uint32_t)((n + THREADS * ITEMS - 1) / (THREADS * ITEMS));
    uint32_t grid = ((nb + 7u) / 8u) * 8u;
    {
        StageTimer t(tc, "rs_hist", n * 4 + (uint64_t)nb * RS_BINS * 4);
        hipLaunchKernelGGL((k_rs_hist<THREADS, ITEMS, true>), dim3(grid), dim3(THREADS), 0, st, ki, n, shift, mask, ghist, nb);
    }
    {
        StageTimer t(tc, "rs_scan", (uint64_t)nb * RS_BINS * 8);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(RS_BINS), dim3(1024), 0, st, ghist, nb, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(RS_BINS), 0, st, tot);
    }
    {
        StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
        uint32_t pg = 256u * PERSIST_BLOCKS_PER_CU;   // multiple of 8
        hipLaunchKernelGGL((k_rs_scatter_p<THREADS, ITEMS, V>), dim3(pg), dim3(THREADS), 0, st, ki, vi, ko, vo, n, shift, mask, ghist, nb, tot);
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
            case 10: rs_pass2<256, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 11: rs_pass2<512, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 16: rs_pass3<512, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 17: rs_pass3<256, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 18: rs_pass4<512, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 19: rs_pass4<256, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 20: rs_pass4<512, 8, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 21: rs_pass4<512, 16, V, true>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 22: rs_pass4<256, 16, V, true>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 30: {   // memory floor of the scatter (see k_rs_copy_floor); the result is NOT sorted
                uint32_t nb = (uint32_t)((n + 8191) / 8192), grid = ((nb + 7u) / 8u) * 8u;
                StageTimer t(tc, "rs_scatter", n * (8 + 2 * sizeof(V)));
                hipLaunchKernelGGL((k_rs_copy_floor<512, 16, V>), dim3(grid), dim3(512), 0, st, ki, vi, ko, vo, n, nb);
                break;
            }
            case 12: rs_pass2<1024, 8, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 13: rs_pass2<1024, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 14: rs_pass2<512, 24, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 15: rs_pass2<256, 32, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 4: rs_pass_p<256, 16, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 5: rs_pass<256, 16, true, V, true>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 6: rs_pass<1024, 4, true, V, false>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 7: rs_pass<512, 8, true, V, false>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 8: rs_pass<256, 8, true, V, false>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 9: rs_pass<1024, 8, true, V, false>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 0: rs_pass<256, 16, false, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 1: rs_pass<256, 16, true, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            case 2: rs_pass<512, 16, false, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
            default: rs_pass<512, 16, true, V>(ki, vi, ko, vo, n, (uint32_t)shift, mask, ghist, tot, st, tc); break;
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
