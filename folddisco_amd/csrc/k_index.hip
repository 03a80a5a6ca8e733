// k_index.hip — posting-list encoding from the sorted (hash, id) stream.
//
// Replaces count_single_entry / allocate_entries / add_single_entry /
// wrapup_offset_and_save_entries / prune_to_sparse of the reference
// (src/index/indextable.rs:88-105, 204-237, 171-202, 239-295, codec :397-418):
//   * duplicates (same hash, same id — the reference's per-structure dedup, mod.rs:343-345) drop out,
//   * first id of a list is absolute, the rest are deltas, each LEB128 (7-bit groups, MSB = continue,
//     value 0 -> one 0x00 byte),
//   * lists are laid out in ascending hash order; only non-empty hashes are kept (sparse form):
//     hashes[H], offsets[H+1] with offsets[k] = first byte of list k, offsets[H] = total bytes.
// Two streaming passes over the sorted pairs (sizes, then write), HBM-bound.
#include "fd_device.h"

#define ENC_THREADS 256
#define ENC_ITEMS 8
#define ENC_TILE (ENC_THREADS * ENC_ITEMS)

__device__ __forceinline__ uint32_t varint_len(uint32_t v) {
    // 1 + ilog2(v)/7 for v > 0, 1 for v == 0 (indextable.rs:93-99)
    return v == 0 ? 1u : 1u + (31u - (uint32_t)__clz(v)) / 7u;
}

struct enc_item { uint32_t len; uint32_t head; uint32_t delta; uint32_t hash; };

// two element encodings of the sorted stream:
//   V = uint32_t : keys[p] = hash,                     vals[p] = structure id
//   V = uint16_t : keys[p] = hash << 2 | (local >> 16), vals[p] = local & 0xffff, id = first_id + local  (6-byte elements)
template <typename V> struct enc_codec;
template <> struct enc_codec<uint32_t> {
    static __device__ __forceinline__ uint32_t hash(uint32_t k) { return k; }
    static __device__ __forceinline__ uint32_t id(uint32_t, uint32_t v, uint32_t) { return v; }
};
template <> struct enc_codec<uint16_t> {
    static __device__ __forceinline__ uint32_t hash(uint32_t k) { return k >> 2; }
    static __device__ __forceinline__ uint32_t id(uint32_t k, uint16_t v, uint32_t first_id) { return first_id + (((k & 3u) << 16) | v); }
};

template <typename V>
__device__ __forceinline__ enc_item enc_classify(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t p, uint32_t first_id) {
    enc_item it;
    uint32_t k = enc_codec<V>::hash(keys[p]), id = enc_codec<V>::id(keys[p], ids[p], first_id);
    bool first = p == 0;
    uint32_t pk = first ? 0u : enc_codec<V>::hash(keys[p - 1]), pid = first ? 0u : enc_codec<V>::id(keys[p - 1], ids[p - 1], first_id);
    it.hash = k;
    bool head = first || pk != k;
    bool dup = !head && pid == id;
    it.head = head ? 1u : 0u;
    it.delta = head ? id : id - pid;
    it.len = dup ? 0u : varint_len(it.delta);
    return it;
}

__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
    uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t t = __shfl_up(v, off, FD_WAVE);
        if ((int)lane >= off) v += t;
    }
    return v;
}
// packs (bytes << 32 | entries... ) no: two independent u64 scans share the shuffles poorly; bytes per
// tile < 2^16 and heads per tile < 2^12, so both fit one u32 pair packed in a u64: hi = bytes, lo = heads.
__device__ __forceinline__ uint64_t block_excl_scan_packed(uint64_t v, uint64_t *sm, uint64_t *total) {
    uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t inc = wave_incl_scan64(v);
    if (lane == 63) sm[wid] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < ENC_THREADS / 64; ++w) {
        uint64_t t = sm[w];
        if (w < wid) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return inc - v + base;
}

// pass 1: per-tile (bytes, heads, postings) sums
template <typename V>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_sizes(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t n, uint32_t first_id,
                                                           uint32_t *__restrict__ tile_bytes, uint32_t *__restrict__ tile_heads,
                                                           uint32_t *__restrict__ tile_posts) {
    __shared__ uint64_t sm[ENC_THREADS / 64];
    uint64_t base = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)threadIdx.x * ENC_ITEMS;
    uint32_t bytes = 0, heads = 0, posts = 0;
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) {
        uint64_t p = base + k;
        if (p < n) {
            enc_item it = enc_classify<V>(keys, ids, p, first_id);
            bytes += it.len;
            heads += it.head;
            posts += it.len ? 1u : 0u;
        }
    }
    uint64_t tot;
    block_excl_scan_packed(((uint64_t)bytes << 32) | heads, sm, &tot);
    // postings: separate reduction
    uint32_t pw = posts;
    for (int off = 32; off > 0; off >>= 1) pw += __shfl_down(pw, off, FD_WAVE);
    __shared__ uint32_t pp[ENC_THREADS / 64];
    if ((threadIdx.x & 63) == 0) pp[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_bytes[blockIdx.x] = (uint32_t)(tot >> 32);
        tile_heads[blockIdx.x] = (uint32_t)tot;
        uint32_t t = 0;
        for (int w = 0; w < ENC_THREADS / 64; ++w) t += pp[w];
        tile_posts[blockIdx.x] = t;
    }
}

// pass 2: write varint bytes, sparse hashes and list start offsets
template <typename V>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_write(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t n, uint32_t first_id,
                                                           const uint64_t *__restrict__ tile_byte_off, const uint64_t *__restrict__ tile_head_off,
                                                           uint8_t *__restrict__ value, uint32_t *__restrict__ hashes, uint64_t *__restrict__ offsets) {
    __shared__ uint64_t sm[ENC_THREADS / 64];
    uint64_t base = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)threadIdx.x * ENC_ITEMS;
    enc_item it[ENC_ITEMS];
    uint32_t bytes = 0, heads = 0;
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) {
        uint64_t p = base + k;
        if (p < n) it[k] = enc_classify<V>(keys, ids, p, first_id);
        else { it[k].len = 0; it[k].head = 0; it[k].delta = 0; it[k].hash = 0; }
        bytes += it[k].len;
        heads += it[k].head;
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_packed(((uint64_t)bytes << 32) | heads, sm, &tot);
    uint64_t boff = tile_byte_off[blockIdx.x] + (ex >> 32);
    uint64_t hoff = tile_head_off[blockIdx.x] + (uint32_t)ex;
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) {
        uint64_t p = base + k;
        if (p >= n) break;
        if (it[k].head) {
            hashes[hoff] = it[k].hash;
            offsets[hoff] = boff;
            ++hoff;
        }
        uint32_t v = it[k].delta;
        for (uint32_t b = 0; b < it[k].len; ++b) {
            uint32_t byte = v & 0x7fu;
            v >>= 7;
            value[boff++] = (uint8_t)(byte | (b + 1 < it[k].len ? 0x80u : 0u));
        }
    }
}

__global__ void k_set_u64(uint64_t *dst, uint64_t idx, const uint64_t *src) { dst[idx] = src[0]; }

uint32_t fd_enc_num_tiles(uint64_t n) { return (uint32_t)((n + ENC_TILE - 1) / ENC_TILE); }
void fd_launch_enc_sizes(const uint32_t *keys, const void *ids, bool ids16, uint32_t first_id, uint64_t n, uint32_t *tb, uint32_t *th, uint32_t *tp,
                         hipStream_t st) {
    if (!n) return;
    if (ids16) hipLaunchKernelGGL(k_enc_sizes<uint16_t>, dim3(fd_enc_num_tiles(n)), dim3(ENC_THREADS), 0, st, keys, (const uint16_t *)ids, n, first_id, tb, th, tp);
    else hipLaunchKernelGGL(k_enc_sizes<uint32_t>, dim3(fd_enc_num_tiles(n)), dim3(ENC_THREADS), 0, st, keys, (const uint32_t *)ids, n, first_id, tb, th, tp);
}
void fd_launch_enc_write(const uint32_t *keys, const void *ids, bool ids16, uint32_t first_id, uint64_t n, const uint64_t *tbo, const uint64_t *tho,
                         uint8_t *value, uint32_t *hashes, uint64_t *offsets, const uint64_t *total_bytes_dev, uint64_t H, hipStream_t st) {
    if (n) {
        if (ids16) hipLaunchKernelGGL(k_enc_write<uint16_t>, dim3(fd_enc_num_tiles(n)), dim3(ENC_THREADS), 0, st, keys, (const uint16_t *)ids, n, first_id, tbo, tho, value, hashes, offsets);
        else hipLaunchKernelGGL(k_enc_write<uint32_t>, dim3(fd_enc_num_tiles(n)), dim3(ENC_THREADS), 0, st, keys, (const uint32_t *)ids, n, first_id, tbo, tho, value, hashes, offsets);
    }
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, offsets, H, total_bytes_dev);
}
