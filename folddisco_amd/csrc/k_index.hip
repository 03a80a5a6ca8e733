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

// three element encodings of the sorted stream:
//   V = uint32_t      : keys[p] = hash,                      vals[p] = structure id
//   V = uint16_t      : keys[p] = hash << 2 | (local >> 16), vals[p] = local & 0xffff, id = first_id + local  (6-byte elements, 2^18 structures)
//   V = uint16_t, B24 : keys[p] = hash << 8 | (local >> 16)  — the MSD build's stream: the top six hash bits ARE the bucket the position p lies
//                       in (bucket b = [seg_off[b * S], seg_off[(b + 1) * S])), the key keeps the other 24 and eight more id bits (2^24 structures)
template <typename V> struct enc_codec;
template <> struct enc_codec<uint32_t> {
    static __device__ __forceinline__ uint32_t hash(uint32_t k) { return k; }
    static __device__ __forceinline__ uint32_t id(uint32_t, uint32_t v, uint32_t) { return v; }
};
template <> struct enc_codec<uint16_t> {
    static __device__ __forceinline__ uint32_t hash(uint32_t k) { return k >> 2; }
    static __device__ __forceinline__ uint32_t id(uint32_t k, uint16_t v, uint32_t first_id) { return first_id + (((k & 3u) << 16) | v); }
};

#define ENC_BUCKETS 40u
struct enc_b24 {              // B24: where the forty buckets start (bo[0..40], compact copy of seg_off[b * S]) and the tile's first / last bucket
    const uint64_t *bo;
    uint32_t b0, b1;
    __device__ __forceinline__ uint32_t at(uint64_t p) const { uint32_t b = b0; while (b < b1 && p >= bo[b + 1]) ++b; return b; }      // tiles on a boundary only
};
// per wave, no LDS and no barrier: lane l holds bo[l]; called AFTER the tile's key loads are in flight
__device__ __forceinline__ enc_b24 enc_b24_setup(const uint64_t *__restrict__ bo, uint64_t n) {
    const uint64_t t0 = (uint64_t)blockIdx.x * (256 * 8);
    const uint64_t t1 = (t0 + 256 * 8 < n ? t0 + 256 * 8 : n) - 1;
    const uint64_t tp = t0 ? t0 - 1 : 0;      // the predecessor of the tile's first element is classified too
    const uint32_t lane = threadIdx.x & 63;
    const bool in = lane >= 1 && lane <= ENC_BUCKETS;
    const uint64_t o = in ? bo[lane] : ~0ull;
    enc_b24 K;
    K.bo = bo;
    K.b0 = (uint32_t)__popcll(__ballot(in && o <= tp));      // buckets 1..40 that start at or before p: the bucket p lies in
    K.b1 = (uint32_t)__popcll(__ballot(in && o <= t1));
    if (K.b0 >= ENC_BUCKETS) K.b0 = ENC_BUCKETS - 1;
    if (K.b1 >= ENC_BUCKETS) K.b1 = ENC_BUCKETS - 1;
    return K;
}
__global__ void k_enc_bucket_starts(const uint64_t *__restrict__ seg_off, uint64_t S, uint64_t *__restrict__ bo) {
    if (threadIdx.x <= ENC_BUCKETS) bo[threadIdx.x] = seg_off[(uint64_t)threadIdx.x * S];
}

// Loads the ENC_ITEMS (= 8) consecutive elements owned by this thread with 16-byte loads (full tiles) and
// classifies them; the predecessor of the thread's first element comes from the neighbouring lane (shuffle)
// or, for lane 0 of a wave, from memory.
template <typename V, bool B24>
__device__ __forceinline__ void enc_load_classify(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t n, uint64_t base,
                                                  uint32_t first_id, enc_item *it, const uint64_t *__restrict__ bo, uint32_t *pred_id = nullptr) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    uint32_t k[ENC_ITEMS], v[ENC_ITEMS];
    const bool full = base + ENC_ITEMS <= n;
    if (full) {
        const u32x4 *kp = reinterpret_cast<const u32x4 *>(keys + base);
        u32x4 a = __builtin_nontemporal_load(&kp[0]), b = __builtin_nontemporal_load(&kp[1]);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
        if (sizeof(V) == 2) {
            u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ids + base));
            v[0] = w.x & 0xffffu; v[1] = w.x >> 16; v[2] = w.y & 0xffffu; v[3] = w.y >> 16;
            v[4] = w.z & 0xffffu; v[5] = w.z >> 16; v[6] = w.w & 0xffffu; v[7] = w.w >> 16;
        } else {
            const u32x4 *vp = reinterpret_cast<const u32x4 *>(ids + base);
            u32x4 c = __builtin_nontemporal_load(&vp[0]), d = __builtin_nontemporal_load(&vp[1]);
            v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ENC_ITEMS; ++j) {
            uint64_t p = base + j;
            k[j] = p < n ? keys[p] : 0u;
            v[j] = p < n ? (uint32_t)ids[p] : 0u;
        }
    }
    // predecessor of element 0
    uint32_t pk = __shfl_up(k[ENC_ITEMS - 1], 1, FD_WAVE), pv = __shfl_up(v[ENC_ITEMS - 1], 1, FD_WAVE);
    if ((threadIdx.x & 63) == 0 && base > 0 && base < n) { pk = keys[base - 1]; pv = (uint32_t)ids[base - 1]; }
    uint32_t ph, pid;
    uint32_t hb[ENC_ITEMS];      // B24: the items' bucket << 24 — one value for the whole tile unless it lies on one of the 40 boundaries
    if (B24) {
        const enc_b24 K = enc_b24_setup(bo, n);
        uint32_t pb = K.b0;
#pragma unroll
        for (int j = 0; j < ENC_ITEMS; ++j) hb[j] = K.b0 << 24;
        if (K.b0 != K.b1) {      // block-uniform
            if (base) pb = K.at(base - 1);
#pragma unroll
            for (int j = 0; j < ENC_ITEMS; ++j) if (base + j < n) hb[j] = K.at(base + j) << 24;
        }
        ph = pb << 24 | pk >> 8; pid = first_id + (((pk & 255u) << 16) | pv);
    } else { ph = enc_codec<V>::hash(pk); pid = enc_codec<V>::id(pk, (V)pv, first_id); }
    if (pred_id) *pred_id = pid;     // id of the element before the thread's first one (undefined for element 0)
    bool have_prev = base > 0;
#pragma unroll
    for (int j = 0; j < ENC_ITEMS; ++j) {
        uint64_t p = base + j;
        uint32_t h, id;
        if (B24) { h = hb[j] | k[j] >> 8; id = first_id + (((k[j] & 255u) << 16) | v[j]); }
        else { h = enc_codec<V>::hash(k[j]); id = enc_codec<V>::id(k[j], (V)v[j], first_id); }
        bool in = p < n;
        bool head = !have_prev || ph != h;
        bool dup = !head && pid == id;
        it[j].hash = h;
        it[j].head = (in && head) ? 1u : 0u;
        it[j].delta = head ? id : id - pid;
        it[j].len = (!in || dup) ? 0u : varint_len(it[j].delta);
        ph = h; pid = id; have_prev = true;
    }
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
template <typename V, bool B24>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_sizes(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t n, uint32_t first_id,
                                                           uint32_t *__restrict__ tile_bytes, uint32_t *__restrict__ tile_heads,
                                                           uint32_t *__restrict__ tile_posts, const uint64_t *__restrict__ bo) {
    __shared__ uint64_t sm[ENC_THREADS / 64];
    uint64_t base = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)threadIdx.x * ENC_ITEMS;
    uint32_t bytes = 0, heads = 0, posts = 0;
    enc_item it[ENC_ITEMS];
    enc_load_classify<V, B24>(keys, ids, n, base, first_id, it, bo);
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) {
        bytes += it[k].len;
        heads += it[k].head;
        posts += it[k].len ? 1u : 0u;
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
template <typename V, bool B24>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_write(const uint32_t *__restrict__ keys, const V *__restrict__ ids, uint64_t n, uint32_t first_id,
                                                           const uint64_t *__restrict__ tile_byte_off, const uint64_t *__restrict__ tile_head_off,
                                                           uint8_t *__restrict__ value, uint32_t *__restrict__ hashes, uint64_t *__restrict__ offsets,
                                                           uint32_t *__restrict__ last_ids, const uint64_t *__restrict__ bo) {
    __shared__ uint64_t sm[ENC_THREADS / 64];
    // varint bytes of the tile are assembled in LDS (pre-shifted by the global misalignment) and leave as
    // 16-byte stores instead of one global byte store per byte
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[ENC_TILE * 5 + 32];
    uint64_t base = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)threadIdx.x * ENC_ITEMS;
    enc_item it[ENC_ITEMS];
    uint32_t bytes = 0, heads = 0;
    uint32_t run_id = 0;   // id of the element before item k (for the per-list last ids)
    enc_load_classify<V, B24>(keys, ids, n, base, first_id, it, bo, &run_id);
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) { bytes += it[k].len; heads += it[k].head; }
    uint64_t tot;
    uint64_t ex = block_excl_scan_packed(((uint64_t)bytes << 32) | heads, sm, &tot);
    const uint64_t tile_b0 = tile_byte_off[blockIdx.x];
    const uint32_t shift = (uint32_t)(tile_b0 & 15ull);
    uint32_t lo = shift + (uint32_t)(ex >> 32);           // LDS position of this thread's first byte
    uint64_t boff = tile_b0 + (ex >> 32);
    uint64_t hoff = tile_head_off[blockIdx.x] + (uint32_t)ex;
#pragma unroll
    for (int k = 0; k < ENC_ITEMS; ++k) {
        if (it[k].head) {
            hashes[hoff] = it[k].hash;
            offsets[hoff] = boff;
            if (hoff) last_ids[hoff - 1] = run_id;            // the list before this head ends on the preceding element
            ++hoff;
        }
        run_id = it[k].head ? it[k].delta : run_id + it[k].delta;
        if (base + k + 1 == n) last_ids[hoff - 1] = run_id;    // last element: its own id
        uint32_t v = it[k].delta;
        for (uint32_t b = 0; b < it[k].len; ++b) {
            uint32_t byte = v & 0x7fu;
            v >>= 7;
            s_bytes[lo++] = (uint8_t)(byte | (b + 1 < it[k].len ? 0x80u : 0u));
        }
        boff += it[k].len;
    }
    __syncthreads();
    const uint32_t tile_len = (uint32_t)(tot >> 32);
    const uint32_t end = shift + tile_len;                 // LDS range [shift, end) holds the tile's bytes
    uint8_t *gbase = value + (tile_b0 - shift);            // 16-byte aligned (value comes from hipMalloc)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t c = threadIdx.x * 16; c < end; c += ENC_THREADS * 16) {
        if (c >= shift && c + 16 <= end) {
            *reinterpret_cast<u32x4 *>(gbase + c) = *reinterpret_cast<const u32x4 *>(s_bytes + c);
        } else {
            for (uint32_t b = c < shift ? shift : c; b < c + 16 && b < end; ++b) gbase[b] = s_bytes[b];
        }
    }
}

__global__ void k_set_u64(uint64_t *dst, uint64_t idx, const uint64_t *src) { dst[idx] = src[0]; }

uint32_t fd_enc_num_tiles(uint64_t n) { return (uint32_t)((n + ENC_TILE - 1) / ENC_TILE); }
// codec: 0 = (u32 hash, u32 id), 1 = 6-byte elements of the structure-major stream, 2 = 6-byte elements of the MSD stream (seg_off = its [40][S]
// table, bo = 41 words of scratch that receive the bucket starts; fd_launch_enc_write reads them again)
void fd_launch_enc_sizes(const uint32_t *keys, const void *ids, int codec, uint32_t first_id, uint64_t n, uint32_t *tb, uint32_t *th, uint32_t *tp,
                         const uint64_t *seg_off, uint64_t S, uint64_t *bo, hipStream_t st) {
    if (!n) return;
    const dim3 g(fd_enc_num_tiles(n)), t(ENC_THREADS);
    if (codec == 2) hipLaunchKernelGGL(k_enc_bucket_starts, dim3(1), dim3(64), 0, st, seg_off, S, bo);
    if (codec == 2) hipLaunchKernelGGL((k_enc_sizes<uint16_t, true>), g, t, 0, st, keys, (const uint16_t *)ids, n, first_id, tb, th, tp, bo);
    else if (codec == 1) hipLaunchKernelGGL((k_enc_sizes<uint16_t, false>), g, t, 0, st, keys, (const uint16_t *)ids, n, first_id, tb, th, tp, bo);
    else hipLaunchKernelGGL((k_enc_sizes<uint32_t, false>), g, t, 0, st, keys, (const uint32_t *)ids, n, first_id, tb, th, tp, bo);
}
void fd_launch_enc_write(const uint32_t *keys, const void *ids, int codec, uint32_t first_id, uint64_t n, const uint64_t *tbo, const uint64_t *tho,
                         uint8_t *value, uint32_t *hashes, uint64_t *offsets, uint32_t *last_ids, const uint64_t *total_bytes_dev, uint64_t H,
                         const uint64_t *bo, hipStream_t st) {
    if (n) {
        const dim3 g(fd_enc_num_tiles(n)), t(ENC_THREADS);
        if (codec == 2) hipLaunchKernelGGL((k_enc_write<uint16_t, true>), g, t, 0, st, keys, (const uint16_t *)ids, n, first_id, tbo, tho, value, hashes, offsets, last_ids, bo);
        else if (codec == 1) hipLaunchKernelGGL((k_enc_write<uint16_t, false>), g, t, 0, st, keys, (const uint16_t *)ids, n, first_id, tbo, tho, value, hashes, offsets, last_ids, bo);
        else hipLaunchKernelGGL((k_enc_write<uint32_t, false>), g, t, 0, st, keys, (const uint32_t *)ids, n, first_id, tbo, tho, value, hashes, offsets, last_ids, bo);
    }
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, offsets, H, total_bytes_dev);
}
