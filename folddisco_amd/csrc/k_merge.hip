// k_merge.hip — device-side merge of resident sub-indices into ONE resident index.
//
// A shard with more than 2^32 residue pairs is built as several fdgpu_index_build calls over consecutive structure-id
// ranges (controller/mod.rs:282-348 walks the input in chunks in the same way).  The reference ends with one table per hash
// whose posting list is the concatenation of the chunks' lists (ids ascending: indextable.rs:171-202 appends in id order).
// Here every chunk is a complete sub-index, so the single index is, per hash, the concatenation of the parts' byte strings
// with the first varint of every continuation re-based from "absolute id" to "delta from the previous part's last id".
//
//   union of the parts' hash sets    bitmap over the hash space (2^30 bits, 2^32 when a part holds an overflowed hash) +
//                                    popcount prefix -> rank(h) = merged slot; coalesced, no sort
//   slot -> position in every part   pos[slot][part] (u32, NONE = absent), filled by one pass over each part's hashes
//   sizes                            thread per merged slot: byte lengths from the parts' offsets, the re-based head from the part's
//                                    first varint and the previous part's LAST id (the encoder stores one u32 per list; for a
//                                    loaded index k_mg_last_ids sums (byte & 0x7f) << 7 * (position inside its varint) over the list)
//   copy                             one wavefront per merged slot, eight lanes per part, 16 bytes per lane and step
// HBM-bound byte work: one read + one write of the value bytes.
#include "fdgpu_internal.h"

#define MG_NONE 0xffffffffu
#define MG_MAX_PARTS 64

struct mg_part { const uint32_t *hashes; const uint64_t *offsets; const uint8_t *value; const uint32_t *last_ids; uint64_t H; };

__global__ void k_mg_bitmap_set(const uint32_t *__restrict__ hashes, uint64_t n, uint32_t *__restrict__ bitmap) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t h = hashes[t];
    atomicOr(&bitmap[h >> 5], 1u << (h & 31u));
}
__global__ void k_mg_popc(const uint32_t *__restrict__ bitmap, uint64_t n_words, uint32_t *__restrict__ cnt) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) cnt[w] = (uint32_t)__popc(bitmap[w]);
}
__global__ void k_mg_expand(const uint32_t *__restrict__ bitmap, const uint64_t *__restrict__ prefix, uint64_t n_words, uint32_t *__restrict__ out) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t m = bitmap[w];
    uint64_t p = prefix[w];
    while (m) {
        const uint32_t b = (uint32_t)__ffs(m) - 1u;
        out[p++] = (uint32_t)(w << 5) | b;
        m &= m - 1u;
    }
}
__global__ void k_mg_pos_fill(const uint32_t *__restrict__ hashes, uint64_t n, const uint32_t *__restrict__ bitmap, const uint64_t *__restrict__ prefix,
                              uint32_t *__restrict__ pos, uint32_t part, uint32_t n_parts) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t h = hashes[t];
    const uint64_t g = prefix[h >> 5] + (uint32_t)__popc(bitmap[h >> 5] & ((1u << (h & 31u)) - 1u));
    pos[g * n_parts + part] = (uint32_t)t;
}

__device__ __forceinline__ uint32_t mg_wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, FD_WAVE);
    return v;
}
__device__ __forceinline__ uint32_t mg_varint_len(uint32_t v) { return v == 0 ? 1u : 1u + (31u - (uint32_t)__clz(v)) / 7u; }

// last structure id of every list of an index that carries no last_ids (fdgpu_index_load): one wavefront per list, the last id is
// the sum over the list's bytes of (byte & 0x7f) << 7 * (position inside its varint)
__global__ __launch_bounds__(256) void k_mg_last_ids(const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ value, uint64_t H, uint32_t *__restrict__ last_ids) {
    const uint64_t t = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= H) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = offsets[t], b1 = offsets[t + 1];
    uint32_t acc = 0, carry = 0;
    for (uint64_t base = b0; base < b1; base += FD_WAVE) {
        const uint64_t p = base + lane;
        const bool in = p < b1;
        const uint32_t byte = in ? value[p] : 0x80u;
        const uint64_t tm = __ballot(in && !(byte & 0x80u));
        const uint64_t below = tm & ((1ull << lane) - 1ull);
        const uint32_t pin = below ? lane - (uint32_t)(63 - __clzll(below)) - 1u : lane + carry;
        acc += in ? (byte & 0x7fu) << (7u * (pin < 5u ? pin : 4u)) : 0u;
        carry = tm ? 63u - (uint32_t)(63 - __clzll(tm)) : carry + 64u;
    }
    acc = mg_wave_sum(acc);
    if (lane == 0) last_ids[t] = acc;
}

// first varint of a byte string: value and length (<= 5 bytes; the 8 bytes behind p are readable: the value buffers carry slack)
__device__ __forceinline__ uint32_t mg_first_varint(const uint8_t *__restrict__ p, uint32_t *nf) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    const unsigned long long stop = ~w & 0x8080808080ull;            // terminator bits of the first five bytes
    const uint32_t n = ((uint32_t)__ffsll((long long)stop) >> 3);     // 1-based byte index of the first terminator
    *nf = n;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) if (k < n) v |= (uint32_t)((w >> (8 * k)) & 0x7full) << (7 * k);
    return v;
}

// sizes: thread = merged slot.  size = sum over the parts holding the hash of (bytes of the part's list), the first varint of every
// continuation re-based from "absolute id" to "delta from the previous part's last id"
// One thread per merged slot walks the parts that hold its hash: byte size of the merged list (a continuation's first varint shrinks from
// an absolute id to a delta against the previous part's last id) and a COPY PLAN per (slot, part) — source offset of the bytes that move
// verbatim, their count, the re-encoded first value and where the piece starts inside the merged list — so that the copy kernel's pieces
// are independent of each other and sit behind ONE dependent load instead of four (position table -> offsets -> first varint -> bytes).
struct mg_plan { uint32_t src_lo, src_hi_dl, n, delta; };      // src_hi_dl: bits 0-15 source offset >> 32, bits 16-18 varint length, bit 31 present
__global__ __launch_bounds__(256) void k_mg_sizes(const mg_part *__restrict__ parts, uint32_t n_parts, const uint32_t *__restrict__ pos, uint64_t n_slots,
                                                  uint32_t *__restrict__ sizes, uint32_t *__restrict__ out_last, mg_plan *__restrict__ plan,
                                                  uint32_t *__restrict__ plan_dst, uint32_t *__restrict__ err_flag) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_slots) return;
    uint32_t total = 0, prev_last = 0;
    bool have_prev = false;
    for (uint32_t k = 0; k < n_parts; ++k) {
        const uint32_t t = pos[g * n_parts + k];
        mg_plan pl = {0u, 0u, 0u, 0u};
        if (t != MG_NONE) {
            const mg_part P = parts[k];
            const uint64_t b0 = P.offsets[t], b1 = P.offsets[t + 1];
            uint32_t len = (uint32_t)(b1 - b0), nf = 0, dl = 0, delta = 0;
            if (b1 - b0 > 0xffffffffull - total) *err_flag = 1u;      // a merged list of 4 GiB or more does not fit the u32 plan: the call fails (FDGPU_ERANGE)
            if (have_prev) {
                const uint32_t first = mg_first_varint(P.value + b0, &nf);
                delta = first - prev_last;
                dl = mg_varint_len(delta);
                len = len - nf + dl;
            }
            const uint64_t src = b0 + nf;
            pl.src_lo = (uint32_t)src; pl.src_hi_dl = (uint32_t)(src >> 32) | (dl << 16) | 0x80000000u; pl.n = len - dl; pl.delta = delta;
            plan_dst[g * n_parts + k] = total;
            total += len;
            prev_last = P.last_ids[t];
            have_prev = true;
        }
        plan[g * n_parts + k] = pl;
    }
    sizes[g] = total;
    out_last[g] = prev_last;
}

// copy: eight lanes per (slot, part) piece, 16 bytes per lane and step (unaligned 16-byte global accesses are native on gfx950); the pieces
// of one slot are adjacent work items, so a merged list still leaves through neighbouring lanes
typedef unsigned int mg_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mg_copy(const mg_part *__restrict__ parts, uint32_t n_parts, const mg_plan *__restrict__ plan,
                                                 const uint32_t *__restrict__ plan_dst, uint64_t n_pieces, const uint64_t *__restrict__ out_off,
                                                 uint8_t *__restrict__ out_value) {
    const uint64_t G = (uint64_t)blockIdx.x * 32u + (threadIdx.x >> 3);
    if (G >= n_pieces) return;
    const mg_plan pl = plan[G];
    if (!(pl.src_hi_dl & 0x80000000u)) return;
    const uint32_t sub = threadIdx.x & 7u, k = (uint32_t)(G % n_parts);
    const uint64_t slot = G / n_parts;
    const uint32_t dl = (pl.src_hi_dl >> 16) & 7u;
    uint8_t *d = out_value + out_off[slot] + plan_dst[G];
    if (sub < dl) d[sub] = (uint8_t)(((pl.delta >> (7u * sub)) & 0x7fu) | (sub + 1u < dl ? 0x80u : 0u));
    const uint8_t *sp = parts[k].value + (((uint64_t)(pl.src_hi_dl & 0xffffu) << 32) | pl.src_lo);
    d += dl;
    const uint64_t n = pl.n;
    uint64_t o = (uint64_t)sub * 16u;
    for (; o + 16 <= n; o += 128) {
        mg_u32x4 v;
        __builtin_memcpy(&v, sp + o, 16);
        __builtin_memcpy(d + o, &v, 16);
    }
    if (o < n) for (uint64_t z = o; z < n; ++z) d[z] = sp[z];     // the lane that owns the ragged tail (< 16 bytes)
}

// ---- launchers
void fd_mg_bitmap_set(const uint32_t *hashes, uint64_t n, uint32_t *bitmap, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_mg_bitmap_set, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hashes, n, bitmap);
}
void fd_mg_popc(const uint32_t *bitmap, uint64_t n_words, uint32_t *cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_mg_popc, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, bitmap, n_words, cnt);
}
void fd_mg_expand(const uint32_t *bitmap, const uint64_t *prefix, uint64_t n_words, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_mg_expand, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, bitmap, prefix, n_words, out);
}
void fd_mg_pos_fill(const uint32_t *hashes, uint64_t n, const uint32_t *bitmap, const uint64_t *prefix, uint32_t *pos, uint32_t part, uint32_t n_parts,
                    hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_mg_pos_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hashes, n, bitmap, prefix, pos, part, n_parts);
}
void fd_mg_last_ids(const uint64_t *offsets, const uint8_t *value, uint64_t H, uint32_t *last_ids, hipStream_t st) {
    if (H) hipLaunchKernelGGL(k_mg_last_ids, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, st, offsets, value, H, last_ids);
}
void fd_mg_sizes(const void *parts, uint32_t n_parts, const uint32_t *pos, uint64_t n_slots, uint32_t *sizes, uint32_t *out_last, void *plan, uint32_t *plan_dst,
                 uint32_t *err_flag, hipStream_t st) {
    if (n_slots) hipLaunchKernelGGL(k_mg_sizes, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, (const mg_part *)parts, n_parts, pos, n_slots, sizes, out_last,
                                    (mg_plan *)plan, plan_dst, err_flag);
}
void fd_mg_copy(const void *parts, uint32_t n_parts, const void *plan, const uint32_t *plan_dst, uint64_t n_slots, const uint64_t *out_off, uint8_t *out_value,
                hipStream_t st) {
    const uint64_t n_pieces = n_slots * n_parts;
    if (n_pieces) hipLaunchKernelGGL(k_mg_copy, dim3((unsigned)((n_pieces + 31) / 32)), dim3(256), 0, st, (const mg_part *)parts, n_parts, (const mg_plan *)plan, plan_dst,
                                     n_pieces, out_off, out_value);
}
