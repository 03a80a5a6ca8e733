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
//   sizes / copy                     one wavefront per merged slot: lanes read the part's byte string 64 bytes at a time;
//                                    the value of a varint stream's LAST id is the sum of all its deltas, i.e. the sum over
//                                    bytes of (byte & 0x7f) << 7 * (position inside its varint) — a ballot over the
//                                    terminator bits gives every byte its position, no reassembly needed
// HBM-bound byte work: 2 reads + 1 write of the value bytes.
#include "fdgpu_internal.h"

#define MG_NONE 0xffffffffu
#define MG_MAX_PARTS 64

struct mg_part { const uint32_t *hashes; const uint64_t *offsets; const uint8_t *value; uint64_t H; };

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

// One part's byte string [b0, b1) of one hash, walked by a wavefront.  Returns (wave-uniform) the first id (absolute), the last
// id (= sum of all values) and the byte length of the first varint.  COPY: bytes behind the first varint go to dst (dst points
// at where the byte b0 + nb_first lands).
template <bool COPY>
__device__ __forceinline__ void mg_walk(const uint8_t *__restrict__ value, uint64_t b0, uint64_t b1, uint32_t *first, uint32_t *last,
                                        uint32_t *nb_first, uint8_t *__restrict__ dst) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0, acc_first = 0, carry = 0, nf = 0;
    for (uint64_t base = b0; base < b1; base += FD_WAVE) {
        const uint64_t p = base + lane;
        const bool in = p < b1;
        const uint32_t byte = in ? value[p] : 0x80u;
        const uint64_t tm = __ballot(in && !(byte & 0x80u));
        const uint64_t below = tm & ((1ull << lane) - 1ull);
        const uint32_t pin = below ? lane - (uint32_t)(63 - __clzll(below)) - 1u : lane + carry;   // position inside the varint
        const uint32_t c = in ? (byte & 0x7fu) << (7u * (pin < 5u ? pin : 4u)) : 0u;
        acc += c;
        if (base == b0) {   // the first varint ends inside the first block (<= 5 bytes)
            const uint32_t ft = (uint32_t)__ffsll((long long)tm) - 1u;
            nf = ft + 1u;
            if (lane <= ft) acc_first = c;
        }
        if (COPY && in && p >= b0 + nf) dst[p - (b0 + nf)] = (uint8_t)byte;
        carry = tm ? 63u - (uint32_t)(63 - __clzll(tm)) : carry + 64u;
    }
    *last = mg_wave_sum(acc);
    *first = mg_wave_sum(acc_first);
    *nb_first = nf;
}

// bytes of merged slot g: sum over the parts that hold the hash of (bytes of the part's list) + (re-based first varint - original)
template <bool COPY>
__global__ __launch_bounds__(256) void k_mg_slot(const mg_part *__restrict__ parts, uint32_t n_parts, const uint32_t *__restrict__ pos, uint64_t n_slots,
                                                 uint32_t *__restrict__ sizes, const uint64_t *__restrict__ out_off, uint8_t *__restrict__ out_value) {
    const uint64_t g = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= n_slots) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t myp = lane < n_parts ? pos[g * n_parts + lane] : MG_NONE;
    uint64_t present = __ballot(myp != MG_NONE);
    uint32_t total = 0, prev_last = 0;
    bool have_prev = false;
    uint8_t *dst = COPY ? out_value + out_off[g] : nullptr;
    while (present) {
        const uint32_t k = (uint32_t)__ffsll((long long)present) - 1u;
        present &= present - 1ull;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)myp, (int)k);
        const mg_part P = parts[k];
        const uint64_t b0 = P.offsets[t], b1 = P.offsets[t + 1];
        uint32_t first, last, nf;
        if (!have_prev) {   // first part holding this hash: bytes unchanged
            if (COPY) {
                for (uint64_t p = b0 + lane; p < b1; p += FD_WAVE) dst[p - b0] = P.value[p];
            }
            mg_walk<false>(P.value, b0, b1, &first, &last, &nf, nullptr);
            total += (uint32_t)(b1 - b0);
            if (COPY) dst += b1 - b0;
        } else {
            // re-based head: delta from the previous part's last id (ids of consecutive parts ascend, so delta >= 1)
            mg_walk<false>(P.value, b0, b0 + 5 < b1 ? b0 + 5 : b1, &first, &last, &nf, nullptr);   // head only: first id + its length
            const uint32_t delta = first - prev_last, dl = mg_varint_len(delta);
            if (COPY) {
                if (lane < dl) dst[lane] = (uint8_t)(((delta >> (7u * lane)) & 0x7fu) | (lane + 1u < dl ? 0x80u : 0u));
                mg_walk<true>(P.value, b0, b1, &first, &last, &nf, dst + dl);
                dst += dl + (uint32_t)(b1 - b0) - nf;
            } else {
                mg_walk<false>(P.value, b0, b1, &first, &last, &nf, nullptr);
            }
            total += dl + (uint32_t)(b1 - b0) - nf;
        }
        prev_last = last;
        have_prev = true;
    }
    if (!COPY && lane == 0) sizes[g] = total;
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
void fd_mg_sizes(const void *parts, uint32_t n_parts, const uint32_t *pos, uint64_t n_slots, uint32_t *sizes, hipStream_t st) {
    if (n_slots) hipLaunchKernelGGL(k_mg_slot<false>, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, st, (const mg_part *)parts, n_parts, pos, n_slots,
                                    sizes, (const uint64_t *)nullptr, (uint8_t *)nullptr);
}
void fd_mg_copy(const void *parts, uint32_t n_parts, const uint32_t *pos, uint64_t n_slots, const uint64_t *out_off, uint8_t *out_value, hipStream_t st) {
    if (n_slots) hipLaunchKernelGGL(k_mg_slot<true>, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, st, (const mg_part *)parts, n_parts, pos, n_slots,
                                    (uint32_t *)nullptr, out_off, out_value);
}
