// k_query.hip — posting-list lookup, varint decode and per-structure scoring on gfx950.
//
// Replaces HOT LOOP C of the reference: FolddiscoIndex::get_raw_entries / get_entries
// (src/index/indextable.rs:53-86, 421-463: binary search + serial varint-delta decode) and
// count_query (src/controller/count_query.rs:82-220: per query node, scan the postings of its
// hashes, match_count += 1, idf_sum += log2(S/len), node / edge occupancy bit-vectors, then merge).
//
// Mapping: posting lists are cut into 2 KB segments (see k_cq_seg).  A segment is consumed in 64-byte blocks (one byte per lane): a
// ballot over the continuation bits finds the terminator lanes, each terminator reassembles its value from the (<= 4) preceding
// lanes with shuffles, a wave prefix sum over the deltas turns them into structure ids.  Scoring sets ONE bit per posting: bit
// `structure` in the occupancy row of ITS QUERY HASH — a posting list holds a structure at most once, so the matrix
// [query hashes][ceil(S/32)] holds exactly which (hash, structure) pairs matched.  A list's ids ascend, so the segment that decodes
// them OWNS a contiguous run of the row's words: it assembles them in an LDS window and stores them as full lines, zeros included
// (no memset of the matrix, no atomic per posting); only the words two neighbouring segments share take atomics (k_cq_bounds).  Everything count_query reports follows
// from it in k_cq_rows_finalize, one thread per structure walking its query's rows (sorted by (node, partner) on the host):
//   match_count  number of set rows;   idf_sum  sum of the rows' idf in 2^-22 fixed point (order-independent, unlike the reference's
//   f32 sum whose order follows FxHashMap iteration; BASELINE.md §2 states the 1e-5 tolerance);   edge_count / node_count  number
//   of (node, partner) / node groups with a set row.
// (Before: four atomics per posting — count, idf, node bit, edge bit — then two; measured 16-30 G atomics/s whatever their mix,
// so the posting's cost IS its atomics: 630 us per 32 motif queries at 542 k structures with two, ~330 with one.)
#include <algorithm>
#include "fdgpu_internal.h"

// Per-structure accumulator: ONE u64 per (query, structure) = match count << 46 | idf sum in units of 2^-22 (a posting costs one
// 64-bit atomic for both; the sum of <= 2^18 addends below 32 fits the low 46 bits).  A query with 2^18 hashes or more, or an idf outside
// [0, 32), takes the wide form instead: u32 counts and u64 sums in separate arrays, same 2^-22 resolution, so both forms give the same bits.
#define IDF_SCALE 4194304.0 /* 2^22 */
#define CQ_CNT_SHIFT 46
#define CQ_SUM_MASK ((1ull << CQ_CNT_SHIFT) - 1ull)
struct fd_count_rec_dev { uint32_t nid, total_match_count, node_count, edge_count; float idf; };

__device__ __forceinline__ int64_t find_hash(const uint32_t *__restrict__ hashes, uint64_t H, uint32_t h) {
    uint64_t lo = 0, hi = H;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (hashes[mid] < h) lo = mid + 1; else hi = mid;
    }
    return (lo < H && hashes[lo] == h) ? (int64_t)lo : -1;
}

// byte length of each query hash's posting list (get_raw_entries(h).len(), indextable.rs:53-81): the varint bytes a scoring pass reads
__global__ void k_posting_bytes(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets, uint64_t H, const uint32_t *__restrict__ q_hash,
                                uint64_t nq, uint64_t *__restrict__ bytes) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int64_t k = find_hash(hashes, H, q_hash[q]);
    bytes[q] = k >= 0 ? offsets[k + 1] - offsets[k] : 0ull;
}
void fd_launch_posting_bytes(const uint32_t *hashes, const uint64_t *offsets, uint64_t H, const uint32_t *q_hash, uint64_t nq, uint64_t *bytes, hipStream_t st) {
    if (nq) hipLaunchKernelGGL(k_posting_bytes, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, hashes, offsets, H, q_hash, nq, bytes);
}

// get_entries (src/index/indextable.rs:83-86, 439-463): posting list of every query hash decoded to structure ids.
// Same wave-parallel varint decode as the scoring kernel; ids go to out[out_off[q] ...] in list order.
__global__ __launch_bounds__(FD_WAVE) void k_get_entries(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets,
                                                         const uint8_t *__restrict__ value, uint64_t H, const uint32_t *__restrict__ q_hash,
                                                         uint64_t nq, const uint64_t *__restrict__ out_off, uint32_t *__restrict__ out) {
    uint64_t q = blockIdx.x;
    if (q >= nq) return;
    int64_t k = find_hash(hashes, H, q_hash[q]);
    if (k < 0) return;
    const uint64_t b0 = offsets[k], b1 = offsets[k + 1];
    const uint32_t lane = threadIdx.x;
    uint32_t *dst = out + out_off[q];
    uint64_t n_done = 0;        // ids written so far (wave-uniform)
    uint32_t run_id = 0, carry_val = 0, carry_shift = 0;
    bool have_first = false;
    for (uint64_t base = b0; base < b1; base += FD_WAVE) {
        uint64_t p = base + lane;
        bool in = p < b1;
        uint32_t byte = in ? value[p] : 0x80u;
        bool term = in && !(byte & 0x80u);
        uint64_t tm = __ballot(term);
        uint64_t below = tm & ((1ull << lane) - 1ull);
        int prev_t = below ? 63 - __clzll(below) : -1;
        uint32_t len_here = lane - (uint32_t)(prev_t + 1) + 1;
        uint32_t v = 0, pay = byte & 0x7fu;
#pragma unroll
        for (int back = 4; back >= 0; --back) {
            uint32_t pb = __shfl(pay, (int)lane - back, FD_WAVE);
            if ((uint32_t)back < len_here) v |= pb << (7u * (len_here - 1u - (uint32_t)back));
        }
        if (term && prev_t < 0) v = carry_val | (v << carry_shift);
        uint32_t s2 = term ? v : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(s2, off, FD_WAVE);
            if ((int)lane >= off) s2 += t;
        }
        uint32_t id = (have_first ? run_id : 0u) + s2;
        if (term) dst[n_done + (uint32_t)__popcll(below)] = id;
        if (tm) {
            int last_t = 63 - __clzll(tm);
            run_id = __shfl(id, last_t, FD_WAVE);
            have_first = true;
            n_done += (uint64_t)__popcll(tm);
            uint32_t tail = 63u - (uint32_t)last_t, pv = 0;
            for (uint32_t t2 = 0; t2 < tail && t2 < 5; ++t2) pv |= __shfl(pay, last_t + 1 + (int)t2, FD_WAVE) << (7u * t2);
            carry_val = pv;
            carry_shift = 7u * tail;
        }   // (a 64-byte block always holds a terminator: varints of u32 ids are at most 5 bytes and a list ends on one)
    }
}
void fd_launch_get_entries(const uint32_t *hashes, const uint64_t *offsets, const uint8_t *value, uint64_t H, const uint32_t *q_hash, uint64_t nq,
                           const uint64_t *out_off, uint32_t *out, hipStream_t st) {
    if (nq) hipLaunchKernelGGL(k_get_entries, dim3((unsigned)nq), dim3(FD_WAVE), 0, st, hashes, offsets, value, H, q_hash, nq, out_off, out);
}

// ------------------------------------------------------------------ segment-parallel scoring
// One wavefront per posting LIST serialises the decode of a long list (Swiss-Prot scale: lists of 100 k ids = 150 KB = 2,300
// dependent 64-byte steps, which alone took 4.9 ms of a 32-query batch).  A delta stream can be cut anywhere once every piece
// knows the id it starts from: a varint belongs to the segment that holds its LAST byte, so
//   plan     per query hash: list position and number of CQ_SEG-byte segments            (k_cq_plan + exclusive scan -> work items)
//   sums     per segment: sum of the varint values that end inside it                     (k_cq_seg<true>; skipped when no list is split)
//   bounds   per list: exclusive prefix of its segment sums = the id every segment starts from; the row words two neighbouring
//            segments share (the word of a segment's start id) are zeroed, rows of absent hashes are zeroed whole   (k_cq_bounds)
//   score    per segment: the same 64-byte block decode; the ids' bits go into an LDS window over the row words the segment owns
//            (everything between its two shared words; the first / last segment own the row's head / tail) and leave as full
//            lines, the shared words take atomic ORs
// Work items are walked by a persistent grid (the count stays on the device).
#define CQ_SEG 2048u
#define CQ_WIN 1024u     // words of an occupancy row in a wavefront's LDS window (32,768 structures)
struct cq_plan {
    const long long *kidx;        // [nq] position of the hash in the index, -1 = absent
    const uint64_t *wstart;       // [nq + 1] first work item of every query hash
    uint32_t *segsum;             // [work items]
};
__global__ void k_cq_plan(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets, uint64_t H, const uint32_t *__restrict__ q_hash, uint64_t nq,
                          long long *__restrict__ kidx, uint32_t *__restrict__ nseg) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t k = find_hash(hashes, H, q_hash[q]);
    kidx[q] = k;
    nseg[q] = k < 0 ? 0u : (uint32_t)((offsets[k + 1] - offsets[k] + CQ_SEG - 1) / CQ_SEG);
}

// number of ids in each query hash's posting list (= bytes without the continuation bit), over the same segments: a list of 100 k
// ids is ~75 independent 2 KB pieces instead of one wavefront's 2,300 dependent steps; 16 bytes per lane and step
__global__ __launch_bounds__(FD_WAVE) void k_pl_count(const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ value, const long long *__restrict__ kidx,
                                                      const uint64_t *__restrict__ wstart, uint64_t nq, unsigned long long *__restrict__ lengths) {
    const uint32_t lane = threadIdx.x;
    const uint64_t W = wstart[nq];
    for (uint64_t w = blockIdx.x; w < W; w += gridDim.x) {
        uint64_t lo = 0, hi = nq;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (wstart[mid] <= w) lo = mid; else hi = mid; }
        const uint64_t q = lo;
        const long long k = kidx[q];
        const uint64_t b0 = offsets[k], b1 = offsets[k + 1];
        const uint64_t s0 = b0 + (w - wstart[q]) * CQ_SEG, s1 = s0 + CQ_SEG < b1 ? s0 + CQ_SEG : b1;
        uint32_t cnt = 0;
        for (uint64_t p = s0 + (uint64_t)lane * 16; p < s1; p += 64 * 16) {
            if (p + 16 <= s1) {
                unsigned long long w0, w1;
                __builtin_memcpy(&w0, value + p, 8); __builtin_memcpy(&w1, value + p + 8, 8);
                cnt += 16u - (uint32_t)__popcll(w0 & 0x8080808080808080ull) - (uint32_t)__popcll(w1 & 0x8080808080808080ull);
            } else for (uint64_t z = p; z < s1; ++z) cnt += (value[z] & 0x80u) ? 0u : 1u;
        }
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, FD_WAVE);
        if (lane == 0 && cnt) atomicAdd(&lengths[q], (unsigned long long)cnt);
    }
}

// row word of a structure id, monotone in the id (ids outside the index's range clamp to the row's ends and set no bit)
__device__ __forceinline__ uint32_t cq_word_of(uint32_t id, uint32_t first_id, uint32_t S) {
    if (id < first_id) return 0u;
    const uint32_t rel = id - first_id;
    return (rel < S ? rel : S - 1u) >> 5;
}
// one wavefront per query hash: segment sums -> start ids (in place), shared words zeroed, absent rows zeroed
__global__ __launch_bounds__(FD_WAVE) void k_cq_bounds(cq_args A, cq_plan P) {
    const uint32_t lane = threadIdx.x;
    for (uint64_t q = blockIdx.x; q < A.nq; q += gridDim.x) {
        const uint64_t w0 = P.wstart[q];
        const uint32_t n = (uint32_t)(P.wstart[q + 1] - w0);
        uint32_t *hb = A.hash_bits + (uint64_t)q * A.words;
        if (n == 0) { for (uint32_t x = lane; x < A.words; x += FD_WAVE) hb[x] = 0u; continue; }
        uint32_t run = 0;
        for (uint32_t t0 = 0; t0 < n && n > 1; t0 += FD_WAVE) {
            const uint32_t t = t0 + lane;
            const uint32_t v = t < n ? P.segsum[w0 + t] : 0u;
            uint32_t s2 = v;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(s2, off, FD_WAVE);
                if ((int)lane >= off) s2 += u;
            }
            const uint32_t excl = run + s2 - v;
            if (t < n) {
                P.segsum[w0 + t] = excl;
                if (t) hb[cq_word_of(excl, A.first_id, A.S)] = 0u;
            }
            run += __shfl(s2, 63, FD_WAVE);
        }
    }
}

template <bool SUMS>
__global__ __launch_bounds__(FD_WAVE) void k_cq_seg(cq_args A, cq_plan P) {
    __shared__ uint32_t win[CQ_WIN];
    const uint32_t lane = threadIdx.x;
    if (!SUMS) for (uint32_t x = lane; x < CQ_WIN; x += FD_WAVE) win[x] = 0u;
    const uint64_t W = P.wstart[A.nq];
    for (uint64_t w = blockIdx.x; w < W; w += gridDim.x) {
        // query hash of this work item: last q with wstart[q] <= w
        uint64_t lo = 0, hi = A.nq;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (P.wstart[mid] <= w) lo = mid; else hi = mid; }
        const uint64_t q = lo;
        const uint64_t w0 = P.wstart[q];
        const uint32_t j = (uint32_t)(w - w0);
        if (SUMS && P.wstart[q + 1] - w0 < 2) continue;           // single-segment list: its base is 0
        const int64_t k = P.kidx[q];
        const uint64_t b0 = A.offsets[k], b1 = A.offsets[k + 1];
        const uint64_t s0 = b0 + (uint64_t)j * CQ_SEG, s1 = s0 + CQ_SEG < b1 ? s0 + CQ_SEG : b1;
        // the varint that straddles the segment start: its leading bytes are the (<= 4) bytes before s0 behind the last terminator
        uint32_t carry_val = 0, carry_shift = 0;
        if (j) {
            const uint64_t p = s0 - 4 + lane;
            const bool in4 = lane < 4;
            const uint32_t byte = (in4 && p >= b0) ? A.value[p] : 0u;              // before the list start counts as a boundary
            const uint32_t tm4 = (uint32_t)__ballot(in4 && !(byte & 0x80u)) & 15u;
            const int lt = tm4 ? 31 - __clz((int)tm4) : -1;
            const uint32_t tail = 3u - (uint32_t)lt;
            const uint32_t pay = byte & 0x7fu;
            uint32_t pv = 0;
            for (uint32_t t2 = 0; t2 < tail; ++t2) pv |= (uint32_t)__shfl((int)pay, lt + 1 + (int)t2, FD_WAVE) << (7u * t2);
            carry_val = pv; carry_shift = 7u * tail;
        }
        uint32_t run_id = 0;
        uint32_t *hb = nullptr;
        // row words [plain_lo, plain_hi) are this segment's alone; lo_word / plain_hi are shared with the neighbours (atomics)
        uint32_t lo_word = 0xffffffffu, plain_lo = 0, plain_hi = 0, win_base = 0;
        if (!SUMS) {
            hb = A.hash_bits + (uint64_t)q * A.words;     // the occupancy row of this query hash
            if (j) { run_id = P.segsum[w]; lo_word = cq_word_of(run_id, A.first_id, A.S); plain_lo = lo_word + 1u; }
            plain_hi = w + 1 < P.wstart[q + 1] ? cq_word_of(P.segsum[w + 1], A.first_id, A.S) : A.words;
            win_base = plain_lo & ~63u;
        }
        uint32_t seg_acc = 0;
        for (uint64_t base = s0; base < s1; base += FD_WAVE) {
            const uint64_t p = base + lane;
            const bool in = p < s1;
            const uint32_t byte = in ? A.value[p] : 0x80u;
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
            if (!SUMS) {
                const uint32_t rel = id - A.first_id;
                const bool valid = term && id >= A.first_id && rel < A.S;
                const uint32_t wd = rel >> 5, bit = 1u << (rel & 31u);
                const bool shared = valid && (wd == lo_word || wd == plain_hi);
                if (shared) atomicOr(&hb[wd], bit);
                bool todo = valid && !shared;
                uint64_t pend = __ballot(todo);
                while (pend) {
                    if (todo && wd - win_base < CQ_WIN) { atomicOr(&win[wd - win_base], bit); todo = false; }      // ds_or_b32
                    pend = __ballot(todo);
                    if (pend) {     // the window leaves as full lines; ids ascend with the lane, so the lowest waiting lane names the next window
                        const uint32_t next_base = (uint32_t)__shfl((int)wd, __ffsll((unsigned long long)pend) - 1, FD_WAVE) & ~63u;
                        for (uint32_t x = lane; x < CQ_WIN; x += FD_WAVE) {
                            const uint32_t g = win_base + x, v = win[x];
                            win[x] = 0u;
                            if (g >= plain_lo) hb[g] = v;
                        }
                        for (uint32_t g = win_base + CQ_WIN + lane; g < next_base; g += FD_WAVE) hb[g] = 0u;
                        win_base = next_base;
                    }
                }
            }
            if (tm) {
                const int last_t = 63 - __clzll(tm);
                run_id = __shfl(id, last_t, FD_WAVE);
                const uint32_t tail = 63u - (uint32_t)last_t;
                uint32_t pv = 0;
                for (uint32_t t2 = 0; t2 < tail && t2 < 5; ++t2) pv |= __shfl(pay, last_t + 1 + (int)t2, FD_WAVE) << (7u * t2);
                carry_val = pv; carry_shift = 7u * tail;
            } else {   // a block of continuation bytes only (cannot happen for 32-bit ids; kept consistent)
                uint32_t pv = carry_val;
                for (uint32_t t2 = 0; t2 < 5; ++t2) pv |= __shfl(pay, (int)t2, FD_WAVE) << (carry_shift + 7u * t2);
                carry_val = pv; carry_shift += 7u * FD_WAVE;
            }
            if (SUMS) seg_acc = run_id;
        }
        if (SUMS && lane == 0) P.segsum[w] = seg_acc;     // run_id started at 0: the sum of the values that end in this segment
        if (!SUMS) {        // the rest of the window, then zeros up to the word shared with the next segment (or the row's end)
            for (uint32_t x = lane; x < CQ_WIN && win_base + (x & ~63u) < plain_hi; x += FD_WAVE) {
                const uint32_t g = win_base + x, v = win[x];
                win[x] = 0u;
                if (g >= plain_lo && g < plain_hi) hb[g] = v;
            }
            for (uint32_t g = win_base + CQ_WIN + lane; g < plain_hi; g += FD_WAVE) hb[g] = 0u;
        }
    }
}
void fd_launch_cq_plan(const cq_args &A, long long *kidx, uint32_t *nseg, hipStream_t st) {
    if (A.nq) hipLaunchKernelGGL(k_cq_plan, dim3((unsigned)((A.nq + 255) / 256)), dim3(256), 0, st, A.hashes, A.offsets, A.H, A.q_hash, A.nq, kidx, nseg);
}
// n_items: number of work items (host copy of wstart[nq]); split: some list has more than one segment
void fd_launch_cq_seg(const cq_args &A, const long long *kidx, const uint64_t *wstart, uint32_t *segsum, uint64_t n_items, bool split,
                      hipStream_t st) {
    if (!A.nq) return;
    cq_plan P;
    P.kidx = kidx; P.wstart = wstart; P.segsum = segsum;
    if (!n_items) { hipLaunchKernelGGL(k_cq_bounds, dim3((unsigned)(A.nq < 16384 ? A.nq : 16384)), dim3(FD_WAVE), 0, st, A, P); return; }
    const unsigned grid = (unsigned)(n_items < 16384 ? n_items : 16384);
    if (split) hipLaunchKernelGGL(k_cq_seg<true>, dim3(grid), dim3(FD_WAVE), 0, st, A, P);
    hipLaunchKernelGGL(k_cq_bounds, dim3((unsigned)(A.nq < 16384 ? A.nq : 16384)), dim3(FD_WAVE), 0, st, A, P);
    hipLaunchKernelGGL(k_cq_seg<false>, dim3(grid), dim3(FD_WAVE), 0, st, A, P);
}

// count_query's per-structure results from the hash occupancy rows.  Thread = one 32-structure word column of one query, loop = the
// rows of its query in (node, partner) order; row_meta[r] = idf in 2^-22 fixed point << 2 | last row of its node << 1 | last row of
// its edge.  The work follows the SET bits: a zero word costs one coalesced load and a test; a set bit adds the row's idf to the
// lane's 32 sums (LDS, stride 33) — match / edge / node counts are bit-sliced carry-save counters in registers (32 structures per
// add, early exit when the carry dies).  Results leave through the LDS tile, structure-major and coalesced.  A query with tens of
// thousands of rows (whole-structure mode) has only S / 4096 workgroups of columns: its rows are cut into slices at node boundaries
// (grid.z) and the slices ADD into zeroed results.
#define CQ_FIN_T 128
template <int NP>
__device__ __forceinline__ void cq_sliced_add(uint32_t (&p)[NP], uint32_t x) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint32_t c = p[k] & x;
        p[k] ^= x;
        x = c;
        if (!x) break;
    }
}
// NP = counter planes: 8 when no query of the call has 256 rows or more (motif queries), else 20
// ALIAS (packed form only): the count tile shares the LDS of the sums — pass 0 folds every thread's counts into its own sums
// (count << CQ_CNT_SHIFT | sum, the packed record itself), passes 1 and 2 reuse the memory — 34 KB per workgroup instead of 51
template <int NP, bool ALIAS = false>
__global__ __launch_bounds__(CQ_FIN_T) void k_cq_rows_finalize(const uint32_t *__restrict__ hash_bits, const unsigned long long *__restrict__ row_meta,
                                                               const uint64_t *__restrict__ q_rows /*[nQ + 1] or null = one query over n_rows*/, uint64_t n_rows,
                                                               const uint64_t *__restrict__ slices /*[gridDim.z + 1] row boundaries or null*/,
                                                               uint32_t words, uint32_t S, int packed, uint32_t *__restrict__ match,
                                                               unsigned long long *__restrict__ acc, uint32_t *__restrict__ node_cnt,
                                                               uint32_t *__restrict__ edge_cnt, uint8_t *__restrict__ flags) {
    __shared__ unsigned long long s_sum[CQ_FIN_T * 33];
    const uint32_t w = blockIdx.x * CQ_FIN_T + threadIdx.x, qy = blockIdx.y;
    const bool live = w < words;
    uint64_t r0 = q_rows ? q_rows[qy] : 0ull, r1 = q_rows ? q_rows[qy + 1] : n_rows;
    const bool add = slices != nullptr;
    if (add) { r0 = slices[blockIdx.z]; r1 = slices[blockIdx.z + 1]; }
    for (int b = 0; b < 32; ++b) s_sum[threadIdx.x * 33 + b] = 0ull;
    uint32_t pc[NP], pe[NP], pn[NP], e_or = 0, n_or = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) { pc[k] = 0; pe[k] = 0; pn[k] = 0; }
    unsigned long long *mine = s_sum + threadIdx.x * 33;
    auto row = [&](uint32_t x, unsigned long long m) {
        if (x) {
            cq_sliced_add(pc, x);
            e_or |= x;
            const unsigned long long fix = m >> 2;
            while (x) { const int b = __builtin_ctz(x); mine[b] += fix; x &= x - 1u; }
        }
        if (m & 1ull) { if (e_or) cq_sliced_add(pe, e_or); n_or |= e_or; e_or = 0; }
        if (m & 2ull) { if (n_or) cq_sliced_add(pn, n_or); n_or = 0; }
    };
    uint64_t r = r0;
    for (; r + 8 <= r1; r += 8) {       // eight rows' words in flight: the walk is latency-bound otherwise
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = live ? hash_bits[(r + u) * words + w] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) row(x[u], row_meta[r + u]);
    }
    for (; r < r1; ++r) row(live ? hash_bits[r * words + w] : 0u, row_meta[r]);
    // counts of this thread's 32 structures -> the tile, packed with the sums where the packed form applies
    __shared__ uint32_t s_cnt_own[ALIAS ? 1 : CQ_FIN_T * 33];
    uint32_t *s_cnt = ALIAS ? reinterpret_cast<uint32_t *>(s_sum) : s_cnt_own;
    const uint32_t nid0 = blockIdx.x * CQ_FIN_T * 32;
    const uint32_t lim = nid0 < S ? (S - nid0 < CQ_FIN_T * 32 ? S - nid0 : CQ_FIN_T * 32) : 0u;
    const uint64_t qbase = (uint64_t)qy * S;
    for (int pass = 0; pass < 3; ++pass) {      // 0: match counts (+ sums), 1: edge counts, 2: node counts
        if (ALIAS && pass) __syncthreads();       // the sums of pass 0 have been read: their memory takes the counts
        for (uint32_t b = 0; b < 32; ++b) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) v |= (((pass == 0 ? pc[k] : pass == 1 ? pe[k] : pn[k]) >> b) & 1u) << k;
            if (ALIAS && pass == 0) s_sum[threadIdx.x * 33 + b] |= (unsigned long long)v << CQ_CNT_SHIFT;
            else s_cnt[threadIdx.x * 33 + b] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < lim; i += CQ_FIN_T) {
            const uint32_t at = (i >> 5) * 33 + (i & 31u);
            const uint64_t g = qbase + nid0 + i;
            const uint32_t v = (ALIAS && pass == 0) ? (uint32_t)(s_sum[at] >> CQ_CNT_SHIFT) : s_cnt[at];
            if (pass == 0) {
                const unsigned long long sum = ALIAS ? (s_sum[at] & CQ_SUM_MASK) : s_sum[at];
                if (!add) {
                    if (packed) acc[g] = ((unsigned long long)v << CQ_CNT_SHIFT) | sum; else { acc[g] = sum; match[g] = v; }
                    flags[g] = v ? 1 : 0;
                } else if (v) {
                    if (packed) atomicAdd(&acc[g], ((unsigned long long)v << CQ_CNT_SHIFT) | sum); else { atomicAdd(&acc[g], sum); atomicAdd(&match[g], v); }
                    flags[g] = 1;       // zeroed by the launcher; every slice that saw the structure says the same
                }
            } else {
                uint32_t *dst = pass == 1 ? edge_cnt : node_cnt;
                if (!add) dst[g] = v; else if (v) atomicAdd(&dst[g], v);
            }
        }
        __syncthreads();
    }
}


__global__ __launch_bounds__(256) void k_cq_compact(const uint32_t *__restrict__ match, const unsigned long long *__restrict__ idf,
                                                    const uint32_t *__restrict__ node_cnt, const uint32_t *__restrict__ edge_cnt,
                                                    const uint8_t *__restrict__ flags, const uint64_t *__restrict__ pos,
                                                    const float *__restrict__ penalty, uint32_t S, uint32_t first_id,
                                                    fd_count_rec_dev *__restrict__ out) {
    uint32_t nid = blockIdx.x * blockDim.x + threadIdx.x;
    if (nid >= S || !flags[nid]) return;
    fd_count_rec_dev r;
    r.nid = nid + first_id;
    const unsigned long long a = idf[nid];
    r.total_match_count = match ? match[nid] : (uint32_t)(a >> CQ_CNT_SHIFT);
    r.node_count = node_cnt[nid];
    r.edge_count = edge_cnt[nid];
    float sum = (float)((double)(match ? a : a & CQ_SUM_MASK) * (1.0 / IDF_SCALE));
    r.idf = sum * penalty[nid];  // count_query.rs:200 idf_sum *= nres^(-lp)
    out[pos[nid]] = r;
}

// ------------------------------------------------------------------ batched scoring (many queries, one launch each)
// Same arithmetic as above; the per-structure results are [n_queries][S], the occupancy matrix has one row per query hash of the batch.
__global__ __launch_bounds__(256) void k_cq_compact_batch(const uint32_t *__restrict__ match, const unsigned long long *__restrict__ idf,
                                                          const uint32_t *__restrict__ node_cnt, const uint32_t *__restrict__ edge_cnt,
                                                          const uint8_t *__restrict__ flags, const uint64_t *__restrict__ pos,
                                                          const float *__restrict__ penalty, uint32_t S, uint64_t total, uint32_t first_id,
                                                          fd_count_rec_dev *__restrict__ out) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total || !flags[g]) return;
    uint32_t nid = (uint32_t)(g % S);
    fd_count_rec_dev r;
    r.nid = nid + first_id;
    const unsigned long long a = idf[g];
    r.total_match_count = match ? match[g] : (uint32_t)(a >> CQ_CNT_SHIFT);
    r.node_count = node_cnt[g];
    r.edge_count = edge_cnt[g];
    float sum = (float)((double)(match ? a : a & CQ_SUM_MASK) * (1.0 / IDF_SCALE));
    r.idf = sum * penalty[nid];
    out[pos[g]] = r;
}

// ------------------------------------------------------------------ per-query top-N preselection (candidate selection, query_pdb.rs:404-411)
// The reference sorts every touched structure by idf (descending) and truncates to --top.  Here a two-level radix select on the
// order-preserving image of the f32 idf (11 + 11 bits) finds, per query, the 22-bit threshold below which a record cannot be in
// the top N; records at or above it (>= N of them unless fewer exist; ties of the threshold bin included) are copied out, so
// the host sorts ~N records instead of every touched structure.  One workgroup per query.
#define TOPN_BINS 2048
__device__ __forceinline__ uint32_t idf_order_key(float v) {
    uint32_t b = __float_as_uint(v + 0.0f);   // -0 -> +0
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// Many workgroups per query: histograms are accumulated in LDS and merged into a global [query][2048] table; the threshold search
// is one small workgroup per query; the survivors take their slots with one global atomic each (~top_n per query).
#define TOPN_SPLIT 32   // workgroups per query
struct topn_state { uint32_t thr_bin, above, thr22, count; };
__global__ __launch_bounds__(256) void k_topn_hist(const fd_count_rec_dev *__restrict__ recs, const uint64_t *__restrict__ off, uint32_t top_n, int level,
                                                   const topn_state *__restrict__ st, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t hist[TOPN_BINS];
    const uint32_t q = blockIdx.y;
    const fd_count_rec_dev *r = recs + off[q];
    const uint64_t m = off[q + 1] - off[q];
    if (m <= top_n) return;
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t b1 = level ? st[q].thr_bin : 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)TOPN_SPLIT * 256) {
        const uint32_t key = idf_order_key(r[i].idf);
        if (!level) atomicAdd(&hist[key >> 21], 1u);
        else if ((key >> 21) == b1) atomicAdd(&hist[(key >> 10) & (TOPN_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) if (hist[k]) atomicAdd(&ghist[(uint64_t)q * TOPN_BINS + k], hist[k]);
}
// threshold bin: the highest bin b with (records above b) + hist[b] >= top_n; one workgroup per query, suffix sums by 256 threads
__global__ __launch_bounds__(256) void k_topn_thr(const uint64_t *__restrict__ off, uint32_t top_n, int level, topn_state *__restrict__ st,
                                                  uint32_t *__restrict__ ghist) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t s_bin, s_above;
    const uint32_t q = blockIdx.x;
    const uint64_t m = off ? off[q + 1] - off[q] : ~0ull;       // dense form: the number of touched structures is not known (nor needed)
    if (m <= top_n) { if (threadIdx.x == 0) { st[q].thr_bin = 0; st[q].above = 0; st[q].thr22 = 0; st[q].count = 0; } return; }
    uint32_t *h = ghist + (uint64_t)q * TOPN_BINS;
    // thread t owns bins [8 t, 8 t + 8); suffix sum over threads from the top
    uint32_t mine = 0;
    for (int k = 0; k < 8; ++k) mine += h[threadIdx.x * 8 + k];
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = level ? st[q].above : 0u;
        int t = 255;
        for (; t > 0; --t) { if (acc + part[t] >= top_n) break; acc += part[t]; }
        int b = t * 8 + 7;
        for (; b > t * 8; --b) { if (acc + h[b] >= top_n) break; acc += h[b]; }
        s_bin = (uint32_t)b; s_above = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!level) { st[q].thr_bin = s_bin; st[q].above = s_above; }
        else { st[q].thr22 = (st[q].thr_bin << 11) | s_bin; st[q].count = 0; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) h[k] = 0;     // ready for the next level / the next call
}
__global__ __launch_bounds__(256) void k_topn_emit(const fd_count_rec_dev *__restrict__ recs, const uint64_t *__restrict__ off, uint32_t cap,
                                                   topn_state *__restrict__ st, fd_count_rec_dev *__restrict__ out) {
    const uint32_t q = blockIdx.y;
    const fd_count_rec_dev *r = recs + off[q];
    const uint64_t m = off[q + 1] - off[q];
    const uint32_t thr22 = st[q].thr22;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)TOPN_SPLIT * 256) {
        const fd_count_rec_dev x = r[i];
        if ((idf_order_key(x.idf) >> 10) >= thr22) {
            const uint32_t pos = atomicAdd(&st[q].count, 1u);
            if (pos < cap) out[(uint64_t)q * cap + pos] = x;
        }
    }
}
// The selected records of every query ranked like the candidate selection ranks them (idf descending, ties by ascending structure id,
// query_pdb.rs:404-411) and cut to top_n: one workgroup per query, bitonic sort of (inverted idf key << 32 | nid) in LDS.  A query whose
// selection overflowed its slots (count > cap) is left to the host.
#define TOPN_SORT_MAX 4096
__global__ __launch_bounds__(256) void k_topn_sort(const fd_count_rec_dev *__restrict__ sel, uint32_t cap, const topn_state *__restrict__ st, uint32_t top_n,
                                                   fd_count_rec_dev *__restrict__ out) {
    __shared__ uint64_t key[TOPN_SORT_MAX];
    __shared__ uint16_t idx[TOPN_SORT_MAX];
    const uint32_t q = blockIdx.x, cnt = st[q].count;
    if (cnt > cap || cnt == 0) return;
    uint32_t n2 = 1;
    while (n2 < cnt) n2 <<= 1;
    const fd_count_rec_dev *r = sel + (uint64_t)q * cap;
    for (uint32_t i = threadIdx.x; i < n2; i += 256) {
        key[i] = i < cnt ? (((uint64_t)(~idf_order_key(r[i].idf)) << 32) | r[i].nid) : ~0ull;
        idx[i] = (uint16_t)i;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n2; i += 256) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const uint64_t a = key[i], b = key[l];
                    if ((a > b) == up) { key[i] = b; key[l] = a; const uint16_t t = idx[i]; idx[i] = idx[l]; idx[l] = t; }
                }
            }
            __syncthreads();
        }
    const uint32_t m = cnt < top_n ? cnt : top_n;
    for (uint32_t i = threadIdx.x; i < m; i += 256) out[(uint64_t)q * top_n + i] = r[idx[i]];
}
void fd_launch_cq_topn_sort(const void *sel, uint32_t cap, const void *state, uint32_t n_queries, uint32_t top_n, void *out, hipStream_t st) {
    if (n_queries) hipLaunchKernelGGL(k_topn_sort, dim3(n_queries), dim3(256), 0, st, (const fd_count_rec_dev *)sel, cap, (const topn_state *)state, top_n,
                                      (fd_count_rec_dev *)out);
}

// Candidate selection of whole queries in the packed form, without the dense per-structure results: only what RANKS a structure is
// computed for all of them, the record (counts, idf) only for the survivors.
//   k_cq_rows_keys     the word-major walk over the occupancy rows of k_cq_rows_finalize, sums only: per structure the ranking key
//                      (order-preserving image of idf sum x penalty; 0 = untouched), 4 bytes instead of the 17 of accumulator, node
//                      count, edge count and flag — plus the first-level histogram of the radix select (LDS, 16-bit counters, bins
//                      of non-negative idf) merged into the global table
//   k_topn_hist_dense  second level over the keys (four loads in flight per thread)
//   k_topn_emit_dense  lists the survivors in LDS, then all threads build their records at once: a survivor walks its query's rows
//                      (tens of independent loads for a motif query) for match / edge / node counts and the exact idf sum
// Fewer touched structures than top_n: the threshold search ends in bin 0 and everything touched is emitted.
struct topn_dense {
    const uint32_t *keys; const float *penalty; const uint32_t *hash_bits; const unsigned long long *row_meta; const uint64_t *q_rows;
    uint32_t words, S, first_id;
};
__global__ __launch_bounds__(CQ_FIN_T) void k_cq_rows_keys(topn_dense D, uint32_t *__restrict__ keys, uint32_t *__restrict__ ghist) {
    __shared__ unsigned long long s_sum[CQ_FIN_T * 33];
    __shared__ uint32_t s_hist[512];       // bins 1024..2047 of the 2048 (idf >= 0), two 16-bit counters per word (<= 4,096 structures per workgroup)
    const uint32_t w = blockIdx.x * CQ_FIN_T + threadIdx.x, qy = blockIdx.y;
    const bool live = w < D.words;
    const uint64_t r0 = D.q_rows[qy], r1 = D.q_rows[qy + 1];
    const uint32_t nid0 = blockIdx.x * CQ_FIN_T * 32;
    const uint32_t lim = nid0 < D.S ? (D.S - nid0 < CQ_FIN_T * 32 ? D.S - nid0 : CQ_FIN_T * 32) : 0u;
    float pen[32];      // the penalties of the 32 structures this thread writes keys for (structure-major), requested up front
#pragma unroll
    for (int k = 0; k < 32; ++k) { const uint32_t i = k * CQ_FIN_T + threadIdx.x; pen[k] = i < lim ? D.penalty[nid0 + i] : 0.0f; }
    for (int k = threadIdx.x; k < 512; k += CQ_FIN_T) s_hist[k] = 0;
    unsigned long long *mine = s_sum + threadIdx.x * 33;
    for (int b = 0; b < 32; ++b) mine[b] = 0ull;
    uint32_t touched = 0;
    uint64_t r = r0;
    for (; r + 8 <= r1; r += 8) {       // eight rows' words in flight
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = live ? D.hash_bits[(r + u) * D.words + w] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t y = x[u];
            if (!y) continue;
            touched |= y;
            const unsigned long long fix = D.row_meta[r + u] >> 2;
            while (y) { const int b = __builtin_ctz(y); mine[b] += fix; y &= y - 1u; }
        }
    }
    for (; r < r1; ++r) {
        uint32_t y = live ? D.hash_bits[r * D.words + w] : 0u;
        if (!y) continue;
        touched |= y;
        const unsigned long long fix = D.row_meta[r] >> 2;
        while (y) { const int b = __builtin_ctz(y); mine[b] += fix; y &= y - 1u; }
    }
    while (touched) { const int b = __builtin_ctz(touched); mine[b] |= 1ull << 63; touched &= touched - 1u; }     // a matched row may carry idf 0
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const uint32_t i = k * CQ_FIN_T + threadIdx.x;
        if (i < lim) {
            const unsigned long long a = s_sum[(i >> 5) * 33 + (i & 31u)];
            uint32_t key = 0;
            if (a >> 63) {
                key = idf_order_key((float)((double)(a & CQ_SUM_MASK) * (1.0 / IDF_SCALE)) * pen[k]);
                const uint32_t bin = key >> 21;
                if (bin >= 1024) atomicAdd(&s_hist[(bin - 1024) >> 1], 1u << ((bin & 1u) * 16));
                else atomicAdd(&ghist[(uint64_t)qy * TOPN_BINS + bin], 1u);      // a negative idf (negative penalty): never in practice
            }
            keys[(uint64_t)qy * D.S + nid0 + i] = key;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += CQ_FIN_T) {
        const uint32_t cn = (s_hist[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
        if (cn) atomicAdd(&ghist[(uint64_t)qy * TOPN_BINS + 1024 + k], cn);
    }
}
__global__ __launch_bounds__(256) void k_topn_hist_dense(topn_dense D, const topn_state *__restrict__ st, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t hist[TOPN_BINS];
    const uint32_t q = blockIdx.y;
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t b1 = st[q].thr_bin;
    const uint32_t *kq = D.keys + (uint64_t)q * D.S;
    for (uint32_t i0 = blockIdx.x * 1024 + threadIdx.x; i0 < D.S; i0 += TOPN_SPLIT * 1024) {
        uint32_t k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k4[u] = i0 + u * 256 < D.S ? kq[i0 + u * 256] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k4[u] && (k4[u] >> 21) == b1) atomicAdd(&hist[(k4[u] >> 10) & (TOPN_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) if (hist[k]) atomicAdd(&ghist[(uint64_t)q * TOPN_BINS + k], hist[k]);
}
__global__ __launch_bounds__(256) void k_topn_emit_dense(topn_dense D, uint32_t cap, topn_state *__restrict__ st, fd_count_rec_dev *__restrict__ out) {
    __shared__ uint32_t s_idx[2048];
    __shared__ uint32_t s_n, s_base;
    const uint32_t q = blockIdx.y;
    const uint32_t thr22 = st[q].thr22;
    const uint64_t r0 = D.q_rows[q], r1 = D.q_rows[q + 1];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    auto drain = [&]() {
        const uint32_t n = s_n;
        if (threadIdx.x == 0 && n) s_base = atomicAdd(&st[q].count, n);
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < n; e += 256) {
            const uint32_t i = s_idx[e], pos = s_base + e;
            if (pos >= cap) continue;
            // the record of structure i from its query's rows, in (node, partner) order like k_cq_rows_finalize
            const uint32_t *col = D.hash_bits + (i >> 5);
            const uint32_t sh = i & 31u;
            uint32_t cnt = 0, edges = 0, nodes = 0;
            bool e_any = false, n_any = false;
            unsigned long long sum = 0;
            auto row = [&](uint32_t x, unsigned long long m) {
                if ((x >> sh) & 1u) { ++cnt; e_any = true; sum += m >> 2; }
                if (m & 1ull) { if (e_any) { ++edges; n_any = true; } e_any = false; }
                if (m & 2ull) { if (n_any) ++nodes; n_any = false; }
            };
            uint64_t r = r0;
            for (; r + 8 <= r1; r += 8) {
                uint32_t x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = col[(r + u) * D.words];
#pragma unroll
                for (int u = 0; u < 8; ++u) row(x[u], D.row_meta[r + u]);
            }
            for (; r < r1; ++r) row(col[r * D.words], D.row_meta[r]);
            fd_count_rec_dev rec;
            rec.nid = i + D.first_id; rec.total_match_count = cnt; rec.node_count = nodes; rec.edge_count = edges;
            rec.idf = (float)((double)sum * (1.0 / IDF_SCALE)) * D.penalty[i];      // count_query.rs:200 idf_sum *= nres^(-lp)
            out[(uint64_t)q * cap + pos] = rec;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    const uint32_t *kq = D.keys + (uint64_t)q * D.S;
    for (uint32_t j0 = blockIdx.x * 1024; j0 < D.S; j0 += TOPN_SPLIT * 1024) {       // block-uniform trip count; four key loads in flight
        uint32_t k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t i = j0 + u * 256 + threadIdx.x; k4[u] = i < D.S ? kq[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k4[u] && (k4[u] >> 10) >= thr22) s_idx[atomicAdd(&s_n, 1u)] = j0 + u * 256 + threadIdx.x;     // <= 1,024 per trip, drained beyond 1,024
        __syncthreads();
        if (s_n > 1024) drain();
    }
    drain();
}
// keys: [n_queries][S] u32 scratch; ghist: [n_queries][2048], zero on entry and left zero; q_rows: device [n_queries + 1] row ranges
void fd_launch_cq_topn_dense(const cq_args &A, const uint64_t *q_rows, const float *penalty, uint32_t *keys, uint32_t n_queries, uint32_t top_n, uint32_t cap,
                             void *out, void *state, uint32_t *ghist, hipStream_t st) {
    if (!n_queries || !A.S) return;
    topn_dense D;
    D.keys = keys; D.penalty = penalty; D.hash_bits = A.hash_bits; D.row_meta = A.row_meta; D.q_rows = q_rows; D.words = A.words; D.S = A.S; D.first_id = A.first_id;
    topn_state *ts = (topn_state *)state;
    (void)hipMemsetAsync(ts, 0, (size_t)n_queries * sizeof(topn_state), st);
    hipLaunchKernelGGL(k_cq_rows_keys, dim3((A.words + CQ_FIN_T - 1) / CQ_FIN_T, n_queries), dim3(CQ_FIN_T), 0, st, D, keys, ghist);
    const dim3 g(TOPN_SPLIT, n_queries);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, (const uint64_t *)nullptr, top_n, 0, ts, ghist);
    hipLaunchKernelGGL(k_topn_hist_dense, g, dim3(256), 0, st, D, ts, ghist);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, (const uint64_t *)nullptr, top_n, 1, ts, ghist);
    hipLaunchKernelGGL(k_topn_emit_dense, g, dim3(256), 0, st, D, cap, ts, (fd_count_rec_dev *)out);
}

// The selection straight from the dense [queries][S] accumulators of k_cq_rows_finalize (packed form) — the path of ONE query with
// thousands of rows, whose finalize runs in row slices (whole-structure mode): no flag scan, no compaction of every touched
// structure, two synchronisations fewer — the survivors' records are built at emit time.  Fewer touched structures than top_n: the
// threshold search ends in bin 0 and everything with a count is emitted.
struct topn_acc { const unsigned long long *acc; const float *penalty; const uint32_t *node_cnt, *edge_cnt; uint32_t S, first_id; };
__device__ __forceinline__ bool topn_acc_key(const topn_acc &D, uint32_t q, uint32_t nid, uint32_t *key, float *idf, uint32_t *cnt) {
    const unsigned long long a = D.acc[(uint64_t)q * D.S + nid];
    *cnt = (uint32_t)(a >> CQ_CNT_SHIFT);
    if (!*cnt) return false;
    const float sum = (float)((double)(a & CQ_SUM_MASK) * (1.0 / IDF_SCALE));
    *idf = sum * D.penalty[nid];
    *key = idf_order_key(*idf);
    return true;
}
__global__ __launch_bounds__(256) void k_topn_hist_acc(topn_acc D, int level, const topn_state *__restrict__ st, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t hist[TOPN_BINS];
    const uint32_t q = blockIdx.y;
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t b1 = level ? st[q].thr_bin : 0u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < D.S; i += TOPN_SPLIT * 256) {
        uint32_t key, cnt; float idf;
        if (!topn_acc_key(D, q, i, &key, &idf, &cnt)) continue;
        if (!level) atomicAdd(&hist[key >> 21], 1u);
        else if ((key >> 21) == b1) atomicAdd(&hist[(key >> 10) & (TOPN_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TOPN_BINS; k += 256) if (hist[k]) atomicAdd(&ghist[(uint64_t)q * TOPN_BINS + k], hist[k]);
}
__global__ __launch_bounds__(256) void k_topn_emit_acc(topn_acc D, uint32_t cap, topn_state *__restrict__ st, fd_count_rec_dev *__restrict__ out) {
    const uint32_t q = blockIdx.y;
    const uint32_t thr22 = st[q].thr22;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = blockIdx.x * 256; i0 < D.S; i0 += TOPN_SPLIT * 256) {       // wave-uniform trip count: one atomic per wavefront and step
        const uint32_t i = i0 + threadIdx.x;
        uint32_t key = 0, cnt = 0; float idf = 0.0f;
        const bool keep = i < D.S && topn_acc_key(D, q, i, &key, &idf, &cnt) && (key >> 10) >= thr22;
        const uint64_t m = __ballot(keep);
        if (!m) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&st[q].count, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), FD_WAVE);
        const uint32_t pos = base + fd_mbcnt(m);
        if (keep && pos < cap) {
            fd_count_rec_dev r;
            r.nid = i + D.first_id; r.total_match_count = cnt;
            r.node_count = D.node_cnt[(uint64_t)q * D.S + i]; r.edge_count = D.edge_cnt[(uint64_t)q * D.S + i]; r.idf = idf;
            out[(uint64_t)q * cap + pos] = r;
        }
    }
}
void fd_launch_cq_topn_acc(const cq_args &A, const float *penalty, const uint32_t *node_cnt, const uint32_t *edge_cnt, uint32_t n_queries, uint32_t top_n,
                             uint32_t cap, void *out, void *state, uint32_t *ghist, hipStream_t st) {
    if (!n_queries) return;
    topn_acc D;
    D.acc = A.idf; D.penalty = penalty; D.node_cnt = node_cnt; D.edge_cnt = edge_cnt; D.S = A.S; D.first_id = A.first_id;
    topn_state *ts = (topn_state *)state;
    (void)hipMemsetAsync(ts, 0, (size_t)n_queries * sizeof(topn_state), st);
    const dim3 g(TOPN_SPLIT, n_queries);
    hipLaunchKernelGGL(k_topn_hist_acc, g, dim3(256), 0, st, D, 0, ts, ghist);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, (const uint64_t *)nullptr, top_n, 0, ts, ghist);
    hipLaunchKernelGGL(k_topn_hist_acc, g, dim3(256), 0, st, D, 1, ts, ghist);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, (const uint64_t *)nullptr, top_n, 1, ts, ghist);
    hipLaunchKernelGGL(k_topn_emit_acc, g, dim3(256), 0, st, D, cap, ts, (fd_count_rec_dev *)out);
}

// state: n_queries topn_state + n_queries * 2048 u32 (zeroed once by the caller; the kernels leave the table zero)
void fd_launch_cq_topn(const void *recs, const uint64_t *off, uint32_t n_queries, uint32_t top_n, uint32_t cap, void *out, void *state, uint32_t *ghist,
                       hipStream_t st) {
    if (!n_queries) return;
    const fd_count_rec_dev *r = (const fd_count_rec_dev *)recs;
    topn_state *ts = (topn_state *)state;
    const dim3 g(TOPN_SPLIT, n_queries);
    hipLaunchKernelGGL(k_topn_hist, g, dim3(256), 0, st, r, off, top_n, 0, ts, ghist);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, off, top_n, 0, ts, ghist);
    hipLaunchKernelGGL(k_topn_hist, g, dim3(256), 0, st, r, off, top_n, 1, ts, ghist);
    hipLaunchKernelGGL(k_topn_thr, dim3(n_queries), dim3(256), 0, st, off, top_n, 1, ts, ghist);
    hipLaunchKernelGGL(k_topn_emit, g, dim3(256), 0, st, r, off, cap, ts, (fd_count_rec_dev *)out);
}

// q_rows: device [n_queries + 1] row ranges (null: one query over all A.nq rows); slices: device [n_slices + 1] row boundaries at node
// boundaries for ONE query with many rows (results are then accumulated into zeroed arrays), or null
void fd_launch_cq_rows_finalize(const cq_args &A, const uint64_t *q_rows, uint32_t n_queries, const uint64_t *slices, uint32_t n_slices, uint32_t *node_cnt,
                                uint32_t *edge_cnt, uint8_t *flags, uint64_t max_rows_per_query, hipStream_t st) {
    if (!A.S || !n_queries) return;
    const bool add = slices && n_slices > 1 && n_queries == 1;
    if (add) {
        const size_t n = (size_t)A.S;
        (void)hipMemsetAsync(A.idf, 0, n * 8, st); (void)hipMemsetAsync(node_cnt, 0, n * 4, st); (void)hipMemsetAsync(edge_cnt, 0, n * 4, st);
        (void)hipMemsetAsync(flags, 0, n, st);
        if (!A.packed) (void)hipMemsetAsync(A.match, 0, n * 4, st);
    }
    const dim3 grid((A.words + CQ_FIN_T - 1) / CQ_FIN_T, n_queries, add ? n_slices : 1);
    static const bool fin_alias = [] { const char *e = getenv("FDGPU_FIN_ALIAS"); return !(e && e[0] == '0'); }();      // 0: the separate count tile (A/B)
    if (max_rows_per_query < 256)
        hipLaunchKernelGGL(k_cq_rows_finalize<8>, grid, dim3(CQ_FIN_T), 0, st, A.hash_bits, A.row_meta, q_rows, A.nq, add ? slices : nullptr, A.words, A.S, A.packed,
                           A.match, A.idf, node_cnt, edge_cnt, flags);
    else if (add && A.packed && fin_alias)
        hipLaunchKernelGGL((k_cq_rows_finalize<20, true>), grid, dim3(CQ_FIN_T), 0, st, A.hash_bits, A.row_meta, q_rows, A.nq, add ? slices : nullptr, A.words, A.S, A.packed,
                           A.match, A.idf, node_cnt, edge_cnt, flags);
    else
        hipLaunchKernelGGL(k_cq_rows_finalize<20>, grid, dim3(CQ_FIN_T), 0, st, A.hash_bits, A.row_meta, q_rows, A.nq, add ? slices : nullptr, A.words, A.S, A.packed,
                           A.match, A.idf, node_cnt, edge_cnt, flags);
}
void fd_launch_cq_compact_batch(const cq_args &A, const uint32_t *node_cnt, const uint32_t *edge_cnt, const uint8_t *flags, const uint64_t *pos,
                                const float *penalty, uint64_t total, void *out, hipStream_t st) {
    if (total) hipLaunchKernelGGL(k_cq_compact_batch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, A.packed ? nullptr : A.match, A.idf, node_cnt, edge_cnt, flags,
                                  pos, penalty, A.S, total, A.first_id, (fd_count_rec_dev *)out);
}

// Posting length of EVERY list of an index (ids per hash = bytes without the continuation bit), one wavefront per list, four per
// workgroup: computed once per index, on the first length request — a batch of queries then looks its lengths up (k_pl_lookup) instead
// of reading its hashes' posting bytes a third time (length pass, segment sums, scoring)
__global__ __launch_bounds__(256) void k_index_lens(const uint64_t *__restrict__ offsets, const uint8_t *__restrict__ value, uint64_t H, uint32_t *__restrict__ lens) {
    const uint64_t t = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= H) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = offsets[t], b1 = offsets[t + 1];
    uint32_t cnt = 0;
    for (uint64_t p = b0 + (uint64_t)lane * 16; p < b1; p += 64 * 16) {
        if (p + 16 <= b1) {
            unsigned long long w0, w1;
            __builtin_memcpy(&w0, value + p, 8); __builtin_memcpy(&w1, value + p + 8, 8);
            cnt += 16u - (uint32_t)__popcll(w0 & 0x8080808080808080ull) - (uint32_t)__popcll(w1 & 0x8080808080808080ull);
        } else for (uint64_t z = p; z < b1; ++z) cnt += (value[z] & 0x80u) ? 0u : 1u;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, FD_WAVE);
    if (lane == 0) lens[t] = cnt;
}
void fd_launch_index_lens(const uint64_t *offsets, const uint8_t *value, uint64_t H, uint32_t *lens, hipStream_t st) {
    if (H) hipLaunchKernelGGL(k_index_lens, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, st, offsets, value, H, lens);
}
__global__ void k_pl_lookup(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lens, uint64_t H,
                            const uint32_t *__restrict__ q_hash, uint64_t nq, unsigned long long *__restrict__ lengths, uint32_t *__restrict__ nseg,
                            long long *__restrict__ kidx) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t k = find_hash(hashes, H, q_hash[q]);
    kidx[q] = k;
    lengths[q] = k < 0 ? 0ull : (unsigned long long)lens[k];
    nseg[q] = k < 0 ? 0u : (uint32_t)((offsets[k + 1] - offsets[k] + CQ_SEG - 1) / CQ_SEG);
}
void fd_launch_posting_lookup(const uint32_t *hashes, const uint64_t *offsets, const uint32_t *lens, uint64_t H, const uint32_t *q_hash, uint64_t nq, uint64_t *lengths,
                              uint32_t *nseg, long long *kidx, hipStream_t st) {
    if (nq) hipLaunchKernelGGL(k_pl_lookup, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, hashes, offsets, lens, H, q_hash, nq, (unsigned long long *)lengths, nseg, kidx);
}

// kidx / nseg / wstart / scan_tmp / total: plan workspaces for nq hashes (k_cq_plan + exclusive scan)
void fd_launch_posting_lengths(const uint32_t *hashes, const uint64_t *offsets, const uint8_t *value, uint64_t H, const uint32_t *q_hash,
                               uint64_t nq, uint64_t *lengths, long long *kidx, uint32_t *nseg, uint64_t *wstart, uint64_t *scan_tmp, uint64_t *total,
                               hipStream_t st) {
    if (!nq) return;
    (void)hipMemsetAsync(lengths, 0, nq * 8, st);
    hipLaunchKernelGGL(k_cq_plan, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, hashes, offsets, H, q_hash, nq, kidx, nseg);
    fd_exclusive_scan<uint32_t>(nseg, nq, wstart, scan_tmp, total, st);
    hipLaunchKernelGGL(k_pl_count, dim3(8192), dim3(FD_WAVE), 0, st, offsets, value, kidx, wstart, nq, (unsigned long long *)lengths);
}
void fd_launch_cq_compact(const uint32_t *match, const unsigned long long *idf, const uint32_t *node_cnt, const uint32_t *edge_cnt,
                          const uint8_t *flags, const uint64_t *pos, const float *penalty, uint32_t S, uint32_t first_id, void *out,
                          hipStream_t st) {
    if (S) hipLaunchKernelGGL(k_cq_compact, dim3((S + 255) / 256), dim3(256), 0, st, match, idf, node_cnt, edge_cnt, flags, pos, penalty, S, first_id, (fd_count_rec_dev *)out);
}
