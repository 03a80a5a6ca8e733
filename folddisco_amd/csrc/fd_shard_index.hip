// fd_shard_index.hip — ONE on-disk index from N ranks without the host (Option A).
//
// After a build sharded by structure every rank holds a resident sub-index of its id range (complete posting lists, restricted to its ids).  The
// reference's output contract is ONE PREFIX / PREFIX.offset pair in ascending hash order (src/index/indextable.rs:239-326).  Rounds 2-4 produced it
// on rank 0's host (fdgpu_merge_subindices: a streaming merge of all exported shards).  Here the hash space is cut into N ranges of about equal
// posting bytes (fdgpu_index_range_bounds on one rank's index, shared with the others), every rank slices its sub-index at those bounds
// (fdgpu_index_slice: lists are stored in ascending hash order, so a range is one contiguous piece of each array), piece j travels to rank j
// (RCCL send / recv in fd_comm.hip; tests and the gloo launch use another transport), rank j concatenates the N pieces of its range per hash with
// the existing device merge (fdgpu_index_merge: id ranges ascend with the source rank) and writes its slice of the two files at its own offsets
// (fdgpu_index_save_part) — the files are byte-identical to the single-GPU build because lists are (hash ascending, id ascending) either way.
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <cstring>
#include <atomic>
#include <string>
#include "fdgpu_internal.h"

#define FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define HIPCHK(ctx, expr)                                                                                                   \
    do {                                                                                                                    \
        hipError_t _e = (expr);                                                                                             \
        if (_e != hipSuccess) { (ctx)->err = std::string(#expr " -> ") + hipGetErrorString(_e); return FDGPU_EHIP; }        \
    } while (0)

// first list whose byte offset reaches k / n of the value bytes, k = 1 .. n - 1 -> its hash (0xffffffff behind the last list): n ranges of about
// equal posting bytes
__global__ void k_si_bounds(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets, uint64_t H, uint32_t n_ranges, uint32_t *__restrict__ bounds) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x + 1u;
    if (k >= n_ranges) return;
    const uint64_t target = (uint64_t)((unsigned __int128)offsets[H] * k / n_ranges);
    uint64_t lo = 0, hi = H;          // first index with offsets[idx] >= target
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] < target) lo = mid + 1; else hi = mid; }
    bounds[k - 1u] = lo < H ? hashes[lo] : 0xffffffffu;
}
// out[0] = first list with hash >= lo, out[1] = first list with hash >= hi (hi = 2^32: H), out[2] / out[3] = their byte offsets
__global__ void k_si_find(const uint32_t *__restrict__ hashes, const uint64_t *__restrict__ offsets, uint64_t H, uint64_t h_lo, uint64_t h_hi, uint64_t *__restrict__ out) {
    if (threadIdx.x > 1 || blockIdx.x) return;
    const uint64_t key = threadIdx.x ? h_hi : h_lo;
    uint64_t lo = 0, hi = H;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)hashes[mid] < key) lo = mid + 1; else hi = mid; }
    out[threadIdx.x] = lo;
    out[2 + threadIdx.x] = offsets[lo];
}
__global__ void k_si_rebase(const uint64_t *__restrict__ src, uint64_t n, uint64_t base, uint64_t *__restrict__ dst) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[k] - base;
}
__global__ __launch_bounds__(256) void k_si_count_postings(const uint8_t *__restrict__ value, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (uint64_t p = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; p < n; p += (uint64_t)gridDim.x * 256 * 16)
        for (uint64_t k = p; k < n && k < p + 16; ++k) acc += !(value[k] & 0x80u);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

extern "C" int fdgpu_index_range_bounds(fdgpu_ctx *c, const fdgpu_index *ix, uint32_t n_ranges, uint32_t *bounds) { FD_LOCK(c);
    if (!c || !ix || !n_ranges || (n_ranges > 1 && !bounds)) return FDGPU_EINVAL;
    if (n_ranges == 1) return FDGPU_OK;
    if (!ix->n_hashes) { for (uint32_t k = 0; k + 1 < n_ranges; ++k) bounds[k] = 0xffffffffu; return FDGPU_OK; }
    HIPCHK(c, c->ws[WS_TOTAL].ensure(std::max<size_t>(64, (size_t)n_ranges * 4)));
    hipLaunchKernelGGL(k_si_bounds, dim3((n_ranges + 63) / 64), dim3(64), 0, c->stream, ix->hashes, ix->offsets, ix->n_hashes, n_ranges, c->ws[WS_TOTAL].as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(bounds, c->ws[WS_TOTAL].p, (size_t)(n_ranges - 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDGPU_OK;
}

// the lists of ix with hash_lo <= hash < hash_hi (hash_hi up to 2^32) as a resident index of their own: same id range, offsets re-based
extern "C" int fdgpu_index_slice(fdgpu_ctx *c, const fdgpu_index *ix, uint64_t hash_lo, uint64_t hash_hi, fdgpu_index **out) { FD_LOCK(c);
    if (!c || !ix || !out || hash_lo > hash_hi || hash_hi > (1ull << 32)) return FDGPU_EINVAL;
    *out = nullptr;
    hipStream_t st = c->stream;
    uint64_t f[4] = {0, 0, 0, 0};
    HIPCHK(c, c->ws[WS_TOTAL].ensure(64));
    if (ix->n_hashes) {
        hipLaunchKernelGGL(k_si_find, dim3(1), dim3(64), 0, st, ix->hashes, ix->offsets, ix->n_hashes, hash_lo, hash_hi, c->ws[WS_TOTAL].as<uint64_t>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(f, c->ws[WS_TOTAL].p, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    const uint64_t i0 = f[0], i1 = f[1], b0 = f[2], b1 = f[3], H = i1 - i0, V = b1 - b0;
    fdgpu_index *s = new (std::nothrow) fdgpu_index();
    if (!s) return FDGPU_ENOMEM;
    s->ctx = c; s->n_hashes = H; s->value_len = V; s->n_structures = ix->n_structures; s->first_id = ix->first_id;
    hipError_t e;
    s->hashes = (uint32_t *)c->pool_alloc(std::max<uint64_t>(H, 1) * 4, &e); s->cap_hashes = c->last_cap;
    if (e == hipSuccess) { s->offsets = (uint64_t *)c->pool_alloc((H + 1) * 8, &e); s->cap_offsets = c->last_cap; }
    if (e == hipSuccess) { s->value = (uint8_t *)c->pool_alloc(V + 16, &e); s->cap_value = c->last_cap; }
    if (e == hipSuccess && ix->last_ids) { s->last_ids = (uint32_t *)c->pool_alloc(std::max<uint64_t>(H, 1) * 4, &e); s->cap_last = c->last_cap; }
    if (e == hipSuccess && H) e = hipMemcpyAsync(s->hashes, ix->hashes + i0, H * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && H && s->last_ids) e = hipMemcpyAsync(s->last_ids, ix->last_ids + i0, H * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && V) e = hipMemcpyAsync(s->value, ix->value + b0, V, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {
        if (ix->n_hashes) hipLaunchKernelGGL(k_si_rebase, dim3((unsigned)((H + 1 + 255) / 256)), dim3(256), 0, st, ix->offsets + i0, H + 1, b0, s->offsets);
        else e = hipMemsetAsync(s->offsets, 0, 8, st);
        if (e == hipSuccess) e = hipGetLastError();
    }
    unsigned long long np = 0;
    if (e == hipSuccess && V) {
        e = hipMemsetAsync(c->ws[WS_TOTAL].p, 0, 8, st);
        if (e == hipSuccess) { hipLaunchKernelGGL(k_si_count_postings, dim3(2048), dim3(256), 0, st, s->value, V, c->ws[WS_TOTAL].as<unsigned long long>()); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(&np, c->ws[WS_TOTAL].p, 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { c->err = std::string("index slice: ") + hipGetErrorString(e); fdgpu_index_destroy(s); return FDGPU_EHIP; }
    s->n_postings = np;
    *out = s;
    return FDGPU_OK;
}

// This rank's hash range of the database's single index -> its regions of PREFIX and PREFIX.offset:
//   PREFIX         [value_before, value_before + V)                                   the range's posting bytes
//   PREFIX.offset  8 + 4 * hashes_before ...                                          its hashes
//                  8 + 4 * total_hashes + 8 * hashes_before ...                       its offsets + value_before (H entries; the rank of the LAST range also
//                                                                                     writes offsets[total_hashes] = total_value)
//                  0 ... 8                                                            u64 total_hashes (the rank with hashes_before == 0 and write_header != 0)
// Every rank opens the two files O_CREAT without truncation and sets their final lengths (the same on all ranks), so the calls may run concurrently on
// the ranks of one node; bytes outside a rank's regions are never touched.  Byte-identical to fdgpu_index_save of the whole index.
extern "C" int fdgpu_index_save_part(fdgpu_ctx *c, const fdgpu_index *part, const char *prefix, uint64_t hashes_before, uint64_t value_before, uint64_t total_hashes,
                                     uint64_t total_value, int write_header, int is_last) { FD_LOCK(c);
    if (!c || !part || !prefix || hashes_before + part->n_hashes > total_hashes || value_before + part->value_len > total_value) return FDGPU_EINVAL;
    const std::string p(prefix);
    const uint64_t H = part->n_hashes;
    std::atomic<int> io_err{0};
    const int fv = open(p.c_str(), O_WRONLY | O_CREAT, 0644);
    const int fo = fv >= 0 ? open((p + ".offset").c_str(), O_WRONLY | O_CREAT, 0644) : -1;
    if (fv < 0 || fo < 0) { if (fv >= 0) close(fv); FAIL(c, FDGPU_EINVAL, "index save: cannot write " + p); }
    if (ftruncate(fv, (off_t)total_value) != 0 || ftruncate(fo, (off_t)(8 + 4 * total_hashes + 8 * (total_hashes + 1))) != 0) io_err = errno ? errno : EIO;
    if (write_header && pwrite(fo, &total_hashes, 8, 0) != 8) io_err = errno ? errno : EIO;
    hipError_t e = fd_d2h_to_file(c, fv, value_before, part->value, part->value_len, &io_err);
    if (e == hipSuccess) e = fd_d2h_to_file(c, fo, 8 + 4 * hashes_before, part->hashes, H * 4, &io_err);
    // offsets + value_before: re-based on the device into the sort workspace's first block, then streamed like the rest
    if (e == hipSuccess) e = c->ws[WS_MISC3].ensure((H + 1) * 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_si_rebase, dim3((unsigned)((H + 1 + 255) / 256)), dim3(256), 0, c->stream, part->offsets, H + 1, (uint64_t)0 - value_before, c->ws[WS_MISC3].as<uint64_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = fd_d2h_to_file(c, fo, 8 + 4 * total_hashes + 8 * hashes_before, c->ws[WS_MISC3].p, (H + (is_last ? 1 : 0)) * 8, &io_err);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (close(fv) != 0 && !io_err) io_err = errno ? errno : EIO;
    if (close(fo) != 0 && !io_err) io_err = errno ? errno : EIO;
    if (e != hipSuccess) { c->err = std::string("index save part: ") + hipGetErrorString(e); return FDGPU_EHIP; }
    if (io_err) FAIL(c, FDGPU_EINVAL, "index save: cannot write " + p + " (" + strerror(io_err) + ")");
    return FDGPU_OK;
}
