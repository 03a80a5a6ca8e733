// k_hash.hip — residue-pair enumeration + PDBTrRosetta hashing on gfx950.
//
// Replaces HOT LOOP A of the reference (src/controller/feature.rs:198-231 driven by
// src/utils/combination.rs:23-44): all ordered residue pairs (i, j), i != j, both residues
// hashable, CA-CA distance <= cutoff.
//
// Mapping: one 64-lane wavefront (= one 64-thread workgroup) per (structure, 64-residue i-tile).
// Lane l owns residue i = tile_start + l and walks j over the whole structure with
// wave-uniform (scalar) loads of CA_j, so the CA-distance filter costs ~15 VALU issues per 64
// pair tests.  Only ~25 % of the tests pass, so running the ~700-instruction descriptor under
// that exec mask would idle 3/4 of the lanes; instead the passing (i, j) are compacted with
// ballot + mbcnt prefix into a per-wave LDS queue and the descriptor is evaluated in full
// 64-entry drains (one queue entry per lane).  No MFMA: this is f32/f64 VALU + irregular gather.
#include "fd_device.h"
#include "fd_geom_other.h"
// waves per SIMD the pair kernel is compiled for: 5 (88 VGPRs, 48 B of scratch) measured 15.0 ms against 16.2 (4 waves,
// 114 VGPRs) and 16.1 (6 waves, 80 VGPRs, 96 B of scratch)
#ifndef FD_EMIT_WAVES
#define FD_EMIT_WAVES 5
#endif
#define FD_MSD_BUCKETS 40u   // MSD build: bucket = top six hash bits = aa_i << 1 | aa_j >> 4 (see drain2)

// ------------------------------------------------------------------ count pass
// counts[s] += number of ordered pairs of structure s that will be emitted
__global__ __launch_bounds__(FD_WAVE) void k_pair_count(fd_batch_view B, fd_hash_consts C, uint32_t *__restrict__ counts) {
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    const uint32_t i = B.wi_i0[w] + threadIdx.x;
    const bool vi = i < r1 && B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (vi) cai = fd_load3(B.ca_xyz, i);
    uint32_t cnt = 0;
    for (uint32_t j = r0; j < r1; ++j) {
        if (!B.hash_ok[j]) continue;  // wave-uniform
        fd_v3 caj = fd_load3(B.ca_xyz, j);
        float d2 = fd_dist2(cai, caj);
        cnt += (vi && i != j && !(d2 > C.d2_max)) ? 1u : 0u;
    }
    // wave reduction
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, FD_WAVE);
    if (threadIdx.x == 0 && cnt) atomicAdd(&counts[s], cnt);
}

// ------------------------------------------------------------------ emit pass
// keys/ids are written into the structure's segment [seg_off[s], seg_off[s+1]); the order inside a
// segment is unspecified (slots are claimed with one atomicAdd per 64-entry drain) — the build
// sorts by hash afterwards, and a stable sort keeps ids ascending because segments are id-major.
template <bool WRITE_IDS>
__device__ __forceinline__ void drain(const fd_batch_view &B, const fd_hash_consts &C, const uint32_t *q, uint32_t n,
                                      uint32_t i0, uint32_t r0, uint32_t s, uint32_t id, const uint64_t *seg_off,
                                      uint32_t *cursor, uint32_t *keys, uint32_t *ids) {
    const uint32_t lane = threadIdx.x;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&cursor[s], n);
    base = __shfl(base, 0, FD_WAVE);
    if (lane < n) {
        uint32_t e = q[lane];
        uint32_t i = i0 + (e >> 16), j = r0 + (e & 0xffffu);
        fd_v3 n1 = fd_load3(B.n_xyz, i), ca1 = fd_load3(B.ca_xyz, i), cb1 = fd_load3(B.cb_xyz, i);
        fd_v3 n2 = fd_load3(B.n_xyz, j), ca2 = fd_load3(B.ca_xyz, j), cb2 = fd_load3(B.cb_xyz, j);
        fd_feature f = fd_pair_feature(n1, ca1, cb1, n2, ca2, cb2);
        uint32_t h = fd_hash_enc(B.aa[i], B.aa[j], f, C.q);
        uint64_t pos = seg_off[s] + base + lane;
        keys[pos] = h;
        if (WRITE_IDS) ids[pos] = id;
    }
}

template <bool WRITE_IDS>
__global__ __launch_bounds__(FD_WAVE) void k_pair_emit(fd_batch_view B, fd_hash_consts C, const uint64_t *__restrict__ seg_off,
                                                       uint32_t *__restrict__ cursor, uint32_t *__restrict__ keys,
                                                       uint32_t *__restrict__ ids, uint32_t first_id) {
    __shared__ uint32_t q[2 * FD_WAVE];
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    const uint32_t i0 = B.wi_i0[w];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = i0 + lane;
    const bool vi = i < r1 && B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (vi) cai = fd_load3(B.ca_xyz, i);
    uint32_t qn = 0;  // wave-uniform
    for (uint32_t j = r0; j < r1; ++j) {
        if (!B.hash_ok[j]) continue;
        fd_v3 caj = fd_load3(B.ca_xyz, j);
        float d2 = fd_dist2(cai, caj);
        bool pass = vi && i != j && !(d2 > C.d2_max);
        uint64_t m = __ballot(pass);
        if (m == 0) continue;
        if (pass) q[qn + fd_mbcnt(m)] = (lane << 16) | (j - r0);
        qn += (uint32_t)__popcll(m);
        if (qn >= FD_WAVE) {
            __syncthreads();
            qn -= FD_WAVE;
            drain<WRITE_IDS>(B, C, q + qn, FD_WAVE, i0, r0, s, first_id + s, seg_off, cursor, keys, ids);
            __syncthreads();
        }
    }
    if (qn) {
        __syncthreads();
        drain<WRITE_IDS>(B, C, q, qn, i0, r0, s, first_id + s, seg_off, cursor, keys, ids);
    }
}

// ------------------------------------------------------------------ frames + unordered-pair fast path
// Index-build path.  Per-residue frames (fd_make_frame) are computed once; the pair kernel visits
// every unordered pair {i < j} once and produces both ordered hashes with fd_pair_both(), which
// shares d_CA, d_CB, theta and the two pair-dependent normalised cross products between the two
// orientations (bit-exact, see fd_geom.h).  Same wave-per-(structure, i-tile) mapping and LDS
// compaction queue as above; a drain of n unordered pairs writes 2n keys.
__global__ __launch_bounds__(256) void k_frames(fd_batch_view B, uint32_t n_res, fd_frame *__restrict__ frames) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_res) return;
    fd_frame F;
    if (B.hash_ok[r]) F = fd_make_frame(fd_load3(B.n_xyz, r), fd_load3(B.ca_xyz, r), fd_load3(B.cb_xyz, r));
    else { F = fd_frame{}; F.ca = fd_load3(B.ca_xyz, r); }
    F.pad = __uint_as_float((uint32_t)B.aa[r]);   // the residue type rides in the frame's spare word (bit pattern, never computed with)
    frames[r] = F;
}

// wave-uniform lane index -> v_readlane_b32 (SGPR broadcast), not the ds_bpermute a general __shfl becomes
__device__ __forceinline__ float bcast_lane(float v, uint32_t k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)k));
}

__global__ __launch_bounds__(FD_WAVE) void k_pair_count2(fd_batch_view B, fd_hash_consts C, uint32_t *__restrict__ counts) {
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r1 = B.res_off[s + 1];
    const uint32_t i0 = B.wi_i0[w];
    const uint32_t i = i0 + threadIdx.x;
    const bool vi = i < r1 && B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (vi) cai = fd_load3(B.ca_xyz, i);
    uint32_t cnt = 0;
    // j is walked in blocks of 64: one coalesced load per block (lane l holds CA of j = jb + l), then the 64
    // candidates are broadcast lane by lane with v_readlane — no per-j memory latency in the filter loop
    for (uint32_t jb = i0; jb < r1; jb += FD_WAVE) {
        const uint32_t jl = jb + threadIdx.x;
        const bool jin = jl < r1;
        fd_v3 cj = {0.f, 0.f, 0.f};
        if (jin) cj = fd_load3(B.ca_xyz, jl);
        const uint64_t okm = __ballot(jin && B.hash_ok[jl]);
        const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
        for (uint32_t k = 0; k < nj; ++k) {
            if (!((okm >> k) & 1ull)) continue;  // wave-uniform
            fd_v3 caj = {bcast_lane(cj.x, k), bcast_lane(cj.y, k), bcast_lane(cj.z, k)};
            float d2 = fd_dist2(cai, caj);
            cnt += (vi && (jb + k) > i && !(d2 > C.d2_max)) ? 2u : 0u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, FD_WAVE);
    if (threadIdx.x == 0 && cnt) atomicAdd(&counts[s], cnt);
}

// Frames in AMINO-ACID ORDER (MSD build): one workgroup per structure sorts its residues by type (counting sort, hashable residues
// first; the order inside a type does not matter — the order of keys inside a structure's segment is arbitrary anyway) and writes the
// frames together with permuted copies of what the pair kernels read per residue (CA, hashable flag, type).  A 64-residue tile then
// holds ~4 residue types instead of ~15, which is what keeps the bucket runs of a drain long (k_pair_emit2<.., MSD>).
__global__ __launch_bounds__(256) void k_frames_perm(fd_batch_view B, fd_frame *__restrict__ frames, float *__restrict__ ca_perm, uint8_t *__restrict__ ok_perm,
                                                     uint8_t *__restrict__ aa_perm, unsigned long long *__restrict__ wide_flag) {
    __shared__ uint32_t cnt[32], cur[32];
    const uint32_t s = blockIdx.x;
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const uint32_t a = B.aa[r];
        atomicAdd(&cnt[(B.hash_ok[r] && a < 20u) ? a : 20u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int k = 0; k < 21; ++k) { cur[k] = run; run += cnt[k]; } }
    __syncthreads();
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const uint32_t a = B.aa[r];
        bool ok = B.hash_ok[r] != 0;
        if (ok && a >= 20u) {      // a residue type outside map_aa_to_u8's 0..19 marked hashable: forty buckets cannot hold it — the build is redone with the
            ok = false;            // structure-major 8-byte path, which hashes whatever the caller passed (wide_flag)
            if (wide_flag) atomicOr(wide_flag, 1ull);
        }
        const uint32_t p = r0 + atomicAdd(&cur[(ok && a < 20u) ? a : 20u], 1u);
        fd_frame F;
        const fd_v3 ca = fd_load3(B.ca_xyz, r);
        if (ok) F = fd_make_frame(fd_load3(B.n_xyz, r), ca, fd_load3(B.cb_xyz, r));
        else { F = fd_frame{}; F.ca = ca; }
        F.pad = __uint_as_float(a);
        frames[p] = F;
        ca_perm[3ull * p] = ca.x; ca_perm[3ull * p + 1] = ca.y; ca_perm[3ull * p + 2] = ca.z;
        ok_perm[p] = ok ? 1 : 0;
        aa_perm[p] = (uint8_t)a;
    }
}

// count pass of the MSD build: counts[bucket * S + s] += keys of structure s that fall into the bucket (both orientations of every
// unordered pair: (aa_i, aa_j >> 4) and (aa_j, aa_i >> 4)).  The forward keys are two per-lane counters (aa_j is wave-uniform per step), the
// reverse keys two population counts of the pass ballot per step; everything meets in forty LDS counters.  B = the permuted view.
__global__ __launch_bounds__(FD_WAVE) void k_pair_count_msd(fd_batch_view B, fd_hash_consts C, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_cnt[FD_WAVE];
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r1 = B.res_off[s + 1];
    const uint32_t i0 = B.wi_i0[w];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = i0 + lane;
    const bool vi = i < r1 && B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    uint32_t aai = 0;
    if (vi) { cai = fd_load3(B.ca_xyz, i); aai = B.aa[i]; }
    s_cnt[lane] = 0;
    __syncthreads();
    // forward keys (aa_i, aa_j >> 4): two counters per lane, the partner's type is wave-uniform per step.  Reverse keys (aa_j, aa_i >> 4): the
    // partners are visited in amino-acid order, so aa_j changes ~20 times per structure — ONE more counter per lane for the current
    // partner type, flushed into the LDS bucket counters when the type changes (a wave-uniform, rare branch)
    uint32_t c_lo = 0, c_hi = 0, rev = 0, cur_aa = 0xffffffffu;
    const uint32_t hi_i = aai >> 4;
    for (uint32_t jb = i0; jb < r1; jb += FD_WAVE) {
        const uint32_t jl = jb + lane;
        const bool jin = jl < r1;
        fd_v3 cj = {0.f, 0.f, 0.f};
        uint32_t aj = 0;
        if (jin) { cj = fd_load3(B.ca_xyz, jl); aj = B.aa[jl]; }
        const uint64_t okm = __ballot(jin && B.hash_ok[jl]);
        const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
        for (uint32_t k = 0; k < nj; ++k) {
            if (!((okm >> k) & 1ull)) continue;  // wave-uniform
            fd_v3 caj = {bcast_lane(cj.x, k), bcast_lane(cj.y, k), bcast_lane(cj.z, k)};
            const uint32_t aaj = (uint32_t)__builtin_amdgcn_readlane((int)aj, (int)k);
            if (aaj != cur_aa) {      // wave-uniform
                if (rev) atomicAdd(&s_cnt[cur_aa * 2u + hi_i], rev);
                rev = 0; cur_aa = aaj;
            }
            const float d2 = fd_dist2(cai, caj);
            const uint32_t pass = (vi && (jb + k) > i && !(d2 > C.d2_max)) ? 1u : 0u;
            rev += pass;
            if (aaj >= 16u) c_hi += pass; else c_lo += pass;
        }
    }
    if (rev) atomicAdd(&s_cnt[cur_aa * 2u + hi_i], rev);
    if (c_lo) atomicAdd(&s_cnt[aai * 2u], c_lo);
    if (c_hi) atomicAdd(&s_cnt[aai * 2u + 1u], c_hi);
    __syncthreads();
    if (lane < FD_MSD_BUCKETS && s_cnt[lane]) atomicAdd(&counts[(uint64_t)lane * B.n_struct + s], s_cnt[lane]);
}

__device__ __forceinline__ fd_frame load_frame(const fd_frame *__restrict__ frames, uint32_t r) {
    const float4 *p = reinterpret_cast<const float4 *>(frames + r);
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    fd_frame F;
    F.ca = {a.x, a.y, a.z}; F.cb = {a.w, b.x, b.y}; F.r1 = {b.z, b.w, c.x}; F.t1 = {c.y, c.z, c.w};
    F.s2 = {d.x, d.y, d.z}; F.nv2 = {d.w, e.x, e.y}; F.len = e.z; F.pad = e.w;
    return F;
}

// IDS16: 6-byte elements — key = hash << 2 | (s >> 16), 16-bit payload = s & 0xffff (s = structure index inside the shard)
// exact table evaluation, out of line: the fallback of the speculative path (and the whole path with FDGPU_EXACT=1)
__device__ __attribute__((noinline)) uint2 pair_both_tab_exact(const fd_frame *__restrict__ frames, uint32_t i, uint32_t j, uint32_t aai,
                                                               uint32_t aaj, float dist_disc, float ang_disc, const uint32_t *tab) {
    fd_frame Fi = load_frame(frames, i), Fj = load_frame(frames, j);
    fd_quant q;
    q.dist_disc = dist_disc; q.ang_disc = ang_disc; q.ang2_disc = 0.0f; q.type = FD_HASH_PDBTR;
    uint32_t a, b;
    fd_pair_both_tab(Fi, Fj, aai, aaj, q, tab, &a, &b);
    return make_uint2(a, b);
}

// MSD (index build, 6-byte elements, default encoding): the keys leave the kernel already partitioned by the top six hash bits — bucket =
// hash >> 24 = aa_i << 1 | aa_j >> 4, forty of them — into a BUCKET-major stream (inside a bucket structure-major, so the stable sort that
// follows still leaves ids ascending inside every hash).  seg_off / cursor are then [bucket][structure] tables; a drain counts its keys
// per bucket in LDS (two ds_add_rtn per lane), claims the slots with one global atomic per touched bucket and stores every key at its
// own position.  The residues of a structure are visited in amino-acid order (k_frames_perm), so a drain touches a handful of buckets and
// the partial lines of one bucket's run are completed in L2 by the drains that follow.  The sort then needs three 8-bit passes, not four.
template <int TAB, bool IDS16, bool MSD, bool DT = false>
__device__ __forceinline__ void drain2(const fd_batch_view &B, const fd_frame *__restrict__ frames, const fd_hash_consts &C,
                                       const uint32_t *tab, const uint32_t *q, uint32_t n, uint32_t i0, uint32_t r0, uint32_t s, uint32_t id,
                                       const uint64_t *seg_off, uint32_t *cursor, uint32_t *keys, void *ids, const float4 *s_fi,
                                       uint32_t *s_bc, uint64_t *s_bb, const uint64_t *s_boff) {
    const uint32_t lane = threadIdx.x;
    uint32_t base = 0, gb = 0, slots = 0;
    if (!MSD) {
        if (lane == 0) base = atomicAdd(&cursor[s], 2u * n);
        base = __shfl(base, 0, FD_WAVE);
    } else {
        // the buckets of a drain's keys are known before any geometry (the queue entry carries the partner's residue type): count per
        // bucket in LDS, claim the slots with one global atomic per touched bucket — its latency then hides behind the descriptor
        s_bc[lane] = 0;
        fd_wave_lds_fence();
        if (lane < n) {
            const uint32_t e = q[lane], il = (e >> 16) & 63u, aj = e >> 22;
            const uint32_t ai = TAB == 2 ? __float_as_uint(s_fi[256 + il].w) : (uint32_t)B.aa[i0 + il];
            const uint32_t bf = ai * 2u + (aj >> 4), br = aj * 2u + (ai >> 4);
            slots = atomicAdd(&s_bc[bf], 1u) | atomicAdd(&s_bc[br], 1u) << 8 | bf << 16 | br << 24;      // one register across the descriptor
        }
        fd_wave_lds_fence();
        if (lane < FD_MSD_BUCKETS) { const uint32_t c = s_bc[lane]; if (c) gb = atomicAdd(&cursor[(uint64_t)lane * B.n_struct + s], c); }
    }
    uint32_t h_ij = 0, h_ji = 0, aai = 0, aaj = 0;
    if (lane < n) {
        uint32_t e = q[lane];
        uint32_t i = i0 + ((e >> 16) & 63u), j = r0 + (e & 0xffffu);
        if (TAB == 2) {
            // frame of i from the work item's LDS-staged residue tile ([k][lane] float4 planes), frame of j from L2
            const uint32_t il = (e >> 16) & 63u;
            const float4 a4 = s_fi[il], b4 = s_fi[64 + il], c4 = s_fi[128 + il], d4 = s_fi[192 + il], e4 = s_fi[256 + il];
            fd_frame Fi;
            Fi.ca = {a4.x, a4.y, a4.z}; Fi.cb = {a4.w, b4.x, b4.y}; Fi.r1 = {b4.z, b4.w, c4.x}; Fi.t1 = {c4.y, c4.z, c4.w};
            Fi.s2 = {d4.x, d4.y, d4.z}; Fi.nv2 = {d4.w, e4.x, e4.y}; Fi.len = e4.z; Fi.pad = e4.w;
            fd_frame Fj = load_frame(frames, j);
            aai = __float_as_uint(Fi.pad); aaj = __float_as_uint(Fj.pad);
#if defined(FD_EMIT_STUB) && FD_EMIT_STUB == 1
            // measurement build (tools/attrib_emit.sh): the drain without the descriptor — frames loaded, slots claimed, keys stored
            h_ij = aai << 25 | aaj << 20 | (__float_as_uint(Fj.ca.x) & 0xfffffu); h_ji = aaj << 25 | aai << 20 | (__float_as_uint(Fi.ca.x) & 0xfffffu);
            if (false)
#endif
            if (!fd_pair_both_spec<DT>(Fi, Fj, aai, aaj, C.q, tab, tab + 32, &h_ij, &h_ji, tab + 64)) {
                uint2 h = pair_both_tab_exact(frames, i, j, B.aa[i], B.aa[j], C.q.dist_disc, C.q.ang_disc, tab);
                h_ij = h.x; h_ji = h.y;
                if (C.spec_miss) atomicAdd(C.spec_miss, 1ull);
            }
        } else if (TAB == 1) {
            uint2 h = pair_both_tab_exact(frames, i, j, B.aa[i], B.aa[j], C.q.dist_disc, C.q.ang_disc, tab);
            h_ij = h.x; h_ji = h.y;
        } else {
            fd_frame Fi = load_frame(frames, i), Fj = load_frame(frames, j);
            fd_pair_both(Fi, Fj, B.aa[i], B.aa[j], C.q, &h_ij, &h_ji);
        }
    }
    if (MSD) {
        if (lane < FD_MSD_BUCKETS) s_bb[lane] = s_boff[lane] + gb;      // first slot of this drain's keys in every bucket (a build call may hold more than 2^32 keys)
        fd_wave_lds_fence();
        if (lane < n) {
            const uint32_t sf = slots & 255u, sr = (slots >> 8) & 255u, bf = (slots >> 16) & 255u, br = slots >> 24;
            // the bucket comes from the residue types like in the count pass (== hash >> 24 unless a saturated distance field bled into
            // the type bits: such a hash raises wide_flag and the build is redone with 8-byte elements; its key still lands in a counted slot)
            if ((((h_ij | h_ji) >> 30) || (h_ij >> 24) != bf || (h_ji >> 24) != br) && C.wide_flag) atomicOr(C.wide_flag, 1ull);
            const uint64_t pf = s_bb[bf] + sf, pr = s_bb[br] + sr;
            const uint32_t hi = s >> 16;      // the position says which bucket (the top six hash bits): the key keeps the other 24 and id bits 23:16
            const uint16_t lo = (uint16_t)(s & 0xffffu);
            keys[pf] = (h_ij << 8) | hi;
            keys[pr] = (h_ji << 8) | hi;
            ((uint16_t *)ids)[pf] = lo;
            ((uint16_t *)ids)[pr] = lo;
        }
        return;
    }
    if (lane < n) {
        // --multiple-bins: the structure's segment holds seg_mul copies of its pair list (one per bin pair), so the stream stays
        // structure-major and the stable sort by hash keeps ids ascending inside every posting list
        const uint64_t so = seg_off[s];
        uint64_t pos = so * C.seg_mul + (uint64_t)C.seg_cfg * (seg_off[s + 1] - so) + base + lane;
        if (IDS16) {
            if (((h_ij | h_ji) >> 30) && C.wide_flag) atomicOr(C.wide_flag, 1ull);   // does not fit hash << 2: the caller rebuilds with 8-byte elements
            uint32_t hi = s >> 16;
            uint16_t lo = (uint16_t)(s & 0xffffu);
            keys[pos] = (h_ij << 2) | hi;
            keys[pos + n] = (h_ji << 2) | hi;
            ((uint16_t *)ids)[pos] = lo;
            ((uint16_t *)ids)[pos + n] = lo;
        } else {
            keys[pos] = h_ij;
            keys[pos + n] = h_ji;
            ((uint32_t *)ids)[pos] = id;
            ((uint32_t *)ids)[pos + n] = id;
        }
    }
}

template <int TAB, bool IDS16, bool MSD, bool DT = false>
__global__ __launch_bounds__(FD_WAVE, FD_EMIT_WAVES) void k_pair_emit2(fd_batch_view B, const fd_frame *__restrict__ frames, fd_hash_consts C,
                                                        const uint64_t *__restrict__ seg_off, uint32_t *__restrict__ cursor,
                                                        uint32_t *__restrict__ keys, void *__restrict__ ids, uint32_t first_id) {
    __shared__ uint32_t q[2 * FD_WAVE];
    __shared__ uint32_t tab[DT ? 64 + FD_DIST_NTHR + 1 : 64];   // [0,27) exact table (bit patterns), [32,59) the same with float thresholds for the speculative path,
                                                                 // DT: [64, 64 + FD_DIST_NTHR) the squared-distance breakpoints of the two distance fields
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    if (TAB) {
        if (DT && threadIdx.x < FD_DIST_NTHR) tab[64 + threadIdx.x] = fd_dist_thr_bits[threadIdx.x];
        if (threadIdx.x == 0) {
            fd_fill_bintab(tab);
            for (int k = 0; k < FD_BINTAB_WORDS; ++k) tab[32 + k] = tab[k];
            for (int m = 0; m < 4; ++m)
                for (int k = 0; k < 4; ++k) { uint32_t v = tab[32 + 7 + 5 * m + k]; tab[32 + 7 + 5 * m + k] = v > 0x7f7fffffu ? 0x7f7fffffu : v; }
        }
        __syncthreads();
    }
    const uint32_t s = B.wi_struct[w];
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    const uint32_t i0 = B.wi_i0[w];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = i0 + lane;
    const bool vi = i < r1 && B.hash_ok[i];
    fd_v3 cai = {0.f, 0.f, 0.f};
    if (vi) cai = fd_load3(B.ca_xyz, i);
    // LDS staging of the work item's residue tile: the 64 frames of the i side (5 KB), plane-major so that the fill is
    // conflict-free; every drain reads its i frames from here instead of gathering them from L2 again
    __shared__ float4 s_fi[5 * FD_WAVE];
    if (TAB == 2 && i < r1) {
        const float4 *fp = reinterpret_cast<const float4 *>(frames + i);
#pragma unroll
        for (int k = 0; k < 5; ++k) s_fi[k * FD_WAVE + lane] = fp[k];
    }
    __shared__ uint32_t s_bc[MSD ? FD_WAVE : 1];
    __shared__ uint64_t s_bb[MSD ? FD_WAVE : 1], s_boff[MSD ? FD_WAVE : 1];
    if (MSD && lane < FD_MSD_BUCKETS) s_boff[lane] = seg_off[(uint64_t)lane * B.n_struct + s];      // where the structure's keys of every bucket start
    __syncthreads();
    uint32_t qn = 0;  // wave-uniform
    // single drain site (two inlined copies of the descriptor code would not fit the I-cache); the queue is
    // flushed on the last candidate. j is walked in blocks of 64: one coalesced load per block, then v_readlane
    // broadcasts — no per-j memory latency in the filter loop.
    for (uint32_t jb = i0; jb < r1; jb += FD_WAVE) {
        const uint32_t jl = jb + lane;
        const bool jin = jl < r1;
        fd_v3 cj = {0.f, 0.f, 0.f};
        if (jin) cj = fd_load3(B.ca_xyz, jl);
        uint32_t aj = 0;
        if (MSD && jin) aj = B.aa[jl];
        const uint64_t okm = __ballot(jin && B.hash_ok[jl]);
        const uint32_t nj = (r1 - jb) < FD_WAVE ? (r1 - jb) : FD_WAVE;
        const bool last_block = jb + FD_WAVE >= r1;
        for (uint32_t k = 0; k < nj; ++k) {
            const bool last = last_block && k + 1 == nj;
            if ((okm >> k) & 1ull) {  // wave-uniform
                fd_v3 caj = {bcast_lane(cj.x, k), bcast_lane(cj.y, k), bcast_lane(cj.z, k)};
                float d2 = fd_dist2(cai, caj);
                const uint32_t j = jb + k;
                bool pass = vi && j > i && !(d2 > C.d2_max);
                uint64_t m = __ballot(pass);
                if (m != 0) {
                    // MSD: bits 22-26 = the partner's residue type (wave-uniform, a scalar OR)
                    const uint32_t tag = MSD ? ((uint32_t)__builtin_amdgcn_readlane((int)aj, (int)k) << 22) | (j - r0) : (j - r0);
                    if (pass) q[qn + fd_mbcnt(m)] = (lane << 16) | tag;
                    qn += (uint32_t)__popcll(m);
                }
            }
            while (qn >= FD_WAVE || (last && qn)) {   // on the last candidate up to two drains may be pending
                fd_wave_lds_fence();
                uint32_t n = qn < FD_WAVE ? qn : FD_WAVE;
                qn -= n;
                drain2<TAB, IDS16, MSD, DT>(B, frames, C, tab, q + qn, n, i0, r0, s, first_id + s, seg_off, cursor, keys, ids, s_fi, s_bc, s_bb, s_boff);
                fd_wave_lds_fence();
            }
        }
    }
}

// ------------------------------------------------------------------ ordered variant (API S1, sort_dedup = 0)
// Same pairs, but written in the reference's row-major (i, j) order: one lane per i computes its
// row offset by a prefix over row counts.  Used for parity tests of the raw hash list; not on the
// index-build fast path.
__global__ __launch_bounds__(FD_WAVE) void k_row_count(fd_batch_view B, fd_hash_consts C, uint32_t *__restrict__ row_cnt, float cutoff) {
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    const uint32_t i = B.wi_i0[w] + threadIdx.x;
    if (i >= r1) return;
    if (fd_own_descriptor(C.q.type)) {   // the encodings with their own descriptor and acceptance rule (fd_geom_other.h)
        uint32_t cnt = 0;
        if (B.aa[i] != 255)
            for (uint32_t j = r0; j < r1; ++j)
                cnt += (j != i && B.aa[j] != 255 && fd_accept_other(C.q.type, B, r0, r1, i, j, cutoff)) ? 1u : 0u;
        row_cnt[i] = cnt;
        return;
    }
    const bool vi = B.hash_ok[i];
    fd_v3 cai = fd_load3(B.ca_xyz, i);
    uint32_t cnt = 0;
    for (uint32_t j = r0; j < r1; ++j) {
        if (!B.hash_ok[j]) continue;
        float d2 = fd_dist2(cai, fd_load3(B.ca_xyz, j));
        cnt += (vi && i != j && !(d2 > C.d2_max)) ? 1u : 0u;
    }
    row_cnt[i] = cnt;
}

// ids (optional): the structure id first_id + s of every entry, or with ids_partner its partner residue j (batch residue index) — the
// row start row_off[i] gives i, so (hash, i, j) is what collect_hash_id_pos walks (controller/summary.rs:632-690)
__global__ __launch_bounds__(FD_WAVE) void k_row_emit(fd_batch_view B, fd_hash_consts C, const uint64_t *__restrict__ row_off,
                                                      uint32_t *__restrict__ keys, float cutoff, uint32_t *__restrict__ ids, uint32_t first_id, int ids_partner) {
    uint32_t w = fd_xcd_remap(blockIdx.x, B.n_work);
    if (w >= B.n_work) return;
    const uint32_t s = B.wi_struct[w];
    const uint32_t r0 = B.res_off[s], r1 = B.res_off[s + 1];
    const uint32_t i = B.wi_i0[w] + threadIdx.x;
    if (i >= r1) return;
    if (fd_own_descriptor(C.q.type)) {
        if (B.aa[i] == 255) return;
        uint64_t pos = row_off[i];
        float f[FD_NFEAT];
        for (uint32_t j = r0; j < r1; ++j) {
            if (j == i || B.aa[j] == 255 || !fd_feature_other(C.q.type, B, r0, r1, i, j, cutoff, f)) continue;
            if (ids) ids[pos] = ids_partner ? j : first_id + s;
            keys[pos++] = fd_hash_other(C.q.type, f, C.q);
        }
        return;
    }
    if (!B.hash_ok[i]) return;
    fd_v3 n1 = fd_load3(B.n_xyz, i), ca1 = fd_load3(B.ca_xyz, i), cb1 = fd_load3(B.cb_xyz, i);
    uint32_t aa1 = B.aa[i];
    uint64_t pos = row_off[i];
    for (uint32_t j = r0; j < r1; ++j) {
        if (!B.hash_ok[j] || i == j) continue;
        fd_v3 ca2 = fd_load3(B.ca_xyz, j);
        if (fd_dist2(ca1, ca2) > C.d2_max) continue;
        fd_v3 n2 = fd_load3(B.n_xyz, j), cb2 = fd_load3(B.cb_xyz, j);
        fd_feature f = fd_pair_feature(n1, ca1, cb1, n2, ca2, cb2);
        if (ids) ids[pos] = ids_partner ? j : first_id + s;
        keys[pos++] = fd_hash_enc(aa1, B.aa[j], f, C.q);
    }
}

// aa / cb_valid -> hash_ok
__global__ void k_hash_ok(const uint8_t *__restrict__ aa, const uint8_t *__restrict__ cb_valid, uint8_t *__restrict__ ok, uint64_t n) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) ok[k] = (aa[k] != 255 && (cb_valid == nullptr || cb_valid[k])) ? 1 : 0;
}

// ------------------------------------------------------------------ start-up known-answer test (fdgpu_create)
// The catalytic triad of query/4CHA.pdb (HIS B57, ASP B102, SER C195) and its six PDBTrRosetta hashes, the literals of the
// reference's own test (controller/graph.rs:71-79).  Every ordered pair is evaluated three ways — the generic chain
// (fd_pair_feature + fd_hash_pdbtr: restated sinf/cosf/acosf/atan2f), the exhaustive-table form (fd_pair_both_tab) and the
// speculative form (fd_pair_both_spec, exact fallback on refusal) — so a build whose arithmetic drifts (fast-math, FMA contraction,
// a different table generation) fails loudly instead of producing a subtly different index.
__global__ void k_selfcheck(fd_quant q, uint32_t *__restrict__ out /*[24]*/) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const fd_v3 N[3] = {{6.661f, 8.291f, 43.860f}, {10.483f, 7.756f, 49.260f}, {5.260f, -1.068f, 41.296f}};
    const fd_v3 CA[3] = {{6.994f, 8.354f, 42.405f}, {9.429f, 7.479f, 48.266f}, {5.547f, 0.158f, 42.050f}};
    const fd_v3 CB[3] = {{8.251f, 7.488f, 42.026f}, {10.033f, 6.489f, 47.255f}, {5.773f, 1.360f, 41.130f}};
    const uint32_t AA[3] = {8u, 3u, 15u};
    const int PI[6] = {1, 1, 0, 0, 2, 2}, PJ[6] = {0, 2, 1, 2, 1, 0};   // B102->B57, B102->C195, B57->B102, B57->C195, C195->B102, C195->B57
    uint32_t tab[64 + FD_DIST_NTHR + 1];
    fd_fill_bintab(tab);
    for (int k = 0; k < FD_BINTAB_WORDS; ++k) tab[32 + k] = tab[k];
    for (int m = 0; m < 4; ++m)
        for (int k = 0; k < 4; ++k) { uint32_t v = tab[32 + 7 + 5 * m + k]; tab[32 + 7 + 5 * m + k] = v > 0x7f7fffffu ? 0x7f7fffffu : v; }
    for (int k = 0; k < FD_DIST_NTHR; ++k) tab[64 + k] = fd_dist_thr_bits[k];
    fd_frame F[3];
    for (int r = 0; r < 3; ++r) F[r] = fd_make_frame(N[r], CA[r], CB[r]);
    for (int k = 0; k < 6; ++k) {
        const int i = PI[k], j = PJ[k];
        out[k] = fd_hash_pdbtr(AA[i], AA[j], fd_pair_feature(N[i], CA[i], CB[i], N[j], CA[j], CB[j]), q);
        uint32_t a, b;
        fd_pair_both_tab(F[i], F[j], AA[i], AA[j], q, tab, &a, &b);
        out[6 + k] = a;
        if (!fd_pair_both_spec(F[i], F[j], AA[i], AA[j], q, tab, tab + 32, &a, &b)) fd_pair_both_tab(F[i], F[j], AA[i], AA[j], q, tab, &a, &b);
        out[12 + k] = a;
        // ... and with the distance fields from the squared-distance table (the index build's form; valid for the default 16 distance bins the check runs with)
        if (!fd_pair_both_spec<true>(F[i], F[j], AA[i], AA[j], q, tab, tab + 32, &a, &b, tab + 64)) fd_pair_both_tab(F[i], F[j], AA[i], AA[j], q, tab, &a, &b);
        out[18 + k] = a;
    }
}

// ------------------------------------------------------------------ launchers (called from fdgpu_api.hip)
extern "C++" {
void fd_launch_selfcheck(const fd_quant &q, uint32_t *out, hipStream_t st) { hipLaunchKernelGGL(k_selfcheck, dim3(1), dim3(64), 0, st, q, out); }
// FDGPU_MSD_PERM=0 (buckets over the caller's residue order): the check k_frames_perm makes while it permutes
__global__ void k_aa_check(const uint8_t *__restrict__ aa, const uint8_t *__restrict__ ok, uint64_t n, unsigned long long *__restrict__ wide_flag) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && ok[k] && aa[k] >= 20u) atomicOr(wide_flag, 1ull);
}
void fd_launch_aa_check(const fd_batch_view &B, uint64_t n_res, unsigned long long *wide_flag, hipStream_t st) {
    if (!n_res) return;
    hipLaunchKernelGGL(k_aa_check, dim3((unsigned)((n_res + 255) / 256)), dim3(256), 0, st, B.aa, B.hash_ok, n_res, wide_flag);
}

void fd_launch_hash_ok(const uint8_t *aa, const uint8_t *cb_valid, uint8_t *ok, uint64_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_hash_ok, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, aa, cb_valid, ok, n);
}
static inline unsigned grid_for(uint32_t n_work) { return ((n_work + 7u) / 8u) * 8u; }
void fd_launch_pair_count(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st) {
    if (!B.n_work) return;
    hipLaunchKernelGGL(k_pair_count, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, counts);
}
void fd_launch_pair_emit(const fd_batch_view &B, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor,
                         uint32_t *keys, uint32_t *ids, uint32_t first_id, hipStream_t st) {
    if (!B.n_work) return;
    if (ids)
        hipLaunchKernelGGL(k_pair_emit<true>, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, seg_off, cursor, keys, ids, first_id);
    else
        hipLaunchKernelGGL(k_pair_emit<false>, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, seg_off, cursor, keys, ids, first_id);
}
void fd_launch_frames(const fd_batch_view &B, uint64_t n_res, void *frames, hipStream_t st) {
    if (!n_res) return;
    hipLaunchKernelGGL(k_frames, dim3((unsigned)((n_res + 255) / 256)), dim3(256), 0, st, B, (uint32_t)n_res, (fd_frame *)frames);
}
void fd_launch_pair_count2(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st) {
    if (!B.n_work) return;
    hipLaunchKernelGGL(k_pair_count2, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, counts);
}
void fd_launch_pair_emit2(const fd_batch_view &B, const void *frames, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor,
                          uint32_t *keys, void *ids, bool ids16, uint32_t first_id, hipStream_t st) {
    if (!B.n_work) return;
    dim3 g(grid_for(B.n_work)), b(FD_WAVE);
    const fd_frame *F = (const fd_frame *)frames;
    if (C.use_tab >= 2 && ids16) hipLaunchKernelGGL((k_pair_emit2<2, true, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
    else if (C.use_tab >= 2) hipLaunchKernelGGL((k_pair_emit2<2, false, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
    else if (C.use_tab && ids16) hipLaunchKernelGGL((k_pair_emit2<1, true, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
    else if (C.use_tab) hipLaunchKernelGGL((k_pair_emit2<1, false, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
    else if (ids16) hipLaunchKernelGGL((k_pair_emit2<0, true, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
    else hipLaunchKernelGGL((k_pair_emit2<0, false, false>), g, b, 0, st, B, F, C, seg_off, cursor, keys, ids, first_id);
}
// the MSD build (6-byte elements): B = the permuted view (ca_xyz / hash_ok / aa = the arrays k_frames_perm wrote), seg_off / cursor = [40][S]
void fd_launch_frames_perm(const fd_batch_view &B, void *frames, float *ca_perm, uint8_t *ok_perm, uint8_t *aa_perm, unsigned long long *wide_flag, hipStream_t st) {
    if (!B.n_struct) return;
    hipLaunchKernelGGL(k_frames_perm, dim3(B.n_struct), dim3(256), 0, st, B, (fd_frame *)frames, ca_perm, ok_perm, aa_perm, wide_flag);
}
void fd_launch_pair_count_msd(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *counts, hipStream_t st) {
    if (!B.n_work) return;
    hipLaunchKernelGGL(k_pair_count_msd, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, counts);
}
void fd_launch_pair_emit_msd(const fd_batch_view &B, const void *frames, const fd_hash_consts &C, const uint64_t *seg_off, uint32_t *cursor, uint32_t *keys,
                             uint16_t *ids, hipStream_t st) {
    if (!B.n_work) return;
    dim3 g(grid_for(B.n_work)), b(FD_WAVE);
    const fd_frame *F = (const fd_frame *)frames;
    if (C.use_tab == 3) hipLaunchKernelGGL((k_pair_emit2<2, true, true, true>), g, b, 0, st, B, F, C, seg_off, cursor, keys, (void *)ids, 0u);      // + squared-distance table
    else if (C.use_tab == 2) hipLaunchKernelGGL((k_pair_emit2<2, true, true>), g, b, 0, st, B, F, C, seg_off, cursor, keys, (void *)ids, 0u);
    else if (C.use_tab) hipLaunchKernelGGL((k_pair_emit2<1, true, true>), g, b, 0, st, B, F, C, seg_off, cursor, keys, (void *)ids, 0u);
    else hipLaunchKernelGGL((k_pair_emit2<0, true, true>), g, b, 0, st, B, F, C, seg_off, cursor, keys, (void *)ids, 0u);
}
void fd_launch_row_count(const fd_batch_view &B, const fd_hash_consts &C, uint32_t *row_cnt, float cutoff, hipStream_t st) {
    if (!B.n_work) return;
    hipLaunchKernelGGL(k_row_count, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, row_cnt, cutoff);
}
void fd_launch_row_emit(const fd_batch_view &B, const fd_hash_consts &C, const uint64_t *row_off, uint32_t *keys, float cutoff, uint32_t *ids,
                        uint32_t first_id, hipStream_t st, int ids_partner) {
    if (!B.n_work) return;
    hipLaunchKernelGGL(k_row_emit, dim3(grid_for(B.n_work)), dim3(FD_WAVE), 0, st, B, C, row_off, keys, cutoff, ids, first_id, ids_partner);
}
}
