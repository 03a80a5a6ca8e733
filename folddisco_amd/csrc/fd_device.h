// fd_device.h — internal device-side types shared by the kernels and the C ABI glue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fd_geom.h"

#define FD_WAVE 64

// Packed structures resident in HBM (fdgpu_batch). Coordinates stay in the caller's
// interleaved xyz layout (12 B per atom, 37 B per residue with aa): they are 37·R bytes against
// ~8·80·R bytes of keys written, so their layout is irrelevant for HBM traffic; what matters is
// that one structure (R·37 B ≈ 12 KB) stays L1/L2 resident while its R/64 tiles are processed.
struct fd_batch_view {
    const float *n_xyz, *ca_xyz, *cb_xyz;
    const uint8_t *aa;
    const uint8_t *hash_ok;      // aa != 255 && cb_valid (precomputed at upload)
    const uint32_t *res_off;     // [S+1]
    const uint32_t *wi_struct;   // [W] work item -> structure
    const uint32_t *wi_i0;       // [W] work item -> first residue (absolute) of its 64-residue i-tile
    uint32_t n_struct;
    uint32_t n_work;
};

struct fd_hash_consts {
    uint32_t seg_mul, seg_cfg;   // --multiple-bins: a structure's segment holds seg_mul copies of its pair list, this launch fills copy seg_cfg
    fd_quant q;
    float d2_max;   // largest f32 whose sqrt is <= dist_cutoff: sqrtf(d2) > cutoff  <=>  d2 > d2_max
    int use_tab;    // 1: default 4 angle bins -> table form of the angle fields (fd_bin_tables.h); 2: + speculative torsions; 3: + default 16 distance
                    // bins -> the distance fields from the squared-distance table (fd_dist_table.h) in the MSD build's pair kernel
    unsigned long long *spec_miss;   // device counter of pairs the speculative path handed to the exact routine (may be null)
    unsigned long long *wide_flag;   // set when a hash does not fit 30 bits (fields are OR-ed unmasked: an infinite distance sets
                                     // bits 30-31) — the 6-byte sort elements keep 30 hash bits, the build is then redone with 8-byte ones
};

__device__ __forceinline__ fd_v3 fd_load3(const float *p, uint32_t r) {
    const float *q = p + 3ull * r;
    return {q[0], q[1], q[2]};
}

// XCD-aware work remap: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md); map
// consecutive logical work items (tiles of the same structure) to the same XCD so that a
// structure's coordinates are fetched into one L2 only.
__device__ __forceinline__ uint32_t fd_xcd_remap(uint32_t b, uint32_t n) {
    uint32_t per = (n + 7u) / 8u;
    uint32_t w = (b & 7u) * per + (b >> 3);
    return w;  // may be >= n for the padded tail; caller checks
}

// One wavefront per workgroup (k_pair_emit2, k_match_pairs): what the lanes need from each other goes through LDS, and a wavefront's LDS operations execute in
// issue order, so a lane reads what another lane wrote earlier without any wait — only the COMPILER must be kept from moving or forwarding the
// accesses (a wavefront-scope fence).  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: the vmcnt(0) made every drain wait for its own
// four scattered key / id stores to reach L2 (and for the partner-frame gathers of the NEXT drain's prologue to...) before the filter loop could go
// on — the waves of this kernel sat in s_waitcnt 39 % of their cycles (profiles/round3_pmc_emit_msd_ab_S67750.txt).
__device__ __forceinline__ void fd_wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t fd_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t fd_mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
