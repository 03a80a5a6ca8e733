// fd_geom_other.h — the four encodings with their own pair descriptors:
//   2 TrRosetta            structure/core.rs:326-345 (get_trrosetta_feature), geometry/trrosetta.rs:56-96
//   4 PointPairFeature     structure/core.rs:310-324 (get_ppf), structure/coordinate.rs:93-102, geometry/ppf.rs:15-49
//   5 TertiaryInteraction  controller/feature.rs:113-160, geometry/tertiary_interaction.rs:21-80
//   6 Hybrid               controller/feature.rs:161-190, structure/core.rs:405-437, geometry/hybrid.rs:20-91
// They run the exact-libm form one ORDERED residue pair at a time (the acceptance rule itself depends on the orientation for
// PointPairFeature, on the CB distance for TrRosetta, on chain-interior residues for the last two), through the row kernels of
// k_hash.hip; the table / speculative fast path exists for the default encoding only.
#pragma once
#include "fd_device.h"

#define FD_HASH_TRROSETTA 2u
#define FD_HASH_PPF 4u
#define FD_HASH_TERTIARY 5u
#define FD_HASH_HYBRID 6u
#define FD_NFEAT 9   // feature container of get_single_feature (controller/feature.rs:205)

FD_HD bool fd_own_descriptor(uint32_t t) { return t == FD_HASH_TRROSETTA || t == FD_HASH_PPF || t == FD_HASH_TERTIARY || t == FD_HASH_HYBRID; }

// calc_angle_radian (structure/coordinate.rs:150-162): angle a-b-c
FD_HD float fd_calc_angle_radian(fd_v3 a, fd_v3 b, fd_v3 c) {
    fd_v3 v1 = {a.x - b.x, a.y - b.y, a.z - b.z};
    fd_v3 v2 = {c.x - b.x, c.y - b.y, c.z - b.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    float l1 = fd_sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    float l2 = fd_sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    return fdd_acosf(dt / (l1 * l2));
}
FD_HD float fd_norm(fd_v3 a) { return fd_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

// map_aa_to_u8_group (utils/convert.rs:85-130) through the residue type: the two tables of the reference agree name by name
FD_HD uint32_t fd_aa_group(uint32_t aa) {
    // A R N D C Q E G H I L K M F P S T W Y V -> 0 3 2 3 0 2 3 0 3 1 1 3 1 1 0 0 2 1 2 1, two bits each
    return aa < 20u ? (uint32_t)((0x6605d738ecull >> (2u * aa)) & 3ull) : 0u;
}

// Ordered pair (i -> j), i != j, both residue types known (the caller checked those two, feature.rs:15-24), of the structure that
// spans residues [r0, r1).  Returns whether the pair has a feature and fills f like get_single_feature does.
__device__ __forceinline__ bool fd_feature_other(uint32_t type, const fd_batch_view &B, uint32_t r0, uint32_t r1, uint32_t i, uint32_t j,
                                                 float cutoff, float f[FD_NFEAT]) {
    const fd_v3 ca1 = fd_load3(B.ca_xyz, i), ca2 = fd_load3(B.ca_xyz, j);
    if (type == FD_HASH_TRROSETTA) {
        if (!B.hash_ok[i] || !B.hash_ok[j]) return false;
        const fd_v3 cb1 = fd_load3(B.cb_xyz, i), cb2 = fd_load3(B.cb_xyz, j), n1 = fd_load3(B.n_xyz, i), n2 = fd_load3(B.n_xyz, j);
        const float cb_dist = fd_dist(cb1, cb2);
        if (cb_dist > cutoff) return false;
        f[0] = (float)B.aa[i]; f[1] = (float)B.aa[j]; f[2] = cb_dist;
        f[3] = fd_calc_torsion(ca1, cb1, cb2, ca2);
        f[4] = fd_calc_torsion(n1, ca1, cb1, cb2);
        f[5] = fd_calc_torsion(cb1, cb2, ca2, n2);
        f[6] = fd_calc_angle_radian(ca1, cb1, cb2);
        f[7] = fd_calc_angle_radian(cb1, cb2, ca2);
        f[8] = 0.0f;
        return true;
    }
    if (type == FD_HASH_PPF) {
        if (!B.hash_ok[i] || !B.hash_ok[j]) return false;
        const fd_v3 a = fd_sub(fd_load3(B.cb_xyz, i), ca1), b = fd_sub(fd_load3(B.cb_xyz, j), ca1);
        const fd_v3 n1 = fd_normalize(a), n2 = fd_normalize(b), d = fd_sub(b, a), nd = fd_normalize(d);
        const float dist = fd_norm(d);
        const float a1 = fdd_acosf(fd_dot(n1, nd)), a2 = fdd_acosf(fd_dot(n2, nd)), a3 = fdd_acosf(fd_dot(n1, n2));
        if (dist > cutoff) return false;
        f[0] = (float)B.aa[i]; f[1] = (float)B.aa[j]; f[2] = dist; f[3] = a1; f[4] = a2; f[5] = a3; f[6] = 0.0f; f[7] = 0.0f; f[8] = 0.0f;
        return true;
    }
    if (i == r0 || j == r0 || i + 1 == r1 || j + 1 == r1) return false;     // first / last residue of the structure (feature.rs:114, 162)
    if (type == FD_HASH_TERTIARY) {
        const float ca_dist = fd_dist(ca1, ca2);
        if (ca_dist > cutoff) return false;
        const fd_v3 u1 = fd_normalize(fd_sub(ca1, fd_load3(B.ca_xyz, i - 1))), u2 = fd_normalize(fd_sub(fd_load3(B.ca_xyz, i + 1), ca1));
        const fd_v3 u3 = fd_normalize(fd_sub(ca2, fd_load3(B.ca_xyz, j - 1))), u4 = fd_normalize(fd_sub(fd_load3(B.ca_xyz, j + 1), ca2));
        const fd_v3 u5 = fd_normalize(fd_sub(ca2, ca1));
        f[0] = fdd_acosf(fd_dot(u1, u2)); f[1] = fdd_acosf(fd_dot(u3, u4)); f[2] = fdd_acosf(fd_dot(u1, u5)); f[3] = fdd_acosf(fd_dot(u3, u5));
        f[4] = fdd_acosf(fd_dot(u1, u4)); f[5] = fdd_acosf(fd_dot(u2, u3)); f[6] = fdd_acosf(fd_dot(u1, u3));
        f[7] = ca_dist; f[8] = (float)(j - r0) - (float)(i - r0);
        return true;
    }
    if (!B.hash_ok[i] || !B.hash_ok[j]) return false;
    const fd_v3 cb1 = fd_load3(B.cb_xyz, i), cb2 = fd_load3(B.cb_xyz, j), n1 = fd_load3(B.n_xyz, i), n2 = fd_load3(B.n_xyz, j);
    const float ca_dist = fd_dist(ca1, ca2);
    if (ca_dist > cutoff) return false;
    f[0] = (float)fd_aa_group(B.aa[i]); f[1] = (float)fd_aa_group(B.aa[j]);
    f[2] = ca_dist; f[3] = fd_dist(cb1, cb2); f[4] = fd_calc_angle(ca1, cb1, ca2, cb2);
    f[5] = fd_calc_torsion(n1, ca1, cb1, cb2); f[6] = fd_calc_torsion(cb1, cb2, ca2, n2);
    f[7] = fd_calc_torsion(fd_load3(B.ca_xyz, i - 1), n1, ca1, fd_load3(B.ca_xyz, i + 1));
    f[8] = fd_calc_torsion(fd_load3(B.ca_xyz, j - 1), n2, ca2, fd_load3(B.ca_xyz, j + 1));
    return true;
}

// acceptance alone (the count pass): the same tests as fd_feature_other, bit for bit, without the angles
__device__ __forceinline__ bool fd_accept_other(uint32_t type, const fd_batch_view &B, uint32_t r0, uint32_t r1, uint32_t i, uint32_t j, float cutoff) {
    if (type == FD_HASH_TRROSETTA) return B.hash_ok[i] && B.hash_ok[j] && !(fd_dist(fd_load3(B.cb_xyz, i), fd_load3(B.cb_xyz, j)) > cutoff);
    if (type == FD_HASH_PPF) {
        if (!B.hash_ok[i] || !B.hash_ok[j]) return false;
        const fd_v3 ca1 = fd_load3(B.ca_xyz, i);
        const fd_v3 a = fd_sub(fd_load3(B.cb_xyz, i), ca1), b = fd_sub(fd_load3(B.cb_xyz, j), ca1);
        return !(fd_norm(fd_sub(b, a)) > cutoff);
    }
    if (i == r0 || j == r0 || i + 1 == r1 || j + 1 == r1) return false;
    if (type == FD_HASH_HYBRID && (!B.hash_ok[i] || !B.hash_ok[j])) return false;
    return !(fd_dist(fd_load3(B.ca_xyz, i), fd_load3(B.ca_xyz, j)) > cutoff);
}

// perfect_hash of the four encodings on a feature container; q.dist_disc / q.ang_disc hold the clamped bin counts' factors
FD_HD uint32_t fd_hash_other(uint32_t type, const float *f, fd_quant q) {
    if (type == FD_HASH_TRROSETTA) {
        uint32_t h = (fd_sat_u32(f[0]) * 20u + fd_sat_u32(f[1])) << 23 | fd_q(f[2], 2.0f, q.dist_disc) << 20;
        for (int k = 0; k < 5; ++k) {
            float s, c;
            fdd_sincosf(f[3 + k], &s, &c);
            h |= fd_q(s, -1.0f, q.ang_disc) << (18 - 4 * k) | fd_q(c, -1.0f, q.ang_disc) << (16 - 4 * k);
        }
        return h;
    }
    if (type == FD_HASH_PPF) {
        uint32_t h = fd_sat_u32(f[0]) << 27 | fd_sat_u32(f[1]) << 22 | fd_q(f[2], 2.0f, q.dist_disc) << 18;
        for (int k = 0; k < 3; ++k) {
            float s, c;
            fdd_sincosf(f[3 + k], &s, &c);
            h |= fd_q(s, -1.0f, q.ang_disc) << (15 - 6 * k) | fd_q(c, -1.0f, q.ang_disc) << (12 - 6 * k);
        }
        return h;
    }
    if (type == FD_HASH_TERTIARY) {
        uint32_t h = 0;
        for (int k = 0; k < 7; ++k) {
            float s, c;
            fdd_sincosf(f[k], &s, &c);
            h |= fd_q(c, -1.0f, q.ang_disc) << (26 - 3 * k);
        }
        const uint32_t seq = f[8] < -4.0f ? 0u : (f[8] > 4.0f ? 8u : fd_sat_u32(f[8]) + 4u);   // `as u32` saturates negative offsets to 0 first
        return h | fd_q(f[7], 2.0f, q.dist_disc) << 4 | seq;
    }
    uint32_t h = fd_sat_u32(f[0]) << 30 | fd_sat_u32(f[1]) << 28 | fd_q(f[2], 2.0f, q.dist_disc) << 24 | fd_q(f[3], 2.0f, q.dist_disc) << 20;
    for (int k = 0; k < 5; ++k) {
        float s, c;
        fdd_sincosf(f[4 + k], &s, &c);
        h |= fd_q(s, -1.0f, q.ang_disc) << (18 - 4 * k) | fd_q(c, -1.0f, q.ang_disc) << (16 - 4 * k);
    }
    return h;
}
