// fd_geom.h — residue-pair descriptor and PDBTrRosetta hash, device side.
// Operation order follows the reference exactly so that every f32 intermediate is
// bit-identical (compile with -ffp-contract=off):
//   descriptor   src/structure/core.rs:378-403 (get_pdb_tr_feature)
//   distance     src/structure/coordinate.rs:109-115
//   angle        src/structure/coordinate.rs:118-133
//   torsion      src/structure/coordinate.rs:204-215 (+ cross/normalize/dot :39-76)
//   quantiser    src/utils/convert.rs:32-36
//   hash         src/geometry/pdb_tr.rs:21-75
#pragma once
#include "fd_libm.h"
#include "fd_bin_tables.h"
#include "fd_dist_table.h"

struct fd_v3 { float x, y, z; };

FD_HD fd_v3 fd_sub(fd_v3 a, fd_v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
FD_HD float fd_dot(fd_v3 a, fd_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FD_HD fd_v3 fd_cross(fd_v3 a, fd_v3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
FD_HD fd_v3 fd_normalize(fd_v3 a) {
    float n = fd_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / n, a.y / n, a.z / n};
}
FD_HD float fd_dist(fd_v3 a, fd_v3 b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return fd_sqrtf(dx * dx + dy * dy + dz * dz);
}
FD_HD float fd_dist2(fd_v3 a, fd_v3 b) {  // the argument of the sqrt above, same rounding
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

// Rust `as u32` on f32: saturating, NaN -> 0
FD_HD uint32_t fd_sat_u32(float v) {
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

// quantiser constants (min, 1/((max-min)/(nb-1))) are computed once on the host in f32
struct fd_quant {
    float dist_disc;   // 1/((20-2)/(nbin_dist-1))
    float ang_disc;    // 1/((max-min)/(nbin_angle-1)): sin/cos fields (-1..1), degrees (0..180, PDBMotif) or radians (-pi..pi, Folddisco*)
    float ang2_disc;   // FolddiscoAngle / FolddiscoDist: the theta field, 1/((pi-0)/(min(nbin_angle, cap)-1))
    uint32_t type;     // HashType index of the reference (geometry/core.rs:26-40): 0 PDBMotif, 1 PDBMotifSinCos, 3 PDBTrRosetta,
                       // 7 FolddiscoAngle, 8 FolddiscoDist — the encodings over the (d_CA, d_CB, theta, tau1, tau2) descriptor
};
#define FD_HASH_PDBMOTIF 0u
#define FD_HASH_PDBMOTIF_SINCOS 1u
#define FD_HASH_PDBTR 3u
#define FD_HASH_FD_ANGLE 7u
#define FD_HASH_FD_DIST 8u
FD_HD uint32_t fd_q(float v, float mn, float disc) { return fd_sat_u32((v - mn) * disc + 0.5f); }

struct fd_feature { float ca_dist, cb_dist, angle, tor1, tor2; };

// angle between (cb1-ca1) and (cb2-ca2) — coordinate.rs:118-133 with a=ca1,b=cb1,c=ca2,d=cb2
FD_HD float fd_calc_angle(fd_v3 a, fd_v3 b, fd_v3 c, fd_v3 d) {
    fd_v3 v1 = {b.x - a.x, b.y - a.y, b.z - a.z};
    fd_v3 v2 = {d.x - c.x, d.y - c.y, d.z - c.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    float l1 = fd_sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    float l2 = fd_sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    return fdd_acosf(dt / (l1 * l2));
}
FD_HD float fd_calc_torsion(fd_v3 a, fd_v3 b, fd_v3 c, fd_v3 d) {
    fd_v3 v1 = fd_sub(b, a), v2 = fd_sub(c, b), v3 = fd_sub(d, c);
    fd_v3 r = fd_normalize(fd_cross(v1, v2));
    fd_v3 s = fd_normalize(fd_cross(v2, v3));
    fd_v3 t = fd_normalize(fd_cross(r, fd_normalize(v2)));
    float x = fd_dot(r, s);
    float y = fd_dot(s, t);
    return -fdd_atan2f(y, x);
}

// full descriptor for an ordered pair; the caller has already applied the rejection rules
// (i != j, aa != 255, CB present, d_CA <= cutoff)
FD_HD fd_feature fd_pair_feature(fd_v3 n1, fd_v3 ca1, fd_v3 cb1, fd_v3 n2, fd_v3 ca2, fd_v3 cb2) {
    fd_feature f;
    f.ca_dist = fd_dist(ca1, ca2);
    f.cb_dist = fd_dist(cb1, cb2);
    f.angle = fd_calc_angle(ca1, cb1, ca2, cb2);
    f.tor1 = fd_calc_torsion(n1, ca1, cb1, cb2);
    f.tor2 = fd_calc_torsion(cb1, cb2, ca2, n2);
    return f;
}

// pdb_tr.rs:21-75; fields OR-ed unmasked
FD_HD uint32_t fd_hash_pdbtr(uint32_t aa1, uint32_t aa2, fd_feature f, fd_quant q) {
    uint32_t ca = fd_q(f.ca_dist, 2.0f, q.dist_disc);
    uint32_t cb = fd_q(f.cb_dist, 2.0f, q.dist_disc);
    float s0, c0, s1, c1, s2, c2;
    fdd_sincosf(f.angle, &s0, &c0);
    fdd_sincosf(f.tor1, &s1, &c1);
    fdd_sincosf(f.tor2, &s2, &c2);
    uint32_t qs0 = fd_q(s0, -1.0f, q.ang_disc), qc0 = fd_q(c0, -1.0f, q.ang_disc);
    uint32_t qs1 = fd_q(s1, -1.0f, q.ang_disc), qc1 = fd_q(c1, -1.0f, q.ang_disc);
    uint32_t qs2 = fd_q(s2, -1.0f, q.ang_disc), qc2 = fd_q(c2, -1.0f, q.ang_disc);
    return aa1 << 25 | aa2 << 20 | ca << 16 | cb << 12 | qs0 << 10 | qc0 << 8 | qs1 << 6 | qc1 << 4 | qs2 << 2 | qc2;
}

// The other encodings over the same descriptor (all fields OR-ed unmasked like the reference):
//   PDBMotif        pdb_motif.rs:26-50         aa1<<20 | aa2<<15 | ca<<10 | cb<<5 | q(theta in degrees, 0..180)
//   PDBMotifSinCos  pdb_motif_sincos.rs:17-53  aa1<<21 | aa2<<16 | ca<<12 | cb<<8 | q(sin theta)<<4 | q(cos theta)
//   FolddiscoAngle  folddisco_angle.rs:25-70   (aa1*20+aa2)<<21 | ca<<18 | cb<<15 | q(theta, 0..pi)<<10 | q(tau1, -pi..pi)<<5 | q(tau2)
//   FolddiscoDist   folddisco_dist.rs:22-63    (aa1*20+aa2)<<21 | ca<<16 | cb<<11 | q(theta)<<8 | q(tau1)<<4 | q(tau2)
// f = the feature vector as get_single_feature leaves it (controller/feature.rs:26-99): theta in DEGREES for PDBMotif, radians otherwise
FD_HD uint32_t fd_hash_enc_feat(uint32_t aa1, uint32_t aa2, fd_feature f, fd_quant q) {
    if (q.type == FD_HASH_PDBTR) return fd_hash_pdbtr(aa1, aa2, f, q);
    const uint32_t ca = fd_q(f.ca_dist, 2.0f, q.dist_disc), cb = fd_q(f.cb_dist, 2.0f, q.dist_disc);
    const float NPI = -3.14159274f;
    if (q.type == FD_HASH_PDBMOTIF) return aa1 << 20 | aa2 << 15 | ca << 10 | cb << 5 | fd_q(f.angle, 0.0f, q.ang_disc);
    if (q.type == FD_HASH_PDBMOTIF_SINCOS) {
        float s0, c0;
        fdd_sincosf(f.angle, &s0, &c0);
        return aa1 << 21 | aa2 << 16 | ca << 12 | cb << 8 | fd_q(s0, -1.0f, q.ang_disc) << 4 | fd_q(c0, -1.0f, q.ang_disc);
    }
    const uint32_t pair = aa1 * 20u + aa2;
    const uint32_t th = fd_q(f.angle, 0.0f, q.ang2_disc), t1 = fd_q(f.tor1, NPI, q.ang_disc), t2 = fd_q(f.tor2, NPI, q.ang_disc);
    if (q.type == FD_HASH_FD_ANGLE) return pair << 21 | ca << 18 | cb << 15 | th << 10 | t1 << 5 | t2;
    return pair << 21 | ca << 16 | cb << 11 | th << 8 | t1 << 4 | t2;
}
// residue types of a hash as the encoding's reverse_hash reports them (prefilter_amino_acid, controller/retrieve.rs:574-577)
FD_HD void fd_hash_aa_pair(uint32_t type, uint32_t h, uint32_t *aa1, uint32_t *aa2) {
    if (type == FD_HASH_PDBMOTIF) { *aa1 = (h >> 20) & 31u; *aa2 = (h >> 15) & 31u; }
    else if (type == FD_HASH_PDBMOTIF_SINCOS) { *aa1 = (h >> 21) & 31u; *aa2 = (h >> 16) & 31u; }
    else if (type == FD_HASH_FD_ANGLE || type == FD_HASH_FD_DIST) { const uint32_t pair = (h >> 21) & 0x1ffu; *aa1 = pair / 20u; *aa2 = pair % 20u; }
    else if (type == 2u) { const uint32_t pair = (h >> 23) & 0x1ffu; *aa1 = pair / 20u; *aa2 = pair % 20u; }   // TrRosetta (trrosetta.rs:98-100)
    else if (type == 4u) { *aa1 = (h >> 27) & 31u; *aa2 = (h >> 22) & 31u; }                                  // PointPairFeature (ppf.rs:56-57)
    else { *aa1 = (h >> 25) & 31u; *aa2 = (h >> 20) & 31u; }
}
FD_HD float fd_to_degrees(float rad) { return rad * 57.2957795130823208767981548141051703f; }   // f32::to_degrees
// f = the descriptor in radians (fd_pair_feature)
FD_HD uint32_t fd_hash_enc(uint32_t aa1, uint32_t aa2, fd_feature f, fd_quant q) {
    if (q.type == FD_HASH_PDBMOTIF) f.angle = fd_to_degrees(f.angle);   // get_ca_cb_angle(i, j, false), coordinate.rs:127-129
    return fd_hash_enc_feat(aa1, aa2, f, q);
}

// =============================================================================================
// Shared-subexpression form used by the index-build kernel: per-residue frames + both orientations
// of an unordered pair at once.  Everything is bit-identical to fd_pair_feature()/fd_hash_pdbtr()
// above; the savings come from IEEE identities that hold exactly:
//   a*b == b*a,  fl(-x) exact,  fl(p - q) == -fl(q - p),  normalize(-v) == -normalize(v)
// so that   cross(a, b) == -cross(b, a)   and   cross(-a, -b) == cross(a, b)   hold bit for bit.
// With v1 = cb_i-ca_i, v2 = cb_j-ca_j, v3 = cb_j-cb_i,  A = n^(v1 x v3),  B = n^(v3 x v2):
//   torsion1(i,j): r = r1_i, s =  A, t = t1_i          torsion2(i,j): r = -B, s = s2_j, t = n^(r x nv2_j)
//   torsion1(j,i): r = r1_j, s =  B, t = t1_j          torsion2(j,i): r = -A, s = s2_i, t = n^(r x nv2_i)
// and d_CA, d_CB, theta are symmetric.  8 normalisations per ordered pair become 2.
// =============================================================================================
struct fd_frame {     // 20 floats = five 16-byte loads (a sixth load for a stored cb-ca costs more than the three subtractions)
    fd_v3 ca, cb;
    fd_v3 r1, t1;    // torsion(n, ca, cb, *):  r = n^((ca-n) x (cb-ca)),  t = n^(r x n^(cb-ca))
    fd_v3 s2, nv2;   // torsion(*, cb, ca, n):  s = n^((ca-cb) x (n-ca)),  n^(ca-cb)
    float len;       // |cb-ca| as evaluated by calc_angle
    float pad;
};

FD_HD fd_frame fd_make_frame(fd_v3 n, fd_v3 ca, fd_v3 cb) {
    fd_frame F;
    F.ca = ca; F.cb = cb;
    fd_v3 v1 = fd_sub(ca, n), v2 = fd_sub(cb, ca);
    F.r1 = fd_normalize(fd_cross(v1, v2));
    F.t1 = fd_normalize(fd_cross(F.r1, fd_normalize(v2)));
    fd_v3 w2 = fd_sub(ca, cb), w3 = fd_sub(n, ca);
    F.s2 = fd_normalize(fd_cross(w2, w3));
    F.nv2 = fd_normalize(w2);
    fd_v3 a = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z};
    F.len = fd_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    F.pad = 0.0f;
    return F;
}
FD_HD fd_v3 fd_neg(fd_v3 a) { return {-a.x, -a.y, -a.z}; }

FD_HD void fd_pair_both(const fd_frame &Fi, const fd_frame &Fj, uint32_t aai, uint32_t aaj, fd_quant q, uint32_t *h_ij, uint32_t *h_ji) {
    fd_feature f, g;
    f.ca_dist = fd_dist(Fi.ca, Fj.ca);
    f.cb_dist = fd_dist(Fi.cb, Fj.cb);
    const fd_v3 v1 = {Fi.cb.x - Fi.ca.x, Fi.cb.y - Fi.ca.y, Fi.cb.z - Fi.ca.z};
    const fd_v3 v2 = {Fj.cb.x - Fj.ca.x, Fj.cb.y - Fj.ca.y, Fj.cb.z - Fj.ca.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    f.angle = fdd_acosf(dt / (Fi.len * Fj.len));
    fd_v3 v3 = fd_sub(Fj.cb, Fi.cb);
    fd_v3 A = fd_normalize(fd_cross(v1, v3));
    fd_v3 B = fd_normalize(fd_cross(v3, v2));
    // (i, j)
    f.tor1 = -fdd_atan2f(fd_dot(A, Fi.t1), fd_dot(Fi.r1, A));
    fd_v3 rB = fd_neg(B);
    fd_v3 tB = fd_normalize(fd_cross(rB, Fj.nv2));
    f.tor2 = -fdd_atan2f(fd_dot(Fj.s2, tB), fd_dot(rB, Fj.s2));
    // (j, i)
    g.ca_dist = f.ca_dist; g.cb_dist = f.cb_dist; g.angle = f.angle;
    g.tor1 = -fdd_atan2f(fd_dot(B, Fj.t1), fd_dot(Fj.r1, B));
    fd_v3 rA = fd_neg(A);
    fd_v3 tA = fd_normalize(fd_cross(rA, Fi.nv2));
    g.tor2 = -fdd_atan2f(fd_dot(Fi.s2, tA), fd_dot(rA, Fi.s2));
    if (q.type != FD_HASH_PDBTR) { *h_ij = fd_hash_enc(aai, aaj, f, q); *h_ji = fd_hash_enc(aaj, aai, g, q); return; }
    // quantise: distances and theta once
    uint32_t ca = fd_q(f.ca_dist, 2.0f, q.dist_disc), cb = fd_q(f.cb_dist, 2.0f, q.dist_disc);
    float s0, c0, s1, c1, s2, c2, s3, c3, s4, c4;
    fdd_sincosf(f.angle, &s0, &c0);
    fdd_sincosf(f.tor1, &s1, &c1);
    fdd_sincosf(f.tor2, &s2, &c2);
    fdd_sincosf(g.tor1, &s3, &c3);
    fdd_sincosf(g.tor2, &s4, &c4);
    uint32_t mid = ca << 16 | cb << 12 | fd_q(s0, -1.0f, q.ang_disc) << 10 | fd_q(c0, -1.0f, q.ang_disc) << 8;
    *h_ij = aai << 25 | aaj << 20 | mid | fd_q(s1, -1.0f, q.ang_disc) << 6 | fd_q(c1, -1.0f, q.ang_disc) << 4 |
            fd_q(s2, -1.0f, q.ang_disc) << 2 | fd_q(c2, -1.0f, q.ang_disc);
    *h_ji = aaj << 25 | aai << 20 | mid | fd_q(s3, -1.0f, q.ang_disc) << 6 | fd_q(c3, -1.0f, q.ang_disc) << 4 |
            fd_q(s4, -1.0f, q.ang_disc) << 2 | fd_q(c4, -1.0f, q.ang_disc);
}

// =============================================================================================
// Table form of the angle fields (default 4 angle bins only).  fd_bin_tables.h lists, from an
// exhaustive scan of every float through glibc, where (q(sin), q(cos)) changes as a function of
//   c = cos(theta) argument of acosf            (theta fields)
//   u = |y/x| per sign quadrant of atan2f(y, x) (torsion fields)
// so acosf / atanf / sinf / cosf / the quantiser collapse into a handful of compares — bit-identical
// by construction.  Special atan2f operands (zero, inf, NaN) take the generic path.
// Packed table (FD_BINTAB_WORDS u32): [0..5] theta thresholds 1..6, [6] theta keys (4 bits each),
// then per quadrant m: 4 thresholds (u bit patterns) + packed keys.
// =============================================================================================
#define FD_BINTAB_WORDS 27
FD_HD void fd_fill_bintab(uint32_t *t) {
    for (int k = 1; k < FD_THETA_NSEG; ++k) t[k - 1] = fd_theta_thr_bits[k];
    uint32_t pk = 0;
    for (int k = 0; k < FD_THETA_NSEG; ++k) pk |= (uint32_t)fd_theta_key[k] << (4 * k);
    t[6] = pk;
    for (int m = 0; m < 4; ++m) {
        uint32_t kk = 0;
        for (int k = 0; k < FD_TOR_MAXSEG; ++k) {
            if (k >= 1) t[7 + 5 * m + (k - 1)] = fd_tor_thr_bits[m][k];
            kk |= (uint32_t)fd_tor_key[m][k] << (4 * k);
        }
        t[7 + 5 * m + 4] = kk;
    }
}
#if FD_THETA_NSEG != 7 || FD_TOR_MAXSEG != 5
#error "fd_bin_tables.h layout changed: update FD_BINTAB_WORDS / fd_fill_bintab"
#endif

FD_HD uint32_t fd_theta_key_of(float c, const uint32_t *t) {
    uint32_t n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // c >= t  <=>  the f32 difference c - t is not negative (it is exact near zero, and x - x = +0): collect the six sign
    // bits with v_alignbit, no compare / carry chains.  A NaN c is rejected by the range test below.
    uint32_t neg = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) neg = __builtin_amdgcn_alignbit(neg, fd_f2u(c - fd_u2f(t[k])), 31);
    n = 6u - (uint32_t)__builtin_popcount(neg & 63u);
#else
#pragma unroll
    for (int k = 0; k < 6; ++k) n += (c >= fd_u2f(t[k])) ? 1u : 0u;
#endif
    uint32_t key = (t[6] >> (4u * n)) & 15u;
    return (c >= -1.0f && c <= 1.0f) ? key : 0u;   // |c| > 1 or NaN: acosf -> NaN -> both bins 0
}
// generic (exact, slow) evaluation of the two torsion bins from the atan2f operands; out of line on the device:
// it is reached only for zero / inf / NaN operands and would otherwise be inlined at every torsion site
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) uint32_t fd_tor_key_generic(float y, float x, float ang_disc) {
#else
static inline uint32_t fd_tor_key_generic(float y, float x, float ang_disc) {
#endif
    float s, c;
    fdd_sincosf(-fd_atan2f(y, x), &s, &c);
    return fd_q(s, -1.0f, ang_disc) << 2 | fd_q(c, -1.0f, ang_disc);
}
FD_HD uint32_t fd_tor_key_of(float y, float x, const uint32_t *t) {
    uint32_t hx = fd_f2u(x), hy = fd_f2u(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix >= 0x7f800000u || iy >= 0x7f800000u || ix == 0u || iy == 0u) return fd_tor_key_generic(y, x, 1.5f);
    uint32_t m = (hy >> 31) | ((hx >> 30) & 2u);
    uint32_t ub = fd_f2u(y / x) & 0x7fffffffu;
    const uint32_t *q = t + 7 + 5 * m;
    uint32_t n = (ub >= q[0] ? 1u : 0u) + (ub >= q[1] ? 1u : 0u) + (ub >= q[2] ? 1u : 0u) + (ub >= q[3] ? 1u : 0u);
    return (q[4] >> (4u * n)) & 15u;
}

// both orientations of an unordered pair, default angle bins, tables in `tab` (LDS on the device)
FD_HD void fd_pair_both_tab(const fd_frame &Fi, const fd_frame &Fj, uint32_t aai, uint32_t aaj, fd_quant q, const uint32_t *tab,
                            uint32_t *h_ij, uint32_t *h_ji) {
    float ca_dist = fd_dist(Fi.ca, Fj.ca);
    const fd_v3 v1 = {Fi.cb.x - Fi.ca.x, Fi.cb.y - Fi.ca.y, Fi.cb.z - Fi.ca.z};
    const fd_v3 v2 = {Fj.cb.x - Fj.ca.x, Fj.cb.y - Fj.ca.y, Fj.cb.z - Fj.ca.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    uint32_t kth = fd_theta_key_of(dt / (Fi.len * Fj.len), tab);
    fd_v3 v3 = fd_sub(Fj.cb, Fi.cb);
    float cb_dist = fd_sqrtf(v3.x * v3.x + v3.y * v3.y + v3.z * v3.z);   // == fd_dist(Fi.cb, Fj.cb): (p-q)^2 == (q-p)^2 bit for bit
    fd_v3 A = fd_normalize(fd_cross(v1, v3));
    fd_v3 B = fd_normalize(fd_cross(v3, v2));
    uint32_t k1 = fd_tor_key_of(fd_dot(A, Fi.t1), fd_dot(Fi.r1, A), tab);
    fd_v3 rB = fd_neg(B);
    fd_v3 tB = fd_normalize(fd_cross(rB, Fj.nv2));
    uint32_t k2 = fd_tor_key_of(fd_dot(Fj.s2, tB), fd_dot(rB, Fj.s2), tab);
    uint32_t k3 = fd_tor_key_of(fd_dot(B, Fj.t1), fd_dot(Fj.r1, B), tab);
    fd_v3 rA = fd_neg(A);
    fd_v3 tA = fd_normalize(fd_cross(rA, Fi.nv2));
    uint32_t k4 = fd_tor_key_of(fd_dot(Fi.s2, tA), fd_dot(rA, Fi.s2), tab);
    uint32_t mid = fd_q(ca_dist, 2.0f, q.dist_disc) << 16 | fd_q(cb_dist, 2.0f, q.dist_disc) << 12 | kth << 8;
    *h_ij = aai << 25 | aaj << 20 | mid | k1 << 4 | k2;
    *h_ji = aaj << 25 | aai << 20 | mid | k3 << 4 | k4;
}

#if defined(__HIPCC__)
// =============================================================================================
// Speculative evaluation of the four torsion fields (device, index build).  The exact chain spends most of its
// instructions on four IEEE normalisations (sqrt + 3 divisions each).  Here the normalisations use v_rsq_f32 and the
// atan2 operand ratio is never formed: for every table threshold t the sign of  g = |y| - t|x|  decides the segment,
// and the decision is accepted only if |g| exceeds a bound M on how far the speculative (y, x) can be from the
// operands the reference arithmetic produces.  Otherwise the pair is re-evaluated by the exact routine.
//
// Bounds (eps = 2^-24; r1, t1, s2, nv2 are unit vectors; c = v1 x v3 is computed by the SAME operations in both
// evaluations, only what follows a normalisation differs):
//   A  = c/|c|:           |A_f - A_ref|  <= 5 eps |A_k|                              (rsq 1 ulp + mul  vs  sqrt, div)
//   y1 = A.t1, x1 = r1.A: |dy|, |dx|     <= 8 eps + 6 eps (product/sum roundings)     -> E1 = 32 eps
//   X  = (-A) x nv2:      |dX_k|         <= 16 eps                                    (|dX| <= 28 eps)
//   tA = X/|X|:           |dtA_k|        <= 56 eps/|X| + 10 eps
//   y4 = s2.tA:           |dy4|          <= 97 eps/|X| + 24 eps                       -> 128 eps/|X| + 32 eps
//   x4 = (-A).s2:         |dx4|          <= 20 eps                                    -> 32 eps
//   reference decision:   |RN(y/x)| >= t  <=>  |y| - t|x| >= -eps t|x|  (|y| <= 1)    -> + 2^-20 in M
// All bounds are multiplied by FD_SPEC_SAFETY = 4.  M = E_y + t E_x; signs must be certain (|y| > E_y, |x| > E_x),
// which also routes zero / NaN / inf operands to the exact path.
// =============================================================================================
#define FD_SPEC_E1 7.62939453125e-06f      /* 4 * 32 eps = 2^-17 */
#define FD_SPEC_EY_SLACK 3.814697265625e-06f /* 4 * 2^-20 */
#define FD_SPEC_EX4 7.62939453125e-06f     /* 4 * 32 eps */
#define FD_SPEC_EY4_A 3.0517578125e-05f    /* 4 * 128 eps = 2^-15, times 1/|X| */
#define FD_SPEC_EY4_B 7.62939453125e-06f   /* 4 * 32 eps */

// tab_f: per quadrant m, 4 float thresholds (padding clamped to FLT_MAX) + packed keys (same layout as the exact table).
// `margin` accumulates min(decision distance - bound) over everything decided here; the result is trusted iff the final
// margin is > 0 (a NaN operand compares false in `fin`).
__device__ __forceinline__ uint32_t fd_tor_key_spec(float y, float x, float Ey, float Ex, const uint32_t *tab_f, float &margin, bool &fin) {
    const float ay = __builtin_fabsf(y), ax = __builtin_fabsf(x);
    fin = fin && (ay < 2.0f) && (ax < 2.0f);
    margin = __builtin_fminf(margin, __builtin_fminf(ay - Ey, ax - Ex));
    const uint32_t m = (fd_f2u(y) >> 31) | ((fd_f2u(x) >> 30) & 2u);
    const uint32_t *q = tab_f + 7 + 5 * m;
    uint32_t neg = 0;   // sign bits of g_0..g_3 (thresholds ascend: the negatives form the tail)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = fd_u2f(q[k]);
        const float g = __builtin_fmaf(-t, ax, ay);
        const float M = __builtin_fmaf(t, Ex, Ey);
        neg = __builtin_amdgcn_alignbit(neg, fd_f2u(g), 31);   // (neg << 1) | sign(g)
        margin = __builtin_fminf(margin, __builtin_fabsf(g) - M);
    }
    const uint32_t n = 4u - (uint32_t)__builtin_popcount(neg & 15u);
    return (q[4] >> (4u * n)) & 15u;
}

__device__ __forceinline__ fd_v3 fd_normalize_spec(fd_v3 v, float *rs_out) {
    const float rs = __builtin_amdgcn_rsqf(v.x * v.x + v.y * v.y + v.z * v.z);
    *rs_out = rs;
    return {v.x * rs, v.y * rs, v.z * rs};
}

// A distance field without the IEEE square root (index build, default 16 distance bins): the bin is a step function of the SQUARED distance x whose
// breakpoints tab_d[k] = the smallest float of bin k were found by pushing every float through sqrtf + the quantiser (fd_dist_table.h, generated by
// tools/gen_dist_table.c).  v_sqrt_f32 (1 ulp) gives a first guess b that is provably within one bin of the truth (the generator checks two ulps to
// either side for every float), and x is compared with the guess's own two breakpoints: bin = b + (x >= T[b + 1]) - (x < T[b]).  ~13 VALU slots
// against ~30 for the correctly rounded sqrt and the saturating cast's guards.  A guess beyond the table (CB distance > ~37 A, inf) clears ok:
// the pair takes the exact routine.  NaN: guess 0, both compares false -> bin 0, what the saturating cast gives.
__device__ __forceinline__ uint32_t fd_dist_bin_tab(float x, float disc, const uint32_t *tab_d, bool &ok) {
    const float g = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_sqrtf(x) - 2.0f, disc, 0.5f), 0.0f, 40.0f);
    const uint32_t b0 = (uint32_t)g;
    const uint32_t b = b0 < (uint32_t)(FD_DIST_NTHR - 2) ? b0 : (uint32_t)(FD_DIST_NTHR - 2);
    ok = ok && b0 == b;
    const float t0 = fd_u2f(tab_d[b]), t1 = fd_u2f(tab_d[b + 1]);
    return b + (x >= t1 ? 1u : 0u) - (x < t0 ? 1u : 0u);
}

// returns false when any field is not provably identical to the reference arithmetic (caller falls back to fd_pair_both_tab)
// DT: the two distance fields from the squared-distance table tab_d (fd_dist_bin_tab) instead of sqrt + quantiser
template <bool DT = false>
__device__ __forceinline__ bool fd_pair_both_spec(const fd_frame &Fi, const fd_frame &Fj, uint32_t aai, uint32_t aaj, fd_quant q,
                                                  const uint32_t *tab, const uint32_t *tab_f, uint32_t *h_ij, uint32_t *h_ji, const uint32_t *tab_d = nullptr) {
    const fd_v3 v1 = {Fi.cb.x - Fi.ca.x, Fi.cb.y - Fi.ca.y, Fi.cb.z - Fi.ca.z};
    const fd_v3 v2 = {Fj.cb.x - Fj.ca.x, Fj.cb.y - Fj.ca.y, Fj.cb.z - Fj.ca.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    uint32_t kth = fd_theta_key_of(dt / (Fi.len * Fj.len), tab);     // exact: one division, no normalisation upstream
    fd_v3 v3 = fd_sub(Fj.cb, Fi.cb);
    const float cb_d2 = v3.x * v3.x + v3.y * v3.y + v3.z * v3.z;        // sqrt of it == fd_dist(Fi.cb, Fj.cb): (p-q)^2 == (q-p)^2 bit for bit
    bool fin = true;
    uint32_t q_ca, q_cb;
    if (DT) {
        q_ca = fd_dist_bin_tab(fd_dist2(Fi.ca, Fj.ca), q.dist_disc, tab_d, fin);
        q_cb = fd_dist_bin_tab(cb_d2, q.dist_disc, tab_d, fin);
    } else {
        q_ca = fd_q(fd_dist(Fi.ca, Fj.ca), 2.0f, q.dist_disc);
        q_cb = fd_q(fd_sqrtf(cb_d2), 2.0f, q.dist_disc);
    }
    float rsA, rsB, rsXA, rsXB;
    fd_v3 A = fd_normalize_spec(fd_cross(v1, v3), &rsA);
    fd_v3 B = fd_normalize_spec(fd_cross(v3, v2), &rsB);
    float margin = 1.0f;
    const float E1y = FD_SPEC_E1 + FD_SPEC_EY_SLACK;
    uint32_t k1 = fd_tor_key_spec(fd_dot(A, Fi.t1), fd_dot(Fi.r1, A), E1y, FD_SPEC_E1, tab_f, margin, fin);
    uint32_t k3 = fd_tor_key_spec(fd_dot(B, Fj.t1), fd_dot(Fj.r1, B), E1y, FD_SPEC_E1, tab_f, margin, fin);
    fd_v3 rB = fd_neg(B), rA = fd_neg(A);
    fd_v3 tB = fd_normalize_spec(fd_cross(rB, Fj.nv2), &rsXB);
    fd_v3 tA = fd_normalize_spec(fd_cross(rA, Fi.nv2), &rsXA);
    const float E4c = FD_SPEC_EY4_B + FD_SPEC_EY_SLACK;
    uint32_t k2 = fd_tor_key_spec(fd_dot(Fj.s2, tB), fd_dot(rB, Fj.s2), __builtin_fmaf(FD_SPEC_EY4_A, rsXB, E4c), FD_SPEC_EX4, tab_f, margin, fin);
    uint32_t k4 = fd_tor_key_spec(fd_dot(Fi.s2, tA), fd_dot(rA, Fi.s2), __builtin_fmaf(FD_SPEC_EY4_A, rsXA, E4c), FD_SPEC_EX4, tab_f, margin, fin);
    uint32_t mid = q_ca << 16 | q_cb << 12 | kth << 8;
    *h_ij = aai << 25 | aaj << 20 | mid | k1 << 4 | k2;
    *h_ji = aaj << 25 | aai << 20 | mid | k3 << 4 | k4;
    // rsq of a zero / denormal cross product is inf: margins become NaN or -inf; NaN must fail, so test "> 0" positively
    return fin && (margin > 0.0f) && (rsXA < 3.0e38f) && (rsXB < 3.0e38f) && (rsA < 3.0e38f) && (rsB < 3.0e38f);
}
#endif
