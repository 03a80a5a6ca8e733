/* fdo_geometry.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates: src/utils/convert.rs, src/structure/coordinate.rs, src/structure/core.rs:378-403,
 * src/controller/feature.rs:11-24,84-99,198-231, src/geometry/pdb_tr.rs:21-162,
 * src/utils/combination.rs:4-44. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

/* Rust `f32 as u32`: saturating, NaN -> 0 */
static uint32_t sat_u32(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

/* src/utils/convert.rs:53-81 map_aa_to_u8 */
uint8_t fdo_map_aa_to_u8(const uint8_t aa[3]) {
    static const struct { const char *names; uint8_t v; } tab[] = {
        {"ALA ABA ORN DAL AIB ALC MDO MAA DAB", 0},
        {"ARG DAR CIR AGM", 1},
        {"ASN DSG MEN SNN", 2},
        {"ASP 0TD DAS IAS PHD BFD ASX", 3},
        {"CYS CSO CSD CME OCS CAS CSX CSS YCM DCY SMC SCH SCY CAF SNC SEC", 4},
        {"GLN DGN CRQ MEQ", 5},
        {"GLU PCA DGL CGU FGA B3E GLX", 6},
        {"GLY CR2 SAR GHP GL3", 7},
        {"HIS HIC DHI NEP CR8 MHS", 8},
        {"ILE DIL", 9},
        {"LEU DLE NLE MLE MK8", 10},
        {"LYS KCX LLP MLY M3L ALY MLZ DLY KPI PYL", 11},
        {"MET MSE FME NRQ CXM SME MHO MED", 12},
        {"PHE DPN PHI MEA PHL", 13},
        {"PRO HYP DPR", 14},
        {"SER CSH SEP DSN SAC GYS DHA OAS", 15},
        {"THR TPO CRO DTH BMT CRF", 16},
        {"TRP DTR TRQ TOX 0AF", 17},
        {"TYR PTR TYS TPQ DTY OMY", 18},
        {"VAL DVA MVA FVA", 19},
    };
    for (size_t t = 0; t < sizeof tab / sizeof tab[0]; ++t) {
        const char *p = tab[t].names;
        while (*p) {
            if (p[0] == (char)aa[0] && p[1] == (char)aa[1] && p[2] == (char)aa[2]) return tab[t].v;
            p += 3;
            if (*p == ' ') ++p;
        }
    }
    return 255;
}

/* src/utils/convert.rs map_u8_to_aa */
const char *fdo_map_u8_to_aa(uint8_t aa) {
    static const char *names[20] = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
                                    "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"};
    return aa < 20 ? names[aa] : "UNK";
}

/* src/utils/convert.rs:32-36 discretize_f32_value_into_u32 */
uint32_t fdo_discretize(float val, float min, float max, float num_bin) {
    float cont_f = (max - min) / (num_bin - 1.0f);
    float disc_f = 1.0f / cont_f;
    return sat_u32((val - min) * disc_f + 0.5f);
}
static float continuize(uint32_t val, float min, float max, float num_bin) {
    float cont_f = (max - min) / (num_bin - 1.0f);
    return (float)val * cont_f + min;
}

typedef struct { float x, y, z; } v3;
static v3 sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static float norm(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static v3 normalize(v3 a) {
    float n = norm(a);
    v3 r = {a.x / n, a.y / n, a.z / n};
    return r;
}
/* coordinate.rs:109-115 */
static float calc_distance(v3 a, v3 b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}
/* coordinate.rs:118-133 (powf(x,2.0) == x*x after LLVM's pow->mul fold) */
static float calc_angle(v3 a, v3 b, v3 c, v3 d) {
    v3 v1 = {b.x - a.x, b.y - a.y, b.z - a.z};
    v3 v2 = {d.x - c.x, d.y - c.y, d.z - c.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    float l1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    float l2 = sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    float cs = dt / (l1 * l2);
    return acosf(cs);
}
/* coordinate.rs:204-215 */
static float calc_torsion_radian(v3 a, v3 b, v3 c, v3 d) {
    v3 v1 = sub(b, a), v2 = sub(c, b), v3_ = sub(d, c);
    v3 r = normalize(cross(v1, v2));
    v3 s = normalize(cross(v2, v3_));
    v3 t = normalize(cross(r, normalize(v2)));
    float x = dot(r, s);
    float y = dot(s, t);
    return -atan2f(y, x);
}

static v3 at(const float *p, int64_t i) { v3 r = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; return r; }

/* HashType the restatement encodes with (geometry/core.rs:26-40): 3 PDBTrRosetta unless a test selects one of the other encodings over
 * the same descriptor (0 PDBMotif, 1 PDBMotifSinCos, 7 FolddiscoAngle, 8 FolddiscoDist).  Test infrastructure: a process-wide switch. */
static uint32_t g_hash_type = 3;
int fdo_set_hash_type(uint32_t t) {
    if (t > 8) return -1;
    g_hash_type = t;
    return 0;
}
uint32_t fdo_get_hash_type(void) { return g_hash_type; }

/* controller/feature.rs:11-24,26-99 + structure/core.rs:255-297,378-403: the five encodings share the pair rules (both aa known,
 * CA and CB present, d_CA <= cutoff); tau1 / tau2 are simply unused by the two PDBMotif forms */
/* coordinate.rs:150-162 */
static float calc_angle_radian(v3 a, v3 b, v3 c) {
    v3 v1 = {a.x - b.x, a.y - b.y, a.z - b.z};
    v3 v2 = {c.x - b.x, c.y - b.y, c.z - b.z};
    float dt = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    float l1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    float l2 = sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    return acosf(dt / (l1 * l2));
}
static v3 at(const float *p, int64_t i);
static float calc_torsion_radian(v3 a, v3 b, v3 c, v3 d);
/* map_aa_to_u8_group (convert.rs:85-130) through the residue type: the two tables agree name by name (checked against the source) */
static const uint8_t AA_GROUP[20] = {0, 3, 2, 3, 0, 2, 3, 0, 3, 1, 1, 3, 1, 1, 0, 0, 2, 1, 2, 1};

/* the four encodings with their own descriptors (controller/feature.rs:67-82, 100-190; structure/core.rs:310-345, 405-437) */
static int pair_feature_other(const fdo_structure *s, int64_t i, int64_t j, float dist_cutoff, float f[9]) {
    const int64_t n = s->n;
    v3 ca1 = at(s->ca_xyz, i), ca2 = at(s->ca_xyz, j);
    if (g_hash_type == 2) {            /* TrRosetta: get_trrosetta_feature, cutoff on the CB distance */
        if (!s->cb_ok[i] || !s->cb_ok[j]) return 0;
        v3 cb1 = at(s->cb_xyz, i), cb2 = at(s->cb_xyz, j), n1 = at(s->n_xyz, i), n2 = at(s->n_xyz, j);
        float cb_dist = calc_distance(cb1, cb2);
        if (cb_dist > dist_cutoff) return 0;
        f[0] = (float)s->aa[i]; f[1] = (float)s->aa[j]; f[2] = cb_dist;
        f[3] = calc_torsion_radian(ca1, cb1, cb2, ca2);
        f[4] = calc_torsion_radian(n1, ca1, cb1, cb2);
        f[5] = calc_torsion_radian(cb1, cb2, ca2, n2);
        f[6] = calc_angle_radian(ca1, cb1, cb2);
        f[7] = calc_angle_radian(cb1, cb2, ca2);
        return 1;
    }
    if (g_hash_type == 4) {            /* PointPairFeature: get_ppf over (cb1 - ca1, cb2 - ca1), coordinate.rs:93-102 */
        if (!s->cb_ok[i] || !s->cb_ok[j]) return 0;
        v3 a = sub(at(s->cb_xyz, i), ca1), b = sub(at(s->cb_xyz, j), ca1);
        v3 n1 = normalize(a), n2 = normalize(b), d = sub(b, a), nd = normalize(d);
        float dist = norm(d);
        float a1 = acosf(dot(n1, nd)), a2 = acosf(dot(n2, nd)), a3 = acosf(dot(n1, n2));
        if (dist > dist_cutoff) return 0;
        f[0] = (float)s->aa[i]; f[1] = (float)s->aa[j]; f[2] = dist; f[3] = a1; f[4] = a2; f[5] = a3;
        return 1;
    }
    if (i == 0 || j == 0 || i == n - 1 || j == n - 1) return 0;
    if (g_hash_type == 5) {            /* TertiaryInteraction: seven angles between consecutive-CA directions */
        float ca_dist = calc_distance(ca1, ca2);
        if (ca_dist > dist_cutoff) return 0;
        v3 u1 = normalize(sub(ca1, at(s->ca_xyz, i - 1))), u2 = normalize(sub(at(s->ca_xyz, i + 1), ca1));
        v3 u3 = normalize(sub(ca2, at(s->ca_xyz, j - 1))), u4 = normalize(sub(at(s->ca_xyz, j + 1), ca2));
        v3 u5 = normalize(sub(ca2, ca1));
        f[0] = acosf(dot(u1, u2)); f[1] = acosf(dot(u3, u4)); f[2] = acosf(dot(u1, u5)); f[3] = acosf(dot(u3, u5));
        f[4] = acosf(dot(u1, u4)); f[5] = acosf(dot(u2, u3)); f[6] = acosf(dot(u1, u3));
        f[7] = ca_dist; f[8] = (float)j - (float)i;
        return 1;
    }
    /* Hybrid: residue groups + the PDBTrRosetta descriptor + the two pseudo backbone torsions (core.rs:405-437) */
    if (!s->cb_ok[i] || !s->cb_ok[j]) return 0;
    v3 cb1 = at(s->cb_xyz, i), cb2 = at(s->cb_xyz, j), n1 = at(s->n_xyz, i), n2 = at(s->n_xyz, j);
    float ca_dist = calc_distance(ca1, ca2);
    if (ca_dist > dist_cutoff) return 0;
    f[0] = (float)AA_GROUP[s->aa[i]]; f[1] = (float)AA_GROUP[s->aa[j]];
    f[2] = ca_dist; f[3] = calc_distance(cb1, cb2); f[4] = calc_angle(ca1, cb1, ca2, cb2);
    f[5] = calc_torsion_radian(n1, ca1, cb1, cb2); f[6] = calc_torsion_radian(cb1, cb2, ca2, n2);
    f[7] = calc_torsion_radian(at(s->ca_xyz, i - 1), n1, ca1, at(s->ca_xyz, i + 1));
    f[8] = calc_torsion_radian(at(s->ca_xyz, j - 1), n2, ca2, at(s->ca_xyz, j + 1));
    return 1;
}

int fdo_pair_feature(const fdo_structure *s, int64_t i, int64_t j, float dist_cutoff, float feature[9]) {
    if (i == j) return 0;
    if (s->aa[i] == 255 || s->aa[j] == 255) return 0;
    if (g_hash_type == 2 || (g_hash_type >= 4 && g_hash_type <= 6)) return pair_feature_other(s, i, j, dist_cutoff, feature);
    if (!s->cb_ok[i] || !s->cb_ok[j]) return 0;
    v3 ca1 = at(s->ca_xyz, i), ca2 = at(s->ca_xyz, j);
    v3 cb1 = at(s->cb_xyz, i), cb2 = at(s->cb_xyz, j);
    v3 n1 = at(s->n_xyz, i), n2 = at(s->n_xyz, j);
    float ca_dist = calc_distance(ca1, ca2);
    if (ca_dist > dist_cutoff) return 0;
    float cb_dist = calc_distance(cb1, cb2);
    float ang = calc_angle(ca1, cb1, ca2, cb2);
    float t1 = calc_torsion_radian(n1, ca1, cb1, cb2);
    float t2 = calc_torsion_radian(cb1, cb2, ca2, n2);
    feature[0] = (float)s->aa[i];
    feature[1] = (float)s->aa[j];
    feature[2] = ca_dist;
    feature[3] = cb_dist;
    feature[4] = g_hash_type == 0 ? ang * 57.2957795130823208767981548141051703f : ang; /* PDBMotif: get_ca_cb_angle(.., false) = degrees */
    feature[5] = t1;
    feature[6] = t2;
    return 1;
}

/* geometry/pdb_tr.rs:21-75 */
uint32_t fdo_hash_pdbtr(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = nbin_dist > 16 ? 16.0f : (nbin_dist == 0 ? 16.0f : (float)nbin_dist);
    float na = nbin_angle > 4 ? 4.0f : (nbin_angle == 0 ? 4.0f : (float)nbin_angle);
    uint32_t res1 = sat_u32(f[0]), res2 = sat_u32(f[1]);
    uint32_t ca = fdo_discretize(f[2], 2.0f, 20.0f, nd);
    uint32_t cb = fdo_discretize(f[3], 2.0f, 20.0f, nd);
    uint32_t s0 = fdo_discretize(sinf(f[4]), -1.0f, 1.0f, na);
    uint32_t c0 = fdo_discretize(cosf(f[4]), -1.0f, 1.0f, na);
    uint32_t s1 = fdo_discretize(sinf(f[5]), -1.0f, 1.0f, na);
    uint32_t c1 = fdo_discretize(cosf(f[5]), -1.0f, 1.0f, na);
    uint32_t s2 = fdo_discretize(sinf(f[6]), -1.0f, 1.0f, na);
    uint32_t c2 = fdo_discretize(cosf(f[6]), -1.0f, 1.0f, na);
    /* Rust `<<` on u32 panics on overflow shifts only; values are OR-ed unmasked */
    return res1 << 25 | res2 << 20 | ca << 16 | cb << 12 | s0 << 10 | c0 << 8 | s1 << 6 | c1 << 4 | s2 << 2 | c2;
}

static float pick_bins(uint64_t n, float cap, float dflt) { return n > (uint64_t)cap ? cap : (n == 0 ? dflt : (float)n); }
/* geometry/pdb_motif.rs:26-50 (theta in degrees, 0..180) */
static uint32_t hash_pdbmotif(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 32.0f, 18.0f), na = pick_bins(nbin_angle, 32.0f, 9.0f);
    uint32_t res1 = sat_u32(f[0]), res2 = sat_u32(f[1]);
    uint32_t ca = fdo_discretize(f[2], 2.0f, 20.0f, nd), cb = fdo_discretize(f[3], 2.0f, 20.0f, nd);
    uint32_t ang = fdo_discretize(f[4], 0.0f, 180.0f, na);
    return res1 << 20 | res2 << 15 | ca << 10 | cb << 5 | ang;
}
/* geometry/pdb_motif_sincos.rs:17-53 */
static uint32_t hash_pdbmotif_sincos(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 16.0f, 8.0f), na = pick_bins(nbin_angle, 16.0f, 3.0f);
    uint32_t res1 = sat_u32(f[0]), res2 = sat_u32(f[1]);
    uint32_t ca = fdo_discretize(f[2], 2.0f, 20.0f, nd), cb = fdo_discretize(f[3], 2.0f, 20.0f, nd);
    uint32_t sn = fdo_discretize(sinf(f[4]), -1.0f, 1.0f, na), cs = fdo_discretize(cosf(f[4]), -1.0f, 1.0f, na);
    return res1 << 21 | res2 << 16 | ca << 12 | cb << 8 | sn << 4 | cs;
}
/* geometry/folddisco_angle.rs:25-70 (8 / 32 / 32 bins) and folddisco_dist.rs:22-63 (32 / 8 / 16 bins): angles binned in radians */
static uint32_t hash_folddisco(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle, int dist_form) {
    const float PI_F = 3.14159274f;
    float nd = dist_form ? pick_bins(nbin_dist, 32.0f, 32.0f) : pick_bins(nbin_dist, 8.0f, 8.0f);
    float na = dist_form ? pick_bins(nbin_angle, 16.0f, 16.0f) : pick_bins(nbin_angle, 32.0f, 32.0f);
    float n180 = dist_form ? 8.0f : 32.0f;
    uint32_t res1 = sat_u32(f[0]), res2 = sat_u32(f[1]);
    uint32_t pair = res1 * 20u + res2; /* map_aa_u32_pair_to_u32 (convert.rs:203-207; asserts < 512) */
    uint32_t ca = fdo_discretize(f[2], 2.0f, 20.0f, nd), cb = fdo_discretize(f[3], 2.0f, 20.0f, nd);
    uint32_t th = fdo_discretize(f[4], 0.0f, PI_F, na < n180 ? na : n180);
    uint32_t p1 = fdo_discretize(f[5], -PI_F, PI_F, na), p2 = fdo_discretize(f[6], -PI_F, PI_F, na);
    return dist_form ? (pair << 21 | ca << 16 | cb << 11 | th << 8 | p1 << 4 | p2) : (pair << 21 | ca << 18 | cb << 15 | th << 10 | p1 << 5 | p2);
}
/* --multiple-bins (test-only process-wide switch like the encoding): the (dist, angle) bin pairs every residue pair is hashed with */
static uint64_t g_multi_n = 0, g_multi[8][2];
int fdo_set_multiple_bins(uint64_t n, const uint64_t *pairs) {
    if (n > 8) return -1;
    g_multi_n = n;
    for (uint64_t k = 0; k < n; ++k) { g_multi[k][0] = pairs[2 * k]; g_multi[k][1] = pairs[2 * k + 1]; }
    return 0;
}
uint64_t fdo_multiple_bins(uint64_t out[16]) {
    for (uint64_t k = 0; k < g_multi_n; ++k) { out[2 * k] = g_multi[k][0]; out[2 * k + 1] = g_multi[k][1]; }
    return g_multi_n;
}
/* geometry/trrosetta.rs:56-96 (caps 8 / 4; its default call passes 8 / 3) */
static uint32_t hash_trrosetta(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 8.0f, 8.0f), na = pick_bins(nbin_angle, 4.0f, 3.0f);
    uint32_t h = (sat_u32(f[0]) * 20u + sat_u32(f[1])) << 23 | fdo_discretize(f[2], 2.0f, 20.0f, nd) << 20;
    for (int k = 0; k < 5; ++k)
        h |= fdo_discretize(sinf(f[3 + k]), -1.0f, 1.0f, na) << (18 - 4 * k) | fdo_discretize(cosf(f[3 + k]), -1.0f, 1.0f, na) << (16 - 4 * k);
    return h;
}
/* geometry/ppf.rs:15-49 */
static uint32_t hash_ppf(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 16.0f, 8.0f), na = pick_bins(nbin_angle, 8.0f, 3.0f);
    uint32_t h = sat_u32(f[0]) << 27 | sat_u32(f[1]) << 22 | fdo_discretize(f[2], 2.0f, 20.0f, nd) << 18;
    for (int k = 0; k < 3; ++k)
        h |= fdo_discretize(sinf(f[3 + k]), -1.0f, 1.0f, na) << (15 - 6 * k) | fdo_discretize(cosf(f[3 + k]), -1.0f, 1.0f, na) << (12 - 6 * k);
    return h;
}
/* geometry/tertiary_interaction.rs:21-80; `feature[8] as u32 + 4` saturates negative offsets to 0 first */
static uint32_t hash_tertiary(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 16.0f, 8.0f), na = pick_bins(nbin_angle, 8.0f, 3.0f);
    uint32_t h = 0;
    for (int k = 0; k < 7; ++k) h |= fdo_discretize(cosf(f[k]), -1.0f, 1.0f, na) << (26 - 3 * k);
    uint32_t seq = f[8] < -4.0f ? 0u : (f[8] > 4.0f ? 8u : sat_u32(f[8]) + 4u);
    return h | fdo_discretize(f[7], 2.0f, 20.0f, nd) << 4 | seq;
}
/* geometry/hybrid.rs:20-91 */
static uint32_t hash_hybrid(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    float nd = pick_bins(nbin_dist, 16.0f, 16.0f), na = pick_bins(nbin_angle, 4.0f, 4.0f);
    uint32_t h = sat_u32(f[0]) << 30 | sat_u32(f[1]) << 28 | fdo_discretize(f[2], 2.0f, 20.0f, nd) << 24 | fdo_discretize(f[3], 2.0f, 20.0f, nd) << 20;
    for (int k = 0; k < 5; ++k)
        h |= fdo_discretize(sinf(f[4 + k]), -1.0f, 1.0f, na) << (18 - 4 * k) | fdo_discretize(cosf(f[4 + k]), -1.0f, 1.0f, na) << (16 - 4 * k);
    return h;
}
/* GeometricHash::perfect_hash_as_u32 (geometry/core.rs:213-246): the encoding's own perfect_hash, each count's 0 / cap handled by itself */
uint32_t fdo_hash_cfg(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    switch (g_hash_type) {
    case 2: return hash_trrosetta(f, nbin_dist, nbin_angle);
    case 4: return hash_ppf(f, nbin_dist, nbin_angle);
    case 5: return hash_tertiary(f, nbin_dist, nbin_angle);
    case 6: return hash_hybrid(f, nbin_dist, nbin_angle);
    case 0: return hash_pdbmotif(f, nbin_dist, nbin_angle);
    case 1: return hash_pdbmotif_sincos(f, nbin_dist, nbin_angle);
    case 7: return hash_folddisco(f, nbin_dist, nbin_angle, 0);
    case 8: return hash_folddisco(f, nbin_dist, nbin_angle, 1);
    default: return fdo_hash_pdbtr(f, nbin_dist, nbin_angle);
    }
}
/* GeometricHash::perfect_hash[_default] (geometry/core.rs:195-246) as the callers use it: either bin count 0 -> the encoding's defaults
 * (controller/feature.rs:216-223, query.rs:72-77) */
uint32_t fdo_hash_any(const float f[9], uint64_t nbin_dist, uint64_t nbin_angle) {
    if (nbin_dist == 0 || nbin_angle == 0) nbin_dist = nbin_angle = 0;
    if (g_hash_type == 2 || (g_hash_type >= 4 && g_hash_type <= 6)) return fdo_hash_cfg(f, nbin_dist, nbin_angle);
    switch (g_hash_type) {
    case 0: return hash_pdbmotif(f, nbin_dist, nbin_angle);
    case 1: return hash_pdbmotif_sincos(f, nbin_dist, nbin_angle);
    case 7: return hash_folddisco(f, nbin_dist, nbin_angle, 0);
    case 8: return hash_folddisco(f, nbin_dist, nbin_angle, 1);
    default: return fdo_hash_pdbtr(f, nbin_dist, nbin_angle);
    }
}

/* geometry/pdb_tr.rs:95-136 reverse_hash (default bins), angles in degrees */
void fdo_reverse_hash_pdbtr(uint32_t h, float out[7]) {
    const float PIS_IN_180 = 57.2957795130823208767981548141051703f; /* f32::to_degrees */
    out[0] = (float)((h >> 25) & 0x1f);
    out[1] = (float)((h >> 20) & 0x1f);
    out[2] = continuize((h >> 16) & 0xf, 2.0f, 20.0f, 16.0f);
    out[3] = continuize((h >> 12) & 0xf, 2.0f, 20.0f, 16.0f);
    float s0 = continuize((h >> 10) & 3, -1.0f, 1.0f, 4.0f), c0 = continuize((h >> 8) & 3, -1.0f, 1.0f, 4.0f);
    float s1 = continuize((h >> 6) & 3, -1.0f, 1.0f, 4.0f), c1 = continuize((h >> 4) & 3, -1.0f, 1.0f, 4.0f);
    float s2 = continuize((h >> 2) & 3, -1.0f, 1.0f, 4.0f), c2 = continuize(h & 3, -1.0f, 1.0f, 4.0f);
    out[4] = atan2f(s0, c0) * PIS_IN_180;
    out[5] = atan2f(s1, c1) * PIS_IN_180;
    out[6] = atan2f(s2, c2) * PIS_IN_180;
}
/* geometry/pdb_tr.rs:158-162 */
int fdo_hash_is_symmetric(uint32_t h) {
    /* pdb_motif.rs:98-102, pdb_motif_sincos.rs:105-109: residue fields equal; folddisco_angle.rs:133-137, folddisco_dist.rs:126-130:
     * residues of the pair equal and the two torsion fields equal (their continuize map is strictly increasing) */
    if (g_hash_type == 2) {   /* trrosetta.rs:158-162 with reverse_hash_default (3 angle bins): residues, theta1 == theta2, phi1 == phi2 */
        uint32_t pr = (h >> 23) & 0x1ffu;
        float a[5];
        for (int k = 0; k < 5; ++k)
            a[k] = atan2f(continuize((h >> (18 - 4 * k)) & 3u, -1.0f, 1.0f, 3.0f), continuize((h >> (16 - 4 * k)) & 3u, -1.0f, 1.0f, 3.0f)) *
                   57.2957795130823208767981548141051703f;
        return pr / 20u == pr % 20u && a[1] == a[2] && a[3] == a[4];
    }
    if (g_hash_type == 4) {   /* ppf.rs:124-127: residues and the first two angles (default 3 bins) */
        float a[2];
        for (int k = 0; k < 2; ++k)
            a[k] = atan2f(continuize((h >> (15 - 6 * k)) & 7u, -1.0f, 1.0f, 3.0f), continuize((h >> (12 - 6 * k)) & 7u, -1.0f, 1.0f, 3.0f)) *
                   57.2957795130823208767981548141051703f;
        return ((h >> 27) & 31u) == ((h >> 22) & 31u) && a[0] == a[1];
    }
    if (g_hash_type == 5) return 0;   /* tertiary_interaction.rs:145-150 */
    if (g_hash_type == 6) {   /* hybrid.rs:185-189: residue groups and the two side-chain torsions (4 angle bins) */
        float a[2];
        for (int k = 0; k < 2; ++k)
            a[k] = atan2f(continuize((h >> (14 - 4 * k)) & 3u, -1.0f, 1.0f, 4.0f), continuize((h >> (12 - 4 * k)) & 3u, -1.0f, 1.0f, 4.0f)) *
                   57.2957795130823208767981548141051703f;
        return ((h >> 30) & 3u) == ((h >> 28) & 3u) && a[0] == a[1];
    }
    if (g_hash_type == 0) return ((h >> 20) & 31u) == ((h >> 15) & 31u);
    if (g_hash_type == 1) return ((h >> 21) & 31u) == ((h >> 16) & 31u);
    if (g_hash_type == 7 || g_hash_type == 8) {
        uint32_t pair = (h >> 21) & 0x1ffu;
        uint32_t p1 = g_hash_type == 7 ? (h >> 5) & 31u : (h >> 4) & 15u, p2 = g_hash_type == 7 ? h & 31u : h & 15u;
        return pair / 20u == pair % 20u && p1 == p2;
    }
    float v[7];
    fdo_reverse_hash_pdbtr(h, v);
    return v[0] == v[1] && v[5] == v[6];
}

/* controller/feature.rs:198-231 with CombinationIterator order (combination.rs:23-44) */
/* the same into a caller-owned buffer that only ever grows (*buf / *cap): the multi-threaded bench driver keeps one per thread, so that
 * hashing a structure costs no mmap / munmap (a 300-residue structure's list is ~128 KB = glibc's mmap threshold; the TLB shootdown of
 * every munmap serialised 256 threads) */
int fdo_hash_structure_buf(const fdo_structure *s, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff, uint32_t **buf, uint64_t *cap_io,
                           uint64_t *n_out) {
    uint64_t cap = *cap_io, n = 0;
    uint32_t *v = *buf;
    if (!v || cap < 1024) { cap = 1024; v = (uint32_t *)realloc(v, cap * sizeof *v); }
    float feat[9] = {0};
    for (int64_t i = 0; i < s->n; ++i) {
        for (int64_t j = 0; j < s->n; ++j) {
            if (i == j) continue;
            if (!fdo_pair_feature(s, i, j, dist_cutoff, feat)) continue;
            for (uint64_t k = 0; k < (g_multi_n ? g_multi_n : 1); ++k) {
                uint32_t h = g_multi_n ? fdo_hash_cfg(feat, g_multi[k][0], g_multi[k][1]) : fdo_hash_any(feat, nbin_dist, nbin_angle);
                if (n == cap) { cap *= 2; v = (uint32_t *)realloc(v, cap * sizeof *v); }
                v[n++] = h;
            }
        }
    }
    *buf = v; *cap_io = cap; *n_out = n;
    return 0;
}
int fdo_hash_structure(const fdo_structure *s, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff,
                       uint32_t **out, uint64_t *n_out) {
    uint64_t cap = 1024, n = 0;
    uint32_t *v = (uint32_t *)malloc(cap * sizeof *v);
    float feat[9] = {0};
    for (int64_t i = 0; i < s->n; ++i) {
        for (int64_t j = 0; j < s->n; ++j) {
            if (i == j) continue;
            if (!fdo_pair_feature(s, i, j, dist_cutoff, feat)) continue;
            /* one hash per bin pair of --multiple-bins (feature.rs:211-215), else the single configuration */
            for (uint64_t k = 0; k < (g_multi_n ? g_multi_n : 1); ++k) {
                uint32_t h = g_multi_n ? fdo_hash_cfg(feat, g_multi[k][0], g_multi[k][1]) : fdo_hash_any(feat, nbin_dist, nbin_angle);
                if (n == cap) { cap *= 2; v = (uint32_t *)realloc(v, cap * sizeof *v); }
                v[n++] = h;
            }
        }
    }
    *out = v;
    *n_out = n;
    return 0;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}
/* controller/mod.rs:343-345 sort_unstable + dedup */
uint64_t fdo_sort_dedup_u32(uint32_t *v, uint64_t n) {
    if (n == 0) return 0;
    qsort(v, n, sizeof *v, cmp_u32);
    uint64_t m = 1;
    for (uint64_t k = 1; k < n; ++k)
        if (v[k] != v[m - 1]) v[m++] = v[k];
    return m;
}
void fdo_free(void *p) { free(p); }
