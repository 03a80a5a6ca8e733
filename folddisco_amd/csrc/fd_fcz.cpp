// fd_fcz.cpp — Foldcomp (.fcz / Foldcomp database) input: compressed entry -> the atom records the reference's ingest consumes.
//
// The reference reads Foldcomp databases through the vendored Foldcomp library (src/structure/io/fcz.rs:76-146 ->
// lib/foldcomp/foldcompffi.cpp:11-52: Foldcomp::read + Foldcomp::decompress, then Structure::update per atom).  AFDB is distributed
// in this form.  This file is an independent decoder of the published format, written for what folddisco needs — the record of
// every atom (name, residue, numbering, B factor) and the coordinates of the backbone N, CA, C plus O and CB — with the float
// arithmetic of the format's reference decoder kept operation for operation, because the hashes are functions of these bits:
//
//   entry      "FCMP" | header (72 B: residue / atom counts and start numbers, anchor count, chain, first / last residue,
//              title length, min + step of the six backbone angle quantisers) | anchor residue indices | title |
//              first N, CA, C | inner anchor N, CA, C triples | last N, CA, C | hasOXT + OXT xyz | 8 B per residue
//              (type 5 b, omega 11 b, psi 12 b, phi 12 b, three bond angles 8 b each) | side-chain torsions (1 B each) |
//              B-factor quantiser + 1 B per residue
//   backbone   per anchor segment: forward NeRF chain from the segment's first three atoms (bond lengths 1.3311 / 1.4581
//              (1.353 behind a proline) / 1.5281), backward chain from the stored anchor atoms with the bond angles measured on
//              the forward chain, position-weighted average of the two
//   O, CB      NeRF from (N, CA, C) and (O, C, CA) with the residue type's bond length / angle and the stored torsion
//              (torsion = -180 + 360 / 255 * byte)
// Distances inside the NeRF step are evaluated in double and rounded to float exactly where the format's decoder does (its norm()
// squares through pow(double), its angle() goes through double sqrt / acos).  Side-chain atoms beyond CB get their records
// (CompactStructure::build looks at record order, names and B factors) but no coordinates: folddisco never reads them.
// tests/test_foldcomp.py compares every atom record and the N / CA / C / O / CB coordinates bit for bit with the reference decoder
// built from lib/foldcomp (oracle/_ref) and with committed golden vectors.
#include "fd_fcz.h"

#include <math.h>
#include <algorithm>
#include <string.h>

#ifndef FD_FCZ_PIPE
#define FD_FCZ_PIPE 0      /* 1: a segment's backward chain placed atom by atom BESIDE the next segment's forward chain instead of after it (see below) */
#endif

namespace {
struct f3 { float x, y, z; };

inline f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float norm(f3 v) { return (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z); }
// angle at b (degrees) as the format's decoder measures it: float dot / sizes, double sqrt, double acos
inline float angle_deg(f3 a, f3 b, f3 c) {
    const f3 d1 = {a.x - b.x, a.y - b.y, a.z - b.z}, d2 = {c.x - b.x, c.y - b.y, c.z - b.z};
    const float inner = (d1.x * d2.x) + (d1.y * d2.y) + (d1.z * d2.z);
    const float s1 = d1.x * d1.x + d1.y * d1.y + d1.z * d1.z, s2 = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
    const float cs = (float)((double)inner / sqrt((double)(s1 * s2)));
    return (float)(acos((double)cs) * 180.0 / M_PI);
}
// sine / cosine of an angle given in degrees, rounded where the format's decoder rounds: degrees -> radians in double, to float, then the
// float routines.  One sincosf call where the C library's sincosf returns what its sinf and cosf return (glibc: one implementation behind the
// three, checked exhaustively for |x| <= 16 by tools/check_sincosf.c and sampled again here at start-up); otherwise the two calls.
struct sc { float c, s; };
static bool sincosf_is_sinf_cosf() {
    uint32_t w = 0x9E3779B9u;
    for (int k = 0; k < 20000; ++k) {
        w = w * 1664525u + 1013904223u;
        const float x = (float)((double)(int32_t)w * (3.5 / 2147483648.0));      // [-3.5, 3.5): every angle here is a float of [-pi, pi]
        float s, c;
        sincosf(x, &s, &c);
        const float s2 = sinf(x), c2 = cosf(x);
        if (memcmp(&s, &s2, 4) || memcmp(&c, &c2, 4)) return false;
    }
    return true;
}
static const bool g_one_call = sincosf_is_sinf_cosf();
inline sc sc_deg(float deg) {
    const float r = (float)((double)deg * M_PI / 180.0);
    sc o;
    if (g_one_call) sincosf(r, &o.s, &o.c);
    else { o.c = cosf(r); o.s = sinf(r); }
    return o;
}
// next atom from three predecessors, bond length, bond angle and torsion (their cosines / sines from sc_deg).  This routine mirrors the
// vendored decoder's place_atom (lib/foldcomp/src/nerf.cpp:39-95) operation for operation — same intermediate roundings — because the
// coordinates must come out bit-identical to it; everything around it is written from the format.  The decoder calls cosf / sinf of the two
// angles inside the routine, five calls per atom; here the values arrive from the caller, which knows that most of them repeat (a torsion
// serves the forward and the backward chain, the bond angles of the forward chain are 8-bit codes, O and CB have per-type angles and
// 8-bit torsions): ~7 libm calls per residue instead of 40, the same bits.
inline f3 place_atom(const f3 &a, const f3 &b, const f3 &c, float bond_length, sc bond, sc tors) {
    const f3 ab = {b.x - a.x, b.y - a.y, b.z - a.z}, bc = {c.x - b.x, c.y - b.y, c.z - b.z};
    const float bc_norm = norm(bc);
    const f3 bcn = {bc.x / bc_norm, bc.y / bc_norm, bc.z / bc_norm};
    const f3 cur = {-1 * bond_length * bond.c, bond_length * tors.c * bond.s, bond_length * tors.s * bond.s};
    f3 n = cross(ab, bcn);
    const float n_norm = norm(n);
    n.x = n.x / n_norm; n.y = n.y / n_norm; n.z = n.z / n_norm;
    const f3 nbc = cross(n, bcn);
    f3 d = {0.0f, 0.0f, 0.0f};
    d.x += bcn.x * cur.x; d.x += nbc.x * cur.y; d.x += n.x * cur.z;
    d.y += bcn.y * cur.x; d.y += nbc.y * cur.y; d.y += n.y * cur.z;
    d.z += bcn.z * cur.x; d.z += nbc.z * cur.y; d.z += n.z * cur.z;
    d.x += c.x; d.y += c.y; d.z += c.z;
    return d;
}

// residue types of the format (5-bit code): atoms behind N, CA, C and the geometry of O and CB
struct aa_info { const char *three; char one; const char *side; float c_o, ca_c_o, ca_cb, c_ca_cb; };
const aa_info AA[20] = {
    {"ALA", 'A', "O CB", 1.23f, 120.31f, 1.52f, 110.852f},
    {"ARG", 'R', "O CB CG CD NE CZ NH1 NH2", 1.23f, 119.745f, 1.53f, 110.579f},
    {"ASN", 'N', "O CB CG OD1 ND2", 1.23f, 120.313f, 1.52f, 110.852f},
    {"ASP", 'D', "O CB CG OD1 OD2", 1.23f, 121.051f, 1.53f, 110.871f},
    {"CYS", 'C', "O CB SG", 1.23f, 120.063f, 1.53f, 111.078f},
    {"GLN", 'Q', "O CB CG CD OE1 NE2", 1.23f, 120.211f, 1.53f, 109.5f},
    {"GLU", 'E', "O CB CG CD OE1 OE2", 1.23f, 120.594f, 1.53f, 110.538f},
    {"GLY", 'G', "O", 1.23f, 120.522f, 0.0f, 0.0f},
    {"HIS", 'H', "O CB CG ND1 CD2 CE1 NE2", 1.23f, 120.548f, 1.53f, 111.329f},
    {"ILE", 'I', "O CB CG1 CG2 CD1", 1.235f, 120.393f, 1.54f, 111.983f},
    {"LEU", 'L', "O CB CG CD1 CD2", 1.235f, 120.211f, 1.53f, 110.418f},
    {"LYS", 'K', "O CB CG CD CE NZ", 1.23f, 120.54f, 1.53f, 109.5f},
    {"MET", 'M', "O CB CG SD CE", 1.23f, 120.148f, 1.53f, 110.833f},
    {"PHE", 'F', "O CB CG CD1 CD2 CE1 CE2 CZ", 1.23f, 120.283f, 1.53f, 110.846f},
    {"PRO", 'P', "O CB CG CD", 1.23f, 120.6f, 1.53f, 111.372f},
    {"SER", 'S', "O CB OG", 1.23f, 120.475f, 1.53f, 110.248f},
    {"THR", 'T', "O CB OG1 CG2", 1.23f, 120.252f, 1.53f, 110.075f},
    {"TRP", 'W', "O CB CG CD1 CD2 NE1 CE2 CE3 CZ2 CZ3 CH2", 1.23f, 120.178f, 1.53f, 110.852f},
    {"TYR", 'Y', "O CB CG CD1 CD2 CE1 CE2 CZ OH", 1.235f, 120.608f, 1.53f, 110.852f},
    {"VAL", 'V', "O CB CG1 CG2", 1.235f, 120.472f, 1.54f, 111.381f}};

const aa_info AA_UNK = {"UNK", 'X', "", 0.0f, 0.0f, 0.0f, 0.0f};      // type 23: N, CA, C only, no side torsions

#pragma pack(push, 1)
struct fcz_header {    // 72 bytes, natural alignment of the original struct reproduced with explicit padding
    uint16_t n_residue, n_atom, idx_residue, idx_atom;
    uint8_t n_anchor;
    char chain;
    uint8_t pad0[2];
    uint32_t n_side_torsion;
    char first_residue, last_residue;
    uint8_t pad1[2];
    uint32_t len_title;
    float mins[6], cont_fs[6];
};
#pragma pack(pop)
static_assert(sizeof(fcz_header) == 72, "Foldcomp header is 72 bytes");

struct res_code { uint32_t type, omega, psi, phi, ca_c_n, c_n_ca, n_ca_c; };
struct res_angles { float phi, psi, omega, n_ca_c, ca_c_n, c_n_ca; };

inline void set_name(char out[4], const char *nm, size_t len) {
    out[0] = ' '; out[1] = nm[0]; out[2] = len > 1 ? nm[1] : ' '; out[3] = len > 2 ? nm[2] : ' ';
}

// one anchor segment under reconstruction
struct seg {
    std::vector<f3> fw, rec;      // forward chain; backward chain from the segment's end
    std::vector<float> bang;      // bond angles measured on the forward chain
    int t0 = 0, n_st = 0, n_range = 0;
    bool last = false;
    size_t na = 0;
};
// per-thread scratch of the decoder: an ingest thread decodes ~10^4 entries a second, none of them allocates once the vectors have grown
struct scratch {
    std::vector<int32_t> anchor_idx;
    std::vector<float> anchors;
    std::vector<res_code> code;
    std::vector<res_angles> ang;
    std::vector<uint8_t> side, bq, tors_have;
    std::vector<sc> tors_sc;
    std::vector<f3> bb;
    seg sg[2];      // forward chain of one segment, backward chain of the one before
    sc bond_sc[3][256];
    uint8_t bond_have[3][256];
};
// O and CB: the bond angle is a constant of the residue type, the torsion one of 256 codes (-180 + 360 / 255 * code) — their sines / cosines
// come from tables filled once per process by the same sc_deg
struct side_tables {
    sc o_bond[20], cb_bond[20], tors[256];
    float tor_deg[256];
    uint32_t n_side[20];            // atoms behind N, CA, C of the residue type
    char side_name[20][12][4];      // their names as PDB columns 13-16
};

struct reader {
    const uint8_t *p; size_t n, at = 0;
    bool ok = true;
    bool get(void *dst, size_t len) { if (!ok || at + len > n) { ok = false; return false; } memcpy(dst, p + at, len); at += len; return true; }
};
}  // namespace

static const side_tables &side_tab();

int fd_fcz_decode(const uint8_t *data, size_t len, std::vector<fd_fcz_atom> *out) {
    static thread_local scratch S;
    out->clear();
    reader r{data, len};
    char magic[4];
    if (!r.get(magic, 4) || memcmp(magic, "FCMP", 4)) return -1;
    fcz_header H;
    if (!r.get(&H, sizeof H)) return -1;
    const int nres = H.n_residue, n_anchor = H.n_anchor;
    if (nres < 1 || n_anchor < 2) return -1;
    // an entry is untrusted input (one corrupt record must cost that entry, not the ingest): every count is bounded by the bytes that
    // remain before anything is allocated or indexed with it
    const size_t remain = len - r.at;
    if ((size_t)n_anchor > remain / 4 || (size_t)nres > remain / 8 || (size_t)H.n_side_torsion > remain || (size_t)H.len_title > remain) return -1;
    std::vector<int32_t> &anchor_idx = S.anchor_idx;
    anchor_idx.resize(n_anchor);
    if (!r.get(anchor_idx.data(), (size_t)n_anchor * 4)) return -1;
    for (int k = 0; k < n_anchor; ++k)      // anchors are residue indices in strictly ascending order
        if (anchor_idx[k] < 0 || anchor_idx[k] >= nres || (k && anchor_idx[k] <= anchor_idx[k - 1])) return -1;
    if (r.at + H.len_title > len) return -1;
    r.at += H.len_title;
    float first[9], last[9], oxt[3];
    if (!r.get(first, sizeof first)) return -1;
    std::vector<float> &anchors = S.anchors;     // per segment end: N, CA, C (inner anchors, then the last atoms)
    anchors.resize((size_t)(n_anchor - 1) * 9);
    if (n_anchor > 2 && !r.get(anchors.data(), (size_t)(n_anchor - 2) * 36)) return -1;
    if (!r.get(last, sizeof last)) return -1;
    memcpy(&anchors[(size_t)(n_anchor - 2) * 9], last, sizeof last);
    char has_oxt;
    if (!r.get(&has_oxt, 1) || !r.get(oxt, sizeof oxt)) return -1;
    std::vector<res_code> &code = S.code;
    code.resize(nres);
    for (int i = 0; i < nres; ++i) {
        uint8_t b[8];
        if (!r.get(b, 8)) return -1;
        code[i].type = (b[0] & 0xF8) >> 3;
        code[i].omega = ((uint32_t)(b[0] & 0x07) << 8) | b[1];
        code[i].psi = ((uint32_t)b[2] << 4) | (b[3] >> 4);
        code[i].phi = ((uint32_t)(b[3] & 0x0F) << 8) | b[4];
        code[i].ca_c_n = b[5]; code[i].c_n_ca = b[6]; code[i].n_ca_c = b[7];
        if (code[i].type >= 20 && code[i].type <= 22) return -2;   // ASX / GLX / STP: the format's decoder has no entry for them (its lookup throws)
        if (code[i].type > 23) code[i].type = 23;                  // every other code reads as UNK (backbone only)
    }
    std::vector<uint8_t> &side = S.side;
    side.resize(H.n_side_torsion);
    if (H.n_side_torsion && !r.get(side.data(), H.n_side_torsion)) return -1;
    float b_min, b_step;
    if (!r.get(&b_min, 4) || !r.get(&b_step, 4)) return -1;
    std::vector<uint8_t> &bq = S.bq;
    bq.resize(nres);
    if (!r.get(bq.data(), nres)) return -1;

    // continuous angles: min + code * step (float)
    std::vector<res_angles> &ang = S.ang;
    ang.resize(nres);
    for (int i = 0; i < nres; ++i) {
        ang[i].phi = H.mins[0] + ((float)code[i].phi * H.cont_fs[0]);
        ang[i].psi = H.mins[1] + ((float)code[i].psi * H.cont_fs[1]);
        ang[i].omega = H.mins[2] + ((float)code[i].omega * H.cont_fs[2]);
        ang[i].n_ca_c = H.mins[3] + ((float)code[i].n_ca_c * H.cont_fs[3]);
        ang[i].ca_c_n = H.mins[4] + ((float)code[i].ca_c_n * H.cont_fs[4]);
        ang[i].c_n_ca = H.mins[5] + ((float)code[i].c_n_ca * H.cont_fs[5]);
    }
    // torsion list of the whole chain: psi, omega, phi of residues 0 .. n-2 — entry t is kind t % 3 of residue t / 3.  Its sine / cosine
    // pairs are computed when the forward chain first needs them and serve the backward chain of the same segment
    const int n_tors = 3 * (nres - 1), tmax = n_tors - 1;
    auto tors_at = [&](int t) { const res_angles &A = ang[t / 3]; return t % 3 == 0 ? A.psi : t % 3 == 1 ? A.omega : A.phi; };
    S.tors_sc.resize((size_t)std::max(n_tors, 1));
    S.tors_have.assign((size_t)std::max(n_tors, 1), 0);
    auto tors_sc = [&](int t) -> sc {
        if (!S.tors_have[t]) { S.tors_sc[t] = sc_deg(tors_at(t)); S.tors_have[t] = 1; }
        return S.tors_sc[t];
    };
    // bond angles of the forward chain: three quantisers of 256 codes each, one sine / cosine pair per (quantiser, code) and entry
    memset(S.bond_have, 0, sizeof S.bond_have);
    auto bond_sc = [&](int q, uint32_t cd, float deg) -> sc {
        if (!S.bond_have[q][cd]) { S.bond_sc[q][cd] = sc_deg(deg); S.bond_have[q][cd] = 1; }
        return S.bond_sc[q][cd];
    };

    // ---- backbone, anchor segment by anchor segment.  Placing an atom is one chain of dependent operations (two double square roots, two
    // rounds of divisions, the accumulations), and the next segment starts from the average of this segment's LAST three atoms, whose
    // backward side is the stored anchor itself: the forward chain of segment s + 1 needs nothing from the backward chain of segment s.  The
    // loop is therefore written so that the two can be placed alternately (FD_FCZ_PIPE = 1).  Measured with tools/fcz_decode_bench.cpp: no gain
    // — 41.6 us per entry one after the other against 42.8 alternately on the MI355X host (EPYC 9575F), 80 against 80 on a 2.1 GHz Xeon; the
    // cores already overlap what there is to overlap.  The default places a segment's backward chain after the next segment's forward chain.
    std::vector<f3> &bb = S.bb;                       // N, CA, C of every residue
    bb.clear();
    bb.reserve((size_t)nres * 3);
    auto avg_at = [](const seg &G, int i) {          // position-weighted average of the forward chain and the backward chain (read back to front)
        const int total = (int)G.na;
        const f3 f = G.fw[i], b = G.rec[G.na - 1 - (size_t)i];
        f3 v;
        v.x = ((f.x * (float)(total - i)) + (b.x * (float)i)) / (float)total;
        v.y = ((f.y * (float)(total - i)) + (b.y * (float)i)) / (float)total;
        v.z = ((f.z * (float)(total - i)) + (b.z * (float)i)) / (float)total;
        return v;
    };
    auto bw_step = [&](seg &G, size_t i) {
        // atom i + 3 of the reversed chain: kinds cycle C, CA, N from the end; the bond runs from atom i + 3 to atom i + 2
        const size_t na = G.na;
        const int kind_cur = (int)((na - 1 - (i + 3)) % 3), kind_prev = (int)((na - 1 - (i + 2)) % 3);   // 0 N, 1 CA, 2 C
        float bl;
        if (kind_cur == 0 && kind_prev == 1) bl = 1.4581f;        // N_TO_CA
        else if (kind_cur == 1 && kind_prev == 2) bl = 1.5281f;   // CA_TO_C
        else bl = 1.3311f;                                        // C_TO_N
        // reversed lists: torsion i = the segment's torsion n_st - 1 - i, bond angle i + 1 = measured angle n_bang - 2 - i (n_bang = na - 2)
        const int j = G.n_st - 1 - (int)i;
        G.rec.push_back(place_atom(G.rec[i], G.rec[i + 1], G.rec[i + 2], bl, sc_deg(G.bang[na - 4 - i]), tors_sc(j < G.n_range ? G.t0 + j : tmax)));
    };
    auto finish = [&](seg &G) {
        const size_t keep = !G.last ? G.na - 3 : G.na;
        for (size_t i = 0; i < keep; ++i) bb.push_back(avg_at(G, (int)i));
    };
    seg *pend = nullptr;          // the segment whose backward chain is under way
    size_t bi = 0, nbk = 0;       // its next step, its number of steps
    f3 prev[3] = {{first[0], first[1], first[2]}, {first[3], first[4], first[5]}, {first[6], first[7], first[8]}};
    for (int sgm = 0; sgm < n_anchor - 1; ++sgm) {
        seg &G = S.sg[sgm & 1];
        std::vector<f3> &fw = G.fw;
        std::vector<float> &bang = G.bang;
        const int max_idx = nres - 1;
        const bool last_sgm = sgm == n_anchor - 2;
        const int i0 = anchor_idx[sgm] < max_idx ? anchor_idx[sgm] : max_idx;
        const int i1 = anchor_idx[sgm + 1] + 1 < max_idx ? anchor_idx[sgm + 1] + 1 : max_idx;
        if (i0 < 0 || i1 < i0) return -1;
        // residues of the segment: i0 .. i1 - 1, plus the chain's last residue behind the last segment; all but the last listed one extend
        // the forward chain
        const int n_sub = i1 - i0 + (last_sgm ? 1 : 0), n_ext = n_sub > 0 ? n_sub - 1 : 0;
        fw.clear();
        fw.push_back(prev[0]); fw.push_back(prev[1]); fw.push_back(prev[2]);
        bang.clear();
        size_t n_ang = 0;      // angles measured so far: entry i sits at atom i + 1
        for (int k = 0; k < n_ext; ++k) {
            const int r = i0 + k;              // (k < n_sub - 1: never the appended last residue)
            const res_angles &A = ang[r];
            const bool cached = 3 * r + 2 <= tmax;
            const f3 p0 = fw[3 * k], p1 = fw[3 * k + 1], p2 = fw[3 * k + 2];
            const f3 nn = place_atom(p0, p1, p2, (float)1.3311, bond_sc(0, code[r].ca_c_n, A.ca_c_n), cached ? tors_sc(3 * r) : sc_deg(A.psi));
            if (FD_FCZ_PIPE && bi < nbk) bw_step(*pend, bi++);
            const f3 ca = place_atom(p1, p2, nn, code[r].type != 14 ? (float)1.4581 : (float)1.353, bond_sc(1, code[r].c_n_ca, A.c_n_ca),
                                     cached ? tors_sc(3 * r + 1) : sc_deg(A.omega));
            if (FD_FCZ_PIPE && bi < nbk) bw_step(*pend, bi++);
            const f3 cc = place_atom(p2, nn, ca, (float)1.5281, bond_sc(2, code[r].n_ca_c, A.n_ca_c), cached ? tors_sc(3 * r + 2) : sc_deg(A.phi));
            if (FD_FCZ_PIPE && bi < nbk) bw_step(*pend, bi++);
            fw.push_back(nn); fw.push_back(ca); fw.push_back(cc);
            // the bond angles the backward chain will want, measured as soon as an atom's successor exists
            for (; n_ang + 2 < fw.size(); ++n_ang) bang.push_back(angle_deg(fw[n_ang], fw[n_ang + 1], fw[n_ang + 2]));
        }
        if (pend) {
            while (bi < nbk) bw_step(*pend, bi++);
            finish(*pend);
        }
        // torsions of the segment: entries t0 .. t1 - 1 of the chain's list, plus its last entry behind the last segment
        G.t0 = 0; G.n_st = 0;
        if (tmax >= 0) {
            const int64_t a0 = (int64_t)anchor_idx[sgm] * 3, a1 = (int64_t)anchor_idx[sgm + 1] * 3;      // 64-bit: 3 * index must not wrap
            G.t0 = (int)std::min<int64_t>(std::max<int64_t>(a0, 0), tmax);
            const int t1 = (int)std::min<int64_t>(std::max<int64_t>(a1, 0), tmax);
            G.n_st = std::max(t1 - G.t0, 0) + (last_sgm ? 1 : 0);
        }
        G.n_range = G.n_st - (last_sgm && tmax >= 0 ? 1 : 0);      // entry j of the segment's torsions = entry t0 + j of the chain's list, the appended one = the list's last
        G.last = last_sgm;
        // backward chain from the stored anchor atoms (C, CA, N of the segment's end first), bond angles measured on the forward chain
        const size_t na = G.na = fw.size();
        const float *an = &anchors[(size_t)sgm * 9];
        for (; n_ang + 2 < na; ++n_ang) bang.push_back(angle_deg(fw[n_ang], fw[n_ang + 1], fw[n_ang + 2]));
        if (na > 3 && na - 3 > (size_t)G.n_st) return -1;      // every backward step needs its torsion
        G.rec.clear();
        G.rec.reserve(na);
        G.rec.push_back({an[6], an[7], an[8]}); G.rec.push_back({an[3], an[4], an[5]}); G.rec.push_back({an[0], an[1], an[2]});
        // the next segment starts from the averages of the last three atoms — forward chain and the anchor itself
        {
            const size_t have = G.rec.size();
            G.rec.resize(na);      // (avg_at reads rec[na - 1 - i]: for the last three atoms these are the three anchors at the front)
            prev[0] = avg_at(G, (int)na - 3); prev[1] = avg_at(G, (int)na - 2); prev[2] = avg_at(G, (int)na - 1);
            G.rec.resize(have);
        }
        pend = &G; bi = 0; nbk = na - 3;
    }
    if (pend) {
        while (bi < nbk) bw_step(*pend, bi++);
        finish(*pend);
    }
    if (bb.size() != (size_t)nres * 3) return -1;

    // ---- atom records: N, CA, C, then the residue type's side atoms (O and CB placed, the rest without coordinates)
    size_t tpos = 0;
    const side_tables &ST = side_tab();
    size_t n_rec = has_oxt ? 1 : 0;
    for (int i = 0; i < nres; ++i) n_rec += 3u + (code[i].type < 20 ? ST.n_side[code[i].type] : 0u);
    out->resize(n_rec);
    fd_fcz_atom *w = out->data();
    for (int i = 0; i < nres; ++i) {
        const uint32_t ty = code[i].type;
        const aa_info &T = ty < 20 ? AA[ty] : AA_UNK;
        const f3 N = bb[3 * i], CA = bb[3 * i + 1], C = bb[3 * i + 2];
        fd_fcz_atom a;      // the residue's record: name and coordinates change from atom to atom
        a.b = ((float)bq[i] * b_step) + b_min;
        memcpy(a.res, T.three, 3);
        a.chain = (uint8_t)H.chain;
        a.rser = (uint64_t)H.idx_residue + (uint64_t)i;
        auto push = [&](const char nm[4], f3 xyz) { a.x = xyz.x; a.y = xyz.y; a.z = xyz.z; memcpy(a.name, nm, 4); *w++ = a; };
        push(" N  ", N); push(" CA ", CA); push(" C  ", C);
        const uint32_t ns = ty < 20 ? ST.n_side[ty] : 0u;
        if (ns > side.size() - tpos) { out->clear(); return -1; }      // (no partly written records for a caller that ignores the code)
        f3 O = {0, 0, 0};
        for (uint32_t k = 0; k < ns; ++k) {
            const uint8_t tc = side[tpos++];
            f3 xyz = {0.0f, 0.0f, 0.0f};
            if (k == 0) O = xyz = place_atom(N, CA, C, T.c_o, ST.o_bond[ty], ST.tors[tc]);
            else if (k == 1) xyz = place_atom(O, C, CA, T.ca_cb, ST.cb_bond[ty], ST.tors[tc]);
            push(ST.side_name[ty][k], xyz);
        }
    }
    if (has_oxt) {
        fd_fcz_atom a;
        a.x = oxt[0]; a.y = oxt[1]; a.z = oxt[2];
        a.b = ((float)bq[nres - 1] * b_step) + b_min;
        set_name(a.name, "OXT", 3);
        const char *hit = H.last_residue ? strchr("ARNDCQEGHILKMFPSTWYV", H.last_residue) : nullptr;
        memcpy(a.res, hit ? AA[hit - "ARNDCQEGHILKMFPSTWYV"].three : "UNK", 3);
        a.chain = (uint8_t)H.chain;
        a.rser = (uint64_t)H.n_residue;
        *w++ = a;
    }
    return 0;
}

static const side_tables &side_tab() {
    static const side_tables T = [] {
        side_tables t;
        for (int a = 0; a < 20; ++a) { t.o_bond[a] = sc_deg(AA[a].ca_c_o); t.cb_bond[a] = sc_deg(AA[a].c_ca_cb); }
        const float tor_step = (180.0f - -180.0f) / (float)255u;            // FixedAngleDiscretizer(255): float division
        for (int c = 0; c < 256; ++c) { t.tor_deg[c] = ((float)c * tor_step) + -180.0f; t.tors[c] = sc_deg(t.tor_deg[c]); }
        for (int a = 0; a < 20; ++a) {
            uint32_t k = 0;
            for (const char *s = AA[a].side; *s;) {
                const char *e = s;
                while (*e && *e != ' ') ++e;
                set_name(t.side_name[a][k++], s, (size_t)(e - s));
                s = *e ? e + 1 : e;
            }
            t.n_side[a] = k;
        }
        return t;
    }();
    return T;
}
