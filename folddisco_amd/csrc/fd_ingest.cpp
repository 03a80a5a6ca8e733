// fd_ingest.cpp — structure ingest at speed: PDB / mmCIF (optionally gzip) text -> the packed
// CompactStructure arrays the GPU path consumes, multi-threaded over files.
//
// Restates, quirks included:
//   * PDB reader          src/structure/io/pdb.rs:37-77, fixed columns src/structure/io/parser.rs:3-56: only "ATOM  "
//                         records, first model, a record with an unparsable number is skipped
//   * mmCIF reader        src/structure/io/cif.rs:102-296: the loop whose header holds _atom_site.group_PDB; ATOM and HETATM
//                         rows alike; first model number only; residue number auth_seq_id else label_seq_id; chain
//                         auth_asym_id if one character else label_asym_id; B factor default 1.0
//   * CompactStructure    src/structure/core.rs:70-214: residue boundary = change of the residue number only; the flush
//                         also fires at the last atom index before that atom is examined; chain and b-factor of residue k
//                         are read from the first atom of residue k+1; C is never reset (a virtual CB may use a stale C);
//                         a GLY N is captured through the GLY branch; virtual CB src/structure/coordinate.rs:167-186
//   * amino-acid map      src/utils/convert.rs:53-81
// Host code only (no device work): text parsing is what remains of an index build once hashing runs on the GPU.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fdgpu.h"
#include "../../include/fdgpu_debug.h"
#include "fd_fcz.h"
#include "fd_inflate.h"

namespace {

struct Atom {
    float x, y, z, b;
    char name[4], res[3];
    uint8_t chain;
    uint64_t rser;
};

struct Compact {
    std::vector<float> n, ca, cb, bfac;
    std::vector<uint8_t> aa, cb_ok, chain, std_name;
    std::vector<uint64_t> serial;
    std::vector<char> resname;
    uint64_t nres_raw = 0;
    uint8_t first_chain = ' ';   // chains[0] of the raw atom list (default chain of a query string)
    bool ok = false;
};

const char *AA_GROUPS[20] = {
    "ALA ABA ORN DAL AIB ALC MDO MAA DAB", "ARG DAR CIR AGM", "ASN DSG MEN SNN", "ASP 0TD DAS IAS PHD BFD ASX",
    "CYS CSO CSD CME OCS CAS CSX CSS YCM DCY SMC SCH SCY CAF SNC SEC", "GLN DGN CRQ MEQ", "GLU PCA DGL CGU FGA B3E GLX",
    "GLY CR2 SAR GHP GL3", "HIS HIC DHI NEP CR8 MHS", "ILE DIL", "LEU DLE NLE MLE MK8",
    "LYS KCX LLP MLY M3L ALY MLZ DLY KPI PYL", "MET MSE FME NRQ CXM SME MHO MED", "PHE DPN PHI MEA PHL", "PRO HYP DPR",
    "SER CSH SEP DSN SAC GYS DHA OAS", "THR TPO CRO DTH BMT CRF", "TRP DTR TRQ TOX 0AF", "TYR PTR TYS TPQ DTY OMY",
    "VAL DVA MVA FVA"};

// residue name -> amino-acid index (src/utils/convert.rs:53-81), is_std = the group's first (canonical) name.  A 1,024-slot open-addressing table over
// the ~120 names, made on first use: the walk over the twenty groups was ~60 three-byte comparisons per residue
struct AaTab {
    uint32_t key[1024];      // name bytes | 1 << 24 (0 = empty)
    uint8_t val[1024];       // index | is_std << 7
    AaTab() {
        memset(key, 0, sizeof key); memset(val, 0, sizeof val);
        for (int k = 0; k < 20; ++k) {
            const char *g = AA_GROUPS[k];
            for (int pos = 0;; pos += 4) {
                const uint32_t nm = (uint32_t)(uint8_t)g[pos] | ((uint32_t)(uint8_t)g[pos + 1] << 8) | ((uint32_t)(uint8_t)g[pos + 2] << 16) | (1u << 24);
                uint32_t at = slot(nm);
                while (key[at] && key[at] != nm) at = (at + 1) & 1023u;
                if (!key[at]) { key[at] = nm; val[at] = (uint8_t)(k | (pos == 0 ? 0x80 : 0)); }      // (a name listed twice keeps its first group, as the walk did)
                if (!g[pos + 3]) break;
            }
        }
    }
    static uint32_t slot(uint32_t nm) { return ((nm * 2654435761u) >> 22) & 1023u; }
};
uint8_t map_aa(const char r[3], bool *is_std) {
    static const AaTab T;
    const uint32_t nm = (uint32_t)(uint8_t)r[0] | ((uint32_t)(uint8_t)r[1] << 8) | ((uint32_t)(uint8_t)r[2] << 16) | (1u << 24);
    uint32_t at = AaTab::slot(nm);
    while (T.key[at]) {
        if (T.key[at] == nm) { *is_std = (T.val[at] & 0x80) != 0; return (uint8_t)(T.val[at] & 0x7f); }
        at = (at + 1) & 1023u;
    }
    *is_std = false;
    return 255;
}

// Rust str::parse::<f32>() on a trimmed field: decimal / exponent / inf / nan, correctly rounded; nothing else
bool parse_f32(const char *s, size_t n, float *out) {
    while (n && (*s == ' ' || *s == '\t')) { ++s; --n; }
    while (n && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\r')) --n;
    if (!n || n > 63) return false;
    {
        // the fields of a PDB file are "%8.3f" / "%6.2f": [sign] digits [. digits] with a mantissa below 2^24 and at most 10 decimals — both
        // operands of mantissa / 10^decimals are then exact floats and ONE IEEE division gives the correctly rounded value (Clinger's fast
        // path), the bits strtof and Rust's parser return (strtof per field was most of the parser's time)
        static const float P10[11] = {1.0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
        size_t k = 0;
        const bool neg = s[0] == '-';
        if (s[0] == '-' || s[0] == '+') k = 1;
        uint32_t m = 0, dec = 0, nd = 0;
        bool dot = false, simple = k < n;
        for (; k < n && simple; ++k) {
            const char c = s[k];
            if (c >= '0' && c <= '9') { m = m * 10u + (uint32_t)(c - '0'); ++nd; dec += dot ? 1u : 0u; simple = m < (1u << 24) && dec <= 10u; }
            else if (c == '.' && !dot) dot = true;
            else simple = false;
        }
        if (simple && nd) { const float v = (float)m / P10[dec]; *out = neg ? -v : v; return true; }
    }
    char buf[64];
    bool digit = false, word = false;
    for (size_t k = 0; k < n; ++k) {
        char c = s[k];
        if (c >= '0' && c <= '9') digit = true;
        else if (c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E') {}
        else if (strchr("infatyINFATY", c)) word = true;
        else return false;
        buf[k] = c;
    }
    buf[n] = 0;
    if (word) {   // only the spellings Rust accepts
        const char *p = buf;
        if (*p == '+' || *p == '-') ++p;
        if (strcasecmp(p, "inf") && strcasecmp(p, "infinity") && strcasecmp(p, "nan")) return false;
    } else if (!digit) return false;
    char *end = nullptr;
    float v = strtof(buf, &end);
    if (end != buf + n) return false;
    *out = v;
    return true;
}

// Rust str::parse::<u64>() on a trimmed field: optional '+', digits, no overflow
bool parse_u64(const char *s, size_t n, uint64_t *out) {
    while (n && *s == ' ') { ++s; --n; }
    while (n && (s[n - 1] == ' ' || s[n - 1] == '\r')) --n;
    if (n && *s == '+') { ++s; --n; }
    if (!n) return false;
    uint64_t v = 0;
    for (size_t k = 0; k < n; ++k) {
        if (s[k] < '0' || s[k] > '9') return false;
        uint64_t d = (uint64_t)(s[k] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

bool read_all(const char *path, std::string *out) {
    {   // a file that does not open with the gzip magic is read as it is (zlib's transparent mode copies through its own buffer)
        int fd = open(path, O_RDONLY);
        if (fd < 0) return false;
        unsigned char magic[2] = {0, 0};
        struct stat sb;
        const bool plain = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && (pread(fd, magic, 2, 0) < 2 || magic[0] != 0x1f || magic[1] != 0x8b);
        if (plain) {
            out->resize((size_t)sb.st_size);
            size_t got = 0;
            while (got < out->size()) {
                const ssize_t r = read(fd, &(*out)[got], out->size() - got);
                if (r <= 0) break;
                got += (size_t)r;
            }
            close(fd);
            out->resize(got);
            return true;
        }
        // gzip: the whole file through the ingest's own decoder (fd_inflate.cpp, ~2.5x zlib's inflate on PDB text); anything it declines
        // (or cannot read) goes through zlib below, which reports damaged files the way it always did
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 18 && !(getenv("FDGPU_ZLIB") && getenv("FDGPU_ZLIB")[0] == '1')) {
            std::string comp((size_t)sb.st_size, '\0');
            size_t got = 0;
            while (got < comp.size()) {
                const ssize_t r = read(fd, &comp[got], comp.size() - got);
                if (r <= 0) break;
                got += (size_t)r;
            }
            if (got == comp.size() && fd_gunzip((const uint8_t *)comp.data(), comp.size(), out)) { close(fd); return true; }
            out->clear();
        }
        close(fd);
    }
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 18);
    out->clear();
    char buf[1 << 16];
    for (;;) {
        int n = gzread(f, buf, sizeof buf);
        if (n < 0) { gzclose(f); return false; }
        if (n == 0) break;
        out->append(buf, (size_t)n);
    }
    gzclose(f);
    return true;
}

// The common shapes of a PDB record's numeric columns, read without the general parsers: a right-aligned "%W.Df" field = [spaces][-]digits '.' D digits
// filling the column, a right-aligned unsigned integer = [spaces]digits.  Value = mantissa / 10^D in ONE f32 division, exactly what parse_f32's own fast
// path computes for the same characters; anything else (exponents, '+', inner blanks, no digit before the point, short lines) -> false and the caller takes
// the general parser, which decides as Rust's str::parse does.
template <int W, int D>
inline bool fixed_f32(const char *s, float *out) {
    static const float P10[4] = {1.0f, 1e1f, 1e2f, 1e3f};
    constexpr int I = W - D - 1;      // index of the decimal point
    if (s[I] != '.') return false;
    uint32_t frac = 0;
    for (int k = I + 1; k < W; ++k) { const uint32_t d = (uint32_t)(s[k] - '0'); if (d > 9u) return false; frac = frac * 10u + d; }
    int k = 0;
    while (k < I && s[k] == ' ') ++k;
    const bool neg = k < I && s[k] == '-';
    k += neg ? 1 : 0;
    if (k >= I) return false;         // no digit before the point
    uint32_t ip = 0;
    for (; k < I; ++k) { const uint32_t d = (uint32_t)(s[k] - '0'); if (d > 9u) return false; ip = ip * 10u + d; }
    uint32_t sc = 1;
    for (int z = 0; z < D; ++z) sc *= 10u;
    const float v = (float)(ip * sc + frac) / P10[D];      // ip < 10^4, D <= 3: the mantissa is below 2^24
    *out = neg ? -v : v;
    return true;
}
template <int W>
inline bool fixed_u64(const char *s, uint64_t *out) {
    int k = 0;
    while (k < W && s[k] == ' ') ++k;
    if (k >= W) return false;
    uint64_t v = 0;
    for (; k < W; ++k) { const uint32_t d = (uint32_t)(s[k] - '0'); if (d > 9u) return false; v = v * 10u + d; }
    *out = v;
    return true;
}

// all_models: the reference's gzip reader (read_structure_from_gz, structure/io/pdb.rs:79-124) has no MODEL handling — it keeps the
// ATOM records of EVERY model, unlike the plain-file reader (pdb.rs:37-77) which stops behind the first one
void parse_pdb(const std::string &txt, std::vector<Atom> *atoms, bool all_models) {
    int model = 0;
    size_t pos = 0, N = txt.size();
    while (pos < N) {
        size_t e = txt.find('\n', pos);
        if (e == std::string::npos) e = N;
        size_t len = e - pos;
        const char *L = txt.data() + pos;
        pos = e + 1;
        if (len && L[len - 1] == '\r') --len;
        if (model > 1 && !all_models) break;
        if (len < 6) continue;
        if (!memcmp(L, "MODEL ", 6)) { ++model; continue; }
        if (memcmp(L, "ATOM  ", 6) || len < 54) continue;
        Atom a;
        uint64_t aser;
        if (!(fixed_f32<8, 3>(L + 30, &a.x) || parse_f32(L + 30, 8, &a.x)) || !(fixed_f32<8, 3>(L + 38, &a.y) || parse_f32(L + 38, 8, &a.y)) ||
            !(fixed_f32<8, 3>(L + 46, &a.z) || parse_f32(L + 46, 8, &a.z))) continue;
        if (!(fixed_u64<5>(L + 6, &aser) || parse_u64(L + 6, 5, &aser)) || !(fixed_u64<4>(L + 22, &a.rser) || parse_u64(L + 22, 4, &a.rser))) continue;
        a.b = 1.0f;
        if (len >= 66 && !(fixed_f32<6, 2>(L + 60, &a.b) || parse_f32(L + 60, 6, &a.b))) continue;
        memcpy(a.name, L + 12, 4);
        memcpy(a.res, L + 17, 3);
        a.chain = (uint8_t)L[21];
        atoms->push_back(a);
    }
}

// ---- mmCIF: just enough of the CIF grammar for the atom_site loop
struct CifTok { const char *p; size_t n; bool quoted; };

// next token starting at *pos (skips whitespace and comments); returns false at end of input
bool cif_next(const std::string &t, size_t *pos, CifTok *tok, bool *at_line_start) {
    size_t i = *pos, N = t.size();
    for (;;) {
        while (i < N && (t[i] == ' ' || t[i] == '\t' || t[i] == '\r' || t[i] == '\n')) ++i;
        if (i < N && t[i] == '#') { while (i < N && t[i] != '\n') ++i; continue; }
        break;
    }
    if (i >= N) { *pos = i; return false; }
    bool bol = i == 0 || t[i - 1] == '\n';
    *at_line_start = bol;
    if (bol && t[i] == ';') {   // text field: up to a line that starts with ';'
        size_t s = i + 1, e = t.find("\n;", s);
        if (e == std::string::npos) e = N;
        *tok = {t.data() + s, e - s, true};
        *pos = e + 2 <= N ? e + 2 : N;
        return true;
    }
    if (t[i] == '\'' || t[i] == '"') {
        char q = t[i];
        size_t s = i + 1, e = s;
        while (e < N && !(t[e] == q && (e + 1 >= N || t[e + 1] == ' ' || t[e + 1] == '\t' || t[e + 1] == '\n' || t[e + 1] == '\r')) && t[e] != '\n') ++e;
        *tok = {t.data() + s, e - s, true};
        *pos = e < N ? e + 1 : N;
        return true;
    }
    size_t s = i;
    while (i < N && t[i] != ' ' && t[i] != '\t' && t[i] != '\n' && t[i] != '\r') ++i;
    *tok = {t.data() + s, i - s, false};
    *pos = i;
    return true;
}

bool tok_is(const CifTok &k, const char *s) { return !k.quoted && k.n == strlen(s) && !strncasecmp(k.p, s, k.n); }
bool tok_missing(const CifTok &k) { return !k.quoted && k.n == 1 && (k.p[0] == '.' || k.p[0] == '?'); }
// lexer's numeric value as the reference reads it (f32); integers must be integral (cif.rs get_isize)
bool tok_f32(const CifTok &k, float *v) { return !k.quoted && !tok_missing(k) && parse_f32(k.p, k.n, v); }
bool tok_int(const CifTok &k, uint64_t *v) {
    float f;
    if (!tok_f32(k, &f) || !(std::trunc(f) == f) || !(f >= -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return false;
    *v = (uint64_t)(int64_t)f;
    return true;
}

void parse_cif(const std::string &txt, std::vector<Atom> *atoms) {
    size_t pos = 0;
    CifTok tk;
    bool bol;
    bool have = cif_next(txt, &pos, &tk, &bol);
    while (have) {
        if (!tok_is(tk, "loop_")) { have = cif_next(txt, &pos, &tk, &bol); continue; }
        std::vector<std::string> header;
        while ((have = cif_next(txt, &pos, &tk, &bol)) && !tk.quoted && tk.n && tk.p[0] == '_') header.emplace_back(tk.p + 1, tk.n - 1);
        auto col = [&](const char *name) { for (size_t k = 0; k < header.size(); ++k) if (header[k] == name) return (int)k; return -1; };
        const bool is_atoms = col("atom_site.group_PDB") >= 0;
        const int c_asym = col("atom_site.label_asym_id"), c_aasym = col("atom_site.auth_asym_id"), c_b = col("atom_site.B_iso_or_equiv"),
                  c_comp = col("atom_site.label_comp_id"), c_id = col("atom_site.id"), c_model = col("atom_site.pdbx_PDB_model_num"),
                  c_name = col("atom_site.label_atom_id"), c_seq = col("atom_site.label_seq_id"), c_aseq = col("atom_site.auth_seq_id"),
                  c_type = col("atom_site.type_symbol"), c_x = col("atom_site.Cartn_x"), c_y = col("atom_site.Cartn_y"), c_z = col("atom_site.Cartn_z");
        const bool usable = is_atoms && c_asym >= 0 && c_comp >= 0 && c_id >= 0 && c_name >= 0 && c_seq >= 0 && c_type >= 0 && c_x >= 0 && c_y >= 0 && c_z >= 0;
        // rows: values until the next loop_ / data name / data_ / save_ keyword
        std::vector<CifTok> row;
        const size_t W = header.size();
        uint64_t first_model = 0, index = 0;
        bool stop = false;
        while (have) {
            if (!tk.quoted && tk.n && (tk.p[0] == '_' || tok_is(tk, "loop_") || (tk.n >= 5 && (!strncasecmp(tk.p, "data_", 5) || !strncasecmp(tk.p, "save_", 5))))) break;
            row.push_back(tk);
            have = cif_next(txt, &pos, &tk, &bol);
            if (row.size() < W || !W) continue;
            if (usable && !stop) {
                uint64_t model = 1, id, rser;
                if (c_model >= 0) { uint64_t m; if (tok_int(row[c_model], &m)) model = m; }
                if (index == 0) first_model = model;
                else if (model != first_model) stop = true;
                ++index;
                Atom a;
                bool good = !stop;
                const CifTok &nm = row[c_name], &rs = row[c_comp];
                if (good && (tok_missing(nm) || nm.n < 1 || nm.n > 4)) good = false;
                if (good) {
                    memset(a.name, ' ', 4);
                    if (nm.n == 4) memcpy(a.name, nm.p, 4); else memcpy(a.name + 1, nm.p, nm.n);
                    memset(a.res, ' ', 3);
                    if (tok_missing(rs)) good = false;
                    else if (rs.n <= 3) memcpy(a.res, rs.p, rs.n);   // longer names become blank (cif.rs:333-336)
                }
                if (good && !tok_int(row[c_id], &id)) good = false;
                if (good && !((c_aseq >= 0 && tok_int(row[c_aseq], &rser)) || tok_int(row[c_seq], &rser))) good = false;
                if (good) {
                    const CifTok *ch = (c_aasym >= 0 && !tok_missing(row[c_aasym]) && row[c_aasym].n == 1) ? &row[c_aasym] : &row[c_asym];
                    if (tok_missing(*ch) || ch->n != 1) good = false; else a.chain = (uint8_t)ch->p[0];
                }
                if (good && !(tok_f32(row[c_x], &a.x) && tok_f32(row[c_y], &a.y) && tok_f32(row[c_z], &a.z))) good = false;
                if (good) { a.b = 1.0f; if (c_b >= 0) { float b; if (tok_f32(row[c_b], &b)) a.b = b; } a.rser = rser; atoms->push_back(a); }
            }
            row.clear();
        }
    }
}

inline void norm3(const float v[3], float o[3]) {
    float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}
// src/structure/coordinate.rs:167-186, every operation rounded to f32 in the reference's order
void approx_cb(const float ca[3], const float n[3], const float c[3], float out[3]) {
    float d1[3] = {c[0] - ca[0], c[1] - ca[1], c[2] - ca[2]}, d2[3] = {n[0] - ca[0], n[1] - ca[1], n[2] - ca[2]};
    float v1[3], v2[3], b1[3], b2[3], u1[3], u2[3], v4[3];
    norm3(d1, v1); norm3(d2, v2);
    const float third = 1.0f / 3.0f;
    for (int k = 0; k < 3; ++k) b1[k] = v2[k] + v1[k] * third;
    b2[0] = v1[1] * b1[2] - v1[2] * b1[1]; b2[1] = v1[2] * b1[0] - v1[0] * b1[2]; b2[2] = v1[0] * b1[1] - v1[1] * b1[0];
    norm3(b1, u1); norm3(b2, u2);
    const float mh = -1.0f / 2.0f, s32 = sqrtf(3.0f) / 2.0f, s83 = sqrtf(8.0f) / 3.0f, mt = -1.0f / 3.0f;
    for (int k = 0; k < 3; ++k) v4[k] = u1[k] * mh - u2[k] * s32;
    for (int k = 0; k < 3; ++k) v4[k] = v4[k] * s83;
    for (int k = 0; k < 3; ++k) v4[k] = v4[k] + v1[k] * mt;
    for (int k = 0; k < 3; ++k) out[k] = ca[k] + v4[k] * 1.5336f;
}

void build_compact(const std::vector<Atom> &atoms, Compact *C) {
    if (!atoms.empty()) C->first_chain = atoms[0].chain;
    uint64_t rec_serial = 0;
    for (const Atom &a : atoms) if (rec_serial != a.rser) { ++C->nres_raw; rec_serial = a.rser; }
    {   // at most one residue per change of the residue number
        const size_t r = (size_t)C->nres_raw;
        C->n.reserve(3 * r); C->ca.reserve(3 * r); C->cb.reserve(3 * r); C->cb_ok.reserve(r); C->serial.reserve(r); C->resname.reserve(3 * r);
        C->aa.reserve(r); C->std_name.reserve(r); C->chain.reserve(r); C->bfac.reserve(r);
    }
    bool have_prev = false, hn = false, hca = false, hcb = false, hc = false, hgn = false, hgc = false;
    uint64_t prev_serial = 0;
    char prev_name[3] = {' ', ' ', ' '};
    float n[3], ca[3], cb[3], c[3], gn[3], gc[3];
    const size_t na = atoms.size();
    for (size_t idx = 0; idx < na; ++idx) {
        const Atom &a = atoms[idx];
        if (!have_prev || prev_serial != a.rser || idx == na - 1) {
            if (hn && hca) {
                float cbv[3] = {0.f, 0.f, 0.f};
                uint8_t ok = 1;
                if (hcb) memcpy(cbv, cb, 12);
                else if (!memcmp(prev_name, "GLY", 3) && hgn && hgc) approx_cb(ca, gn, gc, cbv);
                else if (hc) approx_cb(ca, n, c, cbv);
                else ok = 0;
                C->n.insert(C->n.end(), n, n + 3); C->ca.insert(C->ca.end(), ca, ca + 3); C->cb.insert(C->cb.end(), cbv, cbv + 3);
                C->cb_ok.push_back(ok); C->serial.push_back(prev_serial);
                C->resname.insert(C->resname.end(), prev_name, prev_name + 3);
                bool is_std;
                C->aa.push_back(map_aa(prev_name, &is_std));
                C->std_name.push_back(is_std ? 1 : 0);
                C->chain.push_back(a.chain); C->bfac.push_back(a.b);     // quirk: taken from the atom that triggered the flush
            }
            hca = hcb = hn = false;
            prev_serial = a.rser; memcpy(prev_name, a.res, 3); have_prev = true;
        }
        const float xyz[3] = {a.x, a.y, a.z};
        const bool gly = !memcmp(a.res, "GLY", 3);
        if (!memcmp(a.name, " CA ", 4)) { memcpy(ca, xyz, 12); hca = true; }
        else if (!memcmp(a.name, " CB ", 4)) { memcpy(cb, xyz, 12); hcb = true; }
        else if (!memcmp(a.name, " C  ", 4)) { memcpy(c, xyz, 12); hc = true; if (gly) { /* the C branch wins: GLY C is not recorded as gly_c */ } }
        else if (!memcmp(a.name, " N  ", 4) && !gly) { memcpy(n, xyz, 12); hn = true; }
        else if (gly) {
            if (!memcmp(a.name, " N  ", 4)) { memcpy(gn, xyz, 12); hgn = true; memcpy(n, xyz, 12); hn = true; }
            else if (!memcmp(a.name, " C  ", 4)) { memcpy(gc, xyz, 12); hgc = true; }
        }
    }
    C->ok = true;
}

bool ends_with_ci(const std::string &s, const char *suf) {
    size_t n = strlen(suf);
    return s.size() >= n && !strcasecmp(s.c_str() + s.size() - n, suf);
}

}  // namespace

namespace {
// the packed arrays of fd_parsed from the per-structure Compact parts
int pack_parsed(std::vector<Compact> &parts, uint64_t max_residue, fd_parsed **out, uint32_t n_threads = 1) {
    const auto t_pack0 = std::chrono::steady_clock::now();
    const uint64_t n = parts.size();
    fd_parsed *P = (fd_parsed *)calloc(1, sizeof(fd_parsed));
    if (!P) return FDGPU_ENOMEM;
    uint64_t R = 0;
    for (auto &C : parts) R += C.aa.size();
    P->n_struct = n; P->n_res = R;
    const uint64_t R1 = R ? R : 1, S1 = n ? n : 1;
    P->res_off = (uint64_t *)malloc((n + 1) * 8);
    P->n_xyz = (float *)malloc(R1 * 12); P->ca_xyz = (float *)malloc(R1 * 12); P->cb_xyz = (float *)malloc(R1 * 12);
    P->aa = (uint8_t *)malloc(R1); P->cb_valid = (uint8_t *)malloc(R1); P->chain = (uint8_t *)malloc(R1); P->resname_std = (uint8_t *)malloc(R1);
    P->serial = (uint64_t *)malloc(R1 * 8); P->bfac = (float *)malloc(R1 * 4); P->resname = (char *)malloc(R1 * 3);
    P->nres_raw = (uint64_t *)malloc(S1 * 8); P->plddt = (float *)malloc(S1 * 4); P->ok = (uint8_t *)malloc(S1); P->first_chain = (uint8_t *)malloc(S1);
    if (!P->res_off || !P->n_xyz || !P->ca_xyz || !P->cb_xyz || !P->aa || !P->cb_valid || !P->chain || !P->resname_std || !P->serial || !P->bfac ||
        !P->resname || !P->nres_raw || !P->plddt || !P->ok || !P->first_chain) { fdgpu_parsed_free(P); return FDGPU_ENOMEM; }
    uint64_t off = 0;
    for (uint64_t k = 0; k < n; ++k) { P->res_off[k] = off; off += parts[k].aa.size(); }
    // the copies (and the first touch of the fresh arrays: ~60 bytes per residue, half a gigabyte per 20,000 structures) run on the
    // callers' thread count — as one thread this was a quarter of the ingest's wall time
    std::atomic<uint64_t> next{0};
    auto copy = [&]() {
        for (;;) {
            const uint64_t k0 = next.fetch_add(64);
            if (k0 >= n) break;
            for (uint64_t k = k0; k < std::min(n, k0 + 64); ++k) {
                const Compact &C = parts[k];
                const uint64_t m = C.aa.size(), at = P->res_off[k];
                if (m) {
                    memcpy(P->n_xyz + 3 * at, C.n.data(), m * 12); memcpy(P->ca_xyz + 3 * at, C.ca.data(), m * 12); memcpy(P->cb_xyz + 3 * at, C.cb.data(), m * 12);
                    memcpy(P->aa + at, C.aa.data(), m); memcpy(P->cb_valid + at, C.cb_ok.data(), m); memcpy(P->chain + at, C.chain.data(), m);
                    memcpy(P->resname_std + at, C.std_name.data(), m); memcpy(P->serial + at, C.serial.data(), m * 8); memcpy(P->bfac + at, C.bfac.data(), m * 4);
                    memcpy(P->resname + 3 * at, C.resname.data(), m * 3);
                }
                // get_avg_plddt (structure/core.rs:450-456): sequential f32 sum / n (NaN for an empty structure; the index
                // workflow stores 0 for skipped structures, controller/mod.rs:313-318)
                float s = 0.0f;
                for (uint64_t r = 0; r < m; ++r) s = s + C.bfac[r];
                P->plddt[k] = (max_residue && C.nres_raw > max_residue) ? 0.0f : s / (float)m;
                P->nres_raw[k] = C.nres_raw;
                P->ok[k] = C.ok ? 1 : 0;
                P->first_chain[k] = C.first_chain;
                parts[k] = Compact();      // the part's ten arrays are released here, by the thread that copied them
            }
        }
    };
    const uint32_t T = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(n_threads, 1), (n + 255) / 256 + 1);
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t) th.emplace_back(copy);
    copy();
    for (auto &t : th) t.join();
    P->res_off[n] = off;
    if (getenv("FDGPU_TRACE")) fprintf(stderr, "[fdgpu_ingest] packed %llu structures / %llu residues in %.2f ms on %u threads\n", (unsigned long long)n, (unsigned long long)R,
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pack0).count(), T);
    *out = P;
    return FDGPU_OK;
}
}  // namespace

// where the ingest threads spend their time, summed over the threads of every fdgpu_parse_structures call of the process:
// [0] read + inflate (zlib), [1] text -> atom records, [2] CompactStructure::build, [3] files, [4] inflated bytes
static std::atomic<uint64_t> g_ingest_ns[5];
// the ingest's gzip decoder on a buffer (tests): FDGPU_EINVAL when it declines the input (the ingest then reads the file through zlib)
extern "C" int fdgpu_debug_gunzip(const uint8_t *in, uint64_t n, uint8_t **out, uint64_t *n_out) {
    if (!in || !out || !n_out) return FDGPU_EINVAL;
    std::string o;
    if (!fd_gunzip(in, (size_t)n, &o)) return FDGPU_EINVAL;
    uint8_t *p = (uint8_t *)malloc(o.size() ? o.size() : 1);
    if (!p) return FDGPU_ENOMEM;
    memcpy(p, o.data(), o.size());
    *out = p; *n_out = o.size();
    return FDGPU_OK;
}
extern "C" void fdgpu_ingest_stats(double out[5], int reset) {
    for (int k = 0; k < 5; ++k) { if (out) out[k] = k < 3 ? (double)g_ingest_ns[k].load() * 1e-9 : (double)g_ingest_ns[k].load(); if (reset) g_ingest_ns[k] = 0; }
}

extern "C" int fdgpu_parse_structures(const char *const *paths, uint64_t n, uint32_t n_threads, uint64_t max_residue, fd_parsed **out) {
    if (!out || (n && !paths)) return FDGPU_EINVAL;
    *out = nullptr;
    std::vector<Compact> parts(n);
    std::atomic<uint64_t> next(0);
    auto work = [&]() {
        std::string txt;
        std::vector<Atom> atoms;
        for (;;) {
            uint64_t k = next.fetch_add(1);
            if (k >= n) break;
            Compact &C = parts[k];
            const auto t0 = std::chrono::steady_clock::now();
            if (!paths[k] || !read_all(paths[k], &txt)) continue;
            const auto t1 = std::chrono::steady_clock::now();
            atoms.clear();
            std::string p(paths[k]);
            if (ends_with_ci(p, ".cif") || ends_with_ci(p, ".cif.gz") || ends_with_ci(p, ".mmcif") || ends_with_ci(p, ".mmcif.gz")) parse_cif(txt, &atoms);
            else parse_pdb(txt, &atoms, ends_with_ci(p, ".gz"));
            const auto t2 = std::chrono::steady_clock::now();
            build_compact(atoms, &C);
            const auto t3 = std::chrono::steady_clock::now();
            g_ingest_ns[0] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            g_ingest_ns[1] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
            g_ingest_ns[2] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count();
            g_ingest_ns[3] += 1; g_ingest_ns[4] += txt.size();
            if (max_residue && C.nres_raw > max_residue) {   // controller/mod.rs:313-318: id kept, no hashes, nres = 0
                uint64_t raw = C.nres_raw;
                uint8_t fc = C.first_chain;
                C = Compact();
                C.nres_raw = raw; C.first_chain = fc; C.ok = true;
            }
        }
    };
    // default: every core up to 64 (measured on the 256-thread MI355X host: 19.6 k files/s at 64 threads, 11 k at 256)
    uint32_t T = n_threads ? n_threads : std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    T = (uint32_t)std::min<uint64_t>(T, std::max<uint64_t>(n, 1));
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();

    return pack_parsed(parts, max_residue, out, T);
}

extern "C" void fdgpu_parsed_free(fd_parsed *P) {
    if (!P) return;
    free(P->res_off); free(P->n_xyz); free(P->ca_xyz); free(P->cb_xyz); free(P->aa); free(P->cb_valid); free(P->chain); free(P->resname_std);
    free(P->serial); free(P->bfac); free(P->resname); free(P->nres_raw); free(P->plddt); free(P->ok); free(P->first_chain);
    free(P);
}


// ---- Foldcomp input (fd_fcz.cpp) ---------------------------------------------------------------------------------------------------
// One entry -> its atom records (what foldcomp_process returns to src/structure/io/fcz.rs:82-93); parity-test seam.
extern "C" int fdgpu_foldcomp_decode(const uint8_t *entry, uint64_t len, fd_foldcomp_atom **atoms, uint64_t *n_atoms) {
    if (!entry || !atoms || !n_atoms) return FDGPU_EINVAL;
    *atoms = nullptr; *n_atoms = 0;
    std::vector<fd_fcz_atom> v;
    if (fd_fcz_decode(entry, (size_t)len, &v) != 0) return FDGPU_EINVAL;
    static_assert(sizeof(fd_foldcomp_atom) == sizeof(fd_fcz_atom), "fd_foldcomp_atom layout");
    fd_foldcomp_atom *o = (fd_foldcomp_atom *)malloc(std::max<size_t>(v.size(), 1) * sizeof(fd_foldcomp_atom));
    if (!o) return FDGPU_ENOMEM;
    if (!v.empty()) memcpy(o, v.data(), v.size() * sizeof(fd_foldcomp_atom));
    *atoms = o; *n_atoms = v.size();
    return FDGPU_OK;
}

namespace {
bool read_file_bytes(const std::string &path, std::string *out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    out->clear();
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}
struct db_ent { uint64_t key, start, len; };
// DB.index: "key \t start \t length" per line, sorted by key (FoldcompDbReader::new sorts it, fcz.rs:52-54)
bool read_db_index(const char *db_path, std::vector<db_ent> *idx) {
    std::string txt;
    if (!read_file_bytes(std::string(db_path) + ".index", &txt)) return false;
    size_t pos = 0;
    while (pos < txt.size()) {
        size_t e = txt.find('\n', pos);
        if (e == std::string::npos) e = txt.size();
        unsigned long long k, s, l;
        if (sscanf(txt.substr(pos, e - pos).c_str(), "%llu\t%llu\t%llu", &k, &s, &l) == 3) idx->push_back({k, s, l});
        pos = e + 1;
    }
    std::sort(idx->begin(), idx->end(), [](const db_ent &a, const db_ent &b) { return a.key < b.key; });
    return true;
}
}  // namespace

// The entries of a Foldcomp database in ascending key order: keys (DB.index) and names (DB.lookup column 2, '\n'-joined; empty for a
// key the lookup does not hold).  The reference builds its path vector and db_key vector the same way
// (cli/workflows/build_index.rs:114-123, fcz.rs:208-219, controller/mod.rs:151).
extern "C" int fdgpu_foldcomp_db_list(const char *db_path, uint64_t **keys, char **names, uint64_t *n_entries) {
    if (!db_path || !keys || !names || !n_entries) return FDGPU_EINVAL;
    *keys = nullptr; *names = nullptr; *n_entries = 0;
    std::vector<db_ent> idx;
    std::string lk_txt;
    if (!read_db_index(db_path, &idx) || !read_file_bytes(std::string(db_path) + ".lookup", &lk_txt)) return FDGPU_EINVAL;
    std::vector<std::pair<uint64_t, std::string>> lookup;
    size_t pos = 0;
    while (pos < lk_txt.size()) {
        size_t e = lk_txt.find('\n', pos);
        if (e == std::string::npos) e = lk_txt.size();
        const std::string line = lk_txt.substr(pos, e - pos);
        const size_t t1 = line.find('\t');
        if (t1 != std::string::npos) {
            const size_t t2 = line.find('\t', t1 + 1);
            lookup.push_back({strtoull(line.c_str(), nullptr, 10), line.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1)});
        }
        pos = e + 1;
    }
    std::sort(lookup.begin(), lookup.end());
    const uint64_t n = idx.size();
    uint64_t *k = (uint64_t *)malloc(std::max<uint64_t>(n, 1) * 8);
    std::string all;
    for (uint64_t i = 0; i < n; ++i) {
        if (k) k[i] = idx[i].key;
        auto it = std::lower_bound(lookup.begin(), lookup.end(), std::make_pair(idx[i].key, std::string()));
        if (it != lookup.end() && it->first == idx[i].key) all += it->second;
        all += '\n';
    }
    char *nm = (char *)malloc(all.size() + 1);
    if (!k || !nm) { free(k); free(nm); return FDGPU_ENOMEM; }
    memcpy(nm, all.c_str(), all.size() + 1);
    *keys = k; *names = nm; *n_entries = n;
    return FDGPU_OK;
}

// Entries of a Foldcomp database (by key, in the order given; n_keys = 0: every entry in ascending key order) -> the packed arrays,
// like fdgpu_parse_structures: each entry is decoded (fd_fcz.cpp) and passed through CompactStructure::build, which is what
// FoldcompDbReader::read_single_structure_by_id + to_compact do (fcz.rs:104-126, controller/mod.rs:301-322).  A key that is not in
// DB.index, or an entry that does not decode, gives a structure with ok = 0 and no residues.
extern "C" int fdgpu_parse_foldcomp_db(const char *db_path, const uint64_t *keys, uint64_t n_keys, uint32_t n_threads, uint64_t max_residue, fd_parsed **out) {
    if (!db_path || !out || (n_keys && !keys)) return FDGPU_EINVAL;
    *out = nullptr;
    const auto t_fc0 = std::chrono::steady_clock::now();
    std::vector<db_ent> idx;
    if (!read_db_index(db_path, &idx)) return FDGPU_EINVAL;
    int fdesc = open(db_path, O_RDONLY);
    if (fdesc < 0) return FDGPU_EINVAL;
    struct stat sb;
    if (fstat(fdesc, &sb) != 0) { close(fdesc); return FDGPU_EINVAL; }
    const size_t db_len = (size_t)sb.st_size;
    const uint8_t *db = db_len ? (const uint8_t *)mmap(nullptr, db_len, PROT_READ, MAP_PRIVATE, fdesc, 0) : nullptr;
    close(fdesc);
    if (db_len && db == (const uint8_t *)MAP_FAILED) return FDGPU_EINVAL;
    const uint64_t n = n_keys ? n_keys : idx.size();
    std::vector<Compact> parts(n);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        std::vector<Atom> atoms;
        std::vector<fd_fcz_atom> raw;
        for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= n) break;
            Compact &C = parts[k];
            const db_ent *e = nullptr;
            if (n_keys) {
                auto it = std::lower_bound(idx.begin(), idx.end(), keys[k], [](const db_ent &a, uint64_t key) { return a.key < key; });
                if (it != idx.end() && it->key == keys[k]) e = &*it;
            } else e = &idx[k];
            if (!e || e->start > db_len || e->len > db_len - e->start) continue;     // no wrap-around: both come from DB.index
            if (fd_fcz_decode(db + e->start, (size_t)e->len, &raw) != 0) continue;
            atoms.resize(raw.size());
            for (size_t a = 0; a < raw.size(); ++a) {
                atoms[a].x = raw[a].x; atoms[a].y = raw[a].y; atoms[a].z = raw[a].z; atoms[a].b = raw[a].b;
                memcpy(atoms[a].name, raw[a].name, 4); memcpy(atoms[a].res, raw[a].res, 3);
                atoms[a].chain = raw[a].chain; atoms[a].rser = raw[a].rser;
            }
            build_compact(atoms, &C);
            if (max_residue && C.nres_raw > max_residue) {
                const uint64_t rawn = C.nres_raw;
                const uint8_t fc = C.first_chain;
                C = Compact();
                C.nres_raw = rawn; C.first_chain = fc; C.ok = true;
            }
        }
    };
    uint32_t T = n_threads ? n_threads : std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    T = (uint32_t)std::min<uint64_t>(T, std::max<uint64_t>(n, 1));
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (getenv("FDGPU_TRACE")) fprintf(stderr, "[fdgpu_ingest] %llu Foldcomp entries decoded in %.2f ms on %u threads\n", (unsigned long long)n,
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fc0).count(), T);
    if (db_len) munmap((void *)db, db_len);
    return pack_parsed(parts, max_residue, out, T);
}

// ---- PREFIX.lookup --------------------------------------------------------------------------------------------------------------
// Rust `{}` of an f32 (what src/index/lookup.rs:44 prints for the plddt column): the shortest digits that round-trip, NEVER exponent form,
// integral values without a fraction, "NaN" / "inf" / "-inf".  Digits = the shortest scientific form (std::to_chars), laid out positionally.
#include <charconv>
static size_t fd_f32_display(float v, char *out) {
    if (v != v) { memcpy(out, "NaN", 3); return 3; }
    if (std::isinf(v)) { const char *s = v > 0 ? "inf" : "-inf"; const size_t n = strlen(s); memcpy(out, s, n); return n; }
    char sci[48];
    const auto r = std::to_chars(sci, sci + sizeof sci - 1, v, std::chars_format::scientific);      // d[.ddd]e[+-]xx, shortest round-trip
    *r.ptr = 0;      // (to_chars does not terminate: the exponent is parsed with strtol below)
    const char *p = sci;
    const bool neg = *p == '-';
    if (neg) ++p;
    char digits[24];
    size_t nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p) if (*p != '.') digits[nd++] = *p;
    const int ex = (int)strtol(p + 1, nullptr, 10);
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    size_t o = 0;
    if (neg) out[o++] = '-';        // -0.0 prints as "-0" (Rust's Display keeps the sign of a negative zero)
    if (ex < 0) {
        out[o++] = '0'; out[o++] = '.';
        for (int k = 0; k < -ex - 1; ++k) out[o++] = '0';
        memcpy(out + o, digits, nd); o += nd;
    } else if ((int)nd <= ex + 1) {
        memcpy(out + o, digits, nd); o += nd;
        for (int k = 0; k < ex + 1 - (int)nd; ++k) out[o++] = '0';
    } else {
        memcpy(out + o, digits, (size_t)ex + 1); o += (size_t)ex + 1;
        out[o++] = '.';
        memcpy(out + o, digits + ex + 1, nd - (size_t)ex - 1); o += nd - (size_t)ex - 1;
    }
    return o;
}
// n values -> out[k * 64 ..] NUL-terminated strings (a float's positional form has at most 1 + 39 + 1 + 9 characters)
extern "C" int fdgpu_format_f32_display(const float *v, uint64_t n, char *out) {
    if ((n && !v) || !out) return FDGPU_EINVAL;
    for (uint64_t k = 0; k < n; ++k) { const size_t m = fd_f32_display(v[k], out + 64 * k); out[64 * k + m] = 0; }
    return FDGPU_OK;
}
// id \t tid \t nres \t plddt \t db_key \n per structure (src/index/lookup.rs:35-56, build_index.rs:204-215).  tids: the n ids joined by '\n'
// (an id holds no newline); db_keys NULL = the id itself.
extern "C" int fdgpu_write_lookup(const char *path, const char *tids, uint64_t n, const uint64_t *nres, const float *plddt, const uint64_t *db_keys) {
    if (!path || (n && (!tids || !nres || !plddt))) return FDGPU_EINVAL;
    std::string buf;
    buf.reserve((size_t)n * 64);
    const char *t = tids;
    char num[64];
    for (uint64_t k = 0; k < n; ++k) {
        const char *e = strchr(t, '\n');
        const size_t len = e ? (size_t)(e - t) : strlen(t);
        if (!e && k + 1 < n) return FDGPU_EINVAL;      // fewer ids than structures
        buf.append(num, (size_t)snprintf(num, sizeof num, "%llu\t", (unsigned long long)k));
        buf.append(t, len);
        buf.append(num, (size_t)snprintf(num, sizeof num, "\t%llu\t", (unsigned long long)nres[k]));
        buf.append(num, fd_f32_display(plddt[k], num));
        buf.append(num, (size_t)snprintf(num, sizeof num, "\t%llu\n", (unsigned long long)(db_keys ? db_keys[k] : k)));
        t = e ? e + 1 : t + len;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return FDGPU_EINVAL;
    const bool ok = buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    return fclose(f) == 0 && ok ? FDGPU_OK : FDGPU_EINVAL;
}
