/* fdo_structure.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates: src/structure/io/pdb.rs:37-77, src/structure/io/parser.rs:3-56,
 * src/structure/core.rs:28-43 (Structure::update), :70-214 (CompactStructure::build),
 * :216-223 (get_index), :450-460 (get_avg_plddt), src/structure/coordinate.rs:167-186 (approx_cb). */
#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fd_oracle.h"

typedef struct { float x, y, z; } v3;
static v3 vsub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 vadd(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 vscale(v3 a, float f) { v3 r = {a.x * f, a.y * f, a.z * f}; return r; }
static v3 vcross(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static v3 vnormalize(v3 a) {
    float n = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    v3 r = {a.x / n, a.y / n, a.z / n};
    return r;
}
/* coordinate.rs:167-186 */
static v3 approx_cb(v3 ca, v3 n, v3 c) {
    v3 v1 = vnormalize(vsub(c, ca));
    v3 v2 = vnormalize(vsub(n, ca));
    v3 b1 = vadd(v2, vscale(v1, 1.0f / 3.0f));
    v3 b2 = vcross(v1, b1);
    v3 u1 = vnormalize(b1);
    v3 u2 = vnormalize(b2);
    v3 v4 = vsub(vscale(u1, -1.0f / 2.0f), vscale(u2, sqrtf(3.0f) / 2.0f));
    v4 = vscale(v4, sqrtf(8.0f) / 3.0f);
    v4 = vadd(v4, vscale(v1, -1.0f / 3.0f));
    return vadd(ca, vscale(v4, 1.5336f));
}

typedef struct {
    int32_t n, cap;
    float *xyz;
    uint8_t *name4, *res3, *chain;
    uint64_t *rserial;
    float *bfac;
} atoms_t;

static void atoms_push(atoms_t *a, float x, float y, float z, const uint8_t *name4, const uint8_t *res3,
                       uint64_t rs, uint8_t chain, float b) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4096;
        a->xyz = (float *)realloc(a->xyz, (size_t)a->cap * 3 * sizeof(float));
        a->name4 = (uint8_t *)realloc(a->name4, (size_t)a->cap * 4);
        a->res3 = (uint8_t *)realloc(a->res3, (size_t)a->cap * 3);
        a->chain = (uint8_t *)realloc(a->chain, (size_t)a->cap);
        a->rserial = (uint64_t *)realloc(a->rserial, (size_t)a->cap * sizeof(uint64_t));
        a->bfac = (float *)realloc(a->bfac, (size_t)a->cap * sizeof(float));
    }
    int32_t k = a->n++;
    a->xyz[3 * k] = x; a->xyz[3 * k + 1] = y; a->xyz[3 * k + 2] = z;
    memcpy(a->name4 + 4 * k, name4, 4);
    memcpy(a->res3 + 3 * k, res3, 3);
    a->chain[k] = chain;
    a->rserial[k] = rs;
    a->bfac[k] = b;
}
static void atoms_free(atoms_t *a) {
    free(a->xyz); free(a->name4); free(a->res3); free(a->chain); free(a->rserial); free(a->bfac);
}

/* Rust str::trim + parse::<f32>: accept only a full-token decimal float */
static int parse_f32_field(const char *s, int len, float *out) {
    char buf[32];
    int b = 0, e = len;
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    if (e - b <= 0 || e - b >= (int)sizeof buf) return 0;
    memcpy(buf, s + b, (size_t)(e - b));
    buf[e - b] = 0;
    /* Rust rejects hex floats and leading whitespace; PDB never has them */
    for (int k = 0; buf[k]; ++k)
        if (buf[k] == 'x' || buf[k] == 'X') return 0;
    char *end = NULL;
    errno = 0;
    float v = strtof(buf, &end);
    if (end == buf || *end != 0) return 0;
    *out = v;
    return 1;
}
static int parse_u64_field(const char *s, int len, uint64_t *out) {
    int b = 0, e = len;
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    if (e - b <= 0) return 0;
    uint64_t v = 0;
    int k = b;
    if (s[k] == '+') ++k; /* Rust u64::from_str accepts a leading '+' */
    if (k == e) return 0;
    for (; k < e; ++k) {
        if (s[k] < '0' || s[k] > '9') return 0;
        v = v * 10 + (uint64_t)(s[k] - '0');
    }
    *out = v;
    return 1;
}

/* CompactStructure::build (core.rs:70-214) */
static fdo_structure *build_compact(const atoms_t *a, int32_t num_residues_raw, int32_t num_chains,
                                    const uint8_t *chains) {
    fdo_structure *s = (fdo_structure *)calloc(1, sizeof *s);
    int32_t cap = num_residues_raw + 4;
    s->n_xyz = (float *)calloc((size_t)cap * 3, sizeof(float));
    s->ca_xyz = (float *)calloc((size_t)cap * 3, sizeof(float));
    s->cb_xyz = (float *)calloc((size_t)cap * 3, sizeof(float));
    s->cb_ok = (uint8_t *)calloc((size_t)cap, 1);
    s->resname = (uint8_t *)calloc((size_t)cap * 3, 1);
    s->aa = (uint8_t *)calloc((size_t)cap, 1);
    s->chain = (uint8_t *)calloc((size_t)cap, 1);
    s->serial = (uint64_t *)calloc((size_t)cap, sizeof(uint64_t));
    s->bfac = (float *)calloc((size_t)cap, sizeof(float));
    s->num_residues_raw = num_residues_raw;
    s->num_atoms = a->n;
    s->num_chains = num_chains;
    memcpy(s->chains, chains, (size_t)(num_chains < 256 ? num_chains : 256));

    int have_prev = 0;
    uint64_t prev_serial = 0;
    const uint8_t *prev_name = NULL;
    int hn = 0, hca = 0, hcb = 0, hc = 0, hgn = 0, hgc = 0;
    v3 n = {0, 0, 0}, ca = {0, 0, 0}, cb = {0, 0, 0}, c = {0, 0, 0}, gn = {0, 0, 0}, gc = {0, 0, 0};
    int32_t m = 0;
    for (int32_t idx = 0; idx < a->n; ++idx) {
        uint64_t rs = a->rserial[idx];
        if (!have_prev || prev_serial != rs || idx == a->n - 1) {
            if (hn && hca) { /* (Some(n), Some(ca), cb?) arms; others drop the residue */
                if (m >= cap) { /* cannot happen: at most one flush per serial change (+1) */ abort(); }
                memcpy(s->n_xyz + 3 * m, &n, sizeof n);
                memcpy(s->ca_xyz + 3 * m, &ca, sizeof ca);
                s->serial[m] = prev_serial;
                memcpy(s->resname + 3 * m, prev_name, 3);
                s->chain[m] = a->chain[idx]; /* quirk: current atom = first atom of next residue */
                s->bfac[m] = a->bfac[idx];
                if (hcb) {
                    memcpy(s->cb_xyz + 3 * m, &cb, sizeof cb);
                    s->cb_ok[m] = 1;
                } else if (prev_name[0] == 'G' && prev_name[1] == 'L' && prev_name[2] == 'Y' && hgn && hgc) {
                    v3 v = approx_cb(ca, gn, gc);
                    memcpy(s->cb_xyz + 3 * m, &v, sizeof v);
                    s->cb_ok[m] = 1;
                } else if (hc) {
                    v3 v = approx_cb(ca, n, c);
                    memcpy(s->cb_xyz + 3 * m, &v, sizeof v);
                    s->cb_ok[m] = 1;
                } else {
                    s->cb_ok[m] = 0;
                }
                ++m;
            }
            hca = hcb = hn = 0; /* c, gly_n, gly_c are never reset (core.rs:172-176) */
            have_prev = 1;
            prev_serial = rs;
            prev_name = a->res3 + 3 * idx;
        }
        const uint8_t *an = a->name4 + 4 * idx;
        const uint8_t *rn = a->res3 + 3 * idx;
        v3 p = {a->xyz[3 * idx], a->xyz[3 * idx + 1], a->xyz[3 * idx + 2]};
        int is_gly = rn[0] == 'G' && rn[1] == 'L' && rn[2] == 'Y';
        if (!memcmp(an, " CA ", 4)) { ca = p; hca = 1; }
        else if (!memcmp(an, " CB ", 4)) { cb = p; hcb = 1; }
        else if (!memcmp(an, " C  ", 4)) { c = p; hc = 1; }
        else if (!memcmp(an, " N  ", 4) && !is_gly) { n = p; hn = 1; }
        else if (is_gly) {
            if (!memcmp(an, " N  ", 4)) { gn = p; hgn = 1; n = p; hn = 1; }
            else if (!memcmp(an, " C  ", 4)) { gc = p; hgc = 1; } /* unreachable: caught by the C arm */
        }
    }
    s->n = m;
    for (int32_t k = 0; k < m; ++k) s->aa[k] = fdo_map_aa_to_u8(s->resname + 3 * k);
    return s;
}

static fdo_structure *structure_from_atoms_t(const atoms_t *a) {
    /* Structure::update (core.rs:28-43): record starts at (b' ', 0) */
    uint8_t rec_chain = ' ';
    uint64_t rec_serial = 0;
    int32_t num_chains = 0, num_res = 0;
    uint8_t chains[256];
    for (int32_t k = 0; k < a->n; ++k) {
        if (rec_chain != a->chain[k]) {
            if (num_chains < 256) chains[num_chains] = a->chain[k];
            ++num_chains;
            rec_chain = a->chain[k];
        }
        if (rec_serial != a->rserial[k]) { ++num_res; rec_serial = a->rserial[k]; }
    }
    return build_compact(a, num_res, num_chains, chains);
}

fdo_structure *fdo_structure_from_atoms(int32_t natoms, const float *xyz, const uint8_t *atom_name4,
                                        const uint8_t *res_name3, const uint64_t *res_serial,
                                        const uint8_t *chain, const float *bfac) {
    atoms_t a = {0};
    for (int32_t k = 0; k < natoms; ++k)
        atoms_push(&a, xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], atom_name4 + 4 * k, res_name3 + 3 * k,
                   res_serial[k], chain[k], bfac[k]);
    fdo_structure *s = structure_from_atoms_t(&a);
    atoms_free(&a);
    return s;
}

/* pdb.rs:37-77 read_structure + parser.rs:3-56 parse_line */
fdo_structure *fdo_read_pdb(const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    atoms_t a = {0};
    char *line = NULL;
    size_t lcap = 0;
    ssize_t len;
    int model = 0;
    while ((len = getline(&line, &lcap, fp)) >= 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) --len; /* BufRead::lines strips \n, \r\n */
        if (model > 1) break;
        if (len < 6) continue;
        if (!memcmp(line, "MODEL ", 6)) { ++model; continue; }
        if (memcmp(line, "ATOM  ", 6)) continue;
        if (len < 54) continue; /* the reference would panic on the slice; such files are out of scope */
        float x, y, z, b = 1.0f;
        uint64_t aserial, rserial;
        if (!parse_f32_field(line + 30, 8, &x)) continue;
        if (!parse_f32_field(line + 38, 8, &y)) continue;
        if (!parse_f32_field(line + 46, 8, &z)) continue;
        if (!parse_u64_field(line + 6, 5, &aserial)) continue;
        if (!parse_u64_field(line + 22, 4, &rserial)) continue;
        if (len >= 66) { if (!parse_f32_field(line + 60, 6, &b)) continue; }
        atoms_push(&a, x, y, z, (const uint8_t *)line + 12, (const uint8_t *)line + 17, rserial,
                   (uint8_t)line[21], b);
    }
    free(line);
    fclose(fp);
    fdo_structure *s = structure_from_atoms_t(&a);
    atoms_free(&a);
    return s;
}

fdo_structure *fdo_structure_from_packed(int32_t n, const float *n_xyz, const float *ca_xyz,
                                         const float *cb_xyz, const uint8_t *cb_ok, const uint8_t *aa,
                                         const float *bfac) {
    fdo_structure *s = (fdo_structure *)calloc(1, sizeof *s);
    size_t c = (size_t)(n > 0 ? n : 1);
    s->n = n;
    s->num_residues_raw = n;
    s->n_xyz = (float *)malloc(c * 3 * sizeof(float));
    s->ca_xyz = (float *)malloc(c * 3 * sizeof(float));
    s->cb_xyz = (float *)malloc(c * 3 * sizeof(float));
    s->cb_ok = (uint8_t *)malloc(c);
    s->resname = (uint8_t *)malloc(c * 3);
    s->aa = (uint8_t *)malloc(c);
    s->chain = (uint8_t *)malloc(c);
    s->serial = (uint64_t *)malloc(c * sizeof(uint64_t));
    s->bfac = (float *)malloc(c * sizeof(float));
    memcpy(s->n_xyz, n_xyz, (size_t)n * 3 * sizeof(float));
    memcpy(s->ca_xyz, ca_xyz, (size_t)n * 3 * sizeof(float));
    memcpy(s->cb_xyz, cb_xyz, (size_t)n * 3 * sizeof(float));
    for (int32_t k = 0; k < n; ++k) {
        s->cb_ok[k] = cb_ok ? cb_ok[k] : 1;
        s->aa[k] = aa[k];
        memcpy(s->resname + 3 * k, fdo_map_u8_to_aa(aa[k]), 3);
        s->chain[k] = 'A';
        s->serial[k] = (uint64_t)k + 1;
        s->bfac[k] = bfac ? bfac[k] : 0.0f;
    }
    s->num_chains = 1;
    s->chains[0] = 'A';
    return s;
}

void fdo_structure_free(fdo_structure *s) {
    if (!s) return;
    free(s->n_xyz); free(s->ca_xyz); free(s->cb_xyz); free(s->cb_ok); free(s->resname); free(s->aa);
    free(s->chain); free(s->serial); free(s->bfac);
    free(s);
}

/* core.rs:450-460 (f32 running sum, then / n as f32; n == 0 -> NaN) */
float fdo_avg_plddt(const fdo_structure *s) {
    float sum = 0.0f;
    for (int32_t i = 0; i < s->n; ++i) sum += s->bfac[i];
    return sum / (float)s->n;
}

/* core.rs:216-223 */
int64_t fdo_get_index(const fdo_structure *s, uint8_t chain, uint64_t serial) {
    for (int32_t i = 0; i < s->n; ++i)
        if (s->chain[i] == chain && s->serial[i] == serial) return i;
    return -1;
}
