/* fdo_kabsch.c — TEST INFRASTRUCTURE (see fd_oracle.h).
 * Restates src/structure/kabsch.rs:157-554 `kabsch(x, y, mode)` (TM-align derived closed-form
 * superposition, f64 internally, f32 results; NaN rmsd -> f32::MAX). */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include "fd_oracle.h"

float fdo_kabsch(const float *xf, const float *yf, uint64_t n, int mode, float rot[9], float tran[3]) {
    static const int IP[9] = {0, 1, 3, 1, 2, 4, 3, 4, 5};
    static const int IP2312[4] = {1, 2, 0, 1};
    const double EPSILON = 1.0e-8, TOLERANCE = 0.01, SQRT3 = 1.7320508075688772;
    for (int i = 0; i < 9; ++i) rot[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    tran[0] = tran[1] = tran[2] = 0.0f;
    if (n == 0) return FLT_MAX;

    double rms = 0.0, e0 = 0.0, sigma;
    double s1[3] = {0}, s2[3] = {0}, sx[3] = {0}, sy[3] = {0}, sz[3] = {0}, xc[3], yc[3], t[3] = {0}, e[3];
    double r[3][3], a[3][3] = {{0}}, b[3][3] = {{0}}, u[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double rr[6], ss[6];

    for (uint64_t i = 0; i < n; ++i) {
        double c1[3] = {xf[3 * i], xf[3 * i + 1], xf[3 * i + 2]};
        double c2[3] = {yf[3 * i], yf[3 * i + 1], yf[3 * i + 2]};
        for (int j = 0; j < 3; ++j) { s1[j] += c1[j]; s2[j] += c2[j]; }
        sx[0] += c1[0] * c2[0]; sx[1] += c1[0] * c2[1]; sx[2] += c1[0] * c2[2];
        sy[0] += c1[1] * c2[0]; sy[1] += c1[1] * c2[1]; sy[2] += c1[1] * c2[2];
        sz[0] += c1[2] * c2[0]; sz[1] += c1[2] * c2[1]; sz[2] += c1[2] * c2[2];
    }
    double dn = (double)n;
    for (int j = 0; j < 3; ++j) { xc[j] = s1[j] / dn; yc[j] = s2[j] / dn; }
    if (mode == 0 || mode == 2) {
        for (uint64_t i = 0; i < n; ++i) {
            double d;
            for (int j = 0; j < 3; ++j) {
                double dx = (double)xf[3 * i + j] - xc[j], dy = (double)yf[3 * i + j] - yc[j];
                d = dx * dx + dy * dy; /* powi(2) + powi(2) */
                e0 += d;
            }
        }
    }
    for (int j = 0; j < 3; ++j) {
        r[j][0] = sx[j] - s1[0] * s2[j] / dn;
        r[j][1] = sy[j] - s1[1] * s2[j] / dn;
        r[j][2] = sz[j] - s1[2] * s2[j] / dn;
    }
    double det_r = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) +
                   r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
    sigma = det_r;
    int m = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i <= j; ++i) rr[m++] = r[0][i] * r[0][j] + r[1][i] * r[1][j] + r[2][i] * r[2][j];
    double spur = (rr[0] + rr[2] + rr[5]) / 3.0;
    double cof = (((rr[2] * rr[5] - rr[4] * rr[4]) + rr[0] * rr[5] - rr[3] * rr[3]) + rr[0] * rr[2] - rr[1] * rr[1]) / 3.0;
    double det = det_r * det_r;
    e[0] = e[1] = e[2] = spur;

    if (spur > 0.0) {
        double d = spur * spur;
        double h = d - cof;
        double g = (spur * cof - det) / 2.0 - spur * h;
        if (h > 0.0) {
            double sqrth = sqrt(h);
            double disc = h * h * h - g * g;
            if (disc < 0.0) disc = 0.0;
            double sqrt_disc = sqrt(disc);
            double d_ang = fabs(g) > 1e18 ? (g > 0.0 ? M_PI / 3.0 : 0.0) : atan2(sqrt_disc, -g) / 3.0;
            double cth = sqrth * cos(d_ang);
            double sth = sqrth * SQRT3 * sin(d_ang);
            e[0] = spur + 2.0 * cth;
            e[1] = spur - cth + sth;
            e[2] = spur - cth - sth;
            if (mode != 0) {
                int a_failed = 0, b_failed = 0;
                for (int li = 0; li < 2; ++li) {
                    int l = li == 0 ? 0 : 2;
                    double dl = e[l];
                    ss[0] = (dl - rr[2]) * (dl - rr[5]) - rr[4] * rr[4];
                    ss[1] = (dl - rr[5]) * rr[1] + rr[3] * rr[4];
                    ss[2] = (dl - rr[0]) * (dl - rr[5]) - rr[3] * rr[3];
                    ss[3] = (dl - rr[2]) * rr[3] + rr[1] * rr[4];
                    ss[4] = (dl - rr[0]) * rr[4] + rr[1] * rr[3];
                    ss[5] = (dl - rr[0]) * (dl - rr[2]) - rr[1] * rr[1];
                    for (int k = 0; k < 6; ++k)
                        if (fabs(ss[k]) <= EPSILON) ss[k] = 0.0;
                    double A = fabs(ss[0]), B = fabs(ss[2]), C = fabs(ss[5]);
                    int j = (A >= B && A >= C) ? 0 : (B >= C ? 1 : 2);
                    double dnorm = 0.0;
                    for (int i = 0; i < 3; ++i) {
                        int k = IP[3 * j + i];
                        a[i][l] = ss[k];
                        dnorm += ss[k] * ss[k];
                    }
                    dnorm = dnorm > EPSILON ? 1.0 / sqrt(dnorm) : 0.0;
                    for (int i = 0; i < 3; ++i) a[i][l] *= dnorm;
                }
                double dt = a[0][0] * a[0][2] + a[1][0] * a[1][2] + a[2][0] * a[2][2];
                int m1, mm;
                if (e[0] - e[1] > e[1] - e[2]) { m1 = 2; mm = 0; } else { m1 = 0; mm = 2; }
                double p = 0.0;
                for (int i = 0; i < 3; ++i) {
                    a[i][m1] = a[i][m1] - dt * a[i][mm];
                    p += a[i][m1] * a[i][m1];
                }
                if (p <= TOLERANCE) {
                    int j = 0;
                    p = 1.0;
                    for (int i = 0; i < 3; ++i)
                        if (p < fabs(a[i][mm])) { p = fabs(a[i][mm]); j = i; }
                    int k = IP2312[j], l = IP2312[j + 1];
                    p = sqrt(a[k][mm] * a[k][mm] + a[l][mm] * a[l][mm]);
                    if (p > TOLERANCE) {
                        a[j][m1] = 0.0;
                        a[k][m1] = -a[l][mm] / p;
                        a[l][m1] = a[k][mm] / p;
                    } else a_failed = 1;
                } else {
                    p = 1.0 / sqrt(p);
                    for (int i = 0; i < 3; ++i) a[i][m1] *= p;
                }
                if (!a_failed) {
                    a[0][1] = a[1][2] * a[2][0] - a[1][0] * a[2][2];
                    a[1][1] = a[2][2] * a[0][0] - a[2][0] * a[0][2];
                    a[2][1] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
                    for (int l = 0; l < 2; ++l) {
                        double db = 0.0;
                        for (int i = 0; i < 3; ++i) {
                            b[i][l] = r[i][0] * a[0][l] + r[i][1] * a[1][l] + r[i][2] * a[2][l];
                            db += b[i][l] * b[i][l];
                        }
                        db = db > EPSILON ? 1.0 / sqrt(db) : 0.0;
                        for (int i = 0; i < 3; ++i) b[i][l] *= db;
                    }
                    double dot_b = 0.0;
                    for (int i = 0; i < 3; ++i) dot_b += b[i][0] * b[i][1];
                    double pb = 0.0;
                    for (int i = 0; i < 3; ++i) {
                        b[i][1] -= dot_b * b[i][0];
                        pb += b[i][1] * b[i][1];
                    }
                    if (pb <= TOLERANCE) {
                        pb = 1.0;
                        int j = 0;
                        for (int i = 0; i < 3; ++i)
                            if (pb < fabs(b[i][0])) { pb = fabs(b[i][0]); j = i; }
                        int k = IP2312[j], l = IP2312[j + 1];
                        pb = sqrt(b[k][0] * b[k][0] + b[l][0] * b[l][0]);
                        if (pb > TOLERANCE) {
                            b[j][1] = 0.0;
                            b[k][1] = -b[l][0] / pb;
                            b[l][1] = b[k][0] / pb;
                        } else b_failed = 1;
                    } else {
                        pb = 1.0 / sqrt(pb);
                        for (int i = 0; i < 3; ++i) b[i][1] *= pb;
                    }
                    if (!b_failed) {
                        b[0][2] = b[1][0] * b[2][1] - b[1][1] * b[2][0];
                        b[1][2] = b[2][0] * b[0][1] - b[2][1] * b[0][0];
                        b[2][2] = b[0][0] * b[1][1] - b[0][1] * b[1][0];
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                u[i][j] = b[i][0] * a[j][0] + b[i][1] * a[j][1] + b[i][2] * a[j][2];
                        for (int i = 0; i < 3; ++i)
                            t[i] = yc[i] - (u[i][0] * xc[0] + u[i][1] * xc[1] + u[i][2] * xc[2]);
                    }
                }
            }
        }
    } else {
        for (int i = 0; i < 3; ++i) t[i] = yc[i] - (u[i][0] * xc[0] + u[i][1] * xc[1] + u[i][2] * xc[2]);
    }
    (void)e0; (void)sigma; /* rms1 = e0 - 2*d is computed but unused by the reference (kabsch.rs:507-514) */
    if (mode == 0 || mode == 2) {
        double sum_sq = 0.0;
        for (uint64_t i = 0; i < n; ++i) {
            double X = xf[3 * i], Y = xf[3 * i + 1], Z = xf[3 * i + 2];
            double tr[3] = {u[0][0] * X + u[0][1] * Y + u[0][2] * Z + t[0], u[1][0] * X + u[1][1] * Y + u[1][2] * Z + t[1],
                            u[2][0] * X + u[2][1] * Y + u[2][2] * Z + t[2]};
            for (int j = 0; j < 3; ++j) {
                double diff = tr[j] - (double)yf[3 * i + j];
                sum_sq += diff * diff;
            }
        }
        rms = sqrt(sum_sq / dn);
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot[3 * i + j] = (float)u[i][j];
        tran[i] = (float)t[i];
    }
    float rf = (float)rms;
    if (rf != rf) rf = FLT_MAX;
    return rf;
}

/* Similarity metrics of one superposition (src/structure/metrics.rs:62-251 after KabschSuperimposer::run,
 * src/structure/kabsch.rs:86-95,145-154): ref = fixed points (query), mov = moving points (target), transformed =
 * rot * mov + tran in f32; pairwise f32 distances evaluated in f64; out = {tm_score, gdt_ts, gdt_ha, chamfer, hausdorff}.
 * Quirk kept: tm_score and gdt compare the DISTANCE (not its square) with d0^2 / cutoff^2 (metrics.rs:141-143,160-163). */
void fdo_metrics(const float *ref, const float *mov, uint64_t n, const float rot[9], const float tran[3], float out[5]) {
    if (n == 0) { out[0] = out[1] = out[2] = 0.0f; out[3] = out[4] = INFINITY; return; }
    float *tr = (float *)malloc(n * 3 * sizeof(float));
    for (uint64_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) {
            float v = rot[3 * r] * mov[3 * i] + rot[3 * r + 1] * mov[3 * i + 1] + rot[3 * r + 2] * mov[3 * i + 2];
            tr[3 * i + r] = v + tran[r];
        }
    #define FDO_DIST(c, r) ((float)sqrt(((double)ref[3 * (r)] - (double)tr[3 * (c)]) * ((double)ref[3 * (r)] - (double)tr[3 * (c)]) + \
                                        ((double)ref[3 * (r) + 1] - (double)tr[3 * (c) + 1]) * ((double)ref[3 * (r) + 1] - (double)tr[3 * (c) + 1]) + \
                                        ((double)ref[3 * (r) + 2] - (double)tr[3 * (c) + 2]) * ((double)ref[3 * (r) + 2] - (double)tr[3 * (c) + 2])))
    float d0 = n > 21 ? 1.24f * powf((float)n - 15.0f, 1.0f / 3.0f) - 1.8f : 0.5f;
    double d0_sq = (double)(d0 * d0), tm = 0.0, dn = (double)n;
    for (uint64_t i = 0; i < n; ++i) tm += 1.0 / (1.0 + (double)FDO_DIST(i, i) / d0_sq);
    out[0] = (float)(tm / dn);
    const double ts[4] = {1.0, 2.0, 4.0, 8.0}, ha[4] = {0.5, 1.0, 2.0, 4.0};
    for (int which = 0; which < 2; ++which) {
        const double *cut = which ? ha : ts;
        double sum = 0.0;
        for (int k = 0; k < 4; ++k) {
            uint64_t cnt = 0;
            for (uint64_t i = 0; i < n; ++i) if ((double)FDO_DIST(i, i) <= cut[k] * cut[k]) ++cnt;
            sum += (double)cnt / dn;
        }
        out[1 + which] = (float)(sum / 4.0);
    }
    double ch = 0.0;
    float hd = 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        float mn = FDO_DIST(i, 0);
        for (uint64_t j = 1; j < n; ++j) { float d = FDO_DIST(i, j); if (d < mn) mn = d; }
        ch += (double)mn;
        if (i == 0 || mn > hd) hd = mn;
    }
    out[3] = (float)(ch / dn);
    out[4] = hd;
    #undef FDO_DIST
    free(tr);
}
