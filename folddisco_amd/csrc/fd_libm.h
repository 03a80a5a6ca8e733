// fd_libm.h — bit-exact restatements of the four libm entry points the
// reference's descriptor/hash path reaches through Rust's f32::{sin,cos,acos,atan2}
// (reference call sites: src/geometry/pdb_tr.rs:45-57, src/structure/coordinate.rs:127,214).
//
// On x86_64-linux-gnu those lower to glibc sinf/cosf/acosf/atan2f. To make the
// gfx950 kernels produce the same u32 hashes as a CPU run, the device code does
// not call OCML; it evaluates the same published algorithms with the same
// IEEE-754 operation order:
//   * sinf/cosf : ARM optimized-routines sincosf (glibc >= 2.28), double
//                 precision polynomial, FMA-contracted exactly as glibc's
//                 x86_64 `__sinf_fma/__cosf_fma` ifunc variants are;
//   * acosf     : fdlibm e_acosf.c (float);
//   * atanf     : fdlibm s_atanf.c (float);
//   * atan2f    : fdlibm e_atan2f.c (float).
// Every operation below must stay un-contracted (-ffp-contract=off); the few
// fused operations are spelled fd_fma().  tools/check_libm.c compares these
// against the host glibc (exhaustively for the one-argument functions).
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define FD_HD __host__ __device__ __forceinline__
#else
#define FD_HD static inline
#endif

FD_HD uint32_t fd_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
FD_HD float fd_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
FD_HD double fd_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
FD_HD float fd_sqrtf(float x) { return __builtin_sqrtf(x); }
FD_HD float fd_fabsf(float x) { return fd_u2f(fd_f2u(x) & 0x7fffffffu); }

// ---- sinf / cosf ---------------------------------------------------------
// Polynomial + reduction constants of __sincosf_table[0]; table[1] negates the
// cosine polynomial (used when quadrant bit 1 is set).
#define FD_HPI_INV 0x1.45F306DC9C883p+23
#define FD_HPI     0x1.921FB54442D18p0
#define FD_C0 0x1p0
#define FD_C1 -0x1.ffffffd0c621cp-2
#define FD_C2 0x1.55553e1068f19p-5
#define FD_C3 -0x1.6c087e89a359dp-10
#define FD_C4 0x1.99343027bf8c3p-16
#define FD_S1 -0x1.555545995a603p-3
#define FD_S2 0x1.1107605230bc4p-7
#define FD_S3 -0x1.994eb3774cf24p-13

// sine polynomial on reduced argument x (|x| <= pi/4), x2 = x*x
FD_HD float fd_sin_poly(double x, double x2) {
    double x3 = x * x2;
    double s1 = fd_fma(x2, FD_S3, FD_S2);
    double x7 = x3 * x2;
    double s = fd_fma(x3, FD_S1, x);
    return (float)fd_fma(x7, s1, s);
}
// cosine polynomial; neg selects table[1] (all cosine coefficients negated)
FD_HD float fd_cos_poly(double x2, int neg) {
    double sg = neg ? -1.0 : 1.0;   // exact sign flips of the table constants
    double x4 = x2 * x2;
    double c2 = fd_fma(x2, sg * FD_C4, sg * FD_C3);
    double c1 = fd_fma(x2, sg * FD_C1, sg * FD_C0);
    double x6 = x4 * x2;
    double c = fd_fma(x4, sg * FD_C2, c1);
    return (float)fd_fma(x6, c2, c);
}

// valid for |y| < 120 (the reference only feeds angles in [-pi, pi]); larger
// magnitudes / NaN / inf return NaN like glibc does for inf/NaN.
FD_HD float fd_sinf(float y) {
    uint32_t top = (fd_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {
        double x2 = x * x;
        if (top < 0x398) return y;
        return fd_sin_poly(x, x2);
    } else if (top < 0x42f) {
        double r = x * FD_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
        double xr = fd_fma(-(double)n, FD_HPI, x);
        double x2 = xr * xr;
        if (n & 1) return fd_cos_poly(x2, n & 2);
        double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return fd_sin_poly(xr * sgn, x2);
    }
    return fd_u2f(0x7fc00000u);
}
FD_HD float fd_cosf(float y) {
    uint32_t top = (fd_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {
        double x2 = x * x;
        if (top < 0x398) return 1.0f;
        return fd_cos_poly(x2, 0);
    } else if (top < 0x42f) {
        double r = x * FD_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
        double xr = fd_fma(-(double)n, FD_HPI, x);
        double x2 = xr * xr;
        if (!(n & 1)) return fd_cos_poly(x2, n & 2);
        double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return fd_sin_poly(xr * sgn, x2);
    }
    return fd_u2f(0x7fc00000u);
}

// ---- acosf (fdlibm) --------------------------------------------------------
FD_HD float fd_acosf(float x) {
    const float one = 1.0f, pi = fd_u2f(0x40490fdau), pio2_hi = fd_u2f(0x3fc90fdau),
                pio2_lo = fd_u2f(0x33a22168u),
                pS0 = fd_u2f(0x3e2aaaabu), pS1 = fd_u2f(0xbea6b090u), pS2 = fd_u2f(0x3e4e0aa8u),
                pS3 = fd_u2f(0xbd241146u), pS4 = fd_u2f(0x3a4f7f04u), pS5 = fd_u2f(0x3811ef08u),
                qS1 = fd_u2f(0xc019d139u), qS2 = fd_u2f(0x4001572du), qS3 = fd_u2f(0xbf303361u),
                qS4 = fd_u2f(0x3d9dc62eu);
    int32_t hx = (int32_t)fd_f2u(x);
    int32_t ix = hx & 0x7fffffff;
    float z, p, q, r, w, s, c, df;
    if (ix == 0x3f800000) {
        if (hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    } else if (ix > 0x3f800000) {
        return fd_u2f(0x7fc00000u);  // (x-x)/(x-x): NaN (sign is irrelevant downstream)
    }
    if (ix < 0x3f000000) {
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = fd_sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {
        z = (one - x) * 0.5f;
        s = fd_sqrtf(z);
        df = fd_u2f(fd_f2u(s) & 0xfffff000u);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}

// ---- atanf (fdlibm) ----------------------------------------------------------
FD_HD float fd_atanf(float x) {
    const float aT0 = fd_u2f(0x3eaaaaabu), aT1 = fd_u2f(0xbe4ccccdu), aT2 = fd_u2f(0x3e124925u),
                aT3 = fd_u2f(0xbde38e38u), aT4 = fd_u2f(0x3dba2e6eu), aT5 = fd_u2f(0xbd9d8795u),
                aT6 = fd_u2f(0x3d886b35u), aT7 = fd_u2f(0xbd6ef16bu), aT8 = fd_u2f(0x3d4bda59u),
                aT9 = fd_u2f(0xbd15a221u), aT10 = fd_u2f(0x3c8569d7u);
    int32_t hx = (int32_t)fd_f2u(x);
    int32_t ix = hx & 0x7fffffff;
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        float r = fd_u2f(0x3fc90fdau) + fd_u2f(0x33a22168u);
        return hx > 0 ? r : -r;
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = fd_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x);
                hi = fd_u2f(0x3eed6338u); lo = fd_u2f(0x31ac3769u); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f);
                hi = fd_u2f(0x3f490fdau); lo = fd_u2f(0x33222168u); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x);
                hi = fd_u2f(0x3f7b985eu); lo = fd_u2f(0x33140fb4u); }
            else { id = 3; x = -1.0f / x;
                hi = fd_u2f(0x3fc90fdau); lo = fd_u2f(0x33a22168u); }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -z : z;
}

// ---- atan2f (fdlibm) -----------------------------------------------------------
FD_HD float fd_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = fd_u2f(0x3f490fdbu), pi_o_2 = fd_u2f(0x3fc90fdbu),
                pi = fd_u2f(0x40490fdbu), pi_lo = fd_u2f(0xb3bbbd2eu);
    int32_t hx = (int32_t)fd_f2u(x), hy = (int32_t)fd_f2u(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return fd_atanf(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fd_atanf(fd_fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return fd_u2f(fd_f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// =============================================================================
// Branch-lean forms used inside the gfx950 kernels.  Same IEEE operations on the
// same operands as the functions above (so bit-identical results), but written
// with selects instead of region branches so that a 64-lane wavefront does not
// serialise over the regions.  tools/check_libm.c checks them too.
// =============================================================================

// sinf(y) and cosf(y) together: one sine- and one cosine-polynomial evaluation.
// For |y| < 0.75 glibc skips the reduction; the reduction yields n = 0, xr = y
// exactly there, so the unified path is identical (tiny |y| handled by select).
FD_HD void fdd_sincosf(float y, float *sn, float *cs) {
    uint32_t top = (fd_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    double r = x * FD_HPI_INV;
    // outside |y| < 120 (incl. NaN/inf) the result is NaN; keep the int conversion defined
    int valid = top < 0x42f;
    int n = valid ? (((int32_t)r + 0x800000) >> 24) : 0;
    double xr = fd_fma(-(double)n, FD_HPI, x);
    double x2 = xr * xr;
    double sgn = ((n + 1) & 2) ? -1.0 : 1.0;  // sign[n&3] = {1,-1,-1,1}
    float S = fd_sin_poly(xr * sgn, x2);
    float Cp = fd_cos_poly(x2, n & 2);
    float s_out = (n & 1) ? Cp : S;
    float c_out = (n & 1) ? S : Cp;
    if (top < 0x398) { s_out = y; c_out = 1.0f; }
    if (!valid) { s_out = fd_u2f(0x7fc00000u); c_out = s_out; }
    *sn = s_out;
    *cs = c_out;
}

FD_HD float fdd_acosf(float x) {
    const float one = 1.0f, pi = fd_u2f(0x40490fdau), pio2_hi = fd_u2f(0x3fc90fdau),
                pio2_lo = fd_u2f(0x33a22168u),
                pS0 = fd_u2f(0x3e2aaaabu), pS1 = fd_u2f(0xbea6b090u), pS2 = fd_u2f(0x3e4e0aa8u),
                pS3 = fd_u2f(0xbd241146u), pS4 = fd_u2f(0x3a4f7f04u), pS5 = fd_u2f(0x3811ef08u),
                qS1 = fd_u2f(0xc019d139u), qS2 = fd_u2f(0x4001572du), qS3 = fd_u2f(0xbf303361u),
                qS4 = fd_u2f(0x3d9dc62eu);
    int32_t hx = (int32_t)fd_f2u(x);
    int32_t ix = hx & 0x7fffffff;
    int small = ix < 0x3f000000;
    float z = small ? x * x : ((hx < 0 ? one + x : one - x) * 0.5f);
    float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    float r = p / q;
    float s = fd_sqrtf(z);
    float res_small = pio2_hi - (x - (pio2_lo - x * r));
    float w_neg = r * s - pio2_lo;
    float res_neg = pi - 2.0f * (s + w_neg);
    float df = fd_u2f(fd_f2u(s) & 0xfffff000u);
    float c = (z - df * df) / (s + df);
    float w_pos = r * s + c;
    float res_pos = 2.0f * (df + w_pos);
    float res = small ? res_small : (hx < 0 ? res_neg : res_pos);
    if (small && ix <= 0x32800000) res = pio2_hi + pio2_lo;
    if (ix == 0x3f800000) res = hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) res = fd_u2f(0x7fc00000u);
    return res;
}

FD_HD float fdd_atanf(float x) {
    const float aT0 = fd_u2f(0x3eaaaaabu), aT1 = fd_u2f(0xbe4ccccdu), aT2 = fd_u2f(0x3e124925u),
                aT3 = fd_u2f(0xbde38e38u), aT4 = fd_u2f(0x3dba2e6eu), aT5 = fd_u2f(0xbd9d8795u),
                aT6 = fd_u2f(0x3d886b35u), aT7 = fd_u2f(0xbd6ef16bu), aT8 = fd_u2f(0x3d4bda59u),
                aT9 = fd_u2f(0xbd15a221u), aT10 = fd_u2f(0x3c8569d7u);
    int32_t hx = (int32_t)fd_f2u(x);
    int32_t ix = hx & 0x7fffffff;
    float ax = fd_fabsf(x);
    // region: -1 (|x|<7/16), 0, 1, 2, 3
    int id = ix < 0x3ee00000 ? -1 : (ix < 0x3f300000 ? 0 : (ix < 0x3f980000 ? 1 : (ix < 0x401c0000 ? 2 : 3)));
    float num = id == 0 ? 2.0f * ax - 1.0f : (id == 1 ? ax - 1.0f : (id == 2 ? ax - 1.5f : -1.0f));
    float den = id == 0 ? 2.0f + ax : (id == 1 ? ax + 1.0f : (id == 2 ? 1.0f + 1.5f * ax : ax));
    float xr = id < 0 ? x : num / den;
    float hi = id == 0 ? fd_u2f(0x3eed6338u) : (id == 1 ? fd_u2f(0x3f490fdau) : (id == 2 ? fd_u2f(0x3f7b985eu) : fd_u2f(0x3fc90fdau)));
    float lo = id == 0 ? fd_u2f(0x31ac3769u) : (id == 1 ? fd_u2f(0x33222168u) : (id == 2 ? fd_u2f(0x33140fb4u) : fd_u2f(0x33a22168u)));
    float z = xr * xr;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    float zz = hi - ((xr * (s1 + s2) - lo) - xr);
    float res = id < 0 ? xr - xr * (s1 + s2) : (hx < 0 ? -zz : zz);
    if (ix < 0x31000000) res = x;
    if (ix >= 0x4c000000) {
        float big = fd_u2f(0x3fc90fdau) + fd_u2f(0x33a22168u);
        res = ix > 0x7f800000 ? x + x : (hx > 0 ? big : -big);
    }
    return res;
}

// atan2f with the generic path branch-free; special operands fall back to fd_atan2f's logic.
FD_HD float fdd_atan2f(float y, float x) {
    const float pi = fd_u2f(0x40490fdbu), pi_lo = fd_u2f(0xb3bbbd2eu), pi_o_2 = fd_u2f(0x3fc90fdbu);
    int32_t hx = (int32_t)fd_f2u(x), hy = (int32_t)fd_f2u(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    // anything unusual (NaN, inf, zero operand, x == 1.0): exact slow path
    if (ix >= 0x7f800000 || iy >= 0x7f800000 || ix == 0 || iy == 0 || hx == 0x3f800000) return fd_atan2f(y, x);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    int32_t k = (iy - ix) >> 23;
    float z = fdd_atanf(fd_fabsf(y / x));
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    float r0 = z;
    float r1 = fd_u2f(fd_f2u(z) ^ 0x80000000u);
    float r2 = pi - (z - pi_lo);
    float r3 = (z - pi_lo) - pi;
    return m == 0 ? r0 : (m == 1 ? r1 : (m == 2 ? r2 : r3));
}
