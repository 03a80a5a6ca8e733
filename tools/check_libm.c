// tools/check_libm.c — compares folddisco_amd/csrc/fd_libm.h (compiled for the
// host, -ffp-contract=off) against this machine's glibc.  One-argument
// functions are swept exhaustively over all 2^32 float bit patterns (sinf/cosf
// over |x| < 120, the range the restatement covers); atan2f over N random pairs
// plus a grid of special values.  Exit code 0 iff zero mismatches.
//   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/check_libm tools/check_libm.c -lm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../folddisco_amd/csrc/fd_libm.h"

static int same(float a, float b) {
    if (a != a && b != b) return 1;  // any NaN == any NaN
    return fd_f2u(a) == fd_f2u(b);
}
static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int main(int argc, char **argv) {
    uint64_t step = argc > 1 ? strtoull(argv[1], 0, 10) : 1;  // stride over bit patterns
    uint64_t npairs = argc > 2 ? strtoull(argv[2], 0, 10) : 2000000000ull;
    unsigned long long bad_sin = 0, bad_cos = 0, bad_acos = 0, bad_atan = 0, bad_atan2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad_sin, bad_cos, bad_acos, bad_atan)
    for (uint64_t u = 0; u < (1ull << 32); u += step) {
        float x = fd_u2f((uint32_t)u);
        if (!same(fdd_acosf(x), acosf(x))) { if (bad_acos++ < 3) fprintf(stderr, "fdd_acosf %08x\n", (unsigned)u); }
        if (!same(fdd_atanf(x), atanf(x))) { if (bad_atan++ < 3) fprintf(stderr, "fdd_atanf %08x\n", (unsigned)u); }
        if (fabsf(x) < 120.0f || x != x) {
            float ss, cc; fdd_sincosf(x, &ss, &cc);
            if (!same(ss, sinf(x))) { if (bad_sin++ < 3) fprintf(stderr, "fdd_sin %08x\n", (unsigned)u); }
            if (!same(cc, cosf(x))) { if (bad_cos++ < 3) fprintf(stderr, "fdd_cos %08x\n", (unsigned)u); }
        }
        if (!same(fd_acosf(x), acosf(x))) { if (bad_acos++ < 3) fprintf(stderr, "acosf %08x\n", (unsigned)u); }
        if (!same(fd_atanf(x), atanf(x))) { if (bad_atan++ < 3) fprintf(stderr, "atanf %08x\n", (unsigned)u); }
        if (fabsf(x) < 120.0f) {
            if (!same(fd_sinf(x), sinf(x))) { if (bad_sin++ < 3) fprintf(stderr, "sinf %08x\n", (unsigned)u); }
            if (!same(fd_cosf(x), cosf(x))) { if (bad_cos++ < 3) fprintf(stderr, "cosf %08x\n", (unsigned)u); }
        }
    }
    // atan2f: random bit patterns, random "geometric" magnitudes, and specials
#pragma omp parallel reduction(+ : bad_atan2)
    {
        uint64_t seed = 0x1234567ull;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        seed += 7919ull * (uint64_t)omp_get_thread_num();
#endif
#pragma omp for schedule(static)
        for (uint64_t i = 0; i < npairs; ++i) {
            uint64_t r = splitmix(&seed);
            float y, x;
            if (i & 1) { y = fd_u2f((uint32_t)r); x = fd_u2f((uint32_t)(r >> 32)); }
            else {  // values in [-1,1] like the dot products the torsion feeds
                y = (float)((double)(int32_t)(uint32_t)r / 2147483648.0);
                x = (float)((double)(int32_t)(uint32_t)(r >> 32) / 2147483648.0);
            }
            if (!same(fd_atan2f(y, x), atan2f(y, x))) {
                if (bad_atan2++ < 3) fprintf(stderr, "atan2f %08x %08x\n", fd_f2u(y), fd_f2u(x));
            }
            if (!same(fdd_atan2f(y, x), atan2f(y, x))) {
                if (bad_atan2++ < 3) fprintf(stderr, "fdd_atan2f %08x %08x\n", fd_f2u(y), fd_f2u(x));
            }
        }
    }
    static const float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e-30f, -1e-30f, 1e30f, -1e30f,
                               0.5f, -0.5f, 2.f, 1e-45f, 3.4e38f};
    for (unsigned a = 0; a < sizeof sp / 4; ++a)
        for (unsigned b = 0; b < sizeof sp / 4; ++b)
            if (!same(fd_atan2f(sp[a], sp[b]), atan2f(sp[a], sp[b])) || !same(fdd_atan2f(sp[a], sp[b]), atan2f(sp[a], sp[b]))) {
                bad_atan2++; fprintf(stderr, "atan2f special %g %g\n", sp[a], sp[b]);
            }
    printf("mismatches: sinf %llu cosf %llu acosf %llu atanf %llu atan2f %llu\n", bad_sin, bad_cos, bad_acos,
           bad_atan, bad_atan2);
    return (bad_sin | bad_cos | bad_acos | bad_atan | bad_atan2) ? 1 : 0;
}
