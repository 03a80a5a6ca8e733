// tools/check_geom.cpp — host check that the shared-subexpression pair evaluation of
// folddisco_amd/csrc/fd_geom.h (fd_make_frame + fd_pair_both) is bit-identical to the straightforward
// per-pair evaluation (fd_pair_feature + fd_hash_pdbtr) and to the CPU oracle (glibc libm), on the
// reference's serine_peptidases fixtures and on random near-degenerate geometry.
//   g++ -O2 -ffp-contract=off -I. tools/check_geom.cpp -Loracle -lfdoracle -Wl,-rpath,$PWD/oracle -o /tmp/check_geom
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../folddisco_amd/csrc/fd_geom.h"
#include "../oracle/fd_oracle.h"

static fd_v3 at(const float *p, long i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

static long check_structure(const fdo_structure *s, fd_quant q, float cutoff) {
    std::vector<fd_frame> F(s->n);
    for (int i = 0; i < s->n; ++i) F[i] = fd_make_frame(at(s->n_xyz, i), at(s->ca_xyz, i), at(s->cb_xyz, i));
    long bad = 0, n = 0;
    float feat[9];
    for (int i = 0; i < s->n; ++i)
        for (int j = i + 1; j < s->n; ++j) {
            if (!fdo_pair_feature(s, i, j, cutoff, feat)) continue;
            uint32_t o_ij = fdo_hash_pdbtr(feat, 16, 4);
            fdo_pair_feature(s, j, i, cutoff, feat);
            uint32_t o_ji = fdo_hash_pdbtr(feat, 16, 4);
            uint32_t h_ij, h_ji;
            fd_pair_both(F[i], F[j], s->aa[i], s->aa[j], q, &h_ij, &h_ji);
            uint32_t t_ij, t_ji, tab[FD_BINTAB_WORDS];
            fd_fill_bintab(tab);
            fd_pair_both_tab(F[i], F[j], s->aa[i], s->aa[j], q, tab, &t_ij, &t_ji);
            if (t_ij != o_ij || t_ji != o_ji) { if (bad++ < 5) fprintf(stderr, "table mismatch (%d,%d): %08x/%08x oracle %08x/%08x\n", i, j, t_ij, t_ji, o_ij, o_ji); }
            fd_feature f = fd_pair_feature(at(s->n_xyz, i), at(s->ca_xyz, i), at(s->cb_xyz, i), at(s->n_xyz, j), at(s->ca_xyz, j), at(s->cb_xyz, j));
            uint32_t d_ij = fd_hash_pdbtr(s->aa[i], s->aa[j], f, q);
            ++n;
            if (h_ij != o_ij || h_ji != o_ji || d_ij != o_ij) {
                if (bad++ < 5) fprintf(stderr, "mismatch (%d,%d): both %08x/%08x direct %08x oracle %08x/%08x\n", i, j, h_ij, h_ji, d_ij, o_ij, o_ji);
            }
        }
    printf("  %ld unordered pairs checked, %ld mismatches\n", n, bad);
    return bad;
}

int main(int argc, char **argv) {
    fd_quant q;
    volatile float cd = (20.0f - 2.0f) / (16.0f - 1.0f), ca = 2.0f / 3.0f;
    q.dist_disc = 1.0f / cd;
    q.ang_disc = 1.0f / ca;
    q.ang2_disc = 0.0f;
    q.type = FD_HASH_PDBTR;
    long bad = 0;
    for (int a = 1; a < argc; ++a) {
        fdo_structure *s = fdo_read_pdb(argv[a]);
        if (!s) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        printf("%s (%d residues)\n", argv[a], s->n);
        bad += check_structure(s, q, 20.0f);
        fdo_structure_free(s);
    }
    // random compact clouds incl. collinear / coincident atoms (NaN paths)
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-12.f, 12.f);
    for (int rep = 0; rep < 200; ++rep) {
        int n = 60;
        std::vector<float> N(3 * n), CA(3 * n), CB(3 * n);
        std::vector<uint8_t> aa(n), ok(n, 1);
        for (int i = 0; i < n; ++i) {
            for (int c = 0; c < 3; ++c) {
                CA[3 * i + c] = std::round(U(rng) * 1000.f) / 1000.f;
                N[3 * i + c] = CA[3 * i + c] + std::round(U(rng) * 100.f) / 1000.f;
                CB[3 * i + c] = CA[3 * i + c] + std::round(U(rng) * 120.f) / 1000.f;
            }
            aa[i] = (uint8_t)(rng() % 20);
            if (rep % 10 == 0 && i % 7 == 0) { for (int c = 0; c < 3; ++c) CB[3 * i + c] = CA[3 * i + c]; }       // zero-length CA->CB
            if (rep % 10 == 1 && i % 5 == 0) { for (int c = 0; c < 3; ++c) N[3 * i + c] = 2 * CA[3 * i + c] - CB[3 * i + c]; } // collinear
        }
        fdo_structure *s = fdo_structure_from_packed(n, N.data(), CA.data(), CB.data(), ok.data(), aa.data(), nullptr);
        if (rep < 3 || rep % 50 == 0) printf("random cloud %d\n", rep);
        long b = 0;
        {
            std::vector<fd_frame> F(n);
            for (int i = 0; i < n; ++i) F[i] = fd_make_frame(at(s->n_xyz, i), at(s->ca_xyz, i), at(s->cb_xyz, i));
            float feat[9];
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j) {
                    if (!fdo_pair_feature(s, i, j, 20.0f, feat)) continue;
                    uint32_t o_ij = fdo_hash_pdbtr(feat, 16, 4);
                    fdo_pair_feature(s, j, i, 20.0f, feat);
                    uint32_t o_ji = fdo_hash_pdbtr(feat, 16, 4);
                    uint32_t h_ij, h_ji;
                    fd_pair_both(F[i], F[j], s->aa[i], s->aa[j], q, &h_ij, &h_ji);
                    uint32_t t_ij, t_ji, tab[FD_BINTAB_WORDS];
                    fd_fill_bintab(tab);
                    fd_pair_both_tab(F[i], F[j], s->aa[i], s->aa[j], q, tab, &t_ij, &t_ji);
                    if (t_ij != o_ij || t_ji != o_ji) { if (b++ < 3) fprintf(stderr, "cloud %d table (%d,%d): %08x/%08x vs %08x/%08x\n", rep, i, j, t_ij, t_ji, o_ij, o_ji); }
                    if (h_ij != o_ij || h_ji != o_ji) { if (b++ < 3) fprintf(stderr, "cloud %d (%d,%d): %08x/%08x vs %08x/%08x\n", rep, i, j, h_ij, h_ji, o_ij, o_ji); }
                }
        }
        bad += b;
        fdo_structure_free(s);
    }
    printf("total mismatches: %ld\n", bad);
    return bad ? 1 : 0;
}
