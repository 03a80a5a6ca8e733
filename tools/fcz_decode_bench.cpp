// tools/fcz_decode_bench.cpp — one thread of the Foldcomp decoder (csrc/fd_fcz.cpp) over the entries of tests/golden/foldcomp/example_db.
//   hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math [-DFD_FCZ_PIPE=0] -Ifolddisco_amd/csrc -x c++ folddisco_amd/csrc/fd_fcz.cpp -x c++ tools/fcz_decode_bench.cpp -o /tmp/fcz_bench
//   /tmp/fcz_bench tests/golden/foldcomp/example_db [repetitions]
// Prints microseconds per entry, nanoseconds per residue and a hash of every coordinate's bits (the same for every build of the decoder).
#include "fd_fcz.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s DB [repetitions]\n", argv[0]); return 2; }
    const std::string db = argv[1];
    std::ifstream f(db, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string data = ss.str();
    std::ifstream fi(db + ".index");
    std::vector<std::pair<size_t, size_t>> ents;
    size_t k, s, l;
    while (fi >> k >> s >> l) ents.push_back({s, l});
    if (ents.empty()) { fprintf(stderr, "no entries in %s.index\n", db.c_str()); return 1; }
    std::vector<fd_fcz_atom> out;
    unsigned long long h = 0;
    size_t natoms = 0, nres = 0;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (auto &e : ents) {
            const int rc = fd_fcz_decode((const uint8_t *)data.data() + e.first, e.second, &out);
            if (rc) { printf("entry at %zu: rc %d\n", e.first, rc); return 1; }
            natoms += out.size();
            for (auto &a : out) {
                uint32_t b[3];
                memcpy(b, &a.x, 12);
                h = h * 1000003ull + b[0] + 7 * b[1] + 13 * b[2];
                if (!memcmp(a.name, " CA ", 4)) ++nres;
            }
        }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%.1f us per entry, %.1f ns per residue, %zu atoms, coordinate hash %016llx\n", us / ((double)reps * ents.size()), us * 1e3 / (double)nres, natoms, h);
    return 0;
}
