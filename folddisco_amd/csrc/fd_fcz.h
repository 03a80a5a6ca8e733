// fd_fcz.h — Foldcomp entry decoder (fd_fcz.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

struct fd_fcz_atom {     // one atom record as the reference's Structure::update receives it (foldcompffi.h atom_t)
    float x, y, z, b;
    char name[4], res[3];
    uint8_t chain;
    uint64_t rser;
};
// 0 = ok; -1 malformed entry; -2 a residue type without geometry (ASX / GLX / STP / UNK)
int fd_fcz_decode(const uint8_t *data, size_t len, std::vector<fd_fcz_atom> *out);
