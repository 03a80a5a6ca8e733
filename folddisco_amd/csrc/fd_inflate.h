// fd_inflate.h — gzip decoder of the structure ingest (fd_inflate.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

// every gzip member of in[0 .. n) concatenated into *out.  -> false when the input is not what the decoder expects (not gzip, damaged, a code it
// does not handle): the caller then reads the file through zlib.
bool fd_gunzip(const uint8_t *in, size_t n, std::string *out);
