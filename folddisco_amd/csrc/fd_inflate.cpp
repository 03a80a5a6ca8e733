// fd_inflate.cpp — gzip members -> bytes, for the structure ingest (fd_ingest.cpp).
//
// Why not zlib: `.pdb.gz` ingest is bound by inflate (16 of 21 thread-seconds per 20,500 files with zlib 1.2.11's byte-at-a-time inner loop, 164-187
// MB/s per thread).  This decoder is written from RFC 1951 / RFC 1952 for the case the ingest has — the whole compressed file in memory, the output
// size known from the trailer — with the usual fast-path ingredients: a 64-bit bit buffer refilled eight bytes at a time, two-level decode tables
// (10 bits for literals / lengths, 9 for distances) whose entries carry base value, extra-bit count and the symbol's TOTAL bit count, word-wise match
// copies into a buffer with slack.  PDB text decodes as ~99 % short matches (average length 8: the previous line's columns), so the loop is shaped for the
// chain lookup -> shift -> lookup of a match: one shift per symbol (extra bits are read from the buffer as it was before the shift), the table index taken
// from the bits left over before the refill's load arrives, thirty-two bytes copied without a length test (round 5: 620 -> 880 MB/s per thread here).
// Anything it does not expect (reserved block type, over-subscribed or incomplete code, distance beyond the output, CRC / size mismatch,
// truncated input) makes it return false and the caller falls back to zlib, which then reports the file the way it always did.
// The reference reads gzip through the flate2 crate (src/structure/io/pdb.rs:79-124); a decoder's output is defined by the format.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <mutex>
#include <string>
#include <vector>

#include "fd_inflate.h"

namespace {

constexpr int LB = 10, DB = 9;        // primary table bits: literal / length codes, distance codes
// table entry: bits 0-1 kind, 2-6 bits to consume = code + extra bits, 7-11 extra bits, 12-31 value (literal byte / base length / base distance / subtable
// offset).  ONE shift per symbol: the extra bits come out of the bit buffer as it was before the shift, off the chain lookup -> shift -> lookup
enum : uint32_t { K_LIT = 0, K_BASE = 1, K_END = 2, K_SUB = 3 };
inline uint32_t mk(uint32_t kind, uint32_t nbits, uint32_t extra, uint32_t value) { return kind | ((nbits + (kind == K_SUB ? 0u : extra)) << 2) | (extra << 7) | (value << 12); }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t rev_bits(uint32_t c, int n) { return __builtin_bitreverse32(c) >> (32 - n); }      // n >= 1

struct Table {
    std::vector<uint32_t> e;
    int pbits = 0;
};

// canonical Huffman code of lens[0 .. n) -> two-level table.  kind_of(sym) gives the entry's payload.  A COMPLETE code is required, except the
// one-code distance alphabet RFC 1951 allows (a single code of length 1; its unused half decodes to an invalid entry).  -> false: not decodable here
template <typename Payload>
bool build_table(const uint8_t *lens, int n, int pbits, Table &T, bool allow_single, Payload payload) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int used = 0, max_len = 0;
    for (int l = 1; l <= 15; ++l) if (count[l]) { used += count[l]; max_len = l; }
    if (used == 0) return false;
    long left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
    if (left != 0 && !(allow_single && used == 1 && count[1] == 1)) return false;
    uint32_t next[16];
    {
        uint32_t code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        next[1] = 0;      // (count[0] was zeroed: the recurrence above already gives 0 for l = 1)
    }
    // first pass over the long codes: how many bits the subtable of every primary prefix needs
    T.pbits = pbits;
    const uint32_t psize = 1u << pbits;
    static thread_local std::vector<uint8_t> sub_bits;      // (scratch kept per thread: a dynamic block every ~25 KB of input, two tables each)
    static thread_local std::vector<uint32_t> sub_off;
    sub_bits.clear(); sub_off.clear();
    uint32_t next_l[16];
    memcpy(next_l, next, sizeof next);
    if (max_len > pbits) {
        sub_bits.assign(psize, 0);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = rev_bits(next_l[l]++, l);
            if (l > pbits) { const uint32_t p = r & (psize - 1); if (sub_bits[p] < l - pbits) sub_bits[p] = (uint8_t)(l - pbits); }
        }
    }
    size_t total = psize;
    if (!sub_bits.empty()) {
        sub_off.assign(psize, 0);
        for (uint32_t p = 0; p < psize; ++p) if (sub_bits[p]) { sub_off[p] = (uint32_t)total; total += (size_t)1 << sub_bits[p]; }
    }
    T.e.assign(total, 0xffffffffu);       // 0xffffffff: no code (only reachable through the unused half of a one-code distance alphabet)
    if (!sub_bits.empty())
        for (uint32_t p = 0; p < psize; ++p) if (sub_bits[p]) T.e[p] = mk(K_SUB, (uint32_t)pbits, sub_bits[p], sub_off[p]);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev_bits(next[l]++, l);
        uint32_t kind, extra, value;
        if (!payload(s, &kind, &extra, &value)) return false;
        if (l <= pbits) {
            const uint32_t ent = mk(kind, (uint32_t)l, extra, value);
            for (uint32_t k = r; k < psize; k += 1u << l) T.e[k] = ent;
        } else {
            const uint32_t p = r & (psize - 1), sb = sub_bits[p], hi = r >> pbits, ent = mk(kind, (uint32_t)(l - pbits), extra, value);
            for (uint32_t k = hi; k < (1u << sb); k += 1u << (l - pbits)) T.e[sub_off[p] + k] = ent;
        }
    }
    return true;
}

bool litlen_payload(int s, uint32_t *kind, uint32_t *extra, uint32_t *value) {
    if (s < 256) { *kind = K_LIT; *extra = 0; *value = (uint32_t)s; return true; }
    if (s == 256) { *kind = K_END; *extra = 0; *value = 0; return true; }
    if (s > 285) return false;
    *kind = K_BASE; *extra = LEN_EXTRA[s - 257]; *value = LEN_BASE[s - 257];
    return true;
}
bool dist_payload(int s, uint32_t *kind, uint32_t *extra, uint32_t *value) {
    if (s > 29) return false;
    *kind = K_BASE; *extra = DIST_EXTRA[s]; *value = DIST_BASE[s];
    return true;
}

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int cnt = 0;        // valid bits in buf
    bool over = false;  // a read past the end of the input happened
    inline void refill() {
        if (p + 8 <= end) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(int n) { if (n > cnt) { over = true; cnt = 0; buf = 0; } else { buf >>= n; cnt -= n; } }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    // the symbol loop's forms: no test per field — it looks at cnt once per symbol (a negative count = the input ended inside the symbol)
    inline void skip(int n) { buf >>= n; cnt -= n; }
    inline uint32_t grab(int n) { const uint32_t v = peek(n); skip(n); return v; }
    // position of the next unread byte after dropping the bits up to the next byte boundary
    inline const uint8_t *byte_pos() { drop(cnt & 7); return p - (cnt >> 3); }
};

Table g_fixed_lit, g_fixed_dist;
std::once_flag g_fixed_once;
void make_fixed() {
    uint8_t l[288];
    for (int s = 0; s < 144; ++s) l[s] = 8;
    for (int s = 144; s < 256; ++s) l[s] = 9;
    for (int s = 256; s < 280; ++s) l[s] = 7;
    for (int s = 280; s < 288; ++s) l[s] = 8;
    // symbols 286 / 287 take part in the fixed code's construction but never occur: give them an END-less invalid payload
    build_table(l, 288, LB, g_fixed_lit, false, [](int s, uint32_t *k, uint32_t *e, uint32_t *v) {
        if (s > 285) { *k = K_SUB; *e = 0; *v = 0xfffffu; return true; }      // decodes as "invalid" below (a subtable pointer with 0 bits is never made otherwise)
        return litlen_payload(s, k, e, v);
    });
    uint8_t d[32];
    for (int s = 0; s < 32; ++s) d[s] = 5;
    build_table(d, 32, DB, g_fixed_dist, false, [](int s, uint32_t *k, uint32_t *e, uint32_t *v) {
        if (s > 29) { *k = K_SUB; *e = 0; *v = 0xfffffu; return true; }
        return dist_payload(s, k, e, v);
    });
}

// one deflate stream from B into out (appended); -> false on anything unexpected
// one block's symbols: -> 0 at the end-of-block code, 1 when the output is within 320 bytes of its limit (the caller grows the buffer and calls again:
// the state is in B and o), -1 on anything unexpected.  out0 = first byte of this member's output (matches do not reach behind it).
__attribute__((noinline)) int decode_symbols(const uint32_t *__restrict__ L, const uint32_t *__restrict__ D, Bits &Bref, uint8_t *&oref, const uint8_t *lim, const uint8_t *out0) {
    Bits B = Bref;
    uint8_t *o = oref;
    int ret;
    for (;;) {
        if (o > lim) { ret = 1; break; }
        // the symbol's table index from the bits that are LEFT when they suffice (nearly always: a match takes ~20 of the >= 56 bits of a refill):
        // the refill's load then feeds only the upper bits and stays off the chain lookup -> shift -> lookup
        uint32_t idx;
        if (__builtin_expect(B.cnt >= LB, 1)) { idx = B.peek(LB); B.refill(); } else { B.refill(); idx = B.peek(LB); }
        uint32_t e = L[idx];
        if ((e & 3u) == K_SUB) {
            if (e == 0xffffffffu || ((e >> 7) & 31u) == 0) { ret = -1; goto done; }
            B.skip(LB);
            e = L[(e >> 12) + B.peek((int)((e >> 7) & 31u))];
            if (e == 0xffffffffu || (e & 3u) == K_SUB) { ret = -1; goto done; }
        }
        uint64_t saved = B.buf;
        B.skip((int)((e >> 2) & 31u));
        if ((e & 3u) == K_LIT) {
            *o++ = (uint8_t)(e >> 12);
            // up to two more literals on the bits that are left (a literal / length code is at most 15 bits: 3 x 15 <= 56)
            uint32_t e2 = L[B.peek(LB)];
            if ((e2 & 3u) == K_LIT) {
                B.skip((int)((e2 >> 2) & 31u));
                *o++ = (uint8_t)(e2 >> 12);
                e2 = L[B.peek(LB)];
                if ((e2 & 3u) == K_LIT) { B.skip((int)((e2 >> 2) & 31u)); *o++ = (uint8_t)(e2 >> 12); }
            }
            if (B.cnt < 0) { ret = -1; goto done; }
            continue;
        }
        if ((e & 3u) == K_END) { ret = B.cnt < 0 ? -1 : 0; break; }
        // length (code <= 15 bits + <= 5 extra), distance (<= 15 + <= 13): 48 <= 56 bits since the refill.  The extra bits are read from the buffer as it
        // was before the symbol's one shift
        const uint32_t l_ex = (e >> 7) & 31u;
        const uint32_t len = (e >> 12) + ((uint32_t)(saved >> (((e >> 2) & 31u) - l_ex)) & ((1u << l_ex) - 1u));
        uint32_t d = D[B.peek(DB)];
        if ((d & 3u) == K_SUB) {
            if (d == 0xffffffffu || ((d >> 7) & 31u) == 0) { ret = -1; goto done; }
            B.skip(DB);
            d = D[(d >> 12) + B.peek((int)((d >> 7) & 31u))];
            if (d == 0xffffffffu || (d & 3u) == K_SUB) { ret = -1; goto done; }
        }
        saved = B.buf;
        B.skip((int)((d >> 2) & 31u));
        const uint32_t d_ex = (d >> 7) & 31u;
        const uint32_t dist = (d >> 12) + ((uint32_t)(saved >> (((d >> 2) & 31u) - d_ex)) & ((1u << d_ex) - 1u));
        if (B.cnt < 0 || dist > (size_t)(o - out0)) { ret = -1; goto done; }      // (matches never reach into an earlier member)
        const uint8_t *src = o - dist;
        uint8_t *dst = o;
        o += len;
        if (dist >= 8) {
            // words of eight: the source stays at least eight bytes behind the destination; up to 29 bytes past the match are written (and overwritten
            // later: the buffer keeps 320 bytes of slack).  Four words without a test — half the matches of PDB text are longer than eight bytes, a
            // fifth longer than sixteen: a coin flip per match for the branch predictor; a word may read what the one before it wrote (program order)
            { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); memcpy(&w, src + 8, 8); memcpy(dst + 8, &w, 8);
              memcpy(&w, src + 16, 8); memcpy(dst + 16, &w, 8); memcpy(&w, src + 24, 8); memcpy(dst + 24, &w, 8); }
            if (len > 32) {
                src += 32; dst += 32;
                do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < o);
            }
        } else if (dist == 1) {
            memset(dst, *src, len);
        } else {
            do { *dst++ = *src++; } while (dst < o);
        }
    }
done:
    Bref = B;
    oref = o;
    return ret;
}

// `start` = bytes of out that are output already (earlier members); what out holds behind them is scratch: the string is only ever GROWN here (a
// std::string zero-fills what resize adds — the caller reuses one string per thread, so after the first files nothing is filled) and cut to the
// produced length by the caller at the very end; *end = start + the bytes this stream produced
bool inflate_stream(Bits &B, std::string &out, size_t start, size_t expect, size_t *end) {
    size_t cap = start + (expect ? expect : (size_t)(B.end - B.p) * 4) + 1024;
    if (out.size() < cap) out.resize(cap); else cap = out.size();
    uint8_t *base = (uint8_t *)&out[0], *o = base + start, *lim = base + cap - 320;      // 258 of a match + a word of over-copy + slack
    auto grow = [&]() {
        const size_t at = (size_t)(o - base);
        cap = cap + cap / 2 + 65536;
        out.resize(cap);      // (keeps the bytes written so far)
        base = (uint8_t *)&out[0]; o = base + at; lim = base + cap - 320;
    };
    Table dyn_lit, dyn_dist;
    for (;;) {
        B.refill();
        const uint32_t final = B.take(1), type = B.take(2);
        if (B.over) return false;
        if (type == 0) {
            const uint8_t *q = B.byte_pos();
            if (B.over || q + 4 > B.end) return false;
            const uint32_t len = q[0] | (q[1] << 8), nlen = q[2] | (q[3] << 8);
            if ((len ^ 0xffffu) != nlen) return false;
            q += 4;
            if (q + len > B.end) return false;
            while ((size_t)(lim + 320 - o) < len + 320) grow();
            memcpy(o, q, len);
            o += len;
            B.p = q + len; B.buf = 0; B.cnt = 0;
        } else if (type == 1 || type == 2) {
            const Table *TL, *TD;
            if (type == 1) {
                std::call_once(g_fixed_once, make_fixed);
                TL = &g_fixed_lit; TD = &g_fixed_dist;
            } else {
                B.refill();
                const uint32_t hlit = B.take(5) + 257, hdist = B.take(5) + 1, hclen = B.take(4) + 4;
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t pl[19] = {0};
                for (uint32_t k = 0; k < hclen; ++k) { B.refill(); pl[order[k]] = (uint8_t)B.take(3); }
                if (B.over) return false;
                Table pre;
                if (!build_table(pl, 19, 7, pre, false, [](int s, uint32_t *k, uint32_t *e, uint32_t *v) { *k = K_LIT; *e = 0; *v = (uint32_t)s; return true; })) return false;
                uint8_t lens[286 + 30];
                uint32_t n = 0;
                while (n < hlit + hdist) {
                    B.refill();
                    const uint32_t e = pre.e[B.peek(7)];
                    if (e == 0xffffffffu) return false;
                    B.drop((int)((e >> 2) & 31u));
                    const uint32_t s = e >> 12;
                    if (s < 16) { lens[n++] = (uint8_t)s; continue; }
                    uint32_t rep, val = 0;
                    if (s == 16) { if (!n) return false; val = lens[n - 1]; rep = 3 + B.take(2); }
                    else if (s == 17) rep = 3 + B.take(3);
                    else rep = 11 + B.take(7);
                    if (n + rep > hlit + hdist) return false;
                    memset(lens + n, (int)val, rep);
                    n += rep;
                }
                if (B.over || lens[256] == 0) return false;
                if (!build_table(lens, (int)hlit, LB, dyn_lit, false, litlen_payload)) return false;
                // a block of literals only may send one distance code of zero length (RFC 1951 3.2.7): no table, any match is an error
                bool any_dist = false;
                for (uint32_t k = 0; k < hdist; ++k) any_dist = any_dist || lens[hlit + k];
                if (any_dist) { if (!build_table(lens + hlit, (int)hdist, DB, dyn_dist, true, dist_payload)) return false; }
                else { dyn_dist.e.assign((size_t)1 << DB, 0xffffffffu); dyn_dist.pbits = DB; }
                TL = &dyn_lit; TD = &dyn_dist;
            }
            // the symbols of the block (decode_symbols, its own function: as part of this one its bit count and input pointers lived on the stack);
            // 1 = the output buffer is nearly full: grown here, continued
            for (;;) {
                const int rc = decode_symbols(TL->e.data(), TD->e.data(), B, o, lim, base + start);
                if (rc == 0) break;
                if (rc < 0) return false;
                grow();
            }
        } else return false;
        if (final) break;
    }
    *end = (size_t)(o - base);
    return true;
}

// CRC-32 of RFC 1952 (reflected polynomial 0xedb88320), sixteen bytes per step through sixteen tables made on first use (zlib 1.2.11's
// four-table loop runs at 1 GB/s: a third of this decoder's time)
uint32_t g_crc_tab[16][256];
std::once_flag g_crc_once;
void make_crc() {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t c = b;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        g_crc_tab[0][b] = c;
    }
    for (uint32_t b = 0; b < 256; ++b)
        for (int t = 1; t < 16; ++t) g_crc_tab[t][b] = (g_crc_tab[t - 1][b] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][b] & 0xffu];
}
#if defined(__x86_64__)
// The same CRC by carry-less multiplication (PCLMULQDQ): four 128-bit lanes folded per 64 input bytes, then 4 -> 1, 128 -> 64 -> 32 bits and a Barrett
// reduction, with the folding constants of the reflected gzip polynomial (x^(512+64), x^512, x^(128+64), x^128, x^96 mod P; mu and P for the reduction —
// Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction").  ~5x the sixteen-table loop; the tables finish the tail.
// c = the running CRC register (pre-inverted), n >= 64.  -> the register after the largest multiple of 16 bytes, *done = bytes consumed.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(const uint8_t *buf, size_t n, uint32_t c, size_t *done) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const uint8_t *p = buf;
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64; n -= 64;
    while (n >= 64) {
        const __m128i y1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), y2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i y3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), y4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, y1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, y2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, y3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, y4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
#define FD_CRC_FOLD(a, b) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128((a), k3k4, 0x11), _mm_clmulepi64_si128((a), k3k4, 0x00)), (b))
    x1 = FD_CRC_FOLD(x1, x2); x1 = FD_CRC_FOLD(x1, x3); x1 = FD_CRC_FOLD(x1, x4);
    while (n >= 16) { x1 = FD_CRC_FOLD(x1, _mm_loadu_si128((const __m128i *)p)); p += 16; n -= 16; }
#undef FD_CRC_FOLD
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    // 64 -> 32 bits
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5k0, 0x00), t);
    // Barrett reduction
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    *done = (size_t)(p - buf);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
uint32_t crc32_fast(const uint8_t *p, size_t n) {
    std::call_once(g_crc_once, make_crc);
    uint32_t c = 0xffffffffu;
#if defined(__x86_64__)
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("FDGPU_CRC_TABLES");
    if (have_clmul && n >= 64) { size_t done = 0; c = crc32_clmul(p, n, c, &done); p += done; n -= done; }
#endif
    const uint32_t(*T)[256] = g_crc_tab;
    while (n >= 16) {
        uint32_t a, b, d, e;
        memcpy(&a, p, 4); memcpy(&b, p + 4, 4); memcpy(&d, p + 8, 4); memcpy(&e, p + 12, 4);
        a ^= c;
        c = T[15][a & 0xffu] ^ T[14][(a >> 8) & 0xffu] ^ T[13][(a >> 16) & 0xffu] ^ T[12][a >> 24] ^
            T[11][b & 0xffu] ^ T[10][(b >> 8) & 0xffu] ^ T[9][(b >> 16) & 0xffu] ^ T[8][b >> 24] ^
            T[7][d & 0xffu] ^ T[6][(d >> 8) & 0xffu] ^ T[5][(d >> 16) & 0xffu] ^ T[4][d >> 24] ^
            T[3][e & 0xffu] ^ T[2][(e >> 8) & 0xffu] ^ T[1][(e >> 16) & 0xffu] ^ T[0][e >> 24];
        p += 16; n -= 16;
    }
    while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu];
    return ~c;
}

}  // namespace

bool fd_gunzip(const uint8_t *in, size_t n, std::string *out) {
    size_t at = 0, have = 0;      // have: output bytes so far (out itself is cut to it on return)
    struct Cut { std::string *s; size_t *n; bool ok = false; ~Cut() { s->resize(ok ? *n : 0); } } cut{out, &have};
    bool any = false;
    while (at + 18 <= n && in[at] == 0x1f && in[at + 1] == 0x8b) {
        if (in[at + 2] != 8) return false;
        const uint8_t flg = in[at + 3];
        if (flg & 0xe0) return false;
        size_t h = at + 10;
        if (flg & 4) { if (h + 2 > n) return false; const size_t xl = in[h] | (in[h + 1] << 8); h += 2 + xl; }
        if (flg & 8) { while (h < n && in[h]) ++h; ++h; }
        if (flg & 16) { while (h < n && in[h]) ++h; ++h; }
        if (flg & 2) h += 2;
        if (h + 8 > n) return false;
        // the size a single-member file's trailer announces (mod 2^32) sizes the output buffer; a wrong guess only costs a reallocation
        const size_t tail = n - 4;
        const uint32_t isize_guess = (uint32_t)in[tail] | ((uint32_t)in[tail + 1] << 8) | ((uint32_t)in[tail + 2] << 16) | ((uint32_t)in[tail + 3] << 24);
        Bits B;
        B.p = in + h; B.end = in + n;
        const size_t before = have;
        // (deflate cannot expand beyond 1032 : 1: a trailer that claims more is damaged, and must not size an allocation)
        const size_t max_out = (n - h) * 1032 + 1024;
        if (!inflate_stream(B, *out, before, any ? 0 : (isize_guess < max_out ? isize_guess : max_out), &have)) return false;
        const uint8_t *q = B.byte_pos();
        if (B.over || q + 8 > in + n) return false;
        const uint32_t crc = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        const uint32_t isz = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
        const size_t produced = have - before;
        if ((uint32_t)produced != isz) return false;
        if (crc32_fast((const uint8_t *)out->data() + before, produced) != crc) return false;
        at = (size_t)(q + 8 - in);
        any = true;
    }
    cut.ok = any;
    return any;       // (bytes behind the last member that do not start another one are ignored, as zlib's gzread ignores them)
}
