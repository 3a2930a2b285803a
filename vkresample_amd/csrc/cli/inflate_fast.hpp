// inflate_fast.hpp -- a zlib-stream decoder for the CLI's PNG reader (RFC 1950 / 1951).
//
// With the PNG encoded on the GPU (-gpupng) a batch thread spends four fifths of its time per file in zlib's inflate.  This
// decoder does the same job about twice as fast on image data -- a 64-bit bit buffer refilled eight bytes at a time, one table
// lookup per symbol (11-bit primary table for literal/length codes, 8-bit for distances, second-level tables for longer codes),
// eight-byte match copies -- and is an ACCELERATOR, not an authority: it handles well-formed streams and returns false for
// everything it does not like (bad header, over-subscribed or incomplete code, distance beyond the output, truncated input,
// wrong length, wrong Adler-32), in which case the caller asks zlib, whose verdict counts.  Every read and write is bounds
// checked; tests/sanitize/inflate_driver.cpp runs it under ASan/UBSan against zlib on streams of every level and strategy,
// truncated and bit-flipped ones included.
#pragma once
#include <cstdint>
#include <cstring>

namespace inflate_fast {

struct Entry { uint16_t val; uint8_t bits; uint8_t op; };     // op: 0 literal, 1 length/distance base (+ extra bits in the high
                                                              // nibble of op), 2 end of block, 3 link to a second-level table, 4 invalid
enum { OP_LIT = 0, OP_BASE = 1, OP_END = 2, OP_LINK = 3, OP_BAD = 4 };

struct Pair { uint8_t b0, b1, bits, n; };                     // the next 11 bits as one or two literals (n = 0: not a literal)

struct Tables {
    Entry lit[2048 + 288 * 16];
    Entry dist[256 + 32 * 128];
    Pair lit2[2048];
};

// Filtered image rows are mostly literals with codes of 3-8 bits; the chain "index -> table entry -> shift" is what a decoder
// waits for, and a table that answers eleven bits with TWO literals where both codes fit halves it.
inline void build_pairs(Tables& T)
{
    for (unsigned i = 0; i < 2048; i++) {
        const Entry e1 = T.lit[i];
        Pair q{0, 0, 0, 0};
        if (e1.op == OP_LIT && e1.bits <= 11) {
            q.b0 = (uint8_t)e1.val; q.bits = e1.bits; q.n = 1;
            const unsigned rem = 11u - e1.bits;
            const Entry e2 = T.lit[(i >> e1.bits) & ((1u << rem) - 1)];       // (a code of <= rem bits is decided by those bits alone)
            if (e2.op == OP_LIT && e2.bits <= rem) { q.b1 = (uint8_t)e2.val; q.bits = (uint8_t)(e1.bits + e2.bits); q.n = 2; }
        }
        T.lit2[i] = q;
    }
}

// canonical Huffman decoding table from code lengths (LSB-first bit order: table index = next bits of the stream).
// primary: index bits of the first level.  Returns false for over-subscribed or incomplete sets (the one legal incomplete case --
// a single distance code -- is left to zlib).
inline bool build_table(const uint8_t* len, int n, int primary, Entry* tab, int tab_cap, const uint16_t* base, const uint8_t* extra,
                        int first_base_symbol, int end_symbol)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
        if (count[l]) maxlen = l;
    }
    if (left != 0 || maxlen == 0) return false;
    int next[16];
    next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (next[l] + count[l]) << 1;
    const int psize = 1 << primary;
    for (int i = 0; i < psize; i++) tab[i] = Entry{0, 0, OP_BAD};
    int used = psize;
    for (int sym = 0; sym < n; sym++) {
        const int l = len[sym];
        if (!l) continue;
        unsigned c = (unsigned)next[l]++, r = 0;
        for (int b = 0; b < l; b++) { r = (r << 1) | (c & 1); c >>= 1; }
        Entry e;
        e.bits = (uint8_t)l;
        if (sym == end_symbol) { e.val = 0; e.op = OP_END; }
        else if (sym < first_base_symbol) { e.val = (uint16_t)sym; e.op = OP_LIT; }
        else {
            const int k = sym - first_base_symbol;
            if (base[k] == 0xffff) { e.val = 0; e.op = OP_BAD; }          // symbols the format reserves (286, 287; 30, 31)
            else { e.val = base[k]; e.op = (uint8_t)(OP_BASE | (extra[k] << 4)); }
        }
        if (l <= primary) {
            for (unsigned i = r; i < (unsigned)psize; i += 1u << l) tab[i] = e;
        } else {
            const unsigned low = r & (unsigned)(psize - 1);
            const int sbits = maxlen - primary;                            // every second-level table has the same size
            if (tab[low].op != OP_LINK) {
                if (used + (1 << sbits) > tab_cap) return false;
                tab[low] = Entry{(uint16_t)used, (uint8_t)sbits, OP_LINK};
                for (int i = 0; i < (1 << sbits); i++) tab[used + i] = Entry{0, 0, OP_BAD};
                used += 1 << sbits;
            }
            Entry* sub = tab + tab[low].val;
            for (unsigned i = r >> primary; i < (1u << sbits); i += 1u << (l - primary)) sub[i] = e;
        }
    }
    return true;
}

inline uint32_t adler32(const uint8_t* p, size_t n)
{
    uint64_t a = 1, b = 0;
    while (n >= 256) {
        size_t blocks = n / 256 < 16 ? n / 256 : 16;
        n -= blocks * 256;
        for (; blocks; blocks--, p += 256) {
            uint32_t s1 = 0, s2 = 0;
            for (int i = 0; i < 256; i++) { s1 += p[i]; s2 += (uint32_t)(256 - i) * p[i]; }
            b += 256 * a + s2;
            a += s1;
        }
        a %= 65521;
        b %= 65521;
    }
    for (size_t i = 0; i < n; i++) { a += p[i]; b += a; }
    return (uint32_t)((b % 65521) << 16 | (a % 65521));
}

// Decodes the zlib stream in[0 .. n) into out[0 .. out_len): true only if the stream is well-formed, produces exactly out_len
// bytes and its Adler-32 matches.  out must have 8 bytes of slack behind out_len (match copies move eight bytes at a time).
inline bool zlib_decode(const uint8_t* in, size_t n, uint8_t* out, size_t out_len, Tables& T)
{
    static const uint16_t lbase[31] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0xffff, 0xffff};
    static const uint8_t lextra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
    static const uint16_t dbase[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0xffff, 0xffff};
    static const uint8_t dextra[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
    if (n < 6) return false;
    if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8 | in[1]) % 31) != 0 || (in[1] & 32)) return false;
    const uint8_t* p = in + 2;
    const uint8_t* const end = in + n - 4;                    // the Adler-32 trailer is not deflate data
    uint8_t* o = out;
    uint8_t* const oend = out + out_len;
    uint64_t buf = 0;
    unsigned cnt = 0;
    // at least 56 valid bits in buf while input lasts; beyond its end zero bytes are shifted in and `over` counts them -- they are
    // the topmost bytes of buf, so a stream has consumed bits it did not have exactly when fewer than `over` whole bytes are left
    size_t over = 0;
    auto refill = [&]() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) {
                if (p < end) buf |= (uint64_t)*p++ << cnt; else over++;
                cnt += 8;
            }
        }
    };
    auto take = [&](unsigned k) -> unsigned { const unsigned v = (unsigned)(buf & ((1ull << k) - 1)); buf >>= k; cnt -= k; return v; };
    bool last = false;
    while (!last) {
        refill();
        last = take(1) != 0;
        const unsigned type = take(2);
        if (type == 0) {                                       // stored
            take(cnt & 7);
            refill();
            const unsigned len = take(16), nlen = take(16);
            if ((len ^ nlen) != 0xffffu) return false;
            // bytes still in the bit buffer first, then straight from the input
            unsigned left = len;
            if ((size_t)(oend - o) < left) return false;
            while (left && cnt >= 8) { *o++ = (uint8_t)take(8); left--; }
            if (left) {                                        // (the buffer is drained: with `over` it has handed out bytes that do not exist)
                if (over || (size_t)(end - p) < left) return false;
                // (cnt < 8 here and byte-aligned: nothing of the stream is left in buf)
                buf = 0; cnt = 0;
                memcpy(o, p, left);
                o += left; p += left;
            }
            continue;
        }
        if (type == 3) return false;
        if (type == 1) {                                       // fixed codes
            uint8_t len[288];
            for (int i = 0; i < 144; i++) len[i] = 8;
            for (int i = 144; i < 256; i++) len[i] = 9;
            for (int i = 256; i < 280; i++) len[i] = 7;
            for (int i = 280; i < 288; i++) len[i] = 8;
            if (!build_table(len, 288, 11, T.lit, (int)(sizeof T.lit / sizeof T.lit[0]), lbase, lextra, 257, 256)) return false;
            uint8_t dl[32];
            for (int i = 0; i < 32; i++) dl[i] = 5;
            if (!build_table(dl, 32, 8, T.dist, (int)(sizeof T.dist / sizeof T.dist[0]), dbase, dextra, 0, -1)) return false;
            build_pairs(T);
        } else {                                               // dynamic codes
            const unsigned hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            static const uint8_t perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (unsigned i = 0; i < hclen; i++) { refill(); cl[perm[i]] = (uint8_t)take(3); }
            Entry ct[128];
            if (!build_table(cl, 19, 7, ct, 128, nullptr, nullptr, 19, -1)) return false;
            uint8_t len[286 + 30 + 138];
            unsigned i = 0;
            while (i < hlit + hdist) {
                refill();
                const Entry e = ct[buf & 127];
                if (e.op != OP_LIT) return false;
                take(e.bits);
                if (e.val < 16) { len[i++] = (uint8_t)e.val; continue; }
                unsigned rep, v = 0;
                if (e.val == 16) { if (i == 0) return false; v = len[i - 1]; rep = 3 + take(2); }
                else if (e.val == 17) rep = 3 + take(3);
                else rep = 11 + take(7);
                if (i + rep > hlit + hdist) return false;
                while (rep--) len[i++] = (uint8_t)v;
            }
            if (len[256] == 0) return false;
            if (!build_table(len, (int)hlit, 11, T.lit, (int)(sizeof T.lit / sizeof T.lit[0]), lbase, lextra, 257, 256)) return false;
            if (!build_table(len + hlit, (int)hdist, 8, T.dist, (int)(sizeof T.dist / sizeof T.dist[0]), dbase, dextra, 0, -1)) return false;
            build_pairs(T);
        }
        for (;;) {                                             // the block's symbols
            refill();
            {                                                  // up to three table answers (<= 33 of the >= 56 bits), each one or two literals
                Pair q = T.lit2[buf & 2047];
                if (q.n) {
                    for (int step = 0; step < 3 && q.n; step++) {
                        if ((size_t)(oend - o) < q.n) return false;
                        o[0] = q.b0;
                        o[1] = q.b1;                           // (n = 1: a byte of the slack or of what comes next)
                        o += q.n;
                        take(q.bits);
                        q = T.lit2[buf & 2047];
                    }
                    continue;
                }
            }
            Entry e = T.lit[buf & 2047];
            if (e.op == OP_LINK) e = T.lit[e.val + ((buf >> 11) & ((1u << e.bits) - 1))];
            if (e.op == OP_LIT) {                              // (a literal with a code longer than 11 bits)
                if (o >= oend) return false;
                take(e.bits);
                *o++ = (uint8_t)e.val;
                continue;
            }
            if (e.op == OP_END) { take(e.bits); break; }
            if ((e.op & 15) != OP_BASE) return false;
            take(e.bits);
            const unsigned length = e.val + take(e.op >> 4);   // <= 15 + 5 of the >= 56 bits used so far: the distance's <= 15 + 13 are there
            Entry d = T.dist[buf & 255];
            if (d.op == OP_LINK) d = T.dist[d.val + ((buf >> 8) & ((1u << d.bits) - 1))];
            if ((d.op & 15) != OP_BASE) return false;
            take(d.bits);
            const unsigned distance = d.val + take(d.op >> 4);
            if (distance > (size_t)(o - out) || length > (size_t)(oend - o)) return false;
            const uint8_t* s = o - distance;
            uint8_t* const stop = o + length;
            if (distance >= 8) {
                do { memcpy(o, s, 8); o += 8; s += 8; } while (o < stop);      // (up to 7 bytes beyond `stop`: the slack)
                o = stop;
            } else {
                while (o < stop) *o++ = *s++;
            }
        }
    }
    if (o != oend || (cnt >> 3) < over) return false;
    // whole real bytes still sitting in the bit buffer belong to the input again (the trailer follows the last block's final byte)
    p -= (cnt >> 3) - over;
    if (p != end) return false;
    const uint32_t want = (uint32_t)end[0] << 24 | (uint32_t)end[1] << 16 | (uint32_t)end[2] << 8 | end[3];
    return adler32(out, out_len) == want;
}

}  // namespace inflate_fast
