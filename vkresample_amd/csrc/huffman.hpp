// huffman.hpp -- the Huffman part of a Huffman-only deflate stream (RFC 1951 dynamic blocks without length/distance symbols),
// shared by the CLI's PNG encoder (host, csrc/cli/png_codec.cpp) and the device-side PNG encoder (csrc/kernels_png.hpp): plain
// arrays and loops, no library calls, so that one text compiles for both.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FFTUP_HD __host__ __device__
#else
#define FFTUP_HD
#endif

namespace fftup_huff {

// working storage of the three routines (the caller provides it: stack on the host, LDS on the device -- private arrays of this
// size would become scratch memory there)
struct Work {
    int order[288];
    uint64_t w[2 * 288];
    int parent[2 * 288];
    int depth[2 * 288];
    int count[64];
    int next[17];
    uint8_t seq[259];
    uint32_t cfreq[19];
    uint8_t clen[19];
    uint16_t ccode[19];
};

// the symbols with a non-zero frequency, ascending by (frequency, symbol) -> order[0 .. used); returns used
FFTUP_HD inline int sort_symbols(const uint32_t* freq, int n, int* order)
{
    int used = 0;
    for (int i = 0; i < n; i++)
        if (freq[i]) order[used++] = i;
    for (int i = 1; i < used; i++) {                   // insertion sort (the device sorts by ranks in parallel instead)
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && (freq[order[j]] > freq[v] || (freq[order[j]] == freq[v] && order[j] > v))) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    return used;
}

// Code lengths (<= maxbits) of an optimal prefix code for freq[0..n), n <= 288, 0 for unused symbols, given the used symbols in
// ascending order of frequency (wk.order[0 .. used)).  Two-queue Huffman construction; lengths beyond maxbits are folded back by
// the usual Kraft-sum repair (one code of the longest length is removed, one shorter code made one bit longer, until the sum is
// exactly one).
FFTUP_HD inline void huffman_lengths_sorted(const uint32_t* freq, int n, int used, int maxbits, uint8_t* len, Work& wk)
{
    const int* order = wk.order;
    for (int i = 0; i < n; i++) len[i] = 0;
    if (used == 0) return;
    if (used == 1) {                                   // a complete code needs two codes: one unused sibling
        len[order[0]] = 1;
        len[order[0] == 0 ? 1 : 0] = 1;
        return;
    }
    uint64_t* w = wk.w;
    int* parent = wk.parent;
    for (int i = 0; i < used; i++) w[i] = freq[order[i]];
    int leaf = 0, inner = used, next = used;            // two queues: leaves [leaf, used), inner nodes [inner, next)
    while ((used - leaf) + (next - inner) > 1) {
        const int p0 = (leaf < used && (inner >= next || w[leaf] <= w[inner])) ? leaf++ : inner++;
        const int p1 = (leaf < used && (inner >= next || w[leaf] <= w[inner])) ? leaf++ : inner++;
        w[next] = w[p0] + w[p1];
        parent[p0] = parent[p1] = next;
        next++;
    }
    const int root = next - 1;
    int* depth = wk.depth;
    depth[root] = 0;
    for (int i = root - 1; i >= 0; i--) depth[i] = depth[parent[i]] + 1;      // a parent is always created after its children
    int* count = wk.count;
    for (int l = 0; l < 64; l++) count[l] = 0;
    for (int i = 0; i < used; i++) count[depth[i] < 63 ? depth[i] : 63]++;
    for (int l = maxbits + 1; l < 64; l++) { count[maxbits] += count[l]; count[l] = 0; }
    uint64_t total = 0;
    for (int l = 1; l <= maxbits; l++) total += (uint64_t)count[l] << (maxbits - l);
    while (total > (1ull << maxbits)) {
        count[maxbits]--;
        for (int l = maxbits - 1; l > 0; l--)
            if (count[l]) { count[l]--; count[l + 1] += 2; break; }
        total--;
    }
    int k = 0;                                           // rarest symbols get the longest codes
    for (int l = maxbits; l >= 1; l--)
        for (int c = 0; c < count[l]; c++) len[order[k++]] = (uint8_t)l;
}

FFTUP_HD inline void huffman_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* len, Work& wk)
{
    huffman_lengths_sorted(freq, n, sort_symbols(freq, n, wk.order), maxbits, len, wk);
}

// first canonical code of every length -> wk.next[1 .. maxbits]
FFTUP_HD inline void canonical_first_codes(const uint8_t* len, int n, int maxbits, Work& wk)
{
    int* count = wk.count;
    for (int l = 0; l < 17; l++) { count[l] = 0; wk.next[l] = 0; }
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    for (int l = 1, c = 0; l <= maxbits; l++) { c = (c + count[l - 1]) << 1; wk.next[l] = c; }
}
FFTUP_HD inline uint16_t reverse_bits(unsigned c, int n)
{
    unsigned r = 0;
    for (int b = 0; b < n; b++) { r = (r << 1) | (c & 1); c >>= 1; }
    return (uint16_t)r;
}

// canonical codes for the lengths, bit-reversed (deflate sends Huffman codes most significant bit first)
FFTUP_HD inline void canonical_codes(const uint8_t* len, int n, int maxbits, uint16_t* code, Work& wk)
{
    canonical_first_codes(len, n, maxbits, wk);
    for (int i = 0; i < n; i++) code[i] = len[i] ? reverse_bits((unsigned)wk.next[len[i]]++, len[i]) : (uint16_t)0;
}

// The header of a dynamic block whose literal/length code has the 257 lengths len[] (literals + end of block) and no length or
// distance symbol in use: BFINAL, BTYPE = 2, HLIT = 257, HDIST = 2 (two distance codes of one bit that are never used: a complete
// set, like zlib sends), the code-length code, and the 259 lengths sent one by one (symbols 0..15, no repeat symbols).  The bits go,
// least significant first, into words[0 .. 63] (zeroed here); returns their number (<= 1887).
FFTUP_HD inline int dynamic_header(const uint8_t* len, bool last, uint32_t* words, Work& wk)
{
    uint8_t* seq = wk.seq;
    for (int k = 0; k < 257; k++) seq[k] = len[k];
    seq[257] = seq[258] = 1;
    uint32_t* cfreq = wk.cfreq;
    for (int k = 0; k < 19; k++) cfreq[k] = 0;
    for (int k = 0; k < 259; k++) cfreq[seq[k]]++;
    uint8_t* clen = wk.clen;
    uint16_t* ccode = wk.ccode;
    huffman_lengths(cfreq, 19, 7, clen, wk);
    canonical_codes(clen, 19, 7, ccode, wk);
    const int perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && clen[perm[hclen - 1]] == 0) hclen--;
    for (int k = 0; k < 64; k++) words[k] = 0;
    int nbits = 0;
    auto put = [&](uint32_t v, int n) {                    // n <= 16 bits, least significant first
        const int sh = nbits & 31;
        words[nbits >> 5] |= v << sh;
        if (sh + n > 32) words[(nbits >> 5) + 1] |= v >> (32 - sh);
        nbits += n;
    };
    put(last ? 1 : 0, 1);
    put(2, 2);
    put(257 - 257, 5);
    put(2 - 1, 5);
    put((uint32_t)hclen - 4, 4);
    for (int k = 0; k < hclen; k++) put(clen[perm[k]], 3);
    for (int k = 0; k < 259; k++) put(ccode[seq[k]], clen[seq[k]]);
    return nbits;
}

}  // namespace fftup_huff
