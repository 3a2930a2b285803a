#include "png_codec.hpp"

#include "../huffman.hpp"
#include "inflate_fast.hpp"

#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>

namespace pngio {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

#if defined(__SSE2__)
// Paeth rows of 3- or 4-byte pixels, one pixel per step in 16-bit lanes (the predictor's chain a -> cur -> a is per channel, so
// a pixel's channels go side by side; compare-and-select instead of branches).  Reads and writes four bytes per pixel: the row's
// last pixel is left to the caller.  ~4 x the scalar loop, which was the larger part of reading a photograph.
inline void unfilter_paeth_sse2(const uint8_t* src, uint8_t* cur, const uint8_t* up, size_t npix, int bpp)
{
    const __m128i zero = _mm_setzero_si128();
    __m128i a = zero, c = zero;
    for (size_t k = 0; k < npix; k++, src += bpp, cur += bpp, up += bpp) {
        int32_t wb, wx;
        memcpy(&wb, up, 4);
        memcpy(&wx, src, 4);
        const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(wb), zero);
        const __m128i x = _mm_unpacklo_epi8(_mm_cvtsi32_si128(wx), zero);
        const __m128i pav = _mm_sub_epi16(b, c), pbv = _mm_sub_epi16(a, c), pcv = _mm_add_epi16(pav, pbv);
        const __m128i pa = _mm_max_epi16(pav, _mm_sub_epi16(zero, pav));
        const __m128i pb = _mm_max_epi16(pbv, _mm_sub_epi16(zero, pbv));
        const __m128i pc = _mm_max_epi16(pcv, _mm_sub_epi16(zero, pcv));
        const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        const __m128i ma = _mm_cmpeq_epi16(smallest, pa), mb = _mm_cmpeq_epi16(smallest, pb);      // ties: a, then b, then c
        const __m128i bc = _mm_or_si128(_mm_and_si128(mb, b), _mm_andnot_si128(mb, c));
        const __m128i pred = _mm_or_si128(_mm_and_si128(ma, a), _mm_andnot_si128(ma, bc));
        const __m128i d = _mm_and_si128(_mm_add_epi16(x, pred), _mm_set1_epi16(255));
        const int32_t out = _mm_cvtsi128_si32(_mm_packus_epi16(d, d));
        memcpy(cur, &out, 4);
        a = d;
        c = b;
    }
}
// Sub and Average rows the same way: the left neighbour stays in a register instead of going through a store and a load per byte;
// bytes wrap by themselves, floor((a + b) / 2) = (a & b) + ((a ^ b) >> 1).  `up` NULL: Sub.
inline void unfilter_sub_avg_sse2(const uint8_t* src, uint8_t* cur, const uint8_t* up, size_t npix, int bpp)
{
    __m128i a = _mm_setzero_si128();
    const __m128i m7f = _mm_set1_epi8(0x7f);
    for (size_t k = 0; k < npix; k++, src += bpp, cur += bpp) {
        int32_t wx, wb = 0;
        memcpy(&wx, src, 4);
        __m128i pred = a;
        if (up) {
            memcpy(&wb, up, 4);
            up += bpp;
            const __m128i b = _mm_cvtsi32_si128(wb);
            pred = _mm_add_epi8(_mm_and_si128(a, b), _mm_and_si128(_mm_srli_epi16(_mm_xor_si128(a, b), 1), m7f));
        }
        a = _mm_add_epi8(_mm_cvtsi32_si128(wx), pred);
        const int32_t out = _mm_cvtsi128_si32(a);
        memcpy(cur, &out, 4);
        if (bpp == 3) a = _mm_and_si128(a, _mm_cvtsi32_si128(0x00ffffff));      // (the fourth byte belongs to the next pixel)
    }
}
#endif

// undo the per-row filters of one (sub)image; in: h rows of (1 + rowbytes); out: h rows of rowbytes.  One loop per row and
// filter type (the first bpp bytes have no left neighbour).
bool unfilter(const uint8_t* in, uint8_t* out, int h, size_t rowbytes, int bpp)
{
    const size_t bp = (size_t)bpp < rowbytes ? (size_t)bpp : rowbytes;
    for (int y = 0; y < h; y++) {
        const uint8_t ft = in[(size_t)y * (rowbytes + 1)];
        const uint8_t* src = in + (size_t)y * (rowbytes + 1) + 1;
        uint8_t* cur = out + (size_t)y * rowbytes;
        const uint8_t* up = y ? cur - rowbytes : nullptr;
        switch (ft) {
        case 0: memcpy(cur, src, rowbytes); break;
        case 1: {
            size_t i = 0;
#if defined(__SSE2__)
            if ((bpp == 3 || bpp == 4) && rowbytes >= (size_t)3 * bpp) {          // pixels 0 .. n-2; the last one below
                unfilter_sub_avg_sse2(src, cur, nullptr, rowbytes / bpp - 1, bpp);
                i = (rowbytes / bpp - 1) * bpp;
            }
#endif
            for (; i < bp; i++) cur[i] = src[i];
            for (; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + cur[i - bp]);
            break;
        }
        case 2:
            if (!up) memcpy(cur, src, rowbytes);
            else for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + up[i]);
            break;
        case 3: {
            size_t i = 0;
#if defined(__SSE2__)
            if (up && (bpp == 3 || bpp == 4) && rowbytes >= (size_t)3 * bpp) {    // (pixel 0: a = 0 gives b >> 1, as the loop below)
                unfilter_sub_avg_sse2(src, cur, up, rowbytes / bpp - 1, bpp);
                i = (rowbytes / bpp - 1) * bpp;
            }
#endif
            for (; i < bp; i++) cur[i] = (uint8_t)(src[i] + ((up ? up[i] : 0) >> 1));
            if (up) for (; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + ((cur[i - bp] + up[i]) >> 1));
            else for (; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + (cur[i - bp] >> 1));
            break;
        }
        case 4:
            for (size_t i = 0; i < bp; i++) cur[i] = (uint8_t)(src[i] + (up ? up[i] : 0));             // paeth(0, b, 0) = b
            if (up) {
                size_t i = bp;
#if defined(__SSE2__)
                if ((bpp == 3 || bpp == 4) && rowbytes >= (size_t)3 * bpp) {      // pixels 0 .. n-2 (pixel 0: a = c = 0 gives b, as above)
                    const size_t npix = rowbytes / bpp - 1;
                    unfilter_paeth_sse2(src, cur, up, npix, bpp);
                    i = npix * bpp;
                }
#endif
                for (; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + paeth(cur[i - bp], up[i], up[i - bp]));
            }
            else for (size_t i = bp; i < rowbytes; i++) cur[i] = (uint8_t)(src[i] + cur[i - bp]);         // paeth(a, 0, 0) = a
            break;
        default: return false;
        }
    }
    return true;
}

struct Header { int w, h, depth, ctype, interlace; };

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// sample s (0-based) of pixel x in an unfiltered row -> 8-bit value (raw index for palette images)
inline int sample(const uint8_t* row, int x, int s, int nch, int depth, bool scale)
{
    const size_t idx = (size_t)x * nch + s;
    if (depth == 8) return row[idx];
    if (depth == 16) return row[idx * 2];                      // stb: 16 -> 8 keeps the high byte
    const int per = 8 / depth;
    const int v = (row[idx / per] >> (8 - depth * (1 + (int)(idx % per)))) & ((1 << depth) - 1);
    static const int mul[5] = {0, 255, 85, 0, 17};             // stb scales 1/2/4-bit grey to 0..255
    return scale ? v * mul[depth] : v;
}

void emit_pixel(const Header& hd, const uint8_t* row, int x, const uint8_t* pal, int npal, uint8_t* dst)
{
    const int nch = channels_of(hd.ctype);
    switch (hd.ctype) {
    case 0: case 4: { int g = sample(row, x, 0, nch, hd.depth, true); dst[0] = dst[1] = dst[2] = (uint8_t)g; break; }
    case 2: case 6:
        for (int s = 0; s < 3; s++) dst[s] = (uint8_t)sample(row, x, s, nch, hd.depth, false);
        break;
    default: {
        int i = sample(row, x, 0, 1, hd.depth, false);
        if (i >= npal) i = 0;
        dst[0] = pal[3 * i]; dst[1] = pal[3 * i + 1]; dst[2] = pal[3 * i + 2];
    }
    }
}

// per-thread work buffers, kept between files: a batch thread en/decodes hundreds of equal-sized images, and fresh 25 MB
// vectors per file mean an mmap, 6000 page faults and a munmap each -- with dozens of codec threads in one process those
// serialise on the address-space lock (64 threads: 514 ms per encode instead of 185, profiles/r04_zl_cli_batch.txt)
struct Scratch { std::vector<uint8_t> file, idat, raw, img, comp; inflate_fast::Tables tables; };
Scratch& scratch()
{
    static thread_local Scratch s;
    return s;
}

}  // namespace

bool load_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height, int& channels_in_file,
               std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t>& file = scratch().file;
    file.clear();
    long sz = -1;
    if (fseek(f, 0, SEEK_END) == 0) { sz = ftell(f); rewind(f); }
    try {
        if (sz > 0) {                                   // a regular file: one read of its size
            file.resize((size_t)sz);
            file.resize(fread(file.data(), 1, (size_t)sz, f));
        } else {                                        // a pipe, /dev/stdin, a process substitution: read until the end
            clearerr(f);
            uint8_t chunk[65536];
            size_t n;
            while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) file.insert(file.end(), chunk, chunk + n);
        }
    } catch (const std::bad_alloc&) { fclose(f); err = "image too large"; return false; }
    const bool read_error = ferror(f) != 0;
    fclose(f);
    if (read_error) { err = "read error on " + path; return false; }
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) { err = "not a PNG file"; return false; }
    Header hd{};
    bool have_hdr = false;
    std::vector<uint8_t>& idat = scratch().idat;
    std::vector<uint8_t> pal;
    idat.clear();
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG"; return false; }
        const uint8_t* d = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (have_hdr || len != 13) { err = "bad IHDR"; return false; }
            const uint32_t w32 = be32(d), h32 = be32(d + 4);
            // stb_image's own limits: 2^24 per dimension; the decoded RGB must also fit comfortably in memory
            if (w32 == 0 || h32 == 0 || w32 > (1u << 24) || h32 > (1u << 24) || (uint64_t)w32 * h32 > (1ull << 28)) { err = "bad IHDR (size)"; return false; }
            hd.w = (int)w32; hd.h = (int)h32; hd.depth = d[8]; hd.ctype = d[9]; hd.interlace = d[12];
            if (d[10] != 0 || d[11] != 0 || d[12] > 1) { err = "bad IHDR (compression/filter/interlace method)"; return false; }
            have_hdr = true;
        } else if (!have_hdr) { err = "first chunk is not IHDR"; return false; }
        else if (!memcmp(type, "PLTE", 4)) {
            if (len == 0 || len % 3 != 0 || len > 768) { err = "bad PLTE"; return false; }
            pal.assign(d, d + len);
        }
        else if (!memcmp(type, "IDAT", 4)) {
            try { idat.insert(idat.end(), d, d + len); } catch (const std::bad_alloc&) { err = "image too large"; return false; }
        }
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || hd.w <= 0 || hd.h <= 0) { err = "bad IHDR"; return false; }
    if (!(hd.ctype == 0 || hd.ctype == 2 || hd.ctype == 3 || hd.ctype == 4 || hd.ctype == 6)) { err = "bad colour type"; return false; }
    {
        // legal depth per colour type (PNG spec table 11.1): grey 1/2/4/8/16, palette 1/2/4/8, the others 8/16
        const int dp = hd.depth;
        const bool pow2 = dp == 1 || dp == 2 || dp == 4 || dp == 8 || dp == 16;
        const bool ok = pow2 && (hd.ctype == 0 || (hd.ctype == 3 ? dp <= 8 : dp >= 8));
        if (!ok) { err = "bad bit depth for colour type"; return false; }
    }
    if (hd.ctype == 3 && pal.size() < 3) { err = "palette missing"; return false; }
    if (idat.empty()) { err = "no image data"; return false; }
    const int nch = channels_of(hd.ctype);
    const int bits = nch * hd.depth;
    const int bpp = bits >= 8 ? bits / 8 : 1;
    auto rowbytes = [&](int w) { return ((size_t)w * bits + 7) / 8; };

    // passes: non-interlaced = one; Adam7 = seven sub-images
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = hd.interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < npass; p++) {
        const int pw = hd.interlace ? (hd.w - xs[p] + dx[p] - 1) / dx[p] : hd.w;
        const int ph = hd.interlace ? (hd.h - ys[p] + dy[p] - 1) / dy[p] : hd.h;
        if (pw > 0 && ph > 0) raw_size += (size_t)ph * (rowbytes(pw) + 1);
    }
    std::vector<uint8_t>& raw = scratch().raw;
    try {
        raw.resize(raw_size + 8);                                  // (+ 8: the fast decoder's match copies move eight bytes at a time)
        rgb.resize((size_t)hd.w * hd.h * 3);
    } catch (const std::bad_alloc&) { err = "image too large"; return false; }
    // the fast decoder first (inflate_fast.hpp: about twice zlib's rate, well-formed streams only); whatever it refuses goes to
    // zlib, whose verdict counts
    if (!inflate_fast::zlib_decode(idat.data(), idat.size(), raw.data(), raw_size, scratch().tables)) {
        uLongf dl = (uLongf)raw_size;
        int zr = uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size());
        if (zr != Z_OK || dl != raw_size) { err = "zlib inflate failed"; return false; }
    }

    width = hd.w; height = hd.h;
    channels_in_file = hd.ctype == 3 ? 3 : nch;
    if (hd.ctype == 2 && hd.depth == 8 && !hd.interlace) {         // the common case: the unfiltered rows ARE the image
        if (!unfilter(raw.data(), rgb.data(), hd.h, rowbytes(hd.w), bpp)) { err = "bad filter type"; return false; }
        return true;
    }
    size_t off = 0;
    std::vector<uint8_t>& img = scratch().img;
    for (int p = 0; p < npass; p++) {
        const int pw = hd.interlace ? (hd.w - xs[p] + dx[p] - 1) / dx[p] : hd.w;
        const int ph = hd.interlace ? (hd.h - ys[p] + dy[p] - 1) / dy[p] : hd.h;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rb = rowbytes(pw);
        try { img.resize((size_t)ph * rb); } catch (const std::bad_alloc&) { err = "image too large"; return false; }
        if (!unfilter(raw.data() + off, img.data(), ph, rb, bpp)) { err = "bad filter type"; return false; }
        off += (size_t)ph * (rb + 1);
        if (!hd.interlace && hd.depth == 8 && (hd.ctype == 6 || hd.ctype == 0 || hd.ctype == 4)) {     // RGBA / grey / grey + alpha, 8 bits: tight loops
            const uint8_t* sp = img.data();
            uint8_t* dp = rgb.data();
            const size_t npx = (size_t)pw * ph;
            if (hd.ctype == 6) for (size_t i = 0; i < npx; i++) { dp[3 * i] = sp[4 * i]; dp[3 * i + 1] = sp[4 * i + 1]; dp[3 * i + 2] = sp[4 * i + 2]; }
            else { const int st = hd.ctype == 0 ? 1 : 2; for (size_t i = 0; i < npx; i++) dp[3 * i] = dp[3 * i + 1] = dp[3 * i + 2] = sp[st * i]; }
            continue;
        }
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                const int X = hd.interlace ? xs[p] + x * dx[p] : x, Y = hd.interlace ? ys[p] + y * dy[p] : y;
                emit_pixel(hd, img.data() + (size_t)y * rb, x, pal.data(), (int)pal.size() / 3, &rgb[((size_t)Y * hd.w + X) * 3]);
            }
    }
    return true;
}

// ---- encoder ------------------------------------------------------------------------------------------------------
// The CLI's batched mode is bound by this function (a 4096x2048 frame: 0.06 ms of kernels, 0.5 ms of PCIe, ~200 ms of PNG
// encoding with zlib at its fastest setting), so the two stages are written for speed:
//  * row filters: the five sums of |residual| (the selection heuristic stb_image_write uses too) in reductions the compiler
//    vectorises (AVX2 clone chosen at load time), then only the winner is materialised;
//  * deflate: filtered rows of an interpolated image are small residuals without repeats worth a match search -- zlib's
//    Z_RLE and Z_HUFFMAN_ONLY strategies produce the same size -- so the stream is Huffman-only: per 256 KB block a byte
//    histogram, length-limited canonical codes, and two symbols per 64-bit store.  3-6 x zlib's rate, same size within 1 %.
namespace {

#if defined(__x86_64__) && defined(__GNUC__)
#define PNGIO_SIMD __attribute__((target_clones("avx2", "default")))
#else
#define PNGIO_SIMD
#endif

inline uint8_t mag8(uint8_t r) { const uint8_t n = (uint8_t)(0 - r); return r < n ? r : n; }      // |(int8_t) r|

// sums of |residual| of the five filter types over the bytes [lo, hi) of a row, lo >= 3; `up` may be NULL (first row)
PNGIO_SIMD void filter_sums(const uint8_t* cur, const uint8_t* up, size_t lo, size_t hi, unsigned* s)
{
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (size_t i = lo; i < hi; i++) s0 += mag8(cur[i]);
    for (size_t i = lo; i < hi; i++) s1 += mag8((uint8_t)(cur[i] - cur[i - 3]));
    if (!up) {                                                                               // b = c = 0: Up = None, Paeth = Sub
        for (size_t i = lo; i < hi; i++) s3 += mag8((uint8_t)(cur[i] - (cur[i - 3] >> 1)));
        s2 = s0; s4 = s1;
    } else {
        for (size_t i = lo; i < hi; i++) s2 += mag8((uint8_t)(cur[i] - up[i]));
        for (size_t i = lo; i < hi; i++) s3 += mag8((uint8_t)(cur[i] - (uint8_t)(((unsigned)cur[i - 3] + up[i]) >> 1)));
        for (size_t i = lo; i < hi; i++) {
            const int16_t a = cur[i - 3], b = up[i], c = up[i - 3];
            const int16_t pa = (int16_t)(b > c ? b - c : c - b), pb = (int16_t)(a > c ? a - c : c - a);
            const int16_t t = (int16_t)(a + b - 2 * c), pc = (int16_t)(t < 0 ? -t : t);
            const uint8_t pr = (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
            s4 += mag8((uint8_t)(cur[i] - pr));
        }
    }
    s[0] += s0; s[1] += s1; s[2] += s2; s[3] += s3; s[4] += s4;
}

// PNG filter type with the smallest sum of |residual| for one row of 8-bit RGB (the heuristic of stb_image_write and libpng).
// Rows of more than 3 KB are judged on a quarter of their bytes -- 192 of every 768, spread over the whole row: the heuristic
// is a guess at the entropy anyway, and the five sums were a fifth of the encoder's time.
int choose_filter(const uint8_t* cur, const uint8_t* up, size_t rb)
{
    unsigned s[5] = {0, 0, 0, 0, 0};
    if (rb <= 3072) {
        for (size_t i = 0; i < 3 && i < rb; i++) {                                           // first pixel: a = c = 0
            const uint8_t v = cur[i], b = up ? up[i] : 0;
            s[0] += mag8(v); s[1] += mag8(v); s[2] += mag8((uint8_t)(v - b)); s[3] += mag8((uint8_t)(v - (b >> 1))); s[4] += mag8((uint8_t)(v - b));
        }
        if (rb > 3) filter_sums(cur, up, 3, rb, s);
    } else {
        for (size_t lo = 3; lo < rb; lo += 768) filter_sums(cur, up, lo, lo + 192 < rb ? lo + 192 : rb, s);
    }
    int best = 0;
    for (int ft = 1; ft < 5; ft++)
        if (s[ft] < s[best]) best = ft;
    return best;
}

// residuals of one row under filter ft -> out[0 .. rb)
PNGIO_SIMD void apply_filter(int ft, const uint8_t* cur, const uint8_t* up, size_t rb, uint8_t* out)
{
    const size_t h = rb < 3 ? rb : 3;
    for (size_t i = 0; i < h; i++) {
        const uint8_t v = cur[i], b = up ? up[i] : 0;
        out[i] = ft == 0 || ft == 1 ? v : ft == 3 ? (uint8_t)(v - (b >> 1)) : (uint8_t)(v - b);
    }
    switch (ft) {
    case 0: for (size_t i = 3; i < rb; i++) out[i] = cur[i]; break;
    case 1: for (size_t i = 3; i < rb; i++) out[i] = (uint8_t)(cur[i] - cur[i - 3]); break;
    case 2:
        if (up) for (size_t i = 3; i < rb; i++) out[i] = (uint8_t)(cur[i] - up[i]);
        else for (size_t i = 3; i < rb; i++) out[i] = cur[i];
        break;
    case 3:
        if (up) for (size_t i = 3; i < rb; i++) out[i] = (uint8_t)(cur[i] - (uint8_t)(((unsigned)cur[i - 3] + up[i]) >> 1));
        else for (size_t i = 3; i < rb; i++) out[i] = (uint8_t)(cur[i] - (cur[i - 3] >> 1));
        break;
    default:
        if (!up) { for (size_t i = 3; i < rb; i++) out[i] = (uint8_t)(cur[i] - cur[i - 3]); break; }
        for (size_t i = 3; i < rb; i++) {
            const int16_t a = cur[i - 3], b = up[i], c = up[i - 3];
            const int16_t pa = (int16_t)(b > c ? b - c : c - b), pb = (int16_t)(a > c ? a - c : c - a);
            const int16_t t = (int16_t)(a + b - 2 * c), pc = (int16_t)(t < 0 ? -t : t);
            const uint8_t pr = (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
            out[i] = (uint8_t)(cur[i] - pr);
        }
    }
}

// -- Huffman-only deflate: code construction and the block header are shared with the device-side encoder (../huffman.hpp) --

static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "BitWriter stores its 64-bit accumulator as the stream's next eight bytes");
struct BitWriter {                    // LSB-first bit stream; every put stores 8 bytes at the write position (little-endian host)
    uint8_t* p;
    uint64_t buf = 0;
    unsigned cnt = 0;                 // < 8 between calls
    inline void put(uint64_t bits, unsigned n)        // n <= 48
    {
        buf |= bits << cnt;
        cnt += n;
        memcpy(p, &buf, 8);
        p += cnt >> 3;
        buf >>= cnt & ~7u;
        cnt &= 7;
    }
    void byte_align() { if (cnt) { *p++ = (uint8_t)buf; buf = 0; cnt = 0; } }
};

void huffman_block(BitWriter& bw, const uint8_t* src, size_t n, bool last)
{
    uint32_t f4[4][256];
    memset(f4, 0, sizeof f4);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { f4[0][src[i]]++; f4[1][src[i + 1]]++; f4[2][src[i + 2]]++; f4[3][src[i + 3]]++; }
    for (; i < n; i++) f4[0][src[i]]++;
    uint32_t freq[257];
    for (int s = 0; s < 256; s++) freq[s] = f4[0][s] + f4[1][s] + f4[2][s] + f4[3][s];
    freq[256] = 1;                                       // end of block
    uint8_t len[257];
    uint16_t code[257];
    fftup_huff::Work wk;
    fftup_huff::huffman_lengths(freq, 257, 15, len, wk);
    fftup_huff::canonical_codes(len, 257, 15, code, wk);
    uint32_t hdr[64];
    const int hbits = fftup_huff::dynamic_header(len, last, hdr, wk);
    for (int k = 0; k < hbits; k += 32) bw.put(hdr[k >> 5], hbits - k < 32 ? (unsigned)(hbits - k) : 32u);
    uint32_t tab[256];                                   // code | length << 16
    for (int s = 0; s < 256; s++) tab[s] = code[s] | ((uint32_t)len[s] << 16);
    i = 0;
    for (; i + 3 <= n; i += 3) {                         // three symbols (<= 45 bits) per store
        const uint32_t a = tab[src[i]], b = tab[src[i + 1]], c = tab[src[i + 2]];
        const unsigned la = a >> 16, lb = b >> 16, lc = c >> 16;
        bw.put((uint64_t)(a & 0xffff) | ((uint64_t)(b & 0xffff) << la) | ((uint64_t)(c & 0xffff) << (la + lb)), la + lb + lc);
    }
    for (; i < n; i++) bw.put(tab[src[i]] & 0xffff, tab[src[i]] >> 16);
    bw.put(code[256], len[256]);
}

// Adler-32 (RFC 1950) in blocks of 256 bytes: a' = a + sum x_i, b' = b + 256 a + sum (256 - i) x_i -- two reductions the compiler
// vectorises (zlib 1.2's byte-serial loop was a sixth of the encoder's time)
PNGIO_SIMD uint32_t adler32_blocks(const uint8_t* p, size_t n)
{
    uint64_t a = 1, b = 0;
    while (n >= 256) {
        size_t blocks = n / 256 < 16 ? n / 256 : 16;                  // 4 KB between reductions modulo 65521
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

// zlib stream (RFC 1950) of Huffman-only deflate blocks; returns the number of bytes written to out (capacity: bound below)
// (a Huffman code for 257 symbols spends at most ~8.1 bits per symbol on average -- entropy <= 8.006 plus a redundancy below
// p_max + 0.086 --; nine and an eighth are allowed for, plus 236 bytes of header per block)
size_t huffman_zlib_bound(size_t n) { return n + n / 8 + n / 64 + (n / (256 * 1024) + 2) * 512 + 64; }
size_t huffman_zlib(const uint8_t* src, size_t n, uint8_t* out)
{
    BitWriter bw;
    bw.p = out;
    *bw.p++ = 0x78;                                      // deflate, 32 KB window
    *bw.p++ = 0x01;                                      // fastest level, no dictionary, check bits
    const size_t block = 256 * 1024;
    size_t pos = 0;
    do {
        const size_t m = n - pos < block ? n - pos : block;
        huffman_block(bw, src + pos, m, pos + m == n);
        pos += m;
    } while (pos < n);
    bw.byte_align();
    put32(bw.p, adler32_blocks(src, n));
    return (size_t)(bw.p + 4 - out);
}

}  // namespace

bool write_rgb8(const std::string& path, const uint8_t* rgb, int width, int height, size_t row_stride, std::string& err)
{
    if (width <= 0 || height <= 0) { err = "bad image size"; return false; }
    const size_t rb = (size_t)width * 3;
    std::vector<uint8_t>& raw = scratch().raw;
    std::vector<uint8_t>& comp = scratch().comp;
    try {
        raw.resize((size_t)height * (rb + 1));
        comp.resize(huffman_zlib_bound(raw.size()));
    } catch (const std::bad_alloc&) { err = "image too large"; return false; }
    for (int y = 0; y < height; y++) {
        const uint8_t* cur = rgb + (size_t)y * row_stride;
        const uint8_t* up = y ? cur - row_stride : nullptr;
        uint8_t* out = &raw[(size_t)y * (rb + 1)];
        const int ft = choose_filter(cur, up, rb);
        out[0] = (uint8_t)ft;
        apply_filter(ft, cur, up, rb, out + 1);
    }
    // Which deflate?  Filtered rows of an interpolated photograph are small residuals without repeats worth a match search:
    // there the Huffman-only stream is as small as zlib's and 3-6 x faster (and a third smaller than what the reference's writer
    // produces, stbi_write_png: LZ matching behind FIXED codes; tests/test_cli_png.py).  Flat or periodic content (drawings,
    // borders, screenshots, test patterns) is what a match search shrinks by factors.  Decided by trial on a sample -- whole rows
    // spread over the image, <= 128 KB: both coders run on it (a fraction of a millisecond and ~2 ms); when matching saves more
    // than a tenth, the whole stream goes through zlib at its default level.
    bool matching = false;
    {
        const size_t L = rb + 1, want_rows = std::max<size_t>(1, (128 * 1024) / L);
        const size_t step = std::max<size_t>(1, (size_t)height / want_rows);
        std::vector<uint8_t>& sample = scratch().img;               // (the decoder's scratch: free during an encode)
        sample.clear();
        for (size_t y = 0; y < (size_t)height && sample.size() + L <= 160 * 1024; y += step) sample.insert(sample.end(), &raw[y * L], &raw[y * L] + L);
        const size_t hs = huffman_zlib(sample.data(), sample.size(), comp.data());       // (comp holds the bound of the whole stream)
        uLongf zb = compressBound((uLong)sample.size());
        std::vector<uint8_t> ztmp(zb);
        if (compress2(ztmp.data(), &zb, sample.data(), (uLong)sample.size(), 1) == Z_OK) matching = (size_t)zb * 10 < hs * 9;
    }
    size_t cl;
    if (matching) {
        uLongf bound = (uLongf)comp.size();
        z_stream zs{};
        if (deflateInit2(&zs, 6, Z_DEFLATED, 15, 9, Z_DEFAULT_STRATEGY) != Z_OK) { err = "zlib deflate failed"; return false; }
        if (deflateBound(&zs, (uLong)raw.size()) > bound) { bound = deflateBound(&zs, (uLong)raw.size()); comp.resize(bound); }
        // (zlib counts available bytes in 32 bits: input and output are handed over in pieces of at most 1 GB)
        const size_t piece = (size_t)1 << 30;
        size_t in_pos = 0, out_pos = 0;
        int zr = Z_OK;
        while (zr == Z_OK) {
            if (zs.avail_in == 0 && in_pos < raw.size()) {
                const size_t m = std::min(piece, raw.size() - in_pos);
                zs.next_in = raw.data() + in_pos; zs.avail_in = (uInt)m; in_pos += m;
            }
            if (zs.avail_out == 0 && out_pos < (size_t)bound) {
                const size_t m = std::min(piece, (size_t)bound - out_pos);
                zs.next_out = comp.data() + out_pos; zs.avail_out = (uInt)m; out_pos += m;
            }
            zr = deflate(&zs, in_pos == raw.size() ? Z_FINISH : Z_NO_FLUSH);
        }
        cl = out_pos - zs.avail_out;
        deflateEnd(&zs);
        if (zr != Z_STREAM_END) { err = "zlib deflate failed"; return false; }
    } else cl = huffman_zlib(raw.data(), raw.size(), comp.data());
    if (cl > 0x7fffffffu) { err = "image too large for one IDAT chunk (PNG chunks hold at most 2^31 - 1 bytes)"; return false; }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    fwrite(sig, 1, 8, f);
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
        uint8_t hdr[8];
        put32(hdr, len);
        memcpy(hdr + 4, type, 4);
        fwrite(hdr, 1, 8, f);
        if (len) fwrite(d, 1, len, f);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        for (size_t o = 0; o < len; o += 1u << 30) crc = crc32(crc, d + o, (uInt)(len - o < (1u << 30) ? len - o : (1u << 30)));
        uint8_t c4[4];
        put32(c4, (uint32_t)crc);
        fwrite(c4, 1, 4, f);
    };
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)width); put32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)cl);
    chunk("IEND", nullptr, 0);
    const bool ok = !ferror(f);
    if (fclose(f) != 0 || !ok) { err = "write error"; return false; }
    return true;
}

}  // namespace pngio
