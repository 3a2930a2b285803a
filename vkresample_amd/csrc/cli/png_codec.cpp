#include "png_codec.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
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

// undo the per-row filters of one (sub)image; in: h rows of (1 + rowbytes); out: h rows of rowbytes
bool unfilter(const uint8_t* in, uint8_t* out, int h, size_t rowbytes, int bpp)
{
    for (int y = 0; y < h; y++) {
        const uint8_t ft = in[(size_t)y * (rowbytes + 1)];
        const uint8_t* src = in + (size_t)y * (rowbytes + 1) + 1;
        uint8_t* cur = out + (size_t)y * rowbytes;
        const uint8_t* up = y ? cur - rowbytes : nullptr;
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (uint8_t)v;
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

}  // namespace

bool load_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height, int& channels_in_file,
               std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> file;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) file.insert(file.end(), tmp, tmp + n);
    fclose(f);
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) { err = "not a PNG file"; return false; }
    Header hd{};
    bool have_hdr = false;
    std::vector<uint8_t> idat, pal;
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
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
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
    std::vector<uint8_t> raw;
    try {
        raw.resize(raw_size);
        rgb.assign((size_t)hd.w * hd.h * 3, 0);
    } catch (const std::bad_alloc&) { err = "image too large"; return false; }
    uLongf dl = (uLongf)raw_size;
    int zr = uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || dl != raw_size) { err = "zlib inflate failed"; return false; }

    width = hd.w; height = hd.h;
    channels_in_file = hd.ctype == 3 ? 3 : nch;
    size_t off = 0;
    std::vector<uint8_t> img;
    for (int p = 0; p < npass; p++) {
        const int pw = hd.interlace ? (hd.w - xs[p] + dx[p] - 1) / dx[p] : hd.w;
        const int ph = hd.interlace ? (hd.h - ys[p] + dy[p] - 1) / dy[p] : hd.h;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rb = rowbytes(pw);
        img.resize((size_t)ph * rb);
        if (!unfilter(raw.data() + off, img.data(), ph, rb, bpp)) { err = "bad filter type"; return false; }
        off += (size_t)ph * (rb + 1);
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                const int X = hd.interlace ? xs[p] + x * dx[p] : x, Y = hd.interlace ? ys[p] + y * dy[p] : y;
                emit_pixel(hd, img.data() + (size_t)y * rb, x, pal.data(), (int)pal.size() / 3, &rgb[((size_t)Y * hd.w + X) * 3]);
            }
    }
    return true;
}

bool write_rgb8(const std::string& path, const uint8_t* rgb, int width, int height, size_t row_stride, std::string& err)
{
    const size_t rb = (size_t)width * 3;
    std::vector<uint8_t> raw((size_t)height * (rb + 1));
    // per row: the filter with the smallest sum of absolute values (the heuristic stb_image_write uses too).  One
    // tight, branch-free loop per filter type (they vectorise) instead of a switch per byte.
    std::vector<uint8_t> cand(5 * rb);
    std::vector<uint8_t> zero(rb, 0);
    for (int y = 0; y < height; y++) {
        const uint8_t* cur = rgb + (size_t)y * row_stride;
        const uint8_t* up = y ? rgb + (size_t)(y - 1) * row_stride : zero.data();
        uint8_t* c0 = cand.data();
        uint8_t* c1 = c0 + rb;
        uint8_t* c2 = c1 + rb;
        uint8_t* c3 = c2 + rb;
        uint8_t* c4 = c3 + rb;
        long sum[5] = {0, 0, 0, 0, 0};
        auto mag = [](uint8_t v) -> int { return v < 128 ? v : 256 - v; };     // |(int8_t)v|
        for (size_t i = 0; i < 3 && i < rb; i++) {                             // first pixel: a = c = 0
            const int b = up[i], v = cur[i];
            c0[i] = (uint8_t)v; c1[i] = (uint8_t)v; c2[i] = (uint8_t)(v - b); c3[i] = (uint8_t)(v - (b >> 1)); c4[i] = (uint8_t)(v - b);
        }
        for (size_t i = 3; i < rb; i++) c0[i] = cur[i];
        for (size_t i = 3; i < rb; i++) c1[i] = (uint8_t)(cur[i] - cur[i - 3]);
        for (size_t i = 3; i < rb; i++) c2[i] = (uint8_t)(cur[i] - up[i]);
        for (size_t i = 3; i < rb; i++) c3[i] = (uint8_t)(cur[i] - ((cur[i - 3] + up[i]) >> 1));
        for (size_t i = 3; i < rb; i++) {
            const int a = cur[i - 3], bb = up[i], c = up[i - 3];
            const int pa = abs(bb - c), pb = abs(a - c), pc = abs(a + bb - 2 * c);
            const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
            c4[i] = (uint8_t)(cur[i] - pr);
        }
        for (int ft = 0; ft < 5; ft++) {
            const uint8_t* cc = c0 + (size_t)ft * rb;
            long sacc = 0;
            for (size_t i = 0; i < rb; i++) sacc += mag(cc[i]);
            sum[ft] = sacc;
        }
        int best_ft = 0;
        for (int ft = 1; ft < 5; ft++)
            if (sum[ft] < sum[best_ft]) best_ft = ft;
        uint8_t* out = &raw[(size_t)y * (rb + 1)];
        out[0] = (uint8_t)best_ft;
        memcpy(out + 1, c0 + (size_t)best_ft * rb, rb);
    }
    // filtered image rows are small residuals: the run-length strategy compresses them as well as the default
    // one at level 3 and in two thirds of the time (25 MB per 4096x2048 frame: the encoder is the CLI's bottleneck)
    z_stream zs{};
    if (deflateInit2(&zs, 1, Z_DEFLATED, 15, 9, Z_RLE) != Z_OK) { err = "zlib deflate failed"; return false; }
    uLongf cl = deflateBound(&zs, (uLong)raw.size());
    std::vector<uint8_t> comp(cl);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = comp.data(); zs.avail_out = (uInt)cl;
    const int zr = deflate(&zs, Z_FINISH);
    cl = zs.total_out;
    deflateEnd(&zs);
    if (zr != Z_STREAM_END) { err = "zlib deflate failed"; return false; }
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
        if (len) crc = crc32(crc, d, len);
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
    fclose(f);
    if (!ok) err = "write error";
    return ok;
}

}  // namespace pngio
