// kernels_png.hpp -- the frame leaves the GPU as a finished PNG data stream (round 4).
//
// The batched mode's host time per 4096x2048 frame is PNG work (36 ms of encoding against 0.06 ms of kernels) and its PCIe time
// is the 25 MB of 8-bit pixels; both shrink when the device does the encoding: the row filters are per-byte arithmetic on
// neighbours, and the deflate stream of this encoder is Huffman-only (csrc/huffman.hpp; filtered rows of an interpolated image
// hold no repeats worth a match search), i.e. a prefix sum of code lengths and a scatter of bits -- no serial dependency.
//
//   k_png_filter  one workgroup per image row: the five sums of |residual| (stb_image_write's / libpng's heuristic), the
//                 winning filter's residuals -> raw[y][1 + 3 uW], the row's byte histogram, its Adler-32 partial sums
//   k_png_codes   one workgroup per deflate block (rows_per_block rows): block histogram -> length-limited canonical code
//                 (one lane, the shared host/device routine) -> code table, block header bits, bit size of every row
//   k_png_layout  one workgroup: bit offsets of the blocks (scan), Adler-32 of the whole stream from the rows' sums, zlib
//                 header, trailer, byte count
//   k_png_pack    one workgroup per row: a thread's 1/256 of the row -> its bit offset by a workgroup scan -> bits into the stream
//                 (whole words it owns: plain stores; the first and last word it shares with a neighbour: atomicOr on the
//                 zeroed buffer)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "crc32.hpp"
#include "huffman.hpp"
#include "png_params.hpp"

namespace fftup {

__device__ __forceinline__ unsigned png_mag8(unsigned r) { r &= 255u; return r < 256u - r ? r : 256u - r; }

__global__ __launch_bounds__(256) void k_png_filter(PngParams p)
{
    const int y = blockIdx.x, t = threadIdx.x;
    const unsigned rb = 3u * (unsigned)p.uW, L = rb + 1;
    const uint8_t* cur = p.rgb + (size_t)y * rb;
    const uint8_t* up = y ? cur - rb : nullptr;
    __shared__ unsigned ssum[5];
    __shared__ unsigned shist[256];
    __shared__ unsigned long long sad[2];
    __shared__ int sbest;
    if (t < 5) ssum[t] = 0;
    if (t < 2) sad[t] = 0;
    shist[t] = 0;
    __syncthreads();
    unsigned s[5] = {0, 0, 0, 0, 0};
    for (unsigned i = t; i < rb; i += 256) {
        const int v = cur[i], a = i >= 3 ? cur[i - 3] : 0, b = up ? up[i] : 0, c = (up && i >= 3) ? up[i - 3] : 0;
        const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
        const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        s[0] += png_mag8((unsigned)v);
        s[1] += png_mag8((unsigned)(v - a));
        s[2] += png_mag8((unsigned)(v - b));
        s[3] += png_mag8((unsigned)(v - ((a + b) >> 1)));
        s[4] += png_mag8((unsigned)(v - pr));
    }
    for (int k = 0; k < 5; k++) {
        unsigned v = s[k];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) atomicAdd(&ssum[k], v);
    }
    __syncthreads();
    if (t == 0) {
        int best = 0;
        for (int k = 1; k < 5; k++)
            if (ssum[k] < ssum[best]) best = k;
        sbest = best;
    }
    __syncthreads();
    const int ft = sbest;
    uint8_t* out = p.raw + (size_t)y * L;
    unsigned long long s1 = 0, s2 = 0;
    for (unsigned i = t; i < rb; i += 256) {
        const int v = cur[i], a = i >= 3 ? cur[i - 3] : 0, b = up ? up[i] : 0, c = (up && i >= 3) ? up[i - 3] : 0;
        int pred;
        if (ft == 0) pred = 0;
        else if (ft == 1) pred = a;
        else if (ft == 2) pred = b;
        else if (ft == 3) pred = (a + b) >> 1;
        else {
            const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
            pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        }
        const unsigned r = (unsigned)(v - pred) & 255u;
        out[1 + i] = (uint8_t)r;
        atomicAdd(&shist[r], 1u);
        s1 += r;
        s2 += (unsigned long long)(L - (i + 1)) * r;
    }
    if (t == 0) {
        out[0] = (uint8_t)ft;
        atomicAdd(&shist[ft], 1u);
        s1 += (unsigned)ft;
        s2 += (unsigned long long)L * (unsigned)ft;
    }
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((t & 63) == 0) { atomicAdd(&sad[0], s1); atomicAdd(&sad[1], s2); }
    __syncthreads();
    p.rowhist[(size_t)y * 257 + t] = shist[t];
    if (t == 0) { p.rowhist[(size_t)y * 257 + 256] = 0; p.rowsum[2 * y] = sad[0]; p.rowsum[2 * y + 1] = sad[1]; }
}

__global__ __launch_bounds__(256) void k_png_codes(PngParams p)
{
    const int blk = blockIdx.x, t = threadIdx.x;
    const int y0 = blk * p.rows_per_block, y1 = min(p.uH, y0 + p.rows_per_block);
    __shared__ uint32_t freq[257];
    __shared__ uint8_t len[257];
    __shared__ uint16_t code[257];
    __shared__ uint32_t hdr[64];
    __shared__ int hbits;
    __shared__ unsigned long long red[4];
    __shared__ fftup_huff::Work work;
    __shared__ int first_code[17];
    {
        uint32_t f = 0;
        for (int y = y0; y < y1; y++) f += p.rowhist[(size_t)y * 257 + t];
        freq[t] = f;
        if (t == 0) freq[256] = 1;                      // end of block
    }
    __syncthreads();
    // what one lane would do in ~10^5 dependent steps is done by all where it is a count: the sort of the symbols by frequency
    // (a symbol's rank = how many used symbols come before it) and a symbol's index among the codes of its length; the tree
    // itself (two queues, ~500 steps), the length limit and the block header stay with lane 0.  Working arrays in LDS: private
    // ones of this size would be scratch memory.
    const int used = __syncthreads_count(freq[t] != 0) + 1;            // + the end-of-block symbol
    for (int sym = t; sym < 257; sym += 256) {
        const uint32_t f = freq[sym];
        if (!f) continue;
        int rank = 0;
        for (int j = 0; j < 257; j++) {
            const uint32_t g = freq[j];
            rank += (g != 0) && (g < f || (g == f && j < sym));
        }
        work.order[rank] = sym;
    }
    __syncthreads();
    if (t == 0) {
        fftup_huff::huffman_lengths_sorted(freq, 257, used, 15, len, work);
        fftup_huff::canonical_first_codes(len, 257, 15, work);
        for (int l = 0; l < 17; l++) first_code[l] = work.next[l];      // (the header's own 19-symbol code reuses the work arrays)
        hbits = fftup_huff::dynamic_header(len, blk == p.nblocks - 1, hdr, work);
    }
    __syncthreads();
    for (int sym = t; sym < 257; sym += 256) {
        const int l = len[sym];
        int idx = 0;
        for (int j = 0; j < sym; j++) idx += len[j] == l;
        code[sym] = l ? fftup_huff::reverse_bits((unsigned)(first_code[l] + idx), l) : (uint16_t)0;
    }
    __syncthreads();
    p.tab[(size_t)blk * 257 + t] = code[t] | ((uint32_t)len[t] << 16);
    if (t == 0) p.tab[(size_t)blk * 257 + 256] = code[256] | ((uint32_t)len[256] << 16);
    if (t < 64) p.hdr[(size_t)blk * 64 + t] = hdr[t];
    unsigned long long off = (unsigned long long)hbits;
    for (int y = y0; y < y1; y++) {                     // bits of row y = sum over the symbols of count * length
        unsigned long long v = (unsigned long long)p.rowhist[(size_t)y * 257 + t] * len[t];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        const unsigned long long rowbits = red[0] + red[1] + red[2] + red[3];
        if (t == 0) p.row_off[y] = off;
        off += rowbits;
        __syncthreads();
    }
    if (t == 0) { p.hdr_bits[blk] = (uint32_t)hbits; p.block_bits[blk] = off + len[256]; }
}

// exclusive scan of one value per thread over the 256 threads of the workgroup (all threads call it)
__device__ __forceinline__ unsigned long long png_scan256(unsigned long long v, unsigned long long* buf, unsigned long long* total)
{
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned long long u = t >= o ? buf[t - o] : 0;
        __syncthreads();
        buf[t] += u;
        __syncthreads();
    }
    const unsigned long long incl = buf[t];
    *total = buf[255];
    __syncthreads();
    return incl - v;
}

// One workgroup: bit offsets of the blocks (scan of their sizes), Adler-32 of the whole stream from the rows' partial sums
// (rows of L bytes, a_i = 1 + sum of the earlier rows' byte sums: b = L sum a_i + sum s2_i, a = a_R; everything modulo 65521),
// zlib header, trailer, byte count.  Each thread takes a contiguous share of the blocks and of the rows.
__global__ __launch_bounds__(256) void k_png_layout(PngParams p)
{
    const int t = threadIdx.x;
    __shared__ unsigned long long buf[256];
    unsigned long long total;
    const int bchunk = (p.nblocks + 255) / 256, b0 = min(p.nblocks, t * bchunk), b1 = min(p.nblocks, b0 + bchunk);
    unsigned long long mine = 0;
    for (int b = b0; b < b1; b++) mine += p.block_bits[b];
    unsigned long long pos = 16 + png_scan256(mine, buf, &total);      // 16: the two bytes of the zlib header
    for (int b = b0; b < b1; b++) { p.block_start[b] = pos; pos += p.block_bits[b]; }
    const unsigned long long data_bytes = (16 + total + 7) / 8;
    const unsigned long long M = 65521, L = 3ull * (unsigned long long)p.uW + 1;
    const int rchunk = (p.uH + 255) / 256, y0 = min(p.uH, t * rchunk), y1 = min(p.uH, y0 + rchunk);
    unsigned long long s1 = 0;
    for (int y = y0; y < y1; y++) s1 += p.rowsum[2 * y];
    unsigned long long all1;
    unsigned long long a = 1 + png_scan256(s1, buf, &all1);            // Adler's a before this thread's first row
    unsigned long long part = 0;
    for (int y = y0; y < y1; y++) {
        part += (L % M) * (a % M) + p.rowsum[2 * y + 1] % M;
        a += p.rowsum[2 * y];
    }
    unsigned long long bsum;
    (void)png_scan256(part % M, buf, &bsum);
    if (t == 0) {
        // The code lengths come out of a heuristic length limit; the buffer's bound (9 1/8 bits per symbol) is an argument, not
        // a guarantee the device can lean on: a stream that would not fit is not written at all (k_png_pack, k_png_crc and
        // k_png_deliver read the verdict from meta), fftup_wait_png reports it.  (+ 8: the words the last atomicOr may touch)
        if (data_bytes + 4 + 8 > p.capacity) { p.meta[0] = 0; p.meta[1] = 0; p.meta[2] = data_bytes + 4; return; }
        p.meta[2] = 0;
        const uint32_t adler = (uint32_t)(((bsum % M) << 16) | ((1 + all1) % M));
        atomicOr(&p.stream[0], 0x0178u);                     // bytes 0x78 0x01: deflate, 32 KB window, fastest level
        for (int k = 0; k < 4; k++) {                        // trailer, most significant byte first
            const unsigned long long at = data_bytes + (unsigned)k;
            atomicOr(&p.stream[at >> 2], ((adler >> (24 - 8 * k)) & 255u) << (8 * (at & 3)));
        }
        p.meta[0] = data_bytes + 4;
        p.meta[1] = adler;
    }
}

__global__ __launch_bounds__(256) void k_png_pack(PngParams p)
{
    const int y = blockIdx.x, t = threadIdx.x, blk = y / p.rows_per_block;
    const unsigned L = 3u * (unsigned)p.uW + 1;
    if (p.meta[2]) return;                                // (uniform) the stream does not fit its buffer: k_png_layout
    __shared__ uint32_t tab[257];
    __shared__ unsigned long long scan[256];
    tab[t] = p.tab[(size_t)blk * 257 + t];
    if (t == 0) tab[256] = p.tab[(size_t)blk * 257 + 256];
    __syncthreads();
    // the row goes through LDS: whole words, coalesced, instead of every thread walking its own 1/256 of the row in global memory
    // (rows too long for the launch's LDS are read in place)
    extern __shared__ uint32_t rowbuf[];
    const uint8_t* src = p.raw + (size_t)y * L;
    if (p.row_in_lds) {
        const size_t addr = (size_t)src;
        const unsigned off = (unsigned)(addr & 3), nw = (off + L + 3) / 4;
        const uint32_t* w = (const uint32_t*)(addr - off);             // (raw has eight bytes of slack behind the last row)
        for (unsigned k = t; k < nw; k += 256) rowbuf[k] = w[k];
        __syncthreads();
        src = (const uint8_t*)rowbuf + off;
    }
    const unsigned chunk = (L + 255) / 256, lo = min(L, t * chunk), hi = min(L, lo + chunk);
    unsigned long long bits = 0;
    for (unsigned i = lo; i < hi; i++) bits += tab[src[i]] >> 16;
    const bool last_row = y == p.uH - 1 || (y + 1) % p.rows_per_block == 0;
    const bool tail = last_row && hi == L && lo < L;     // the thread with the row's last symbol also sends the end of block
    scan[t] = bits;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                  // inclusive scan
        const unsigned long long v = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const unsigned long long start = p.block_start[blk] + p.row_off[y] + scan[t] - bits;
    if (y % p.rows_per_block == 0 && t == 0) {           // block header: <= 59 words, all by atomicOr (it may share words at both ends)
        const unsigned long long at = p.block_start[blk];
        const unsigned sh = (unsigned)(at & 31);
        const int nw = ((int)p.hdr_bits[blk] + 31) / 32;
        for (int k = 0; k < nw; k++) {
            const uint32_t w = p.hdr[(size_t)blk * 64 + k];
            if (!w) continue;
            atomicOr(&p.stream[(at >> 5) + k], w << sh);
            if (sh && (w >> (32 - sh))) atomicOr(&p.stream[(at >> 5) + k + 1], w >> (32 - sh));
        }
    }
    if (lo >= hi) return;
    // the low (start & 31) bits of the first word belong to the neighbour before: zeros here, the word goes out by atomicOr
    uint32_t* outw = p.stream + (start >> 5);
    unsigned long long acc = 0;
    unsigned n = (unsigned)(start & 31);
    bool first = true;
    auto flush = [&]() {
        if (first) { atomicOr(outw, (uint32_t)acc); first = false; }
        else *outw = (uint32_t)acc;
        outw++;
        acc >>= 32;
        n -= 32;
    };
    for (unsigned i = lo; i < hi; i++) {
        const uint32_t e = tab[src[i]];
        acc |= (unsigned long long)(e & 0xffffu) << n;
        n += e >> 16;
        if (n >= 32) flush();
    }
    if (tail) {
        acc |= (unsigned long long)(tab[256] & 0xffffu) << n;
        n += tab[256] >> 16;
        if (n >= 32) flush();
    }
    if (n) atomicOr(outw, (uint32_t)acc);
}

// CRC-32 (the PNG chunk checksum) of every whole 4 KB piece of the finished stream, one piece per thread, four table lookups
// per word (tables built in LDS).  The host joins the pieces -- crc(A || B) = shift_|B|(crc(A)) ^ crc(B), one fixed 32 x 32
// matrix over GF(2) for |B| = 4 KB -- in ~0.1 ms; its own pass over the 14 MB was 6 ms, nine tenths of fftup_wait_png.
__global__ __launch_bounds__(256) void k_png_crc(PngParams p)
{
    __shared__ uint32_t T[4][256];
    const int t = threadIdx.x;
    T[0][t] = fftup_crc::crc32_table_entry((uint32_t)t);
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        T[k][t] = (T[k - 1][t] >> 8) ^ T[0][T[k - 1][t] & 255];
        __syncthreads();
    }
    const unsigned long long pieces = p.meta[0] / 4096;
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + t;
    if (i >= pieces) return;
    const uint32_t* w = p.stream + i * 1024;
    uint32_t crc = 0xFFFFFFFFu;
    for (int k = 0; k < 1024; k++) {
        const uint32_t a = crc ^ w[k];
        crc = T[3][a & 255] ^ T[2][(a >> 8) & 255] ^ T[1][(a >> 16) & 255] ^ T[0][a >> 24];
    }
    p.crc_parts[i] = ~crc;
}

// The stream goes to the caller's page-locked buffer by the GPU's own stores (the buffer is mapped into the device's address space),
// sized by the byte count the device already knows -- no host round trip between "size known" and "copy issued".  The file's
// stream starts at byte 41 of the buffer (signature, IHDR, IDAT framing): every aligned output word is the top byte of one stream
// word and the low three of the next.  Word 10 (bytes 40-43) is written with a zero byte 40, which the host fills in afterwards.
__global__ __launch_bounds__(256) void k_png_deliver(PngParams p, uint32_t* dst /* the buffer, 16-byte aligned */)
{
    const unsigned long long n = p.meta[0];                              // stream bytes
    const unsigned long long words = (41 + n + 3) / 4;                   // buffer words that hold stream bytes: 10 .. words - 1
    const unsigned long long groups = (words + 3) / 4;
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (unsigned long long)gridDim.x * 256) {
        if (g < 2) continue;                                             // words 0 .. 7: header bytes only (host)
        uint32_t o[4];
        for (int k = 0; k < 4; k++) {
            const long long m = (long long)(4 * g + k);                  // o[k] = stream bytes 4m - 41 .. 4m - 38
            const uint32_t lo = m >= 11 ? p.stream[m - 11] : 0u, hi = m >= 10 ? p.stream[m - 10] : 0u;
            o[k] = (lo >> 24) | (hi << 8);
        }
        if (g == 2) { dst[10] = o[2]; dst[11] = o[3]; }                  // (words 8, 9: header)
        else *(uint4*)(dst + 4 * g) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace fftup
