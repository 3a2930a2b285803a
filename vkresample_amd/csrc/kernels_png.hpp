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

// ---- Huffman code lengths by the whole workgroup.  In: the used symbols in ascending order of (frequency, symbol) -- order[k],
// wt[k] = frequency, k < used -- and len[] zeroed.  Out: len[symbol] <= MAXBITS, first_code[1 .. MAXBITS] (canonical), cnt[l] =
// codes of length l.  The same lengths as fftup_huff::huffman_lengths_sorted (csrc/huffman.hpp, the host's routine).
// What is a count or a walk is done by all threads: the depths (every leaf walks up its parent chain), the codes per length, the
// length of the k-th rarest symbol.  The two-queue merge is serial by nature (used - 1 steps, ~500 cycles each on one lane: a
// dependent chain of LDS round trips; the same chain on wave-resident registers through v_readlane measured the same -- a single
// wave issues a dependent instruction every ~5 cycles either way), the Kraft repair runs on counts in registers.
// s_memtime marks, a block of 80 used symbols (2.2 GHz): ranks 45 k cycles, these lengths 81 k, the 19-symbol code 12 k, the
// symbols' code indices 12 k, header bits 7 k, row sizes 5 k -- 84 us per block, 137 blocks side by side on 137 compute units.
template <int MAXBITS>
__device__ __forceinline__ void png_huffman_lengths(int used, const int* order, uint32_t* wt, int* par, int* cnt, uint8_t* len, int* first_code)
{
    const int t = threadIdx.x;
    if (t < 64) cnt[t] = 0;
    if (used == 1) {                                    // a complete code needs two codes: one unused sibling
        if (t == 0) { len[order[0]] = 1; len[order[0] == 0 ? 1 : 0] = 1; }
        __syncthreads();
        if (t == 0) { cnt[1] = 2; for (int l = 1; l <= MAXBITS; l++) first_code[l] = 0; first_code[1] = 0; }
        __syncthreads();
        return;
    }
    if (t == 0) {                                       // two queues: leaves [leaf, used), inner nodes [inner, next); heads in registers
        int leaf = 0, inner = used, next = used;
        uint32_t wl = wt[0], wi = 0;
        while ((used - leaf) + (next - inner) > 1) {
            int pq[2];
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (leaf < used && (inner >= next || wl <= wi)) { pq[k] = leaf++; sum += wl; if (leaf < used) wl = wt[leaf]; }
                else { pq[k] = inner++; sum += wi; if (inner < next) wi = wt[inner]; }
            }
            wt[next] = sum;
            if (inner == next) wi = sum;                // the inner queue was empty: the new node is its head
            par[pq[0]] = par[pq[1]] = next;
            next++;
        }
    }
    __syncthreads();
    const int root = 2 * used - 2;
    for (int i = t; i < used; i += 256) {               // depth of leaf i = steps to the root
        int d = 0;
        for (int n = i; n != root; n = par[n]) d++;
        atomicAdd(&cnt[d < 63 ? d : 63], 1);
    }
    __syncthreads();
    if (t == 0) {
        // Lengths beyond MAXBITS are folded back by the usual Kraft-sum repair: one code of the longest length is removed, one
        // shorter code made one bit longer, until the sum is exactly one -- as many steps as the fold overshoots, a hundred for the
        // residuals of a photograph (rare symbols sit 20-25 levels deep).  On the counts in REGISTERS: with cnt[] in LDS every
        // step was half a dozen dependent round trips, and this loop was half of the kernel's 83 us.
        int c[MAXBITS + 1];
#pragma unroll
        for (int l = 0; l <= MAXBITS; l++) c[l] = cnt[l];
        for (int l = MAXBITS + 1; l < 64; l++) c[MAXBITS] += cnt[l];
        unsigned long long total = 0;
#pragma unroll
        for (int l = 1; l <= MAXBITS; l++) total += (unsigned long long)c[l] << (MAXBITS - l);
        while (total > (1ull << MAXBITS)) {
            c[MAXBITS]--;
            bool done = false;
#pragma unroll
            for (int l = MAXBITS - 1; l > 0; l--)
                if (!done && c[l]) { c[l]--; c[l + 1] += 2; done = true; }
            total--;
        }
        c[0] = 0;
        int fc = 0;
#pragma unroll
        for (int l = 1; l <= MAXBITS; l++) { fc = (fc + c[l - 1]) << 1; first_code[l] = fc; cnt[l] = c[l]; }
        cnt[0] = 0;
        for (int l = MAXBITS + 1; l < 64; l++) cnt[l] = 0;
    }
    __syncthreads();
    for (int k = t; k < used; k += 256) {               // the rarest symbols get the longest codes
        int s0 = 0, l = MAXBITS;
        for (; l >= 1; l--) { if (k < s0 + cnt[l]) break; s0 += cnt[l]; }
        len[order[k]] = (uint8_t)l;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_png_codes(PngParams p)
{
    const int blk = blockIdx.x, t = threadIdx.x;
    const int y0 = blk * p.rows_per_block, y1 = min(p.uH, y0 + p.rows_per_block);
    __shared__ uint32_t freq[257];
    __shared__ uint8_t len[257], seq[259], clen[19];
    __shared__ uint16_t code[257], ccode[19];
    __shared__ uint32_t hdr[64], cfreq[19];
    __shared__ int hbits, hbase;
    __shared__ unsigned long long sbuf[256];
    __shared__ int first_code[17], cfirst[17], order[257];
    __shared__ uint32_t wt[2 * 257];                    // node weights: the used symbols ascending, then the inner nodes as they are made
    __shared__ int par[2 * 257], cnt[64];
    {
        uint32_t f = 0;
        for (int y = y0; y < y1; y++) f += p.rowhist[(size_t)y * 257 + t];
        freq[t] = f;
        if (t == 0) freq[256] = 1;                      // end of block
    }
    __syncthreads();
    // The sort of the symbols by frequency is a count (a symbol's rank = how many used symbols come before it), as are a symbol's
    // index among the codes of its length, the bit positions of the header's 259 code lengths (a scan) and their bits (atomicOr):
    // all threads.  Same lengths, same header bits as the host's routines (csrc/huffman.hpp: huffman_lengths, dynamic_header).
    const int used = __syncthreads_count(freq[t] != 0) + 1;            // + the end-of-block symbol
    for (int sym = t; sym < 257; sym += 256) {
        const uint32_t f = freq[sym];
        len[sym] = 0;
        if (!f) continue;
        int rank = 0;
        for (int j = 0; j < 256; j++) {
            const uint32_t g = freq[j];
            rank += (g != 0) && (g < f || (g == f && j < sym));
        }
        rank += sym == 256 ? 0 : (1u < f || (1u == f && 256 < sym));      // the end-of-block symbol, frequency 1
        order[rank] = sym;
        wt[rank] = f;
    }
    __syncthreads();
    png_huffman_lengths<15>(used, order, wt, par, cnt, len, first_code);
    // ---- block header (fftup_huff::dynamic_header): the 259 lengths, their 19-symbol code, and their bits
    for (int k = t; k < 259; k += 256) seq[k] = k < 257 ? len[k] : 1;
    if (t < 19) { cfreq[t] = 0; clen[t] = 0; }
    if (t < 64) hdr[t] = 0;
    __syncthreads();
    for (int k = t; k < 259; k += 256) atomicAdd(&cfreq[seq[k]], 1u);
    __syncthreads();
    const int cused = __syncthreads_count(t < 19 && cfreq[t] != 0);
    if (t < 19 && cfreq[t]) {
        int rank = 0;
        for (int j = 0; j < 19; j++) rank += cfreq[j] != 0 && (cfreq[j] < cfreq[t] || (cfreq[j] == cfreq[t] && j < t));
        order[rank] = t;
        wt[rank] = cfreq[t];
    }
    __syncthreads();
    png_huffman_lengths<7>(cused, order, wt, par, cnt, clen, cfirst);
    if (t < 19) {
        const int l = clen[t];
        int idx = 0;
        for (int j = 0; j < t; j++) idx += clen[j] == l;
        ccode[t] = l ? fftup_huff::reverse_bits((unsigned)(cfirst[l] + idx), l) : (uint16_t)0;
    }
    __syncthreads();
    if (t < 64) {                                       // the fixed part of the header: 17 bits + 3 per code-length-code length (<= 74 bits)
        const int perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const unsigned cl = t < 19 ? clen[perm[t]] : 0u;
        const unsigned long long nonzero = __ballot(cl != 0);           // hclen = highest sent position + 1, at least 4
        const int hclen = max(4, 64 - __builtin_clzll(nonzero | 1ull));
        if (t == 0) {
            // BFINAL, BTYPE = 2, HLIT - 257 = 0, HDIST - 1 = 1, HCLEN - 4   (1 + 2 + 5 + 5 + 4 bits)
            atomicOr(&hdr[0], (blk == p.nblocks - 1 ? 1u : 0u) | (2u << 1) | (0u << 3) | (1u << 8) | ((uint32_t)(hclen - 4) << 13));
            hbase = 17 + 3 * hclen;
        }
        if (t < hclen && cl) {
            const int pos = 17 + 3 * t, sh = pos & 31;
            atomicOr(&hdr[pos >> 5], cl << sh);
            if (sh + 3 > 32) atomicOr(&hdr[(pos >> 5) + 1], cl >> (32 - sh));
        }
    }
    __syncthreads();
    {
        unsigned long long tot;
        const unsigned nb = clen[seq[t]];
        const unsigned long long off = png_scan256((unsigned long long)nb, sbuf, &tot);
        const unsigned pos = (unsigned)hbase + (unsigned)off, sh = pos & 31;
        const uint32_t v = ccode[seq[t]];
        if (nb) {
            atomicOr(&hdr[pos >> 5], v << sh);
            if (sh + nb > 32) atomicOr(&hdr[(pos >> 5) + 1], v >> (32 - sh));
        }
        if (t == 0) {                                   // items 256 .. 258 behind the scanned 256
            unsigned pos2 = (unsigned)hbase + (unsigned)tot;
            for (int k = 256; k < 259; k++) {
                const unsigned n2 = clen[seq[k]], sh2 = pos2 & 31;
                const uint32_t v2 = ccode[seq[k]];
                if (n2) {
                    atomicOr(&hdr[pos2 >> 5], v2 << sh2);
                    if (sh2 + n2 > 32) atomicOr(&hdr[(pos2 >> 5) + 1], v2 >> (32 - sh2));
                }
                pos2 += n2;
            }
            hbits = (int)pos2;
        }
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
    // bits of row y = sum over the symbols of count * length: wave w sums rows w, w + 4, .. (its 64 lanes over the 256 symbols) and
    // parks them in row_off; then an exclusive scan over the block's rows turns them into offsets -- no barrier per row
    {
        const int w = t >> 6, l = t & 63;
        const unsigned l0 = len[l], l1 = len[64 + l], l2 = len[128 + l], l3 = len[192 + l];
        for (int y = y0 + w; y < y1; y += 4) {
            const uint32_t* h = p.rowhist + (size_t)y * 257;
            unsigned long long v = (unsigned long long)h[l] * l0 + (unsigned long long)h[64 + l] * l1 + (unsigned long long)h[128 + l] * l2 +
                                   (unsigned long long)h[192 + l] * l3;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            if (l == 0) p.row_off[y] = v;
        }
    }
    __syncthreads();                                    // (row_off written by this workgroup's waves: visible to it behind the barrier)
    unsigned long long carry = (unsigned long long)hbits;
    for (int base = y0; base < y1; base += 256) {
        const int y = base + t;
        const unsigned long long v = y < y1 ? p.row_off[y] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = png_scan256(v, sbuf, &tot);
        if (y < y1) p.row_off[y] = carry + ex;
        carry += tot;
    }
    if (t == 0) { p.hdr_bits[blk] = (uint32_t)hbits; p.block_bits[blk] = carry + len[256]; }
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

// CRC-32 (the PNG chunk checksum) of every whole 4 KB piece of the finished stream.  A thread takes 256 bytes (four table
// lookups per word, tables built in LDS); the sixteen threads of a piece join their values -- crc(A || B) = shift_|B|(crc(A)) ^
// crc(B), shift_n = the GF(2)-linear operator "append n zero bytes": thread k of the piece applies shift_256^(15-k), a 32 x 32 bit
// matrix from a table the host computed once (p.crc_shift), and the sixteen results are exclusive-ored across lanes.
// (One thread per 4 KB piece -- round 4 -- left 3 500 threads with 1 024 dependent steps each: 75 us alone for 14 MB.)
// The host joins the pieces the same way (shift_4096, crc32.hpp) in ~0.1 ms; its own pass over the 14 MB was 6 ms.
__global__ __launch_bounds__(256) void k_png_crc(PngParams p)
{
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t M[16][32];                       // M[j] = shift_256^j, column n = image of bit n
    const int t = threadIdx.x;
    T[0][t] = fftup_crc::crc32_table_entry((uint32_t)t);
    M[t >> 5][t & 31] = p.crc_shift[t];
    M[8 + (t >> 5)][t & 31] = p.crc_shift[256 + t];
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        T[k][t] = (T[k - 1][t] >> 8) ^ T[0][T[k - 1][t] & 255];
        __syncthreads();
    }
    const unsigned long long pieces = p.meta[0] / 4096;
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + t, piece = g >> 4;
    const int sub = (int)(g & 15);
    if (piece >= pieces) return;                          // (whole groups of sixteen lanes)
    const uint32_t* w = p.stream + piece * 1024 + sub * 64;
    uint32_t crc = 0xFFFFFFFFu;
#pragma unroll 8
    for (int k = 0; k < 64; k++) {
        const uint32_t a = crc ^ w[k];
        crc = T[3][a & 255] ^ T[2][(a >> 8) & 255] ^ T[1][(a >> 16) & 255] ^ T[0][a >> 24];
    }
    crc = ~crc;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) v ^= ((crc >> i) & 1u) ? M[15 - sub][i] : 0u;       // shift_256^(15 - sub) applied to this part's value
    for (int o = 8; o; o >>= 1) v ^= __shfl_xor(v, o);
    if (sub == 0) p.crc_parts[piece] = v;
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
