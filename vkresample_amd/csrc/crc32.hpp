// crc32.hpp -- the ONE CRC-32 of the library (ISO 3309 / the zlib polynomial, the PNG chunk checksum): table entry, a host
// routine (eight bytes per step), the join of per-piece values, and the same table for the device kernel (k_png_crc builds it in
// LDS from crc32_table_entry).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>

namespace fftup_crc {

constexpr uint32_t kPoly = 0xEDB88320u;

// T[0][i]: the CRC register after feeding byte i into a zero register
__host__ __device__ constexpr uint32_t crc32_table_entry(uint32_t i)
{
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? kPoly ^ (c >> 1) : c >> 1;
    return c;
}

// crc of (data so far || p[0..n)); start with crc = 0
inline uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) T[0][i] = crc32_table_entry(i);
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) T[k][i] = (T[k - 1][i] >> 8) ^ T[0][T[k - 1][i] & 255];
    });
    crc = ~crc;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = T[7][a & 255] ^ T[6][(a >> 8) & 255] ^ T[5][(a >> 16) & 255] ^ T[4][a >> 24] ^
              T[3][b & 255] ^ T[2][(b >> 8) & 255] ^ T[1][(b >> 16) & 255] ^ T[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 255] ^ (crc >> 8);
    return ~crc;
}

// The operator "append 2^log2_bits zero bits" on a CRC value is linear over GF(2): a 32 x 32 bit matrix, column n = image of bit n.
// The matrix of one zero bit is the polynomial and a shift; squaring it log2_bits times gives the operator.
inline uint32_t crc32_matrix_times(const uint32_t* m, uint32_t v)
{
    uint32_t s = 0;
    for (int i = 0; v; v >>= 1, i++)
        if (v & 1) s ^= m[i];
    return s;
}
inline void crc32_shift_matrix(int log2_bits, uint32_t out[32])
{
    uint32_t a[32], b[32];
    a[0] = kPoly;
    for (int n = 1; n < 32; n++) a[n] = 1u << (n - 1);
    for (int k = 0; k < log2_bits; k++) {
        for (int n = 0; n < 32; n++) b[n] = crc32_matrix_times(a, a[n]);
        memcpy(a, b, sizeof a);
    }
    memcpy(out, a, sizeof a);
}
// crc(A || B) from crc(A) and crc(B) for |B| = 4096: crc32_shift_4096(crc(A)) ^ crc(B)   (2^15 bits = 4096 bytes)
inline uint32_t crc32_shift_4096(uint32_t crc)
{
    static uint32_t M[32];
    static std::once_flag once;
    std::call_once(once, [] { crc32_shift_matrix(15, M); });
    return crc32_matrix_times(M, crc);
}
// powers 0 .. 15 of the operator "append 256 zero bytes" (2^11 bits): what the sixteen threads of a 4 KB piece apply to their
// 256-byte values before the exclusive-or across lanes (k_png_crc)
inline void crc32_shift_256_powers(uint32_t out[16][32])
{
    uint32_t m1[32];
    crc32_shift_matrix(11, m1);
    for (int n = 0; n < 32; n++) { out[0][n] = 1u << n; out[1][n] = m1[n]; }
    for (int j = 2; j < 16; j++)
        for (int n = 0; n < 32; n++) out[j][n] = crc32_matrix_times(m1, out[j - 1][n]);
}

}  // namespace fftup_crc
