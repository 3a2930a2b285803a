// png_params.hpp -- argument block of the device-side PNG encoder's kernels (csrc/kernels_png.hpp)
#pragma once
#include <cstdint>

namespace fftup {

struct PngParams {
    const uint8_t* rgb;          // [uH][uW][3], the frame's 8-bit image
    uint8_t* raw;                // [uH][1 + 3 uW] filter type + residuals
    uint32_t* rowhist;           // [uH][257]   (symbol 256 unused: the end-of-block symbol is counted per block)
    unsigned long long* rowsum;  // [uH][2]: sum of the row's stream bytes, sum of (L - i) * byte_i  (Adler-32 partials)
    uint32_t* tab;               // [nblocks][257] code | length << 16
    uint32_t* hdr;               // [nblocks][64] block header bits
    uint32_t* hdr_bits;          // [nblocks]
    unsigned long long* block_bits;   // [nblocks] header + symbols + end of block
    unsigned long long* block_start;  // [nblocks] bit offset in the stream
    unsigned long long* row_off; // [uH] bit offset of the row inside its block (header included)
    uint32_t* stream;            // zlib stream, zeroed before the frame
    unsigned long long* meta;    // [0] bytes of the stream (with header and trailer), [1] Adler-32, [2] != 0: the stream would not fit
                                 //     `capacity` -- nothing was packed, [0] is 0 and [2] the bytes it needs
    unsigned long long capacity; // bytes of `stream`
    uint32_t* crc_parts;         // CRC-32 of every whole 4 KB piece of the stream (k_png_crc)
    const uint32_t* crc_shift;   // [16][32]: powers of the operator "append 256 zero bytes" (crc32.hpp: crc32_shift_256_powers)
    int uW, uH, rows_per_block, nblocks;
    int row_in_lds;              // k_png_pack was given LDS for a whole row
};

}  // namespace fftup
