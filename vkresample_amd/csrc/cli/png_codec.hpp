// png_codec.hpp -- minimal PNG reader/writer over zlib for the CLI.
// Replaces the two stb_image entry points the reference uses: stbi_load(name,&w,&h,&ch,3)
// (VkResample.cpp:1362, 1630: decode forced to 3 channels, alpha dropped, 16-bit -> high byte, low-bit
// grey scaled to 0..255) and stbi_write_png(name,w,h,3,data,stride) (VkResample.cpp:1754).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pngio {
// Decodes any non-animated PNG (colour types 0,2,3,4,6; bit depths 1..16; Adam7 or not) to 8-bit RGB.
// Returns false and sets err on failure.  channels_in_file mirrors stbi_load's comp output.
bool load_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height, int& channels_in_file,
               std::string& err);
// Writes 8-bit RGB (row stride in bytes) as a PNG.
bool write_rgb8(const std::string& path, const uint8_t* rgb, int width, int height, size_t row_stride, std::string& err);
}  // namespace pngio
