// ref_host_shim.cpp -- thin C exports around the REFERENCE's own header-only host libraries,
// compiled where they lie under /root/reference (never copied): half_lib/half.hpp (fp16 host type
// used by the -p 2 pack loop, VkResample.cpp:1670-1683) and stb_image (PNG decode forced to 3
// channels, VkResample.cpp:1362; PNG encode VkResample.cpp:1754).  Output: oracle/_ref/libref_host.so.
// TEST INFRASTRUCTURE: used to pin oracle/fftup_oracle.c's load conversion and the PNG codec of the
// CLI against the reference's real code.  Absent on the GPU box (no /root/reference there).
#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image/stb_image.h"
#include "stb_image/stb_image_write.h"
#include "half_lib/half.hpp"

#include <cstdint>
#include <cstring>

using half_float::half;

extern "C" {

// the expression of VkResample.cpp:1676, evaluated by the reference's half type
__attribute__((visibility("default"))) double ref_pack_half(unsigned char v)
{
    half h;
    h = (half)v / 255.0;          // assignment, as in the reference: double -> float -> half
    return (double)(float)h;
}
// VkResample.cpp:1644
__attribute__((visibility("default"))) double ref_pack_float(unsigned char v)
{
    float f = (float)v / 255.0;
    return (double)f;
}
// VkResample.cpp:1715 / 1741 with the value held in float / half
__attribute__((visibility("default"))) unsigned char ref_unpack_float(float x)
{
    unsigned char o = 255.0 * x;
    return o;
}
__attribute__((visibility("default"))) double ref_half_round(double x)
{
    return (double)(float)half((float)x);
}
// stbi_load(file, &w, &h, &ch, 3) -> caller frees with ref_free
__attribute__((visibility("default"))) unsigned char* ref_png_load_rgb(const char* path, int* w, int* h, int* ch)
{
    return stbi_load(path, w, h, ch, 3);
}
__attribute__((visibility("default"))) void ref_free(void* p) { stbi_image_free(p); }
__attribute__((visibility("default"))) int ref_png_write_rgb(const char* path, int w, int h, const unsigned char* rgb)
{
    return stbi_write_png(path, w, h, 3, rgb, w * 3);
}
}
