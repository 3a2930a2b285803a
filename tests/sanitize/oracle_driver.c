/* Sanitizer harness for the CPU oracle (tests/test_sanitize.py builds oracle/fftup_oracle.c together with this file
 * under -fsanitize=address,undefined): a sweep of small configurations incl. odd radices, non-integer scales and all
 * three precisions; prints a checksum so that the run cannot be optimised away. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { uint32_t width, height; float upscale; uint32_t precision; float sharpen; uint32_t u8_wrap; } orc_config;
int orc_out_dims(const orc_config*, uint32_t*, uint32_t*);
int orc_check(const orc_config*);
int orc_upscale_rgb8(const orc_config*, const uint8_t*, double*, double*, uint8_t*);

int main(void)
{
    static const struct { uint32_t w, h; float u; } cases[] = {{16, 8, 2.0f}, {20, 12, 2.0f}, {60, 42, 2.0f}, {16, 8, 1.5f},
                                                               {32, 16, 1.0f}, {24, 16, 3.0f}, {14, 10, 2.5f}, {2, 2, 2.0f}};
    uint64_t sum = 0;
    for (unsigned i = 0; i < sizeof cases / sizeof cases[0]; i++)
        for (uint32_t p = 0; p < 3; p++)
            for (uint32_t wrap = 0; wrap < 2; wrap++) {
                orc_config c = {cases[i].w, cases[i].h, cases[i].u, p, 0.2f, wrap};
                if (orc_check(&c)) continue;
                uint32_t uW, uH;
                orc_out_dims(&c, &uW, &uH);
                uint8_t* rgb = malloc(3u * c.width * c.height);
                uint32_t s = 12345u + i;
                for (uint32_t k = 0; k < 3u * c.width * c.height; k++) { s = s * 1664525u + 1013904223u; rgb[k] = (uint8_t)(s >> 24); }
                double* pre = malloc(sizeof(double) * 3u * uW * uH);
                double* out = malloc(sizeof(double) * 3u * uW * uH);
                uint8_t* o8 = malloc(3u * uW * uH);
                if (orc_upscale_rgb8(&c, rgb, pre, out, o8)) { printf("case %u failed\n", i); return 1; }
                for (uint32_t k = 0; k < 3u * uW * uH; k++) sum += o8[k];
                free(rgb); free(pre); free(out); free(o8);
            }
    printf("oracle sanitizer sweep ok, checksum %llu\n", (unsigned long long)sum);
    return 0;
}
