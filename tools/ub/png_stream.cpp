// Host-streamed frames through the C ABI from ONE C++ thread (no interpreter between the calls): pixels back (fftup_submit_rgb8)
// against finished PNG files back (fftup_submit_png with the destination named: the GPU delivers the stream itself).
//   g++ -O2 -std=c++17 -I include tools/ub/png_stream.cpp -L vkresample_amd -lfftup -Wl,-rpath,$PWD/vkresample_amd -o /tmp/png_stream
//   /tmp/png_stream [frames 2048] [ring 4]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fftup.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 2048, ring = argc > 2 ? atoi(argv[2]) : 4;
    const uint32_t W = 2048, H = 1024;
    fftup_config cfg{};
    cfg.width = W; cfg.height = H; cfg.channels = 3; cfg.upscale = 2.0f; cfg.precision = 0; cfg.sharpen = 0.2f; cfg.device = 0; cfg.ring = (uint32_t)ring;
    fftup_plan* plan = nullptr;
    if (fftup_plan_create(&plan, &cfg) != FFTUP_OK) { printf("plan: %s\n", fftup_last_error()); return 1; }
    const size_t inB = (size_t)W * H * 3, outB = (size_t)4 * W * H * 3, cap = (fftup_png_bound(plan) + 63) / 64 * 64;
    std::vector<uint8_t*> pin(ring), pout(ring), ppng(ring);
    for (int s = 0; s < ring; s++) {
        pin[s] = (uint8_t*)fftup_host_alloc(inB); pout[s] = (uint8_t*)fftup_host_alloc(outB); ppng[s] = (uint8_t*)fftup_host_alloc(cap);
        // smooth structure + a little noise (what PNG carries is images; uniform noise does not compress)
        for (size_t i = 0; i < inB; i++) {
            const double x = (double)(i / 3 % W), y = (double)(i / 3 / W);
            const double v = 128 + 60 * std::sin(x / 37.0 + y / 91.0 + s + (double)(i % 3)) + 40 * std::cos(y / 53.0 - x / 201.0);
            pin[s][i] = (uint8_t)(v + (double)((i * 2654435761u >> 29) & 7) - 3.5);
        }
    }
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {                       // first pass warms
            std::vector<uint64_t> tk((size_t)ring);
            std::vector<char> live((size_t)ring, 0);
            size_t bytes = 0;
            const double t0 = now_s();
            for (int k = 0; k < frames + ring; k++) {
                const int s = k % ring;
                if (live[(size_t)s]) {
                    size_t n = outB;
                    int rc = mode ? fftup_wait_png(plan, tk[(size_t)s], ppng[s], cap, &n) : fftup_wait(plan, tk[(size_t)s]);
                    if (rc != FFTUP_OK) { printf("wait: %s\n", fftup_last_error()); return 1; }
                    bytes += n;
                    live[(size_t)s] = 0;
                }
                if (k < frames) {
                    int rc = mode ? fftup_submit_png(plan, pin[s], (size_t)W * 3, ppng[s], cap, &tk[(size_t)s])
                                  : fftup_submit_rgb8(plan, pin[s], (size_t)W * 3, pout[s], (size_t)2 * W * 3, &tk[(size_t)s]);
                    if (rc != FFTUP_OK) { printf("submit: %s\n", fftup_last_error()); return 1; }
                    live[(size_t)s] = 1;
                }
            }
            const double dt = now_s() - t0;
            if (rep) printf("%s: %d frames, ring %d: %.0f frames/s, %.1f MB back per frame, %.1f GB/s on the link\n", mode ? "PNG from the device" : "pixels", frames, ring,
                            frames / dt, bytes / 1e6 / frames, (inB + (double)bytes / frames) * frames / dt / 1e9);
        }
    }
    fftup_plan_destroy(plan);
    return 0;
}
