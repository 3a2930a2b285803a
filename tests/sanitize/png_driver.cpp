// Sanitizer harness for the CLI's PNG codec (tests/test_sanitize.py builds it with -fsanitize=address,undefined):
// decodes every file given on the command line, re-encodes what decoded, prints one line per file.
#include <cstdio>

#include "png_codec.hpp"

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; i++) {
        std::vector<uint8_t> rgb;
        int w = 0, h = 0, ch = 0;
        std::string err;
        if (!pngio::load_rgb8(argv[i], rgb, w, h, ch, err)) {
            printf("%s: rejected (%s)\n", argv[i], err.c_str());
            continue;
        }
        std::string out = std::string(argv[i]) + ".out.png";
        const bool ok = pngio::write_rgb8(out, rgb.data(), w, h, (size_t)w * 3, err);
        printf("%s: %dx%d ch %d %s\n", argv[i], w, h, ch, ok ? "ok" : err.c_str());
    }
    return 0;
}
