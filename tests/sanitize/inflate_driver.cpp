// Sanitizer + differential harness for csrc/cli/inflate_fast.hpp (tests/test_sanitize.py): streams produced by zlib at every level
// and strategy from data of different statistics must decode to the input; damaged streams (truncated, bit-flipped, wrong length)
// must be refused or -- where the damage does not matter -- decode to exactly what zlib makes of them.  Never a wild access
// (ASan/UBSan).
#include <zlib.h>

#include <cstdio>
#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#include "../../vkresample_amd/csrc/cli/inflate_fast.hpp"

static std::vector<uint8_t> deflate_with(const std::vector<uint8_t>& src, int level, int strategy)
{
    z_stream zs{};
    deflateInit2(&zs, level, Z_DEFLATED, 15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, (uLong)src.size()) + 64);
    zs.next_in = (Bytef*)src.data(); zs.avail_in = (uInt)src.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

int main()
{
    std::mt19937 rng(777);
    static inflate_fast::Tables T;
    int fails = 0, accepted_damaged = 0, refused_damaged = 0, streams = 0;
    std::vector<std::vector<uint8_t>> inputs;
    inputs.push_back({});
    inputs.push_back({42});
    inputs.push_back(std::vector<uint8_t>(100000, 0));
    { std::vector<uint8_t> v(300000); for (auto& x : v) x = (uint8_t)rng(); inputs.push_back(v); }
    { std::vector<uint8_t> v(300000); for (auto& x : v) { int k = 0; while (k < 30 && (rng() & 1)) k++; x = (uint8_t)(k * 7); } inputs.push_back(v); }
    { std::vector<uint8_t> v(400000); for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)((i % 613) * (i % 7) + (rng() % 3)); inputs.push_back(v); }   // long repeats
    { std::vector<uint8_t> v(200000); for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)(128 + 100 * std::sin(i / 300.0) + (rng() % 5)); inputs.push_back(v); }
    { std::vector<uint8_t> v; for (int s = 0; s < 26; s++) v.insert(v.end(), (size_t)1 << (s < 18 ? s : 17), (uint8_t)(s * 9)); std::shuffle(v.begin(), v.end(), rng); inputs.push_back(v); }  // deep trees
    const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    for (const auto& src : inputs)
        for (int level : {0, 1, 3, 6, 9})
            for (int st : strategies) {
                const std::vector<uint8_t> z = deflate_with(src, level, st);
                std::vector<uint8_t> out(src.size() + 8, 0xCD);
                streams++;
                if (!inflate_fast::zlib_decode(z.data(), z.size(), out.data(), src.size(), T) || (!src.empty() && memcmp(out.data(), src.data(), src.size()))) {
                    printf("FAILED: %zu bytes, level %d, strategy %d\n", src.size(), level, st);
                    fails++;
                }
                // wrong expected length
                if (!src.empty() && inflate_fast::zlib_decode(z.data(), z.size(), out.data(), src.size() - 1, T)) { printf("accepted a short output\n"); fails++; }
                std::vector<uint8_t> big(src.size() + 9);
                if (inflate_fast::zlib_decode(z.data(), z.size(), big.data(), src.size() + 1, T)) { printf("accepted a long output\n"); fails++; }
                // damage: truncations and bit flips; whatever is accepted must be what zlib decodes
                for (int k = 0; k < 24 && z.size() > 8; k++) {
                    std::vector<uint8_t> bad = z;
                    if (k % 2) bad.resize(rng() % z.size());
                    else bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
                    std::vector<uint8_t> mine(src.size() + 8), ref(src.size() + 8);
                    const bool ok = inflate_fast::zlib_decode(bad.data(), bad.size(), mine.data(), src.size(), T);
                    if (ok) {
                        uLongf dl = (uLongf)src.size();
                        const int zr = uncompress(ref.data(), &dl, bad.data(), (uLong)bad.size());
                        if (zr != Z_OK || dl != src.size() || (!src.empty() && memcmp(mine.data(), ref.data(), src.size()))) { printf("accepted a stream zlib refuses or reads differently\n"); fails++; }
                        accepted_damaged++;
                    } else refused_damaged++;
                }
            }
    // garbage
    for (int k = 0; k < 2000; k++) {
        std::vector<uint8_t> g(6 + rng() % 300);
        for (auto& x : g) x = (uint8_t)rng();
        if (k % 2) { g[0] = 0x78; g[1] = 0x9c; }
        std::vector<uint8_t> out(5000 + 8);
        if (inflate_fast::zlib_decode(g.data(), g.size(), out.data(), 5000, T)) {
            std::vector<uint8_t> ref(5000);
            uLongf dl = 5000;
            if (uncompress(ref.data(), &dl, g.data(), (uLong)g.size()) != Z_OK || dl != 5000 || memcmp(ref.data(), out.data(), 5000)) { printf("accepted garbage\n"); fails++; }
        }
    }
    printf("%d streams, damaged: %d refused, %d accepted (equal to zlib)\n", streams, refused_damaged, accepted_damaged);
    printf(fails ? "FAILED %d\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
