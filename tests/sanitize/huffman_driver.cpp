// Sanitizer harness for the PNG encoder's Huffman-only deflate (tests/test_sanitize.py): the translation unit of the codec is
// included so that its internal functions can be called.  Checks (1) code lengths: complete (Kraft sum exactly one), within
// the limit, zero exactly for unused symbols -- for Fibonacci frequencies (the deepest trees there are), single symbols and
// random sets; (2) streams: huffman_zlib output inflates (zlib) to the input for skewed, flat, random and tiny buffers.
#include "../../vkresample_amd/csrc/cli/png_codec.cpp"

#include <algorithm>
#include <random>

using namespace pngio;

static int check_lengths(const std::vector<uint32_t>& f, int maxbits, const char* what)
{
    const int n = (int)f.size();
    std::vector<uint8_t> len(n);
    fftup_huff::Work wk;
    fftup_huff::huffman_lengths(f.data(), n, maxbits, len.data(), wk);
    int used = 0;
    unsigned long long kraft = 0;
    for (int i = 0; i < n; i++) {
        used += f[i] != 0;
        if (len[i] > maxbits) { printf("%s: length %d beyond %d\n", what, len[i], maxbits); return 1; }
        if (len[i]) kraft += 1ull << (maxbits - len[i]);
    }
    if (used == 0) return 0;
    int bad = 0;
    for (int i = 0; i < n; i++) {
        if (f[i] && !len[i]) bad++;
        if (!f[i] && len[i] && used != 1) bad++;
    }
    if (bad || kraft != (1ull << maxbits)) { printf("%s: bad code (kraft %llu of %llu, %d symbols wrong)\n", what, kraft, 1ull << maxbits, bad); return 1; }
    // more frequent symbols never get longer codes
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (f[i] > f[j] && f[j] && len[i] > len[j]) { printf("%s: order violated\n", what); return 1; }
    return 0;
}

static int check_stream(const std::vector<uint8_t>& src, const char* what)
{
    std::vector<uint8_t> comp(huffman_zlib_bound(src.size()) + 8, 0xAB), back(src.size() + 1);
    const size_t cl = huffman_zlib(src.data(), src.size(), comp.data());
    if (cl > huffman_zlib_bound(src.size())) { printf("%s: bound exceeded\n", what); return 1; }
    uLongf dl = (uLongf)back.size();
    const int zr = uncompress(back.data(), &dl, comp.data(), (uLong)cl);
    if (zr != Z_OK || dl != src.size() || (!src.empty() && memcmp(back.data(), src.data(), src.size()))) { printf("%s: round trip failed (zlib %d, %lu of %zu bytes)\n", what, zr, (unsigned long)dl, src.size()); return 1; }
    return 0;
}

int main()
{
    int fails = 0;
    std::mt19937 rng(12345);
    for (int n : {2, 3, 19, 30, 40, 257, 286}) {                         // Fibonacci: depth n - 1 without a limit
        std::vector<uint32_t> f(n);
        uint64_t a = 1, b = 1;
        for (int i = 0; i < n; i++) { f[i] = (uint32_t)std::min<uint64_t>(a, 0xffffffffu); const uint64_t c = a + b; a = b; b = c; }
        fails += check_lengths(f, 15, "fibonacci/15");
        if (n <= 19) fails += check_lengths(f, 7, "fibonacci/7");
    }
    for (int t = 0; t < 3000; t++) {
        const int n = t % 3 == 0 ? 19 : 257;
        std::vector<uint32_t> f(n, 0);
        const int used = 1 + (int)(rng() % n);
        for (int k = 0; k < used; k++) f[rng() % n] = 1 + (rng() % (1u << (rng() % 24)));
        fails += check_lengths(f, n == 19 ? 7 : 15, "random");
    }
    {
        std::vector<uint32_t> one(257, 0);
        one[0] = 5; fails += check_lengths(one, 15, "single symbol 0");
        one[0] = 0; one[200] = 9; fails += check_lengths(one, 15, "single symbol 200");
    }
    // streams
    fails += check_stream({}, "empty");
    fails += check_stream({7}, "one byte");
    fails += check_stream(std::vector<uint8_t>(700000, 0), "zeros, three blocks");
    {
        std::vector<uint8_t> v(600000);
        for (auto& x : v) x = (uint8_t)rng();
        fails += check_stream(v, "uniform noise");
        for (auto& x : v) { int k = 0; while (k < 40 && (rng() & 1)) k++; x = (uint8_t)k; }                 // geometric: P(k) = 2^-(k+1)
        fails += check_stream(v, "geometric");
        std::vector<uint8_t> fib;
        uint64_t a = 1, b = 1;
        for (int s = 0; s < 27; s++) { fib.insert(fib.end(), (size_t)a, (uint8_t)(s * 9)); const uint64_t c = a + b; a = b; b = c; }   // deeper than 15 bits
        std::shuffle(fib.begin(), fib.end(), rng);
        fails += check_stream(fib, "fibonacci bytes");
        for (size_t n : {1u, 2u, 3u, 4u, 5u, 262143u, 262144u, 262145u, 524288u}) { v.resize(n); fails += check_stream(v, "sizes around the block length"); }
    }
    printf(fails ? "FAILED %d\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
