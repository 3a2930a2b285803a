// jit.cpp -- the plan-time compiler (see jit.hpp)
#include <dlfcn.h>
#include "jit.hpp"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#if __has_include("kernel_sources.inc")
#include "kernel_sources.inc"          // static const char* const fftup_kernel_sources[][2] = {{name, text}, ...}
#define FFTUP_HAVE_EMBEDDED_SOURCES 1
#else
#define FFTUP_HAVE_EMBEDDED_SOURCES 0
#endif

namespace fftup_jit {

static const char* const kHeaderNames[] = {"fft_engine.hpp", "kernels_generic.hpp", "kernels_pow2.hpp", "kernels_mixed.hpp", "kernels_dswap.hpp"};
static constexpr int kNumHeaders = (int)(sizeof kHeaderNames / sizeof kHeaderNames[0]);

// radices the register engines have butterflies for (fft_engine.hpp bfly<R>, kernels_pow2.hpp twiddle_all<R>)
static const int kRadices[] = {2, 3, 4, 5, 7, 8, 9, 10, 12, 14, 15, 16};


static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static bool is_radix(int r) { for (int s : kRadices) if (s == r) return true; return false; }

// Test knobs (jit.hpp): the parser exists in a -DFFTUP_TEST_KNOBS build only
const char* experiment(const char* key)
{
#ifdef FFTUP_TEST_KNOBS
    const char* e = getenv("FFTUP_EXPERIMENT");
    if (!e || !*e) return nullptr;
    static thread_local std::string val;
    const std::string s = e, k = key;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t end = s.find(';', pos);
        if (end == std::string::npos) end = s.size();
        const std::string item = s.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (item.substr(0, eq) == k) { val = eq == std::string::npos ? "" : item.substr(eq + 1); return val.c_str(); }
        pos = end + 1;
    }
    return nullptr;
#else
    (void)key;
    return nullptr;
#endif
}

// "r0,r1,..." (optionally "T:r0,r1,...") from an experiment key: experiments and tests pin a factorization
static bool env_radices(const char* name, std::vector<int>& r, int* threads)
{
    const char* e = experiment(name);
    if (!e || !*e) return false;
    r.clear();
    std::string s = e;
    const size_t colon = s.find(':');
    if (colon != std::string::npos) { if (threads) *threads = atoi(s.substr(0, colon).c_str()); s = s.substr(colon + 1); }
    size_t pos = 0;
    while (pos < s.size()) {
        r.push_back(atoi(s.c_str() + pos));
        const size_t c = s.find(',', pos);
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return !r.empty();
}

// Every stage moves all n points through LDS once and multiplies them by twiddles, whatever its radix, so the work of a
// factorization is decided by its NUMBER of stages; between factorizations with equally many stages the lane slots
// count (threads x points per thread, idle lanes included): balanced radices need the fewest threads.

// three stages, one butterfly per thread and stage (MrFftT): n = r0 * r1 * r2, tk interleaved sequences per workgroup,
// at most tmax threads per sequence.  An odd first radix spreads the stage-0 scatter over the LDS banks without an
// index map; an even one costs bank conflicts in that scatter (2-way for 10 and 12 with one sequence, 4- to 16-way with four).
static bool choose3(int n, int tk, int tmax, int r[3], int* threads, const char* env)
{
    std::vector<int> pin;
    if (env_radices(env, pin, nullptr) && pin.size() == 3 && pin[0] * pin[1] * pin[2] == n && is_radix(pin[0]) && is_radix(pin[1]) && is_radix(pin[2])) {
        r[0] = pin[0]; r[1] = pin[1]; r[2] = pin[2];
        *threads = std::max(n / r[0], std::max(n / r[1], n / r[2]));
        return *threads <= tmax;
    }
    double best = 0;
    int best_min = 0;
    bool found = false;
    for (int a : kRadices) {
        if (n % a) continue;
        for (int b : kRadices) {
            if ((n / a) % b || !is_radix(n / a / b)) continue;
            const int c = n / a / b;
            const int T = std::max(n / a, std::max(n / b, n / c));
            if (T > tmax) continue;
            double cost = (double)T * (a + b + c);
            if (a % 2 == 0) cost *= (tk > 1) ? 2.0 : ((a == 10 || a == 12) ? 1.25 : 1.5);
            const int mn = std::min(a, std::min(b, c));
            if (!found || cost < best || (cost == best && mn > best_min)) { best = cost; best_min = mn; found = true; r[0] = a; r[1] = b; r[2] = c; *threads = T; }
        }
    }
    return found;
}

// any number of stages on T threads (MrFftN / FusedPlanN): first radix a multiple of 4, one butterfly per thread in the
// first and the last stage, at most 16 points per thread in between.  Measured over a dozen sizes (profiles/
// r02_k_jit_factorizations.txt): a first radix of 8 -- every thread of a UW/8-thread workgroup loads and transforms its own
// eight inputs -- beats 16 (half the threads idle in the prefetch and in the first stage) by 10-25 %, even at one stage
// more; after that the fewest stages (every stage is an LDS exchange with two workgroup barriers), the fewest lane
// slots, the largest smallest radix.
static bool choose_fused_n(int n, int D, int DD, std::vector<int>& out, int* threads)
{
    {
        int T = 0;
        std::vector<int> pin;
        if (env_radices("jit_fused", pin, &T) && T >= 64 && T <= 1024 && T % 64 == 0) {
            long prod = 1;
            bool ok = pin.size() >= 2 && (pin[0] * DD) % D == 0;
            for (int q : pin) { ok &= is_radix(q); prod *= q; }
            if (ok && prod == n && T >= n / pin[0] && T >= n / pin.back()) { out = pin; *threads = T; return true; }
        }
    }
    bool found = false;
    int best_ns = 0, best_min = 0, best_r0 = 0;
    double best = 0;
    std::vector<int> cur;
    auto r0_rank = [](int r0) { return r0 == 8 ? 0 : r0 == 12 ? 1 : r0 == 16 ? 2 : 3; };
    auto eval = [&]() {
        const int ns = (int)cur.size();
        if (ns < 2 || (cur[0] * DD) % D) return;
        const int tmin = std::max(n / cur[0], n / cur[ns - 1]);
        // whole multiples of 256 threads (the waves spread evenly over the four SIMDs, fewer sharpen passes) beat the
        // minimum in 30 of 43 tuner decisions; small workgroups take one wave more than they need
        const int t0 = tmin > 128 ? (tmin + 255) / 256 * 256 : (tmin + 63) / 64 * 64 + 64;
        for (int T = t0; T <= 1024; T += 64) {
            int vn = 0, mn = 99;
            double cost = 0;
            for (int s = 0; s < ns; s++) {
                const int bpt = (n / cur[s] + T - 1) / T;
                vn = std::max(vn, bpt * cur[s]);
                mn = std::min(mn, cur[s]);
                cost += (double)bpt * T * cur[s];
            }
            if (vn > 16) continue;
            if (T > 768) cost *= 1.5;                                          // 64-VGPR territory
            if ((n + 4 * T - 1) / (4 * T) > 4) cost *= 1.3;                    // ring rows no longer fit the registers
            const int rk = r0_rank(cur[0]);
            mn = std::min(mn, 4);                                              // radix-2/3 stages: all exchange, hardly any arithmetic
            const bool better = !found || rk < best_r0 ||
                                (rk == best_r0 && (ns < best_ns || (ns == best_ns && (mn > best_min || (mn == best_min && cost < best)))));
            if (better) { found = true; best_r0 = rk; best_ns = ns; best = cost; best_min = mn; out = cur; *threads = T; }
            break;
        }
    };
    // depth-first over ordered factorizations of at most five factors
    struct Rec {
        static void go(int m, std::vector<int>& cur, const std::function<void()>& leaf)
        {
            if (m == 1) { leaf(); return; }
            if (cur.size() >= 5) return;
            for (int r : kRadices)
                if (m % r == 0) { cur.push_back(r); go(m / r, cur, leaf); cur.pop_back(); }
        }
    };
    Rec::go(n, cur, eval);
    return found;
}

// The chooser's alternatives for the fused kernel, for the plan-time tuner (fftup_plan_create with FFTUP_FLAG_TUNE_PLAN /
// experiment jit_tune=1): the best factorization (by the ranking above, first radix aside) of every (first radix, number of
// stages, thread count) class, at most `max` of them, the chooser's own pick first.
static std::string cache_dir();
static bool read_file(const std::string& path, std::string& out);
static std::string join(const std::vector<int>& v);
std::vector<FusedCand> fused_candidates(int n, int D, size_t max, int DD)
{
    struct Best { int r0, ns, T, mn; double cost; std::vector<int> r; };
    std::vector<Best> classes;
    std::vector<int> cur;
    auto eval = [&]() {
        const int ns = (int)cur.size();
        if (ns < 2 || ns > 4 || (cur[0] * DD) % D) return;
        const int tmin = std::max(n / cur[0], n / cur[ns - 1]);
        for (int T = (tmin + 63) / 64 * 64, tries = 0; T <= 1024 && tries < 2; T += 64) {
            int vn = 0, mn = 99;
            double cost = 0;
            for (int s = 0; s < ns; s++) {
                const int bpt = (n / cur[s] + T - 1) / T;
                vn = std::max(vn, bpt * cur[s]);
                mn = std::min(mn, cur[s]);
                cost += (double)bpt * T * cur[s];
            }
            if (vn > 16) continue;
            tries++;
            mn = std::min(mn, 4);
            bool placed = false;
            for (auto& b : classes)
                if (b.r0 == cur[0] && b.ns == ns && b.T == T) {
                    if (mn > b.mn || (mn == b.mn && cost < b.cost)) { b.mn = mn; b.cost = cost; b.r = cur; }
                    placed = true;
                }
            if (!placed) classes.push_back({cur[0], ns, T, mn, cost, cur});
            if (T % 256 == 0) break;                                           // (also try the next multiple of 256 threads)
            T = T / 256 * 256 + 256 - 64;
        }
    };
    struct Rec {
        static void go(int m, std::vector<int>& cur, const std::function<void()>& leaf)
        {
            if (m == 1) { leaf(); return; }
            if (cur.size() >= 4) return;
            for (int r : kRadices)
                if (m % r == 0) { cur.push_back(r); go(m / r, cur, leaf); cur.pop_back(); }
        }
    };
    Rec::go(n, cur, eval);
    auto r0_rank = [](int r0) { return r0 == 8 ? 0 : r0 == 12 ? 1 : r0 == 16 ? 2 : 3; };
    std::sort(classes.begin(), classes.end(), [&](const Best& a, const Best& b) {
        if (a.ns != b.ns) return a.ns < b.ns;
        if (r0_rank(a.r0) != r0_rank(b.r0)) return r0_rank(a.r0) < r0_rank(b.r0);
        if (a.mn != b.mn) return a.mn > b.mn;
        return a.cost < b.cost;
    });
    // variety before depth: at most two candidates per (first radix, number of stages)
    std::vector<FusedCand> out;
    for (const auto& b : classes) {
        if (out.size() >= max) break;
        int same = 0;
        for (const auto& o : out) same += (o.r[0] == b.r0 && (int)o.r.size() == b.ns);
        if (same < 2) out.push_back({b.T, b.r});
    }
    return out;
}

// What the plan-time tuner found on an MI355X (tools/gpu_wisdom.py, profiles/r02_o_wisdom_mi355x.txt: 107 row lengths x
// factors, frames overlapping on three streams) where it differed from the chooser's pick by more than 3 %: row length,
// D = 2 x upscale factor, "threads:radices".  Consulted after the user's wisdom.txt, for both precisions.
static const struct { int uw, d; const char* plan; } kBuiltinWisdom[] = {
    {768, 3, "128:12,8,8"},
    {1080, 3, "256:12,9,10"},
    {1200, 3, "256:12,10,10"},
    {1280, 4, "256:8,10,16"},
    {1280, 5, "192:10,16,8"},
    {1344, 3, "192:12,16,7"},
    {1440, 3, "192:12,10,12"},
    {1500, 3, "192:15,10,10"},
    {1536, 3, "192:12,16,8"},
    {1536, 6, "192:12,16,8"},
    {1600, 4, "256:16,10,10"},
    {1600, 5, "256:10,10,16"},
    {1728, 3, "256:12,12,12"},
    {1800, 5, "256:10,12,15"},
    {1920, 3, "256:12,10,16"},
    {1920, 4, "256:12,10,16"},
    {1920, 6, "256:12,10,16"},
    {2240, 5, "448:5,7,8,8"},
    {2688, 3, "512:12,4,7,8"},
    {2688, 6, "512:12,4,7,8"},
    {2880, 4, "256:12,15,16"},
    {2880, 8, "256:16,12,15"},
    {3200, 5, "448:10,4,10,8"},
    {3200, 8, "512:8,4,10,10"},
    {3200, 10, "448:10,4,10,8"},
    {3600, 10, "512:10,4,9,10"},
    {3840, 3, "256:15,16,16"},
    {3840, 6, "384:12,4,8,10"},
    {4000, 5, "512:10,4,10,10"},
    {4000, 10, "512:10,4,10,10"},
    {4480, 5, "640:10,8,8,7"},
    {4480, 10, "640:10,8,8,7"},
    {4608, 8, "768:8,8,8,9"},
    {4800, 5, "512:10,4,10,12"},
    {5040, 6, "512:12,5,7,12"},
    {5120, 5, "768:10,8,8,8"},
    {5120, 8, "768:8,8,8,10"},
    {5120, 10, "768:10,8,8,8"},
    {5376, 6, "768:12,7,8,8"},
    {5760, 10, "768:10,8,8,9"},
    {6000, 6, "768:12,5,10,10"},
    {6144, 6, "768:12,8,8,8"},
    {8000, 10, "1024:10,8,10,10"},
};

// plan-time tuner's memory: <cache dir>/wisdom.txt, one "key = value" per line, last one wins
static std::string wisdom_path() { const std::string d = cache_dir(); return d.empty() ? "" : d + "/wisdom.txt"; }
bool wisdom_lookup(const std::string& key, std::string& value)
{
    std::string text;
    const std::string path = wisdom_path();
    if (path.empty() || !read_file(path, text)) return false;
    bool found = false;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        const std::string line = text.substr(pos, eol - pos);
        const size_t eq = line.find(" = ");
        if (eq != std::string::npos && line.compare(0, eq, key) == 0) { value = line.substr(eq + 3); found = true; }
        pos = eol + 1;
    }
    return found;
}
void wisdom_store(const std::string& key, const std::string& value)
{
    const std::string path = wisdom_path();
    if (path.empty()) return;
    if (FILE* f = fopen(path.c_str(), "a")) { fprintf(f, "%s = %s\n", key.c_str(), value.c_str()); fclose(f); }
}

// any number of stages for the row and column kernels (MrFftNT, XOR index map: no preference for odd first radices):
// one butterfly per thread in the first and the last stage, at most 16 points per thread, T a multiple of `granule`.
static bool choose_n(int n, int tmax, int granule, std::vector<int>& out, int* threads, const char* env)
{
    {
        std::vector<int> pin;
        if (env_radices(env, pin, nullptr) && pin.size() >= 2) {
            long prod = 1;
            bool ok = true;
            for (int q : pin) { ok &= is_radix(q); prod *= q; }
            const int T = ok && prod == n ? (std::max(n / pin[0], n / pin.back()) + granule - 1) / granule * granule : 0;
            if (T > 0 && T <= tmax) { out = pin; *threads = T; return true; }
        }
    }
    bool found = false;
    int best_ns = 0, best_min = 0;
    double best = 0;
    std::vector<int> cur;
    auto eval = [&]() {
        const int ns = (int)cur.size();
        if (ns < 2) return;
        const int T = (std::max(n / cur[0], n / cur[ns - 1]) + granule - 1) / granule * granule;
        if (T > tmax) return;
        int vn = 0, mn = 99;
        double cost = 0;
        for (int s = 0; s < ns; s++) {
            const int bpt = (n / cur[s] + T - 1) / T;
            vn = std::max(vn, bpt * cur[s]);
            mn = std::min(mn, cur[s]);
            cost += (double)bpt * T * cur[s];
        }
        if (vn > 16) return;
        // rows (one sequence per workgroup): no radix-2/3 stage if it can be avoided (3584 = 8*8*8*7 runs 15 % faster than
        // 16*2*7*16); columns: the fewest lane slots decide (four sequences per workgroup: the block size is what hurts)
        const int mnk = granule >= 64 ? std::min(mn, 4) : 0;
        if (!found || ns < best_ns || (ns == best_ns && (mnk > best_min || (mnk == best_min && cost < best)))) {
            found = true; best_ns = ns; best = cost; best_min = mnk; out = cur; *threads = T;
        }
    };
    struct Rec {
        static void go(int m, std::vector<int>& cur, const std::function<void()>& leaf)
        {
            if (m == 1) { leaf(); return; }
            if (cur.size() >= 5) return;
            for (int r : kRadices)
                if (m % r == 0) { cur.push_back(r); go(m / r, cur, leaf); cur.pop_back(); }
        }
    };
    Rec::go(n, cur, eval);
    return found;
}

// the fused kernel of `c` := FusedPlanN<UW, T, 2, wpe, rr, radices...>
void set_fused_n(Choice& c, int T, const std::vector<int>& radices)
{
    c.fused_kind = 2; c.fr = radices; c.fused_t = T; c.fused_rr = true;
    c.fused_wpe = std::max((T + 255) / 256, std::min(T * 2 / 256, 4));         // >= 128 VGPRs; load() relaxes it when the kernel spills
    const size_t xb = sizeof(float2) * (size_t)((c.UW + 15) & ~15);
    const int npass = (c.UW + 4 * T - 1) / (4 * T);
    c.fused_lds = (npass <= 4 ? 1 : 2) * xb + 32 * sizeof(float);
}
std::string fused_key(const Choice& c, const std::string& arch)
{
    return "fused v1 " + arch + " " + std::to_string(c.UW) + " " + std::to_string(c.D) + (c.DD != 1 ? "/" + std::to_string(c.DD) : "") + (c.half ? " h" : " f");
}
std::string fused_value(const Choice& c)
{
    if (c.fused_kind == 0) return "pow2";
    if (c.fused_kind == 1) return "mr16";
    return std::to_string(c.fused_t) + ":" + join(c.fr);
}

// Factorizations for a W x H -> (D/2) W x (D/2) H plan.  D = 2 x the upscale factor: even = integer factor U = D/2
// (polyphase column pass), odd = half-integer factor.  false: some dimension has no supported factorization (the plan
// then stays on the size-generic kernels).  ct_radices: the stage list of the size-generic plan for the output width
// (radices <= 8); arch: device + mode key of the tuner's wisdom file ("" = built-in wisdom only); use_wisdom = false: the
// structural default (pow2 / 16*16*R / the chooser's pick), whatever the wisdom says -- the tuner times it as a candidate.
bool choose(int W, int H, int D, bool half, const std::vector<int>& ct_radices, Choice& c, const std::string& arch, bool use_wisdom, int DD)
{
    // (DD = 2 / 4: quarter- / eighth-integer factor D / 4, D / 8, D odd -- like the half-integer ones one spectrum buffer with all rows, U = 1)
    const int U = (DD == 1 && D % 2 == 0) ? D / 2 : 1;
    c.W = W; c.H = H; c.U = U; c.D = D; c.DD = DD; c.UW = D * W / (2 * DD); c.UH = D * H / (2 * DD); c.half = half;
    if (W < 64 || H < 64 || W > 8192 || H > 8192 || D < 3 || c.UW > 8192 || (D * W) % (2 * DD) || (D * H) % (2 * DD)) return false;
    {   // (the factor D / (2 DD) in lowest terms as far as DD goes, above 1)
        int a = D, b = DD;
        while (b) { const int t = a % b; a = b; b = t; }
        if (DD < 1 || DD > 7 || (DD != 1 && (a != 1 || D <= 2 * DD))) return false;
    }
    if (c.UW % 4) return false;        // the sharpen works on quads of pixels (-u 5 with W = 2 * odd: the size-generic kernels)
    // ---- row R2C
    if (is_pow2(W) && W >= 256) { c.row_kind = 0; c.row_block = W / 8; }
    else if (!experiment("jit_row_nstage") && choose3(W, 1, 1024, c.rr, &c.row_t, "jit_row")) { c.row_kind = 1; c.row_t = (c.row_t + 63) / 64 * 64; c.row_block = c.row_t; }
    else if (choose_n(W, 1024, 64, c.rn, &c.row_t, "jit_row")) { c.row_kind = 3; c.row_block = c.row_t; }
    else c.row_kind = 2;               // no supported factorization: the size-generic row kernel (same S1 layout) stays
    // ---- column (four columns of a spectrum tile per workgroup; two when a stage of a long column needs more than 256 threads)
    // (columns beyond 5120 points: four of them no longer fit the 160 KB of LDS side by side -- two per workgroup, up to 8192 points)
    auto col_lds = [](int len, int cols) { return sizeof(float2) * (size_t)((len * cols + 15) & ~15); };
    auto col_n = [&](int len, std::vector<int>& r, int* tpc, const char* env) {
        if (col_lds(len, 4) <= 160 * 1024 && choose_n(len, 256, 16, r, tpc, env)) return 4;
        if (col_lds(len, 2) <= 160 * 1024 && choose_n(len, 512, 32, r, tpc, env)) return 2;
        return 0;
    };
    if (U == 1) {
        int ti = 0;
        if (c.UH > 8192) return false;
        const int cf = col_n(H, c.cn, &c.col_tpc, "jit_col"), ci = col_n(c.UH, c.ci, &ti, "jit_coli");
        if (!cf || !ci) return false;
        c.col_cols = std::min(cf, ci);
        c.col_tpc = std::max(c.col_tpc, ti);
        c.col_kind = 5; c.col_block = c.col_cols * c.col_tpc; c.col_lds = sizeof(float2) * (size_t)((c.UH * c.col_cols + 15) & ~15);
        if (c.col_block > 1024 || c.col_lds > 160 * 1024) return false;
    } else if (U > 2) {
        if (!(c.col_cols = col_n(H, c.cn, &c.col_tpc, "jit_col"))) return false;
        c.col_kind = 4; c.col_block = c.col_cols * c.col_tpc; c.col_lds = sizeof(float2) * (size_t)((H * c.col_cols + 15) & ~15);
    } else if (is_pow2(H) && H >= 128 && H <= 2048) {
        c.col_kind = 0; c.col_block = 4 * H / 8; c.col_lds = sizeof(float2) * (size_t)((H * 4 + 15) & ~15);      // lswz_size
    } else if (!experiment("jit_col_nstage") && col_lds(H, 4) <= 160 * 1024 && choose3(H, 4, 256, c.cr, &c.col_tpc, "jit_col")) {
        c.col_kind = 1; c.col_block = 4 * c.col_tpc; c.col_lds = sizeof(float2) * (size_t)H * 4;
    } else if ((c.col_cols = col_n(H, c.cn, &c.col_tpc, "jit_col"))) {
        c.col_kind = 3; c.col_block = c.col_cols * c.col_tpc; c.col_lds = sizeof(float2) * (size_t)((H * c.col_cols + 15) & ~15);
    } else return false;
    // ---- fused C2R + sharpen
    const int UW = c.UW;
    size_t xb = sizeof(float2) * (size_t)((UW + 15) & ~15);                    // lswz_size(UW)
    int nbuf = 2;
    if ((UW == 1024 || UW == 2048 || UW == 4096) && (8 * DD) % D == 0) { c.fused_kind = 0; c.fused_t = UW / 8; nbuf = 3; }
    else {
        const bool mr16 = UW % 256 == 0 && is_radix(UW / 256) && (16 * DD) % D == 0 && !experiment("jit_fused");
        if (mr16) {
            c.fused_kind = 1; c.fused_t = 256;
            xb = (sizeof(float2) * (size_t)(UW + (UW >> 4) + 1) + 15) & ~(size_t)15;                               // lpad_size(UW)
        } else if (choose_fused_n(UW, D, DD, c.fr, &c.fused_t)) {
            set_fused_n(c, c.fused_t, std::vector<int>(c.fr));
            if (const char* e = experiment("jit_fused_opt")) { int w = 0, r = 1; if (sscanf(e, "%d,%d", &w, &r) == 2) { c.fused_wpe = std::max(1, w); c.fused_rr = r != 0; } }
        } else return false;
    }
    {
        const int npass = (UW + 4 * c.fused_t - 1) / (4 * c.fused_t);
        const size_t nx = (npass <= 4 && (c.fused_kind != 2 || c.fused_rr)) ? 1 : 2;    // FusedGLds::RR
        c.fused_lds = (nx + (nbuf == 3 ? 1 : 0)) * xb + (nbuf == 3 ? 0 : 32 * sizeof(float));
        if (c.fused_lds > 160 * 1024) return false;
    }
    // what the plan-time tuner found best on this device for rows of this length (wisdom.txt)
    if (use_wisdom && !experiment("jit_fused")) {
        std::string w;
        bool have = !arch.empty() && wisdom_lookup(fused_key(c, arch), w);
        if (!have && !experiment("jit_no_builtin_wisdom"))
            for (const auto& e : kBuiltinWisdom)
                if (e.uw == UW && e.d == D && DD == 1) { w = e.plan; have = true; }
        if (have && w != fused_value(c) && w != "pow2" && w != "mr16") {
            int T = 0;
            std::vector<int> r;
            const size_t colon = w.find(':');
            if (colon != std::string::npos) {
                T = atoi(w.c_str());
                size_t pos = colon + 1;
                while (pos < w.size()) { r.push_back(atoi(w.c_str() + pos)); const size_t cm = w.find(',', pos); if (cm == std::string::npos) break; pos = cm + 1; }
            }
            long prod = 1;
            bool ok = r.size() >= 2 && T >= 64 && T <= 1024 && T % 64 == 0 && (r[0] * DD) % D == 0;
            for (int q : r) { ok &= is_radix(q); prod *= q; }
            // (a hand-edited or stale entry must pass the chooser's own bound: at most 16 points per thread in every stage --
            // 16,3,16 on UW/16 threads needs 18 and would not compile, leaving the plan on the size-generic kernels for good)
            if (ok && prod == UW)
                for (int q : r) ok &= ((UW / q + T - 1) / T) * q <= 16;
            if (ok && prod == UW && T >= UW / r[0] && T >= UW / r.back() && sizeof(float2) * (size_t)UW * 2 + 1024 <= 160 * 1024) set_fused_n(c, T, r);
        }
    }
    // ---- stand-alone C2R for the pre-sharpen tap (LDS ping-pong, compile-time radices)
    c.ct = ct_radices;
    c.ct_t = std::min(1024, std::max(64, (UW / 8 + 63) / 64 * 64));
    c.ct_lds = 2 * sizeof(float2) * (size_t)(UW + (UW >> 4) + 1);
    return true;
}

static std::string join(const std::vector<int>& v)
{
    std::string s;
    for (size_t i = 0; i < v.size(); i++) s += (i ? ", " : "") + std::to_string(v[i]);
    return s;
}


// A plan's kernels come from two translation units: part 0 holds the row and column kernels (they depend on W, H and the
// factor), part 1 the fused C2R+sharpen kernel and the stand-alone C2R (they depend on the output row length only), so
// that plans of different heights share part 1's code object, and the register-bound relaxation and the tuner recompile
// part 1 alone.  Returns the source of `part` and the name expressions of its kernels (others "").
static std::string make_source(const Choice& c, std::string names[K_COUNT], int part)
{
    const std::string W = std::to_string(c.W), H = std::to_string(c.H), UW = std::to_string(c.UW);
    for (int k = 0; k < K_COUNT; k++) names[k] = "";
    std::string s;
    if (part == 0) {
        s += "// generated by fftup (jit.hpp): row and column kernels, " + W + "x" + H + " -> " + UW + "x" + std::to_string(c.UH) + (c.half ? ", binary16 storage\n" : ", fp32\n");
        s += "#include \"kernels_mixed.hpp\"\n#include \"kernels_dswap.hpp\"\nnamespace fftup {\n";
        s += "struct JitCfg {\n    static constexpr int W = " + W + ", H = " + H + ";\n";
        if (c.row_kind == 1)
            s += "    static constexpr int RR0 = " + std::to_string(c.rr[0]) + ", RR1 = " + std::to_string(c.rr[1]) + ", RR2 = " + std::to_string(c.rr[2]) +
                 ", ROW_T = " + std::to_string(c.row_t) + ";\n";
        if (c.col_kind == 1)
            s += "    static constexpr int CR0 = " + std::to_string(c.cr[0]) + ", CR1 = " + std::to_string(c.cr[1]) + ", CR2 = " + std::to_string(c.cr[2]) +
                 ", COL_TPC = " + std::to_string(c.col_tpc) + ";\n";
        if (c.row_kind == 3)
            s += "    static constexpr int ROW_T = " + std::to_string(c.row_t) + ";\n    using RowN = MrFftNT<W, +1, ROW_T, 1, " + join(c.rn) + ">;\n";
        const std::string cc = std::to_string(c.col_cols);
        if (c.col_kind == 5)
            s += "    static constexpr int UH = " + std::to_string(c.UH) + ";\n    using ColIU = MrFftNT<UH, -1, " + std::to_string(c.col_tpc) + ", " + cc + ", " + join(c.ci) + ">;\n";
        if (c.col_kind >= 3)
            s += "    static constexpr int COL_TPC = " + std::to_string(c.col_tpc) + ", COL_COLS = " + cc + ";\n    using ColF = MrFftNT<H, +1, COL_TPC, COL_COLS, " + join(c.cn) +
                 ">;\n    using ColI = MrFftNT<H, -1, COL_TPC, COL_COLS, " + join(c.cn) + ">;\n";
        s += "};\n}\n";
        const std::string fm = c.half ? "fftup::IN_F16" : "fftup::IN_F32", um = c.half ? "fftup::IN_U8_F16" : "fftup::IN_U8_F32";
        if (c.row_kind == 0) {
            names[K_ROW_PLANAR] = "fftup::k_row_r2c_t<" + W + ", " + fm + ", 4>";
            names[K_ROW_U8] = "fftup::k_row_r2c_t<" + W + ", " + um + ", 4>";
        } else if (c.row_kind != 2) {
            const std::string k = c.row_kind == 3 ? "fftup::k_row_r2c_n" : "fftup::k_row_r2c_m";
            names[K_ROW_PLANAR] = k + "<fftup::JitCfg, " + fm + ">";
            names[K_ROW_U8] = k + "<fftup::JitCfg, " + um + ">";
        }
        names[K_COL] = c.col_kind == 5 ? "fftup::k_col_pad<fftup::JitCfg>" : c.col_kind == 0 ? ((c.H == 1024 || c.H == 512 || c.H == 256) ? "fftup::k_col_v<4, " + H + ">" : "fftup::k_col_t<" + H + ", 4>") :      // (H = 256, 512, 1024: the digit-swap column kernel, kernels_dswap.hpp)
                        c.col_kind == 3 ? "fftup::k_col_n<fftup::JitCfg>" :
                       c.col_kind == 4 ? "fftup::k_col_u<fftup::JitCfg, " + std::to_string(c.U) + ">" : "fftup::k_col_m<fftup::JitCfg>";
        return s;
    }
    s += "// generated by fftup (jit.hpp): fused C2R + sharpen and stand-alone C2R for rows of " + UW + (c.half ? ", binary16 storage\n" : ", fp32\n");
    s += "#include \"kernels_mixed.hpp\"\nnamespace fftup {\n";
    if (c.fused_kind == 0) s += "using JitFused = FusedPlanPow2<" + UW + ">;\n";
    else if (c.fused_kind == 1) s += "using JitFused = FusedPlanMr16<" + UW + ", " + std::to_string(c.UW / 256) + ">;\n";
    else s += "using JitFused = FusedPlanN<" + UW + ", " + std::to_string(c.fused_t) + ", 2, " + std::to_string(c.fused_wpe) + ", " + (c.fused_rr ? "true" : "false") +
              ", " + join(c.fr) + ">;\n";
    s += "using JitCT = CtPlan<" + UW + ", " + std::to_string(c.ct_t) + ", " + join(c.ct) + ">;\n";
    s += "static_assert(FusedGLds<JitFused>::TOTAL == " + std::to_string(c.fused_lds) + " && JitFused::T == " + std::to_string(c.fused_t) +
         ", \"host and device disagree on the fused kernel's geometry\");\n";
    s += "}\n";
    const std::string hb = c.half ? "true" : "false";
    const std::string U = std::to_string(c.U) + ", " + std::to_string(c.D), DD = std::to_string(c.DD);
    names[K_FUSED] = "fftup::k_c2r_sharpen_g<fftup::JitFused, " + hb + ", 4, " + U + (c.u8out ? ", true, " : ", false, ") + DD + ">";
    names[K_C2R_CT] = "fftup::k_row_c2r_ct<fftup::JitCT, " + hb + ", " + U + ", " + DD + ">";
    return s;
}

std::string describe(const Choice& c)
{
    std::string s = (c.DD == 3 || c.DD >= 5) ? "u" + std::to_string(c.D % 2 ? c.D : c.D / 2) + "/" + std::to_string(c.D % 2 ? 2 * c.DD : c.DD) + " row "
                    : c.DD == 4 ? "u" + std::to_string(c.D / 8) + "." + std::to_string(c.D % 8 * 125) + " row "
                    : c.DD == 2 ? "u" + std::to_string(c.D / 4) + (c.D % 4 == 1 ? ".25 row " : ".75 row ")
                    : c.D == 4 ? "row " : (c.D % 2 ? "u" + std::to_string(c.D / 2) + ".5 row " : "u" + std::to_string(c.U) + " row ");
    auto star = [](const std::vector<int>& v) { std::string t; for (size_t i = 0; i < v.size(); i++) t += (i ? "*" : "") + std::to_string(v[i]); return t; };
    s += c.row_kind == 2 ? "generic" : c.row_kind == 0 ? "pow2/8" : c.row_kind == 3 ? star(c.rn) : std::to_string(c.rr[0]) + "*" + std::to_string(c.rr[1]) + "*" + std::to_string(c.rr[2]);
    s += " x" + std::to_string(c.row_block) + ", col ";
    s += c.col_kind == 0 ? ((c.H == 1024 || c.H == 512 || c.H == 256) ? "pow2/8 digit-swap" : "pow2/8") : c.col_kind == 5 ? star(c.cn) + " -> " + star(c.ci) : c.col_kind >= 3 ? star(c.cn) : std::to_string(c.cr[0]) + "*" + std::to_string(c.cr[1]) + "*" + std::to_string(c.cr[2]);
    s += " x" + std::to_string(c.col_block) + (c.col_kind >= 3 && c.col_cols == 2 ? " (2 columns)" : "") + ", fused ";
    if (c.fused_kind == 0) s += "pow2/8";
    else if (c.fused_kind == 1) s += "16*16*" + std::to_string(c.UW / 256);
    else for (size_t i = 0; i < c.fr.size(); i++) s += (i ? "*" : "") + std::to_string(c.fr[i]);
    s += " x" + std::to_string(c.fused_t) + " (" + std::to_string(c.fused_lds) + " B LDS";
    if (c.fused_kind == 2) s += ", " + std::to_string(c.fused_wpe) + " waves/SIMD" + (c.fused_rr ? "" : ", ring rows in LDS");
    s += ")";
    return s;
}

// ------------------------------------------------------------------------------------------------ hipRTC through dlopen
struct Rtc {
    void* lib = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    hiprtcResult (*AddNameExpression)(hiprtcProgram, const char*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetLoweredName)(hiprtcProgram, const char*, const char**) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    hiprtcResult (*Version)(int*, int*) = nullptr;
    bool ok = false;
};

static const Rtc& rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* e = getenv("FFTUP_HIPRTC_LIB")) r.lib = dlopen(e, RTLD_NOW | RTLD_LOCAL);       // (this one or none)
        else
            for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        if (!r.lib) return;
        bool all = true;
#define FFTUP_RTC_SYM(field, sym) all &= ((*(void**)&r.field = dlsym(r.lib, sym)) != nullptr)
        FFTUP_RTC_SYM(CreateProgram, "hiprtcCreateProgram");
        FFTUP_RTC_SYM(AddNameExpression, "hiprtcAddNameExpression");
        FFTUP_RTC_SYM(CompileProgram, "hiprtcCompileProgram");
        FFTUP_RTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
        FFTUP_RTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
        FFTUP_RTC_SYM(GetCodeSize, "hiprtcGetCodeSize");
        FFTUP_RTC_SYM(GetCode, "hiprtcGetCode");
        FFTUP_RTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
        FFTUP_RTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
        FFTUP_RTC_SYM(Version, "hiprtcVersion");
#undef FFTUP_RTC_SYM
        r.ok = all;
    });
    return r;
}

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

static bool read_file(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

static void kernel_dir_anchor() {}
static std::string kernel_dir()
{
    if (const char* e = getenv("FFTUP_KERNEL_DIR")) return e;
    Dl_info info;
    if (dladdr((void*)&kernel_dir_anchor, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t slash = p.rfind('/');
        return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/csrc";
    }
    return "csrc";
}

static std::string cache_dir()
{
    std::string d;
    if (const char* e = getenv("FFTUP_CACHE_DIR")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/fftup";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/fftup";
    else return "";
    // (mkdir -p of the last two components; failures simply disable the disk cache)
    const size_t slash = d.rfind('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(d.substr(0, slash).c_str(), 0755);
    (void)mkdir(d.c_str(), 0755);
    return d;
}


static bool load_cached(const std::string& path, Binary& b)
{
    std::string raw;
    if (!read_file(path, raw) || raw.size() < 24 || raw.compare(0, 6, "FJIT2\n") != 0) return false;
    // FJIT2: magic, fnv1a of everything behind the checksum, lowered names, code -- a file that another process or thread
    // is still writing, or a torn one, does not load
    uint64_t sum;
    memcpy(&sum, raw.data() + 6, 8);
    if (sum != fnv1a(raw.substr(14))) return false;
    size_t off = 14;
    auto rd = [&](void* dst, size_t n) { if (off + n > raw.size()) return false; memcpy(dst, raw.data() + off, n); off += n; return true; };
    for (int k = 0; k < K_COUNT; k++) {
        uint32_t len;
        if (!rd(&len, 4) || off + len > raw.size()) return false;
        b.lowered[k].assign(raw.data() + off, len);
        off += len;
    }
    uint64_t cs;
    if (!rd(&cs, 8) || off + cs != raw.size()) return false;
    b.code.assign(raw.data() + off, cs);
    return true;
}

static void store_cached(const std::string& path, const Binary& b)
{
    std::string body;
    for (int k = 0; k < K_COUNT; k++) {
        const uint32_t len = (uint32_t)b.lowered[k].size();
        body.append((const char*)&len, 4);
        body += b.lowered[k];
    }
    const uint64_t cs = b.code.size();
    body.append((const char*)&cs, 8);
    body += b.code;
    const uint64_t sum = fnv1a(body);
    // a temporary of its own per writer (mkstemp: threads of one process share the pid), published by rename
    std::string tmp = path + ".XXXXXX";
    const int fd = mkstemp(&tmp[0]);
    if (fd < 0) return;
    FILE* f = fdopen(fd, "wb");
    if (!f) { (void)close(fd); (void)unlink(tmp.c_str()); return; }
    bool ok = fwrite("FJIT2\n", 1, 6, f) == 6 && fwrite(&sum, 8, 1, f) == 1 && fwrite(body.data(), 1, body.size(), f) == body.size();
    ok &= fclose(f) == 0;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

// compile (or fetch) the translation unit of `c` for `arch` ("gfx950:sramecc+:xnack-").  No device needed.
static bool compile(const Choice& c, const std::string& arch, int part, Binary& out, std::string& err)
{
    const Rtc& R = rtc();
    if (!R.ok) { err = "hipRTC (libhiprtc.so) not available"; return false; }
    // kernel headers: embedded text (default) or a directory
    std::vector<std::string> hdr(kNumHeaders);
    std::string kdir;
    const bool from_dir = getenv("FFTUP_KERNEL_DIR") || !FFTUP_HAVE_EMBEDDED_SOURCES;
    if (from_dir) {
        kdir = kernel_dir();
        for (int i = 0; i < kNumHeaders; i++)
            if (!read_file(kdir + "/" + kHeaderNames[i], hdr[i])) { err = "kernel header " + kdir + "/" + kHeaderNames[i] + " not found (set FFTUP_KERNEL_DIR)"; return false; }
    }
#if FFTUP_HAVE_EMBEDDED_SOURCES
    else {
        for (int i = 0; i < kNumHeaders; i++)
            for (const auto& e : fftup_kernel_sources)
                if (!strcmp(e[0], kHeaderNames[i])) hdr[i] = e[1];
    }
#endif
    std::string hdr_text;
    for (const auto& h : hdr) hdr_text += h;
    std::string names[K_COUNT];
    const std::string src = make_source(c, names, part);
    if (const char* dump = experiment("jit_dump")) {       // the generated translation unit, for inspection (tools/jit_resources.sh)
        if (FILE* f = fopen((std::string(dump) + (part ? ".fused.hip" : ".rowcol.hip")).c_str(), "w")) {
            fputs(src.c_str(), f);
            for (int k = 0; k < K_COUNT; k++) if (!names[k].empty()) fprintf(f, "template __global__ decltype(%s) %s;\n", names[k].c_str(), names[k].c_str());
            fclose(f);
        }
    }
    const char* rocm = getenv("ROCM_PATH");
    const std::string inc_rocm = std::string("-I") + (rocm ? rocm : "/opt/rocm") + "/include";
    const std::string arch_opt = "--offload-arch=" + arch;
    // the flags of the ahead-of-time build (__graft_entry__.py): results must not depend on which of the two compiled a kernel
    const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=on", inc_rocm.c_str()};
    int vmaj = 0, vmin = 0;
    R.Version(&vmaj, &vmin);
    uint64_t key = fnv1a(src);
    for (int k = 0; k < K_COUNT; k++) key = fnv1a(names[k] + ";", key);      // (the instantiations: factor and precision live here)
    for (const char* o : opts) key = fnv1a(o, key);
    key = fnv1a(std::to_string(vmaj) + "." + std::to_string(vmin), key);
    key = fnv1a(hdr_text, key);
    char keyhex[32];
    snprintf(keyhex, sizeof keyhex, "%016llx", (unsigned long long)key);

    // The lock guards the tables only (the two parts of a plan compile side by side, compile_both()).  Threads that ask for
    // a translation unit another thread is compiling right now wait for it instead of compiling it again (the CLI's
    // -numthreads mode: N threads create plans of one size at the same moment).
    static std::mutex mu;
    static std::condition_variable cv;
    static std::map<uint64_t, Binary> memo;
    static std::set<uint64_t> in_flight;
    {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            auto it = memo.find(key);
            if (it != memo.end()) { out = it->second; return true; }
            if (!in_flight.count(key)) break;
            cv.wait(lock);
        }
        in_flight.insert(key);
    }
    struct Done {                                   // whatever way this call ends: let the waiters look again
        uint64_t key;
        ~Done() { { std::lock_guard<std::mutex> lock(mu); in_flight.erase(key); } cv.notify_all(); }
    } done{key};
    const std::string cdir = cache_dir();
    const std::string cpath = cdir.empty() ? "" : cdir + "/" + keyhex + ".fjit";
    if (!cpath.empty() && load_cached(cpath, out)) { std::lock_guard<std::mutex> lock(mu); memo[key] = out; return true; }

    hiprtcProgram prog = nullptr;
    const char *hdr_ptr[kNumHeaders], *hdr_names[kNumHeaders];
    for (int i = 0; i < kNumHeaders; i++) { hdr_ptr[i] = hdr[i].c_str(); hdr_names[i] = kHeaderNames[i]; }
    if (R.CreateProgram(&prog, src.c_str(), "fftup_jit.hip", kNumHeaders, hdr_ptr, hdr_names) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return false; }
    for (int k = 0; k < K_COUNT; k++) if (!names[k].empty()) R.AddNameExpression(prog, names[k].c_str());
    const hiprtcResult rc = R.CompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        R.GetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) R.GetProgramLog(prog, &log[0]);
        err = "hipRTC compilation failed (" + describe(c) + "): " + log.substr(0, 2000);
        R.DestroyProgram(&prog);
        return false;
    }
    size_t cs = 0;
    R.GetCodeSize(prog, &cs);
    out.code.assign(cs, '\0');
    R.GetCode(prog, &out.code[0]);
    bool ok = cs > 0;
    for (int k = 0; k < K_COUNT; k++) {
        out.lowered[k].clear();
        if (names[k].empty()) continue;
        const char* low = nullptr;
        ok &= R.GetLoweredName(prog, names[k].c_str(), &low) == HIPRTC_SUCCESS && low;
        if (low) out.lowered[k] = low;
    }
    R.DestroyProgram(&prog);
    if (!ok) { err = "hipRTC returned no code / no lowered names"; return false; }
    { std::lock_guard<std::mutex> lock(mu); memo[key] = out; }
    if (!cpath.empty()) store_cached(cpath, out);
    return true;
}

// both parts of a plan.  (experiment jit_threads=1 compiles the second one on a thread of its own: measured, no gain --
// hipRTC serialises its compilations internally, 48 plans take 54 s either way -- so one after the other is the default.)
bool compile_both(const Choice& c, const std::string& arch, Binary b[2], std::string& err)
{
    const char* e = experiment("jit_threads");
    if (!e || atoi(e) == 0) return compile(c, arch, 0, b[0], err) && compile(c, arch, 1, b[1], err);
    std::string err1;
    bool ok1 = false;
    std::thread t([&] { ok1 = compile(c, arch, 1, b[1], err1); });
    const bool ok0 = compile(c, arch, 0, b[0], err);
    t.join();
    if (ok0 && !ok1) err = err1;
    return ok0 && ok1;
}


static Module* load_once(const Choice& c, const std::string& arch, std::string& err);

// the fused kernel's scratch bytes per lane (register spills), 0 if unknown
static int fused_scratch(const Module* m);

// Load the plan's code object; when the fused kernel of an N-stage plan spills under the two-strips-per-unit register
// bound, rebuild it for one strip per unit (twice the registers), then with the ring rows in LDS.
Module* load(Choice c, const std::string& arch, std::string& err)
{
    // `choose` filled the geometry for c.fused_wpe / c.fused_rr as they are; the variants below only change those two
    for (;;) {
        Module* m = load_once(c, arch, err);
        if (!m || c.fused_kind != 2 || experiment("jit_fused_opt") || fused_scratch(m) == 0) return m;
        const int one_strip = std::max(1, c.fused_t / 256);
        const int npass = (c.UW + 4 * c.fused_t - 1) / (4 * c.fused_t);
        if (c.fused_wpe > one_strip) c.fused_wpe = one_strip;
        else if (c.fused_rr && npass <= 4 && 2 * (c.fused_lds - 32 * sizeof(float)) + 32 * sizeof(float) <= 160 * 1024) {
            c.fused_rr = false;
            c.fused_lds = 2 * (c.fused_lds - 32 * sizeof(float)) + 32 * sizeof(float);
        } else return m;                                                        // nothing left to relax: it runs, with spills
        delete m;
    }
}

static Module* load_once(const Choice& c, const std::string& arch, std::string& err)
{
    Binary bin[2];
    if (!compile_both(c, arch, bin, err)) return nullptr;
    Module* m = new Module();
    m->choice = c;
    for (int part = 0; part < 2; part++) {
        const Binary& b = bin[part];
        hipError_t e = hipModuleLoadData(&m->mod[part], b.code.data());
        if (e != hipSuccess) { err = std::string("hipModuleLoadData: ") + hipGetErrorString(e); m->mod[part] = nullptr; delete m; return nullptr; }
        for (int k = 0; k < K_COUNT; k++) {
            if (b.lowered[k].empty()) continue;
            e = hipModuleGetFunction(&m->fn[k], m->mod[part], b.lowered[k].c_str());
            if (e != hipSuccess) { err = "hipModuleGetFunction(" + b.lowered[k] + "): " + hipGetErrorString(e); delete m; return nullptr; }
        }
    }
    return m;
}

static int fused_scratch(const Module* m)
{
    int bytes = 0;
    if (hipFuncGetAttribute(&bytes, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, m->fn[K_FUSED]) != hipSuccess) return 0;
    return bytes;
}


}  // namespace fftup_jit
