// fftup_plan.hip -- plan construction behind the C ABI (include/fftup.h): what launchResample() derives from its configuration
// (VkResample.cpp:1409-1617) -- sizes, zero-padding ranges, the R2C rule, factorizations -- device buffers, twiddle tables, the
// plan-time tuner; device enumeration and error text.  The kernels themselves are named in fftup_launch.hip only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_fp16.h>

#include "plan.hpp"

using namespace fftup;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

// the marketing name, or -- some driver builds leave it empty -- the architecture name ("gfx950:sramecc+:xnack-")
static const char* device_label(const hipDeviceProp_t& prop) { return prop.name[0] ? prop.name : prop.gcnArchName; }

// ------------------------------------------------------------------------------------------------
static bool is_smooth(uint32_t n)
{
    if (n == 0) return false;
    for (uint32_t p : {2u, 3u, 5u, 7u})
        while (n % p == 0) n /= p;
    return n == 1;
}

// Four-step split of a row of n points that does not fit the LDS (k_row4_a / k_row4_b): n = n1 * n2.  Pass A transforms tka
// sequences of n1 points side by side -- tka consecutive elements of the row per piece it reads and of the transposed row per
// piece it writes -- pass B tkb sequences of n2 points (tkb consecutive output elements per piece): each the widest of 16, 8, 4, 2, 1
// that divides the other factor and whose two Stockham buffers fit 160 KB.  Widest tiles first (8-byte pieces are a quarter
// of the bandwidth of 32-byte ones, profiles/r05_*_four_step.txt), then as square as possible.
static bool split_four(uint32_t n, size_t el, int* n1, int* n2, int* tka, int* tkb)
{
    long best = -1;
    auto widest = [&](uint32_t len, uint32_t other) -> int {        // sequences of `len` points, tile width must divide `other`
        for (int t : {16, 8, 4, 2, 1})
            if (other % (uint32_t)t == 0 && 2 * el * (size_t)lpad_size((int)len * t) <= (size_t)160 * 1024) return t;
        return 0;
    };
    for (uint32_t d = 2; d * d <= n; d++) {
        if (n % d) continue;
        const uint32_t a = d, b = n / d;                     // a <= b
        const int ta = widest(a, b), tb = widest(b, a);
        if (!ta || !tb) continue;
        // time ~ bytes / piece width: pass A reads and writes the row in pieces of ta elements, pass B writes it in pieces of tb
        const long score = (long)(2000 / ta + 1000 / tb) * (1l << 32) + (long)(b - a);
        if (best < 0 || score < best) { best = score; *n1 = (int)a; *n2 = (int)b; *tka = ta; *tkb = tb; }
    }
    return best >= 0;
}

// radix sequence: as many 8s as possible, then 4/2, then 3,5,7 (VkFFTScheduler vkFFT.h:4707-5189
// makes the same kind of choice; order only affects speed)
static StagePlan make_stage_plan(uint32_t n)
{
    StagePlan p{};
    p.n = (int)n;
    uint32_t m = n;
    int e2 = 0;
    while (m % 2 == 0) { m /= 2; e2++; }
    int ns = 0;
    while (e2 >= 3) { p.radix[ns++] = 8; e2 -= 3; }
    if (e2 == 2) p.radix[ns++] = 4;
    if (e2 == 1) p.radix[ns++] = 2;
    for (uint32_t q : {3u, 5u, 7u})
        while (m % q == 0) { p.radix[ns++] = (uint8_t)q; m /= q; }
    p.nstages = ns;
    return p;
}

int dev_alloc(fftup_plan* P, void** ptr, size_t bytes)
{
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        *ptr = nullptr;
        return fail(FFTUP_E_OUT_OF_MEMORY, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    P->allocs.push_back(*ptr);
    P->device_bytes += bytes;
    return FFTUP_OK;
}

static int make_twiddles(fftup_plan* P, float2** dptr, uint32_t n)
{
    if (P->dbl) {                     // double2 table behind the same pointer member
        std::vector<double2> h(n);
        for (uint32_t k = 0; k < n; k++) {
            double a = 2.0 * M_PI * (double)k / (double)n;
            h[k] = make_double2(std::cos(a), std::sin(a));
        }
        int rc = dev_alloc(P, (void**)dptr, sizeof(double2) * n);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(*dptr, h.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
        return FFTUP_OK;
    }
    std::vector<float2> h(n);
    for (uint32_t k = 0; k < n; k++) {
        // exact octant reduction is unnecessary in double; rounded once to fp32
        double a = 2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    int rc = dev_alloc(P, (void**)dptr, sizeof(float2) * n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*dptr, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    return FFTUP_OK;
}

// the sharpen constants reach the reference's shader as "%f" text (VkResample.cpp:893-901, 920)
static float const_via_percent_f(double v, bool half)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%f", v);
    float f = (float)strtod(buf, nullptr);
    if (half) f = __half2float(__float2half_rn(f));
    return f;
}

// the row kernel reads uint8 RGB directly (fp32 / fp16 plans only)
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static bool jit_enabled()
{
    const char* e = getenv("FFTUP_JIT");
    return !e || atoi(e) != 0;
}
// The upscale factor as D / (2 DD) when the specialised kernels' assumptions hold: an integer or half-integer factor in [1.5, 8]
// (DD = 1, D = 2u), a quarter-integer one (DD = 2, D = 4u odd: -u 1.25, 1.75, 2.25 ...; round 5), an odd number of eighths (DD = 4: -u 1.125, 1.875)
// or a ratio with denominator 3, 5 or 7 (DD: -u 4/3, 5/3, 1.4, 1.6 ...), output sizes exactly u W and u H,
// and the reference's zero-padding guard of the column pass (float arithmetic, VkResample.cpp:1494-1495) exactly
// [H/2, uH - H/2).  Returns D (0: none of that) and sets *DD.
static int jit_factor(float upscale, uint32_t W, uint32_t H, uint32_t uW, uint32_t uH, int zly, int zry, int* DD)
{
    // denominators in the order of their use; lowest terms follow from taking the first that fits (2 DD uW = D W rules out the rest).
    // A factor that is no binary fraction (4/3, 1.6 ...) is whatever float the caller passed: it joins when the reference's float
    // arithmetic makes the output sizes come out exact for THIS size (-u 1.3333334 at 1920x1080 does); the guard may sit a row off
    // the symmetric one (-u 1.2 at 1600x900: [449, 630)): k_col_pad takes it as it is
    *DD = 1;
    for (int dd : {1, 2, 4, 3, 5, 7}) {
        const float t = 2.0f * (float)dd * upscale;
        const int d = (int)lrintf(t);
        if (fabsf(t - (float)d) > 1e-5f * t) continue;
        if (d < 3 || d > 16 * dd || d <= 2 * dd - (dd == 1)) continue;
        if (2 * (uint64_t)dd * uW != (uint64_t)d * W || 2 * (uint64_t)dd * uH != (uint64_t)d * H) continue;
        // integer factors run the polyphase column kernels (k_col_u: residues of the symmetric guard only); every other factor runs
        // k_col_pad, which takes the guard as the reference's float arithmetic puts it -- as long as it leaves the halves apart
        const bool polyphase = dd == 1 && d % 2 == 0;
        if (polyphase ? (zly != (int)(H / 2) || zry != (int)(uH - H / 2)) : (zly < 1 || zly > (int)H || zry < zly || zry > (int)uH)) return 0;
        *DD = dd;
        return d;
    }
    return 0;
}
static void tune_fused(fftup_plan* P);

// HIP streams ("lanes") the frames of a plan alternate on
int lane_count()
{
    int nl = 3;
    if (const char* e = getenv("FFTUP_STREAMS")) nl = atoi(e);
    return std::max(1, std::min(nl, 4));
}
// Do consecutive frames of this plan overlap on several streams?  A ring of slots (fftup_execute_ring, fftup_submit_rgb8) or
// FFTUP_FLAG_OVERLAP_ITERATIONS (fftup_execute's extension) -- as long as there is more than one stream.
static bool frames_overlap(const fftup_plan* P)
{
    return lane_count() > 1 && (P->ring > 1 || (P->cfg.flags & FFTUP_FLAG_OVERLAP_ITERATIONS));
}
// what the tuner's findings are filed under: the device and whether consecutive frames overlap (what fits beside a strip
// decides) or run one after the other (the kernel's own time decides)
static std::string wisdom_device_key(const fftup_plan* P)
{
    return std::string(P->prop.gcnArchName) + (frames_overlap(P) ? " overlapped" : " sequential");
}

// Row pairs per workgroup (strip) of the fused C2R+sharpen kernel -- a property of the PLAN (results depend on the cuts in
// their last bits, tests/test_gpu_parity.py: test_fused_output_independent_of_strip_length), chosen by how its frames run.
// Frames that overlap on several streams: ONE strip per compute unit -- the rest of every unit is left to the row and column
// kernels of the frames on the other streams, and the frame time is what counts.
// Frames that run one after the other (a plan without a ring: fftup_execute's ordered iterations, the CLI's -n N):
// nothing runs beside a strip, and a workgroup of at most 512 threads (one or two waves per SIMD) does not hide its own
// latencies: two strips per unit (1080p 100 -> 91 us per iteration, 1000x1000 75 -> 62, 2048x1024 77.2 -> 76.0, -p 2
// 82.7 -> 79.7; 768 and 1024 threads: 2-7 % slower with two; profiles/r04_s_strips_per_unit_sequential.txt).
// How many workgroups are resident is the hardware's business.
void set_strip_length(fftup_plan* P)
{
    const int fused_threads = P->tuned ? (int)P->uW / 8 : P->mixed == 3 ? P->jit->choice.fused_t : 256;
    int per_cu = (!frames_overlap(P) && fused_threads <= 512) ? 2 : 1;
    if (const char* e = fftup_jit::experiment("g_per_cu")) per_cu = std::max(1, std::min(4, atoi(e)));
    const int total_pairs = 3 * (int)P->uH / 2, slots = std::max(1, P->prop.multiProcessorCount) * per_cu;
    P->pairs_per_strip = std::max(2, (total_pairs + slots - 1) / slots);
    if (P->u8out) {
        // fused 8-bit store: strips per plane, the three planes' strips of the same rows on ONE of the 8 XCDs (fused_grid):
        // whole triples per XCD, or one compute unit of an XCD gets two strips and the launch takes twice as long
        const int per_xcd = std::max(3, slots / 8) / 3, ppp = (int)P->uH / 2;
        P->pairs_per_strip = std::max(2, (ppp + 8 * per_xcd - 1) / (8 * per_xcd));
    }
    if (const char* e = fftup_jit::experiment("pairs_per_strip")) P->pairs_per_strip = std::max(1, atoi(e));
}
static bool jit_tune_enabled()
{
    const char* e = fftup_jit::experiment("jit_tune");
    return e && atoi(e) != 0;
}

static std::vector<int> stage_radices(const StagePlan& p)
{
    std::vector<int> r;
    for (int s = 0; s < p.nstages; s++) r.push_back(p.radix[s]);
    return r;
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int fftup_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fftup_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(FFTUP_E_INVALID_ARG, "null buffer");
    hipDeviceProp_t prop;
    if (device < 0 || device >= fftup_device_count()) return fail(FFTUP_E_NO_DEVICE, "bad device id");
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s", device_label(prop));
    return FFTUP_OK;
}

int fftup_device_pci_bus_id(int device, char* buf, size_t buflen)
{
    if (!buf || buflen < 16) return fail(FFTUP_E_INVALID_ARG, "buffer of at least 16 bytes needed");
    if (device < 0 || device >= fftup_device_count()) return fail(FFTUP_E_NO_DEVICE, "bad device id");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buflen, device));
    return FFTUP_OK;
}

int fftup_jit_check(uint32_t width, uint32_t height, float upscale, uint32_t precision, const char* arch, char* desc, size_t desclen)
{
    if (desc && desclen) desc[0] = 0;
    if (precision != 0 && precision != 2) return fail(FFTUP_E_UNSUPPORTED_PRECISION, "run-time specialised plans exist for -p 0 and -p 2");
    if (width < 2 || height < 2 || (width & 1) || (height & 1) || width > 65536 || height > 65536 || !is_smooth(width) || !is_smooth(height))
        return fail(FFTUP_E_UNSUPPORTED_SIZE, "sizes must be even and factor into 2,3,5,7");
    fftup_jit::Choice ch;
    if (!(upscale >= 1.0f && upscale <= 8.0f)) return fail(FFTUP_E_INVALID_ARG, "upscale out of range");
    const uint32_t uW = (uint32_t)(upscale * (float)width), uH = (uint32_t)(upscale * (float)height);
    int DD = 1;
    const int D = (uW & 1) || (uH & 1) || !is_smooth(uW) || !is_smooth(uH) || uW > 8192 ? 0 :
                  jit_factor(upscale, width, height, uW, uH, (int)(uint32_t)((float)uH / (2 * upscale)), (int)(uint32_t)((2 * upscale - 1) * (float)uH / (2 * upscale)), &DD);
    if (!D || !fftup_jit::choose((int)width, (int)height, D, precision == 2, stage_radices(make_stage_plan(uW)), ch, "", true, DD))
        return fail(FFTUP_E_UNSUPPORTED_SIZE, "no specialised factorization for this size: the size-generic kernels run it");
    if (desc && desclen) snprintf(desc, desclen, "%s", fftup_jit::describe(ch).c_str());
    if (arch && !*arch) return FFTUP_OK;                     // "": the factorizations only, nothing is compiled
    fftup_jit::Binary bin[2];
    std::string err;
    if (!fftup_jit::compile_both(ch, arch ? arch : "gfx950", bin, err)) return fail(FFTUP_E_HIP, err);
    return FFTUP_OK;
}

void fftup_plan_destroy(fftup_plan* P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    if (P->stream) (void)hipStreamSynchronize(P->stream);
    for (size_t l = 1; l < P->lanes.size(); l++) {
        if (P->lanes[l].stream) { (void)hipStreamSynchronize(P->lanes[l].stream); (void)hipStreamDestroy(P->lanes[l].stream); }
        if (P->lanes[l].done) (void)hipEventDestroy(P->lanes[l].done);
    }
    for (auto& qs : P->q) {
        if (qs.done) (void)hipEventDestroy(qs.done);
        if (qs.png.copied) (void)hipEventDestroy(qs.png.copied);
        if (qs.png.meta_host) (void)hipHostFree(qs.png.meta_host);
        if (qs.png.parts_host) (void)hipHostFree(qs.png.parts_host);
    }
    if (P->png_copy) (void)hipStreamDestroy(P->png_copy);
    for (void* p : P->allocs) (void)hipFree(p);
    delete P->jit;
    if (P->ev0) (void)hipEventDestroy(P->ev0);
    if (P->ev1) (void)hipEventDestroy(P->ev1);
    if (P->stream) (void)hipStreamDestroy(P->stream);
    delete P;
}

// Threads a workgroup needs to run every stage of `sp` in place on `tk` interleaved sequences with `pt` points per thread
// (stage_fits_inplace_tk; at least one eighth of the points: the loads and stores around the transform), a multiple of 64 --
// or 0 when that is more than `tmax`.  Radices 3, 5, 7 need more threads than N / 8: one butterfly of 5 or 7 per thread.
static int inplace_threads(const StagePlan& sp, int tk, int pt, int tmax)
{
    long need = (long)sp.n * tk / 8;
    for (int st = 0; st < sp.nstages; st++) {
        const int r = sp.radix[st], per = pt / r;
        if (per < 1) return 0;
        need = std::max(need, ((long)(sp.n / r) * tk + per - 1) / per);
    }
    const long thr = std::max(64l, (need + 63) / 64 * 64);
    return thr <= tmax ? (int)thr : 0;
}

int fftup_plan_create(fftup_plan** out, const fftup_config* cfg)
{
    if (!out || !cfg) return fail(FFTUP_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->channels != 3) return fail(FFTUP_E_INVALID_ARG, "channels must be 3 (VkResample.cpp:1368)");
    if (cfg->precision > 2) return fail(FFTUP_E_UNSUPPORTED_PRECISION, "precision must be 0 (single), 1 (double) or 2 (half)");
    const uint32_t W = cfg->width, H = cfg->height;
    // the float -> uint32 casts below are undefined for NaN / out-of-range products: bound the inputs first
    if (!(cfg->upscale >= 1.0f && cfg->upscale <= 64.0f)) return fail(FFTUP_E_INVALID_ARG, "upscale must be a finite number in [1, 64]");
    if (W > (1u << 16) || H > (1u << 16)) return fail(FFTUP_E_INVALID_ARG, "width/height above 65536");
    if (cfg->ring > 1024) return fail(FFTUP_E_INVALID_ARG, "ring must be <= 1024");
    if (!(cfg->sharpen == cfg->sharpen)) return fail(FFTUP_E_INVALID_ARG, "sharpen is NaN");
    const uint32_t uW = (uint32_t)(cfg->upscale * (float)W);     // VkResample.cpp:1417-1418
    const uint32_t uH = (uint32_t)(cfg->upscale * (float)H);
    if (W < 2 || H < 2 || (W & 1) || (H & 1) || (uW & 1) || (uH & 1) || uW < W || uH < H)
        return fail(FFTUP_E_INVALID_ARG, "width/height (and upscaled sizes) must be even, upscale >= 1");
    if (!is_smooth(W) || !is_smooth(H) || !is_smooth(uW) || !is_smooth(uH))
        return fail(FFTUP_E_UNSUPPORTED_SIZE, "sizes must factor into 2,3,5,7 (vkFFT.h:4719-4726)");
    // R2C rule of the reference: uW <= maxComputeSharedMemorySize/8 with 64 KB (VkResample.cpp:1424; complexSizeCalc = 16
    // for -p 1, VkResample.cpp:1334-1336, halves the limit); beyond it the full complex path runs
    const bool cplx = uW > (cfg->precision == 1 ? 4096u : 8192u);
    // (checked here, before any device access: gfx950 has 160 KB of LDS per workgroup.)  Non-R2C rows whose two Stockham
    // buffers do not fit run in ONE buffer (fft_lds_inplace: up to 16 384 complex fp32 points, 1024 threads, every stage
    // N/R <= (16/R) * 1024: radix 7 up to 14336 points, 3 and 5 up to 15360); the reference switches to multi-upload plans there (vkFFT.h:4773-4992)
    auto rows_fit = [&](uint32_t n) -> int {             // 2: two buffers, 1: one buffer (in place), 0: not at all
        const size_t el = cfg->precision == 1 ? 16 : 8, lds = (size_t)160 * 1024;
        if (2 * el * (size_t)lpad_size((int)n) <= lds) return 2;
        if (cfg->precision == 1 || el * (size_t)lpad_size((int)n) > lds) return 0;
        const StagePlan sp = make_stage_plan(n);
        for (int st = 0; st < sp.nstages; st++)
            if (!stage_fits_inplace((int)n, sp.radix[st], 1024, 16)) return 0;
        return 1;
    };
    // ... and rows beyond one buffer run in four steps through HBM (k_row4_a / k_row4_b), as the reference's multi-upload plans
    {
        int a, b, t, t2;
        const size_t el = cfg->precision == 1 ? 16 : 8;
        if (cplx && ((!rows_fit(uW) && !split_four(uW, el, &a, &b, &t, &t2)) || (!rows_fit(W) && !split_four(W, el, &a, &b, &t, &t2))))
            return fail(FFTUP_E_UNSUPPORTED_SIZE, "row too long: no four-step split of the row length fits the LDS");
    }

    int ndev = fftup_device_count();
    if (ndev <= 0) return fail(FFTUP_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(FFTUP_E_NO_DEVICE, "device id out of range");

    fftup_plan* P = new fftup_plan();
    P->cfg = *cfg;
    P->W = W; P->H = H; P->uW = uW; P->uH = uH;
    P->ring = cfg->ring ? cfg->ring : 1;
    P->half = cfg->precision == 2;
    P->dbl = cfg->precision == 1;
    P->cplx = cplx;
    P->ncols = cplx ? (int)W : (int)(W / 2 + 1);
    P->esz = P->dbl ? 8 : (P->half ? 2 : 4);
    P->csz = P->dbl ? 16 : 8;
    P->device = cfg->device;
    int rc = FFTUP_OK;
#define PLAN_TRY(expr)                                                                             \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            rc = fail(FFTUP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
            goto bad;                                                                              \
        }                                                                                          \
    } while (0)
#define PLAN_RC(expr)                                                                              \
    do {                                                                                           \
        rc = (expr);                                                                               \
        if (rc) goto bad;                                                                          \
    } while (0)

    {
        PLAN_TRY(hipSetDevice(P->device));
        PLAN_TRY(hipGetDeviceProperties(&P->prop, P->device));
        PLAN_TRY(hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking));
        PLAN_TRY(hipEventCreate(&P->ev0));
        PLAN_TRY(hipEventCreate(&P->ev1));

        // zero-padding ranges exactly as launchResample computes them (float math, uint32 store)
        const float u = cfg->upscale;
        P->zlx = (int)(W / 2);
        P->zrx = cplx ? (int)(uint32_t)((2 * u - 1) * (float)uW / (2 * u)) : (int)(uW / 2);      // VR:1498 / VR:1493
        P->zly = (int)(uint32_t)((float)uH / (2 * u));
        P->zry = (int)(uint32_t)((2 * u - 1) * (float)uH / (2 * u));

        P->planW = make_stage_plan(W);
        P->planH = make_stage_plan(H);
        P->planUW = make_stage_plan(uW);
        P->planUH = make_stage_plan(uH);

        const size_t lds_max = P->prop.sharedMemPerBlock ? P->prop.sharedMemPerBlock : 65536;
        // size-specialised kernels: u == 2 and power-of-two sizes with instantiated plans
        // (experiment aot=0: the sizes with ahead-of-time kernels go through the plan-time compiler as well)
        const char* const aot_e = fftup_jit::experiment("aot");
        const bool aot = !(aot_e && atoi(aot_e) == 0);
        P->tuned = aot && !P->dbl && !cplx && !(cfg->flags & FFTUP_FLAG_GENERIC_KERNELS) && uW == 2 * W && uH == 2 * H &&
                   (W == 512 || W == 1024 || W == 2048) && (H == 256 || H == 512 || H == 1024);
        P->TK = 0;
        if (P->tuned) {
            P->TK = TUNED_TK;
            P->ldsCol = kernels_tuned_col_lds(H);
        } else {
            // u = 2 with the symmetric guard: the polyphase column kernel (k_col_poly: forward, phase, length-H inverse in ONE buffer of
            // H TK points, odd rows out; the C2R kernel takes the even rows from S1) where its stages run in place
            const char* const poly_e = fftup_jit::experiment("generic_poly");
            if (!cplx && uW == 2 * W && uH == 2 * H && P->zly == (int)(H / 2) && P->zry == (int)(uH - H / 2) && !(poly_e && atoi(poly_e) == 0)) {
                for (int tk : {8, 4, 2, 1}) {
                    const size_t need = P->csz * (size_t)lpad_size((int)H * tk);
                    const int thr = inplace_threads(P->planH, tk, COL_INPLACE_PT, kernels_generic_max_threads(P->dbl));
                    if (thr && need <= lds_max / 2) { P->TK = tk; P->ldsCol = need; P->poly = true; P->thrCol = thr; break; }    // (two workgroups per compute unit)
                }
            }
            // -p 1 R2C plans: ONE buffer where every stage of both column transforms runs in place with COL_INPLACE_PT points per thread (k_col<.., true>)
            if (!P->TK && P->dbl && !cplx) {
                for (int tk : {8, 4, 2, 1}) {
                    const size_t need = P->csz * (size_t)lpad_size((int)uH * tk);
                    const int tmax = kernels_generic_max_threads(true);
                    const int thr = std::max(inplace_threads(P->planH, tk, COL_INPLACE_PT, tmax), inplace_threads(P->planUH, tk, COL_INPLACE_PT, tmax));
                    const bool ok = inplace_threads(P->planH, tk, COL_INPLACE_PT, tmax) && inplace_threads(P->planUH, tk, COL_INPLACE_PT, tmax) &&
                                    need <= lds_max / 2 && (size_t)(H / 2) * tk <= (size_t)COL_INPLACE_PT * thr;  // (two workgroups per compute unit)
                    if (ok) { P->TK = tk; P->ldsCol = need; P->inplaceC = true; P->thrCol = thr; break; }
                }
            }
            // column tile width: widest of 8,4,2,1 whose ping-pong buffers fit in LDS
            if (!P->TK) for (int tk : {8, 4, 2, 1}) {
                size_t need = 2 * P->csz * (size_t)lpad_size((int)uH * tk);
                if (need <= lds_max) { P->TK = tk; P->ldsCol = need; break; }
            }
        }
        if (!P->TK) {
            // not even one column fits: tiles of one column, both column transforms in four steps through HBM (k_row4_a / k_row4_b)
            P->TK = 1; P->ldsCol = 0;
            for (auto fh : {std::make_pair(&P->colF, H), std::make_pair(&P->colI, uH)}) {
                fftup_plan::Four& f = *fh.first;
                f.on = split_four(fh.second, P->csz, &f.n1, &f.n2, &f.tka, &f.tkb);
                if (!f.on) { rc = fail(FFTUP_E_UNSUPPORTED_SIZE, "column too long: no four-step split of the height fits the LDS"); goto bad; }
                f.p1 = make_stage_plan((uint32_t)f.n1); f.p2 = make_stage_plan((uint32_t)f.n2);
                f.ldsA = 2 * P->csz * (size_t)lpad_size(f.n1 * f.tka); f.ldsB = 2 * P->csz * (size_t)lpad_size(f.n2 * f.tkb);
                const int tmax = kernels_generic_max_threads(P->dbl);
                f.thrA = std::min(tmax, std::max(64, round_up(f.n1 * f.tka / 8, 64)));
                f.thrB = std::min(tmax, std::max(64, round_up(f.n2 * f.tkb / 8, 64)));
            }
        }
        if (aot && !P->dbl && !cplx && !P->tuned && !(cfg->flags & FFTUP_FLAG_GENERIC_KERNELS) && uW == 2 * W && uH == 2 * H && P->TK >= 4) {
            P->mixed = kernels_aot_mixed_plan(W, H);                    // 1920x1080, 1280x720
        }
        if (P->mixed) { P->TK = 4; P->ldsCol = sizeof(float2) * (size_t)H * 4; }             // k_col_m: one in-place buffer
        // any other size with an integer or half-integer upscale factor: kernels specialised for it now (the counterpart
        // of VkFFT generating its shaders at plan time)
        if (!P->dbl && !cplx && !P->tuned && !P->mixed && !(cfg->flags & (FFTUP_FLAG_GENERIC_KERNELS | FFTUP_FLAG_UNFUSED_SHARPEN)) && jit_enabled()) {
            int DD = 1;
            const int D = jit_factor(cfg->upscale, W, H, uW, uH, P->zly, P->zry, &DD);
            if (D) {
                fftup_jit::Choice ch;
                std::string jerr;
                if (fftup_jit::choose((int)W, (int)H, D, P->half, stage_radices(P->planUW), ch, wisdom_device_key(P), true, DD)) {
                    ch.u8out = (cfg->flags & FFTUP_FLAG_FUSE_U8_STORE) != 0;         // (such a plan is always fused)
                    P->jit = fftup_jit::load(ch, P->prop.gcnArchName, jerr);
                    if (P->jit) {
                        P->mixed = 3; P->U = ch.U; P->TK = 4; P->ldsCol = P->jit->choice.col_lds;
                        // (inputs taller than 4800 rows: the size-generic plan above would have run its columns in four steps through
                        // HBM -- the specialised column kernel holds two whole columns in LDS instead)
                        P->colF = fftup_plan::Four{}; P->colI = fftup_plan::Four{};
                    }
                    else if (getenv("FFTUP_JIT_VERBOSE")) fprintf(stderr, "fftup: run-time specialisation failed, size-generic kernels in use: %s\n", jerr.c_str());
                }
            }
        }
        if (P->tuned || P->mixed) P->poly = false;               // (their own column kernels)
        P->fused = (P->tuned || P->mixed) && !(cfg->flags & FFTUP_FLAG_UNFUSED_SHARPEN);
        P->u8out = P->fused && (cfg->flags & FFTUP_FLAG_FUSE_U8_STORE);
        set_strip_length(P);
        P->NT = (P->ncols + P->TK - 1) / P->TK;
        P->ldsRowF = 2 * P->csz * (size_t)lpad_size((int)W);
        P->ldsRowI = 2 * P->csz * (size_t)lpad_size((int)uW);
        if (cplx) {                                          // long non-R2C rows: one buffer, in place (rows_fit above)
            P->inplaceF = rows_fit(W) == 1; P->inplaceI = rows_fit(uW) == 1;
            if (P->inplaceF) P->ldsRowF /= 2;
            if (P->inplaceI) P->ldsRowI /= 2;
            auto four = [&](fftup_plan::Four& f, uint32_t n) {           // ... or four steps through HBM
                f.on = split_four(n, P->csz, &f.n1, &f.n2, &f.tka, &f.tkb);
                f.p1 = make_stage_plan((uint32_t)f.n1); f.p2 = make_stage_plan((uint32_t)f.n2);
                f.ldsA = 2 * P->csz * (size_t)lpad_size(f.n1 * f.tka); f.ldsB = 2 * P->csz * (size_t)lpad_size(f.n2 * f.tkb);
                const int tmax = kernels_generic_max_threads(P->dbl);
                f.thrA = std::min(tmax, std::max(64, round_up(f.n1 * f.tka / 8, 64)));
                f.thrB = std::min(tmax, std::max(64, round_up(f.n2 * f.tkb / 8, 64)));
            };
            if (!rows_fit(W)) { four(P->fourF, W); P->ldsRowF = 0; }
            if (!rows_fit(uW)) { four(P->fourI, uW); P->ldsRowI = 0; }
        }
        if (P->ldsRowI > lds_max) { rc = fail(FFTUP_E_UNSUPPORTED_SIZE, "upscaled width too large for LDS"); goto bad; }
        {
            const int tmax = kernels_generic_max_threads(P->dbl);
            P->thrW = std::min(tmax, std::max(64, round_up((int)W / 8, 64)));
            P->thrUW = std::min(tmax, std::max(64, round_up((int)uW / 8, 64)));
            if (!(P->poly || P->inplaceC)) P->thrCol = std::min(tmax, std::max(64, round_up((int)uH * P->TK / 8, 64)));     // (in-place column plans chose theirs above)
            // -p 1 R2C rows: one LDS buffer where every stage runs in place with 8 points per thread (two workgroups per compute unit)
            if (P->dbl && !cplx) {
                const int tf = inplace_threads(P->planW, 1, 8, tmax), ti = inplace_threads(P->planUW, 1, 8, tmax);
                if (tf) { P->inplaceF = true; P->thrW = tf; P->ldsRowF /= 2; }
                if (ti) { P->inplaceI = true; P->thrUW = ti; P->ldsRowI /= 2; }
            }
        }

        P->upsq = const_via_percent_f((double)(cfg->upscale * cfg->upscale), P->half);   // VkResample.cpp:1615
        P->coef = const_via_percent_f((double)cfg->sharpen, P->half);                    // VkResample.cpp:1616

        PLAN_RC(make_twiddles(P, &P->twW, W));
        PLAN_RC(make_twiddles(P, &P->twH, H));
        PLAN_RC(make_twiddles(P, &P->twUW, uW));
        PLAN_RC(make_twiddles(P, &P->twUH, uH));
        for (fftup_plan::Four* f : {&P->fourF, &P->fourI, &P->colF, &P->colI})
            if (f->on) { PLAN_RC(make_twiddles(P, &f->tw1, (uint32_t)f->n1)); PLAN_RC(make_twiddles(P, &f->tw2, (uint32_t)f->n2)); }

        const size_t esz = P->esz;
        P->in_plane_stride = (size_t)(W + 2) * H;                    // VkResample.cpp:1644
        P->in_planar.assign(P->ring, nullptr);
        P->in_u8.assign(P->ring, nullptr);
        P->in_kind.assign(P->ring, 0);
        P->out.assign(P->ring, nullptr);
        for (uint32_t s = 0; s < P->ring; s++) {
            PLAN_RC(dev_alloc(P, &P->in_planar[s], 3 * P->in_plane_stride * esz));
            PLAN_RC(dev_alloc(P, (void**)&P->in_u8[s], (size_t)3 * W * H));
            PLAN_RC(dev_alloc(P, &P->out[s], (size_t)3 * uW * uH * (P->u8out ? 1 : esz) + 8));          // (+ 8: readers of whole words)
        }
        // tuned plans (k_col_t): S2 holds the odd rows only and sits right behind S1 in ONE allocation (the fused
        // kernel addresses both with 32-bit offsets from one base)
        const size_t s1_elems = (size_t)3 * P->NT * H * P->TK;
        auto alloc_spectra = [&](float2** s1, float2** s2) -> int {
            if ((P->tuned || P->mixed) && P->U >= 2) {
                int r = dev_alloc(P, (void**)s1, P->csz * (size_t)P->U * s1_elems);       // S1 + the U-1 residue buffers
                *s2 = r ? nullptr : *s1 + s1_elems;
                return r;
            }
            int r = dev_alloc(P, (void**)s1, P->csz * s1_elems);
            return r ? r : dev_alloc(P, (void**)s2, P->csz * 3 * (size_t)P->NT * uH * P->TK);
        };
        PLAN_RC(alloc_spectra(&P->S1, &P->S2));
        // the pre-sharpen image (the reference's tempBuffer): every frame of an unfused plan goes through it; a fused plan
        // only needs one for the fftup_download_presharpen tap, which allocates it on first use (ensure_R)
        P->r_bytes = (size_t)3 * uW * uH * (cplx ? (P->half ? 4 : P->csz) : esz);  // non-R2C path: complex pre-sharpen image (binary16 pairs for -p 2)
        if (!P->fused) PLAN_RC(dev_alloc(P, &P->R, P->r_bytes));
        if (!P->u8out) PLAN_RC(dev_alloc(P, (void**)&P->out_u8, (size_t)3 * uW * uH + 8));   // staging of the conversion launch (+ 8: k_png_filter reads whole words)
        {
            P->nlanes = lane_count();
            P->lanes.resize(P->nlanes);
            P->lanes[0].stream = P->stream; P->lanes[0].S1 = P->S1; P->lanes[0].S2 = P->S2; P->lanes[0].R = P->R;
            const size_t t4_bytes = P->csz * 3 * std::max(std::max(P->fourF.on ? (size_t)W * H : 0, P->fourI.on ? (size_t)uW * uH : 0),
                                                          P->colI.on ? (size_t)P->ncols * uH : 0);
            if (t4_bytes) PLAN_RC(dev_alloc(P, &P->lanes[0].T4, t4_bytes));
            for (int l = 1; l < P->nlanes; l++) {
                PLAN_TRY(hipStreamCreateWithFlags(&P->lanes[l].stream, hipStreamNonBlocking));
                PLAN_TRY(hipEventCreateWithFlags(&P->lanes[l].done, hipEventDisableTiming));
                PLAN_RC(alloc_spectra(&P->lanes[l].S1, &P->lanes[l].S2));
                if (!P->fused) PLAN_RC(dev_alloc(P, &P->lanes[l].R, P->r_bytes));
                if (t4_bytes) PLAN_RC(dev_alloc(P, &P->lanes[l].T4, t4_bytes));
            }
        }
        PLAN_RC(kernels_set_attributes(P));          // dynamic LDS above 64 KB for the kernels THIS plan launches (fftup_launch.hip)
        if (P->mixed == 3 && ((cfg->flags & FFTUP_FLAG_TUNE_PLAN) || jit_tune_enabled())) tune_fused(P);
    }
    png_geometry(P);                       // (fixed per plan: fftup_png_bound may be asked by several threads at once)
    *out = P;
    return FFTUP_OK;
bad:
    fftup_plan_destroy(P);
    return rc;
#undef PLAN_TRY
#undef PLAN_RC
}

int fftup_plan_describe(const fftup_plan* P, char* buf, size_t buflen)
{
    if (!P || !buf || !buflen) return fail(FFTUP_E_INVALID_ARG, "null argument");
    std::string s;
    if (P->mixed == 3) s = "specialised at plan time: " + fftup_jit::describe(P->jit->choice);
    else if (P->tuned) s = "ahead-of-time power-of-two kernels (radix 8, 8 points per thread; fused C2R+sharpen " + std::string(P->fused ? "on" : "off") + ")"
                           + "; column kernel with digit-swap exchanges";
    else if (P->mixed) s = std::string("ahead-of-time mixed-radix kernels: ") + (P->mixed == 1 ? "row 15*8*16, col 9*10*12, fused 16*16*15" : "row 5*16*16, col 9*8*10, fused 16*16*10");
    else if (P->cplx) s = "size-generic kernels, non-R2C path (full complex transforms)";
    else s = std::string("size-generic kernels (") + ((P->inplaceF || P->inplaceI || P->inplaceC) ? "in place in one LDS buffer" : "LDS ping-pong") + ", run-time radix lists"
             + (P->poly ? ", polyphase column pass)" : ")")
             + (P->dbl ? ", double" : "");
    auto four = [&](const char* what, const fftup_plan::Four& f) {
        if (f.on) s += std::string("; ") + what + " in four steps " + std::to_string(f.n1) + "*" + std::to_string(f.n2) + " (tiles of " + std::to_string(f.tka) + " / " + std::to_string(f.tkb) + ")";
    };
    four("forward rows", P->fourF); four("inverse rows", P->fourI); four("forward columns", P->colF); four("inverse columns", P->colI);
    if (P->u8out) s += "; fused 8-bit RGB store";
    snprintf(buf, buflen, "%s", s.c_str());
    return FFTUP_OK;
}

int fftup_plan_info(const fftup_plan* P, fftup_info* info)
{
    if (!P || !info) return fail(FFTUP_E_INVALID_ARG, "null argument");
    memset(info, 0, sizeof *info);
    info->out_width = P->uW;
    info->out_height = P->uH;
    info->num_kernels = P->fused ? 3 : 4;
    info->tuned = P->mixed == 3 ? 2 : ((P->tuned || P->mixed) ? 1 : 0);
    // B_alg = in + 2*S1 + 2*S2 + 2*R + out
    const double C = 3.0, W = P->W, H = P->H, uW = P->uW, uH = P->uH;
    const bool fused_u8 = fuse_u8(P);
    const double b_in = fused_u8 ? 1.0 : (double)P->esz;
    const double b_r = (double)P->esz, b_out = P->u8out ? 1.0 : b_r, b_c = (double)P->csz;
    const double in = C * W * H * b_in;
    const double S1 = C * P->ncols * H * b_c;
    const double S2 = C * P->ncols * uH * b_c;
    const double R = C * uW * uH * (P->cplx ? b_c : b_r);
    const double o = C * uW * uH * b_out;
    info->alg_bytes_per_frame = in + 2 * S1 + 2 * S2 + 2 * R + o;
    info->kernel_alg_bytes[0] = in + S1;
    info->kernel_alg_bytes[1] = S1 + S2;
    // a fused C2R+sharpen launch does the work of the reference's I2 and C dispatches: its algorithmic
    // bytes stay S2 + 2R + out although R never reaches HBM
    info->kernel_alg_bytes[2] = P->fused ? S2 + 2 * R + o : S2 + R;
    info->kernel_alg_bytes[3] = P->fused ? 0.0 : R + o;
    {
        // what the launches really have to move: polyphase plans write/read only the odd half of S2; a fused strip
        // re-reads one halo pair of spectrum rows
        const bool poly = (P->tuned || P->mixed) && P->U >= 2;
        const double S2w = poly ? S1 * (P->U - 1) : S2;               // odd rows (residues 1..U-1) only
        const double halo = P->fused ? (double)(P->pairs_per_strip + 1) / P->pairs_per_strip : 1.0;
        info->kernel_min_bytes[0] = in + S1;
        info->kernel_min_bytes[1] = S1 + S2w;
        info->kernel_min_bytes[2] = P->fused ? S2 * halo + o : S2 + R;
        info->kernel_min_bytes[3] = P->fused ? 0.0 : R + o;
    }
    info->device_bytes = P->device_bytes;
    info->abi_version = FFTUP_ABI_VERSION;
    info->u8_store = P->u8out ? 1 : 0;
    snprintf(info->device_name, sizeof info->device_name, "%s", device_label(P->prop));
    snprintf(info->kernel_names[0], 64, P->cplx ? "row_c2c" : "row_r2c");
    snprintf(info->kernel_names[1], 64, "col_fwd_pad_inv");
    snprintf(info->kernel_names[2], 64, P->fused ? "row_c2r_sharpen" : (P->cplx ? "row_c2c_inv" : "row_c2r"));
    snprintf(info->kernel_names[3], 64, P->fused ? "-" : "sharpen");
    return FFTUP_OK;
}

const char* fftup_strerror(int code)
{
    switch (code) {
    case FFTUP_OK: return "success";
    case FFTUP_E_INVALID_ARG: return "invalid argument";
    case FFTUP_E_UNSUPPORTED_SIZE: return "unsupported size (not 2,3,5,7-smooth)";
    case FFTUP_E_UNSUPPORTED_PRECISION: return "unsupported precision";
    case FFTUP_E_NO_DEVICE: return "no usable HIP device";
    case FFTUP_E_HIP: return "HIP runtime error";
    case FFTUP_E_OUT_OF_MEMORY: return "out of device memory";
    case FFTUP_E_NO_INPUT: return "no input uploaded / nothing executed";
    case FFTUP_E_INCOMPLETE: return "incomplete (image not found)";
    case FFTUP_E_WOULD_BLOCK: return "the call would wait for the calling thread itself";
    case FFTUP_E_OVERFLOW: return "an internal buffer bound was exceeded";
    default: return "unknown error";
    }
}

const char* fftup_last_error(void) { return g_last_error.c_str(); }
const char* fftup_version(void) { return "fftup 0.5.0 (gfx950, ABI 2)"; }

}  // extern "C"

// Plan-time tuner (FFTUP_FLAG_TUNE_PLAN / experiment jit_tune=1) for a run-time specialised plan: the chooser's alternatives for
// the fused C2R+sharpen kernel -- the one that takes two thirds of a frame -- are compiled and the PLAN is timed with
// each of them on this device, the way it will run (frames overlapping on the plan's streams when it has a ring of slots,
// else one after the other: a kernel that is faster alone but fills the compute units' registers makes overlapping
// frames slower), with the plan's own buffers (their contents do not matter: no data-dependent control flow).
// The fastest one is kept and remembered in <cache dir>/wisdom.txt, which later plans for the same row length, device
// and mode read instead of measuring again.  Different factorizations give the same pixels up to fp32 rounding (tests).
static void tune_fused(fftup_plan* P)
{
    const std::string arch = P->prop.gcnArchName;
    const fftup_jit::Choice base = P->jit->choice;
    const std::string key = fftup_jit::fused_key(base, wisdom_device_key(P));
    std::string known;
    if (fftup_jit::experiment("jit_fused") || fftup_jit::wisdom_lookup(key, known)) return;
    const std::vector<int> kinds = P->in_kind;
    const int executed = P->executed;
    for (auto& k : P->in_kind) if (!k) k = 1;                                  // (uninitialised planar input: fine for timing)
    const uint32_t frames = 4 * (uint32_t)std::max(1, std::min(P->nlanes, (int)P->ring));
    auto time_plan = [&]() -> double {
        double best = 1e30, ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            if (execute_ring_impl(P, frames, 0, &ms, nullptr, 1) != FFTUP_OK) return 1e30;
            if (rep > 0) best = std::min(best, ms / frames);                    // (the first repetition warms up)
        }
        return best;
    };
    const double t_base = time_plan();
    if (t_base >= 1e30) {                                                       // the plan does not even run: nothing to learn, nothing to file
        P->in_kind = kinds;
        P->executed = executed;
        return;
    }
    double t_best = t_base;
    fftup_jit::Module* const original = P->jit;
    fftup_jit::Module* best = nullptr;
    // candidates: the chooser's alternatives -- and the structural default (pow2 / 16*16*R), when built-in wisdom made the
    // plan start from something else
    std::vector<fftup_jit::Choice> cands;
    {
        fftup_jit::Choice d;
        if (fftup_jit::choose(base.W, base.H, base.D, base.half, stage_radices(P->planUW), d, "", false, base.DD) &&
            fftup_jit::fused_value(d) != fftup_jit::fused_value(base)) {
            d.u8out = base.u8out;
            cands.push_back(d);
        }
    }
    for (const auto& cand : fftup_jit::fused_candidates(base.UW, base.D, 5, base.DD)) {
        if (base.fused_kind == 2 && cand.T == base.fused_t && cand.r == base.fr) continue;
        fftup_jit::Choice c = base;
        fftup_jit::set_fused_n(c, cand.T, cand.r);
        cands.push_back(c);
    }
    for (const fftup_jit::Choice& c : cands) {
        if (c.fused_lds > 160 * 1024) continue;
        std::string err;
        fftup_jit::Module* m = fftup_jit::load(c, arch, err);
        if (!m) continue;
        P->jit = m;
        set_strip_length(P);
        const double t = time_plan();
        P->jit = original;
        set_strip_length(P);
        if (getenv("FFTUP_JIT_VERBOSE"))
            fprintf(stderr, "fftup: tuning %s: %s %.1f us/frame (default %s %.1f)\n", key.c_str(), fftup_jit::fused_value(m->choice).c_str(), t * 1e3,
                    fftup_jit::fused_value(base).c_str(), t_base * 1e3);
        if (t < 0.97 * t_best) { delete best; best = m; t_best = t; }           // (3 %: do not chase noise)
        else delete m;
    }
    if (best) { delete original; P->jit = best; set_strip_length(P); }
    P->in_kind = kinds;
    P->executed = executed;
    fftup_jit::wisdom_store(key, fftup_jit::fused_value(P->jit->choice));
}
