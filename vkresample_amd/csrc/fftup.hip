// fftup.hip -- host side of the C ABI (include/fftup.h): plan construction, device buffers,
// kernel launches, timing.  Mirrors the plan semantics of launchResample() (VkResample.cpp:1409-1617)
// and the frame executor performVulkanUpscale() (VkResample.cpp:1249-1279) on one HIP stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fftup.h"
#include "kernels_generic.hpp"
#include "kernels_pow2.hpp"
#include "kernels_mixed.hpp"
#include "kernels_dswap.hpp"
#include "kernels_png.hpp"
#include "jit.hpp"

using namespace fftup;

static constexpr int TUNED_TK = 4;     // column tile width of the size-specialised kernels

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(_e == hipErrorOutOfMemory ? FFTUP_E_OUT_OF_MEMORY : FFTUP_E_HIP,           \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                        \
    } while (0)

// events that are destroyed on every exit path
struct EventList {
    std::vector<hipEvent_t> ev;
    ~EventList() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
    int create(size_t n)
    {
        ev.assign(n, nullptr);
        for (auto& e : ev) {
            hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) { e = nullptr; return fail(FFTUP_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(r)); }
        }
        return FFTUP_OK;
    }
    hipEvent_t& operator[](size_t i) { return ev[i]; }
};

struct fftup_plan {
    fftup_config cfg{};
    uint32_t W = 0, H = 0, uW = 0, uH = 0;
    uint32_t ring = 1;
    bool half = false;                // -p 2: binary16 storage
    bool dbl = false;                 // -p 1: double storage and arithmetic (size-generic kernels, double2 spectra)
    size_t esz = 4, csz = 8;          // bytes per real / complex element in HBM
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop{};

    // geometry
    int TK = 8, NT = 0;
    int zlx = 0, zrx = 0, zly = 0, zry = 0;
    StagePlan planW{}, planH{}, planUW{}, planUH{};
    int thrW = 0, thrCol = 0, thrUW = 0;
    size_t ldsRowF = 0, ldsCol = 0, ldsRowI = 0;
    float upsq = 0, coef = 0;
    bool tuned = false;
    bool fused = false;               // sharpen fused into the C2R kernel (tuned plans)
    bool u8out = false;               // FFTUP_FLAG_FUSE_U8_STORE in effect: the fused kernel stores 8-bit RGB, `out` slots hold [uH][uW][3] bytes
    int mixed = 0;                    // compile-time mixed-radix plans: 1 = 1920x1080 -> 3840x2160, 2 = 1280x720 -> 2560x1440,
                                      // 3 = specialised at plan time for this size (jit.hpp), kernels in `jit`
    fftup_jit::Module* jit = nullptr;
    int U = 2;                        // integer upscale factor of a polyphase plan (tuned / mixed): S1 + U-1 residue buffers
    bool cplx = false;                // non-R2C path (VR:1424 false): full complex transforms, uW beyond the R2C limit
    bool inplaceF = false, inplaceI = false;   // ... whose forward / inverse rows are too long for two LDS buffers: fft_lds_inplace
    // ... and rows too long for ONE buffer: four steps through HBM (k_row4_a / k_row4_b), row length = n1 * n2
    struct Four { bool on = false; int n1 = 0, n2 = 0, tk = 1; StagePlan p1{}, p2{}; float2 *tw1 = nullptr, *tw2 = nullptr; size_t ldsA = 0, ldsB = 0; int thrA = 64, thrB = 64; };
    Four fourF, fourI;
    Four colF, colI;                  // columns longer than the LDS (TK = 1): the same two kernels on dense columns
    int ncols = 0;                    // spectrum columns kept: W/2 + 1, or W on the non-R2C path
    int pairs_per_strip = 6;
    bool R_valid = false;             // pre-sharpen buffer holds the last frame (unfused path only)

    // device memory
    std::vector<void*> in_planar;     // per slot: planar float/half, row stride W, plane stride (W+2)*H
    std::vector<uint8_t*> in_u8;      // per slot: RGB u8 [H][W][3] (staging and fused-load source)
    std::vector<int> in_kind;         // per slot: 0 none, 1 planar valid, 2 u8 valid (fused)
    float2 *S1 = nullptr, *S2 = nullptr;
    void* R = nullptr;                // pre-sharpen, dense [3][uH][uW]
    // batched mode runs consecutive frames on `nlanes` streams (the reference's -numthreads does the same with
    // several queues on one device); every lane owns its scratch spectra.  Lane 0 = the members above.
    struct Lane { hipStream_t stream = nullptr; float2 *S1 = nullptr, *S2 = nullptr; void* R = nullptr; hipEvent_t done = nullptr;
                  void* T4 = nullptr; };             // T4: the four-step rows' transposition buffer
    std::vector<Lane> lanes;
    int nlanes = 1, cur = 0, last_lane = 0;
    std::vector<void*> out;           // per slot: dense [3][uH][uW]
    uint8_t* out_u8 = nullptr;        // staging for download_rgb8
    // host-streamed queue (fftup_submit_rgb8): created on first use
    // (png: the device-side PNG encoder's buffers of the slot, created on the first fftup_submit_png; state 1 = a stream waits for
    // its fftup_wait_png -- a later submission of the slot waits for that on q_cv)
    struct PngSlot { PngParams p{}; unsigned long long* meta_host = nullptr; uint32_t* parts_host = nullptr; hipEvent_t copied = nullptr; int state = 0; uint64_t ticket = 0; uint8_t* dest = nullptr; size_t dest_cap = 0; };
    struct QSlot { uint8_t* out_u8 = nullptr; hipEvent_t done = nullptr; PngSlot png; };
    std::vector<QSlot> q;
    std::atomic<uint64_t> q_next{0};   // next ticket; written under q_mu, read by fftup_wait without it
    std::mutex q_mu;                   // fftup_submit_rgb8 may be called by several host threads (codec workers sharing a plan)
    std::condition_variable q_cv;
    hipStream_t png_copy = nullptr;    // the sized D2H copies of fftup_wait_png
    int png_rpb = 0, png_nblocks = 0;  // rows per deflate block, blocks per frame
    size_t png_stream_bytes = 0;       // capacity of a slot's stream buffer
    float2 *twW = nullptr, *twH = nullptr, *twUW = nullptr, *twUH = nullptr;
    uint64_t device_bytes = 0;
    size_t r_bytes = 0;               // bytes of one pre-sharpen image
    uint64_t* d_sum = nullptr;        // fftup_output_checksum accumulator (created on first use)
    size_t in_plane_stride = 0;
    int executed = 0;

    std::vector<void*> allocs;
};

// ------------------------------------------------------------------------------------------------
static bool is_smooth(uint32_t n)
{
    if (n == 0) return false;
    for (uint32_t p : {2u, 3u, 5u, 7u})
        while (n % p == 0) n /= p;
    return n == 1;
}

// Four-step split of a row of n points that does not fit the LDS (k_row4_a / k_row4_b): n = n1 * n2, both transforms with
// their two Stockham buffers of tk interleaved sequences in 160 KB, as square as possible, tk = 4 where both factors allow it
static bool split_four(uint32_t n, size_t el, int* n1, int* n2, int* tk)
{
    long best = -1;
    for (uint32_t d = 2; d * d <= n; d++) {
        if (n % d) continue;
        const uint32_t a = d, b = n / d;                     // a <= b
        const int t = (a % 4 == 0 && b % 4 == 0) ? 4 : 1;
        if (2 * el * (size_t)lpad_size((int)b * t) > (size_t)160 * 1024) continue;
        const long score = (t == 4 ? 0 : (1l << 40)) + (long)(b - a);
        if (best < 0 || score < best) { best = score; *n1 = (int)a; *n2 = (int)b; *tk = t; }
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

static void png_geometry(fftup_plan* P);

static int dev_alloc(fftup_plan* P, void** ptr, size_t bytes)
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
static bool fuse_u8(const fftup_plan* P) { return (P->cfg.flags & FFTUP_FLAG_FUSE_U8_LOAD) && !P->dbl; }

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static bool jit_enabled()
{
    const char* e = getenv("FFTUP_JIT");
    return !e || atoi(e) != 0;
}
// 2 x the upscale factor when the specialised kernels' assumptions hold: the factor is an integer or a half-integer in
// [1.5, 8], the output sizes are exactly u W and u H, and the reference's zero-padding guard of the column pass
// (float arithmetic, VkResample.cpp:1494-1495) is exactly [H/2, uH - H/2).  0 otherwise.
static int jit_factor_x2(float upscale, uint32_t W, uint32_t H, uint32_t uW, uint32_t uH, int zly, int zry)
{
    const float two_u = 2.0f * upscale;
    const int D = (int)two_u;
    if ((float)D != two_u || D < 3 || D > 16) return 0;
    if (2 * (uint64_t)uW != (uint64_t)D * W || 2 * (uint64_t)uH != (uint64_t)D * H) return 0;
    if (zly != (int)(H / 2) || zry != (int)(uH - H / 2)) return 0;
    return D;
}
static void tune_fused(fftup_plan* P);

// HIP streams ("lanes") the frames of a plan alternate on
static int lane_count()
{
    int nl = 3;
    if (const char* e = getenv("FFTUP_STREAMS")) nl = atoi(e);
    return std::max(1, std::min(nl, 4));
}
// Do consecutive frames of this plan overlap on several streams?  A ring of slots (fftup_execute_ring, fftup_submit_rgb8) or the
// pipelined fftup_execute (every plan without FFTUP_FLAG_SEQUENTIAL_EXECUTE) -- as long as there is more than one stream.
static bool frames_overlap(const fftup_plan* P)
{
    return lane_count() > 1 && (P->ring > 1 || !(P->cfg.flags & FFTUP_FLAG_SEQUENTIAL_EXECUTE));
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
// Frames that run one after the other (FFTUP_FLAG_SEQUENTIAL_EXECUTE on a plan without a ring: the CLI's single image, -n 1):
// nothing runs beside a strip, and a workgroup of at most 512 threads (one or two waves per SIMD) does not hide its own
// latencies: two strips per unit (1080p 100 -> 91 us per iteration, 1000x1000 75 -> 62, 2048x1024 77.2 -> 76.0, -p 2
// 82.7 -> 79.7; 768 and 1024 threads: 2-7 % slower with two; profiles/r04_s_strips_per_unit_sequential.txt).
// How many workgroups are resident is the hardware's business.
static void set_strip_length(fftup_plan* P)
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
// four-step rows (k_row4_a / k_row4_b): launch both passes; ATTR: only allow their dynamic LDS sizes (plan creation)
template <typename C, int DIR, int MODE, int OUT, int TKS, bool ATTR>
static hipError_t four_passes(const fftup_plan::Four& f, const Row4Params<C>& q, int rows, hipStream_t st)
{
    if constexpr (ATTR) {
        hipError_t e = hipFuncSetAttribute((const void*)(k_row4_a<DIR, TKS, MODE, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.ldsA);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void*)(k_row4_b<DIR, TKS, OUT, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.ldsB);
    } else {
        hipLaunchKernelGGL((k_row4_a<DIR, TKS, MODE, C>), dim3(rows, f.n2 / TKS, 3), dim3(f.thrA), f.ldsA, st, q);
        hipLaunchKernelGGL((k_row4_b<DIR, TKS, OUT, C>), dim3(rows, f.n1 / TKS, 3), dim3(f.thrB), f.ldsB, st, q);
        return hipSuccess;
    }
}
template <typename C, int DIR, int MODE, int OUT, bool ATTR>
static hipError_t four_run(const fftup_plan::Four& f, const Row4Params<C>& q, int rows, hipStream_t st)
{
    return f.tk == 4 ? four_passes<C, DIR, MODE, OUT, 4, ATTR>(f, q, rows, st) : four_passes<C, DIR, MODE, OUT, 1, ATTR>(f, q, rows, st);
}
// forward rows of a plan: input mode from the slot's kind and the precision; inverse rows: output type from the precision
template <typename C, bool ATTR> static hipError_t four_forward(fftup_plan* P, const Row4Params<C>& q, int kind, hipStream_t st)
{
    if constexpr (sizeof(scalar_t<C>) == 8) return four_run<C, +1, IN_F64, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
    else {
        if (kind == 2) return P->half ? four_run<C, +1, IN_U8_F16, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st) : four_run<C, +1, IN_U8_F32, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
        return P->half ? four_run<C, +1, IN_F16, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st) : four_run<C, +1, IN_F32, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
    }
}
template <typename C, bool ATTR> static hipError_t four_inverse(fftup_plan* P, const Row4Params<C>& q, hipStream_t st)
{
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->half) return four_run<C, -1, IN4_TILES, OUT4_HALF, ATTR>(P->fourI, q, (int)P->uH, st);
    }
    return four_run<C, -1, IN4_TILES, OUT4_DENSE, ATTR>(P->fourI, q, (int)P->uH, st);
}
// columns longer than the LDS: forward in place in S1 (tiles of one column = dense columns), inverse S1 -> S2 with shift and guard
template <typename C, bool ATTR> static hipError_t four_columns(fftup_plan* P, hipStream_t st)
{
    using S = scalar_t<C>;
    Row4Params<C> q{};
    const fftup_plan::Four &f = P->colF, &g = P->colI;
    q.spec = (const C*)P->lanes[P->cur].S1; q.T = (C*)P->lanes[P->cur].T4; q.R = P->lanes[P->cur].S1;
    q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twH; q.plan1 = f.p1; q.plan2 = f.p2;
    q.N = (int)P->H; q.N1 = f.n1; q.N2 = f.n2; q.rows = P->ncols; q.W = (int)P->H; q.TK = 1; q.NT = P->ncols; q.inv_norm = (S)1;
    hipError_t e = four_run<C, +1, IN4_DENSE, OUT4_DENSE, ATTR>(f, q, P->ncols, st);
    if (e != hipSuccess) return e;
    q.R = P->lanes[P->cur].S2; q.tw1 = (const C*)g.tw1; q.tw2 = (const C*)g.tw2; q.twN = (const C*)P->twUH; q.plan1 = g.p1; q.plan2 = g.p2;
    q.N = (int)P->uH; q.N1 = g.n1; q.N2 = g.n2; q.zlx = P->zly; q.zrx = P->zry; q.inv_norm = (S)(1.0 / (double)P->uH);
    return four_run<C, -1, IN4_DENSE_SHIFT, OUT4_DENSE, ATTR>(g, q, P->ncols, st);
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
    snprintf(buf, buflen, "%s", prop.name);
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
    const int D = (uW & 1) || (uH & 1) || !is_smooth(uW) || !is_smooth(uH) || uW > 8192 ? 0 :
                  jit_factor_x2(upscale, width, height, uW, uH, (int)(uint32_t)((float)uH / (2 * upscale)), (int)(uint32_t)((2 * upscale - 1) * (float)uH / (2 * upscale)));
    if (!D || !fftup_jit::choose((int)width, (int)height, D, precision == 2, stage_radices(make_stage_plan(uW)), ch))
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
        int a, b, t;
        const size_t el = cfg->precision == 1 ? 16 : 8;
        if (cplx && ((!rows_fit(uW) && !split_four(uW, el, &a, &b, &t)) || (!rows_fit(W) && !split_four(W, el, &a, &b, &t))))
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
            P->ldsCol = sizeof(float2) * (size_t)lswz_size((int)H * TUNED_TK);     // both transforms of k_col_t have length H
        } else {
            // column tile width: widest of 8,4,2,1 whose ping-pong buffers fit in LDS
            for (int tk : {8, 4, 2, 1}) {
                size_t need = 2 * P->csz * (size_t)lpad_size((int)uH * tk);
                if (need <= lds_max) { P->TK = tk; P->ldsCol = need; break; }
            }
        }
        if (!P->TK) {
            // not even one column fits: tiles of one column, both column transforms in four steps through HBM (k_row4_a / k_row4_b)
            P->TK = 1; P->ldsCol = 0;
            for (auto fh : {std::make_pair(&P->colF, H), std::make_pair(&P->colI, uH)}) {
                fftup_plan::Four& f = *fh.first;
                f.on = split_four(fh.second, P->csz, &f.n1, &f.n2, &f.tk);
                if (!f.on) { rc = fail(FFTUP_E_UNSUPPORTED_SIZE, "column too long: no four-step split of the height fits the LDS"); goto bad; }
                f.p1 = make_stage_plan((uint32_t)f.n1); f.p2 = make_stage_plan((uint32_t)f.n2);
                f.ldsA = 2 * P->csz * (size_t)lpad_size(f.n1 * f.tk); f.ldsB = 2 * P->csz * (size_t)lpad_size(f.n2 * f.tk);
                const int tmax = P->dbl ? GenericMaxThreads<double2>::value : GenericMaxThreads<float2>::value;
                f.thrA = std::min(tmax, std::max(64, round_up(f.n1 * f.tk / 8, 64)));
                f.thrB = std::min(tmax, std::max(64, round_up(f.n2 * f.tk / 8, 64)));
            }
        }
        if (aot && !P->dbl && !cplx && !P->tuned && !(cfg->flags & FFTUP_FLAG_GENERIC_KERNELS) && uW == 2 * W && uH == 2 * H && P->TK >= 4) {
            if (W == MixedCfg1080::W && H == MixedCfg1080::H) P->mixed = 1;
            if (W == MixedCfg720::W && H == MixedCfg720::H) P->mixed = 2;
        }
        if (P->mixed) { P->TK = 4; P->ldsCol = sizeof(float2) * (size_t)H * 4; }             // k_col_m: one in-place buffer
        // any other size with an integer or half-integer upscale factor: kernels specialised for it now (the counterpart
        // of VkFFT generating its shaders at plan time)
        if (!P->dbl && !cplx && !P->tuned && !P->mixed && !(cfg->flags & (FFTUP_FLAG_GENERIC_KERNELS | FFTUP_FLAG_UNFUSED_SHARPEN)) && jit_enabled()) {
            const int D = jit_factor_x2(cfg->upscale, W, H, uW, uH, P->zly, P->zry);
            if (D) {
                fftup_jit::Choice ch;
                std::string jerr;
                if (fftup_jit::choose((int)W, (int)H, D, P->half, stage_radices(P->planUW), ch, wisdom_device_key(P))) {
                    ch.u8out = (cfg->flags & FFTUP_FLAG_FUSE_U8_STORE) != 0;         // (such a plan is always fused)
                    P->jit = fftup_jit::load(ch, P->prop.gcnArchName, jerr);
                    if (P->jit) { P->mixed = 3; P->U = ch.U; P->TK = 4; P->ldsCol = P->jit->choice.col_lds; }
                    else if (getenv("FFTUP_JIT_VERBOSE")) fprintf(stderr, "fftup: run-time specialisation failed, size-generic kernels in use: %s\n", jerr.c_str());
                }
            }
        }
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
                f.on = split_four(n, P->csz, &f.n1, &f.n2, &f.tk);
                f.p1 = make_stage_plan((uint32_t)f.n1); f.p2 = make_stage_plan((uint32_t)f.n2);
                f.ldsA = 2 * P->csz * (size_t)lpad_size(f.n1 * f.tk); f.ldsB = 2 * P->csz * (size_t)lpad_size(f.n2 * f.tk);
                const int tmax = P->dbl ? GenericMaxThreads<double2>::value : GenericMaxThreads<float2>::value;
                f.thrA = std::min(tmax, std::max(64, round_up(f.n1 * f.tk / 8, 64)));
                f.thrB = std::min(tmax, std::max(64, round_up(f.n2 * f.tk / 8, 64)));
            };
            if (!rows_fit(W)) { four(P->fourF, W); P->ldsRowF = 0; }
            if (!rows_fit(uW)) { four(P->fourI, uW); P->ldsRowI = 0; }
        }
        if (P->ldsRowI > lds_max) { rc = fail(FFTUP_E_UNSUPPORTED_SIZE, "upscaled width too large for LDS"); goto bad; }
        {
            const int tmax = P->dbl ? GenericMaxThreads<double2>::value : GenericMaxThreads<float2>::value;
            P->thrW = std::min(tmax, std::max(64, round_up((int)W / 8, 64)));
            P->thrUW = std::min(tmax, std::max(64, round_up((int)uW / 8, 64)));
            P->thrCol = std::min(tmax, std::max(64, round_up((int)uH * P->TK / 8, 64)));
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

        // allow > 64 KB dynamic LDS -- for the kernels THIS plan launches, nothing else
#define SET_LDS(kern, bytes) PLAN_TRY(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
        const bool generic = !P->tuned && !P->mixed;
        // (same predicate as launch_frame: a plan-time plan without a row factorization runs the size-generic row kernel)
        const bool generic_rows = generic || (P->mixed == 3 && P->jit->choice.row_kind == 2);
        if (generic_rows && !cplx && !P->dbl) {
            if (P->half) { SET_LDS(k_row_r2c<IN_F16>, P->ldsRowF); SET_LDS(k_row_r2c<IN_U8_F16>, P->ldsRowF); }
            else { SET_LDS(k_row_r2c<IN_F32>, P->ldsRowF); SET_LDS(k_row_r2c<IN_U8_F32>, P->ldsRowF); }
        }
        if (generic && !cplx && !P->dbl) {
            if (P->half) SET_LDS(k_row_c2r<true>, P->ldsRowI); else SET_LDS(k_row_c2r<false>, P->ldsRowI);
        }
        if (P->colF.on) {                                    // columns in four steps (k_row4_a / k_row4_b on dense columns)
            if (P->dbl) PLAN_TRY((four_columns<double2, true>(P, nullptr))); else PLAN_TRY((four_columns<float2, true>(P, nullptr)));
        }
        if (generic && !P->dbl && !P->colF.on) {
            switch (P->TK) {
            case 8: SET_LDS(k_col<8>, P->ldsCol); break;
            case 4: SET_LDS(k_col<4>, P->ldsCol); break;
            case 2: SET_LDS(k_col<2>, P->ldsCol); break;
            default: SET_LDS(k_col<1>, P->ldsCol); break;
            }
        }
        if (cplx) {
            // (one instantiation per input type / output type / one-or-two-buffer form: only this plan's)
            const bool f1 = !P->fourF.on, i1 = !P->fourI.on;         // rows in one launch (else: four steps, below)
            if (P->dbl) { if (f1) SET_LDS((k_row_c2c_fwd<IN_F64, double2>), P->ldsRowF); if (i1) SET_LDS((k_row_c2c_inv<double2>), P->ldsRowI); }
            else if (P->half) {
                if (!f1) {}
                else if (P->inplaceF) { SET_LDS((k_row_c2c_fwd<IN_F16, float2, true>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F16, float2, true>), P->ldsRowF); }
                else { SET_LDS((k_row_c2c_fwd<IN_F16, float2>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F16, float2>), P->ldsRowF); }
                if (!i1) {}
                else if (P->inplaceI) SET_LDS((k_row_c2c_inv<float2, true, true>), P->ldsRowI);
                else SET_LDS((k_row_c2c_inv<float2, true, false>), P->ldsRowI);
            } else {
                if (!f1) {}
                else if (P->inplaceF) { SET_LDS((k_row_c2c_fwd<IN_F32, float2, true>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F32, float2, true>), P->ldsRowF); }
                else { SET_LDS((k_row_c2c_fwd<IN_F32, float2>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F32, float2>), P->ldsRowF); }
                if (!i1) {}
                else if (P->inplaceI) SET_LDS((k_row_c2c_inv<float2, false, true>), P->ldsRowI);
                else SET_LDS((k_row_c2c_inv<float2, false, false>), P->ldsRowI);
            }
            if (P->fourF.on) {
                if (P->dbl) PLAN_TRY((four_forward<double2, true>(P, Row4Params<double2>{}, 1, nullptr)));
                else { PLAN_TRY((four_forward<float2, true>(P, Row4Params<float2>{}, 1, nullptr))); PLAN_TRY((four_forward<float2, true>(P, Row4Params<float2>{}, 2, nullptr))); }
            }
            if (P->fourI.on) {
                if (P->dbl) PLAN_TRY((four_inverse<double2, true>(P, Row4Params<double2>{}, nullptr)));
                else PLAN_TRY((four_inverse<float2, true>(P, Row4Params<float2>{}, nullptr)));
            }
        }
        if (P->dbl) {
            if (!cplx) { SET_LDS((k_row_r2c<IN_F64, double2>), P->ldsRowF); SET_LDS((k_row_c2r<false, double2>), P->ldsRowI); }
            if (!P->colF.on) switch (P->TK) {
            case 8: SET_LDS((k_col<8, double2>), P->ldsCol); break;
            case 4: SET_LDS((k_col<4, double2>), P->ldsCol); break;
            case 2: SET_LDS((k_col<2, double2>), P->ldsCol); break;
            default: SET_LDS((k_col<1, double2>), P->ldsCol); break;
            }
        }
#define SET_FUSED(PL, TKK) do { if (P->u8out) { if (P->half) SET_LDS((k_c2r_sharpen_g<PL, true, TKK, 2, 4, true>), FusedGLds<PL>::TOTAL); \
                                                else SET_LDS((k_c2r_sharpen_g<PL, false, TKK, 2, 4, true>), FusedGLds<PL>::TOTAL); } \
                                else if (P->half) SET_LDS((k_c2r_sharpen_g<PL, true, TKK>), FusedGLds<PL>::TOTAL); \
                                else SET_LDS((k_c2r_sharpen_g<PL, false, TKK>), FusedGLds<PL>::TOTAL); } while (0)
#define SET_MIXED(CFG) do { SET_LDS(k_col_m<CFG>, P->ldsCol); \
        if (P->half) SET_LDS((k_row_c2r_ct<CFG::CT, true>), P->ldsRowI); else SET_LDS((k_row_c2r_ct<CFG::CT, false>), P->ldsRowI); \
        SET_FUSED(CFG::FUSED, 4); } while (0)
        if (P->mixed == 1) { SET_MIXED(MixedCfg1080); }
        if (P->mixed == 2) { SET_MIXED(MixedCfg720); }
#undef SET_MIXED
        if (P->tuned) {
            switch (uW) {
            case 1024: SET_FUSED(FusedPlanPow2<1024>, TUNED_TK); break;
            case 2048: SET_FUSED(FusedPlanPow2<2048>, TUNED_TK); break;
            default: SET_FUSED(FusedPlanPow2<4096>, TUNED_TK); break;
            }
            switch (H) {                                      // (digit-swap column kernels: 4 KB of LDS per wave)
            case 256: SET_LDS((k_col_v<TUNED_TK, 256>), 8192); break;
            case 512: SET_LDS((k_col_v<TUNED_TK, 512>), 16384); break;
            default: SET_LDS((k_col_v<TUNED_TK, 1024>), 32768); break;
            }
        }
#undef SET_FUSED
#undef SET_LDS
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
    else s = std::string("size-generic kernels (LDS ping-pong, run-time radix lists)") + (P->dbl ? ", double" : "");
    auto four = [&](const char* what, const fftup_plan::Four& f) {
        if (f.on) s += std::string("; ") + what + " in four steps " + std::to_string(f.n1) + "*" + std::to_string(f.n2) + (f.tk == 4 ? "" : " (one sequence per workgroup)");
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
    snprintf(info->device_name, sizeof info->device_name, "%s", P->prop.name);
    snprintf(info->kernel_names[0], 64, P->cplx ? "row_c2c" : "row_r2c");
    snprintf(info->kernel_names[1], 64, "col_fwd_pad_inv");
    snprintf(info->kernel_names[2], 64, P->fused ? "row_c2r_sharpen" : (P->cplx ? "row_c2c_inv" : "row_c2r"));
    snprintf(info->kernel_names[3], 64, P->fused ? "-" : "sharpen");
    return FFTUP_OK;
}

// ------------------------------------------------------------------------------------------------
// the two host loops of the reference as kernels (VR:1636-1685, VR:1708-1748)
static void launch_unpack(fftup_plan* P, uint32_t slot, hipStream_t st)
{
    dim3 grid((P->W + 255) / 256, P->H);
    if (P->dbl)
        hipLaunchKernelGGL(k_unpack_u8_f64, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, (double*)P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
    else if (P->half)
        hipLaunchKernelGGL(k_unpack_u8<true>, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
    else
        hipLaunchKernelGGL(k_unpack_u8<false>, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
}

static void launch_pack(fftup_plan* P, uint32_t slot, uint8_t* dst, hipStream_t st)
{
    dim3 grid(P->dbl ? (P->uW + 255) / 256 : (P->uW + 1023) / 1024, P->uH);      // (float / half: four pixels per thread)
    const int wrap = (P->cfg.flags & FFTUP_FLAG_U8_WRAP) ? 1 : 0;
    if (P->dbl) hipLaunchKernelGGL(k_pack_u8_f64, grid, dim3(256), 0, st, (const double*)P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
    else if (P->half) hipLaunchKernelGGL(k_pack_u8<true>, grid, dim3(256), 0, st, P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
    else hipLaunchKernelGGL(k_pack_u8<false>, grid, dim3(256), 0, st, P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
}

static int check_slot(fftup_plan* P, uint32_t slot)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (slot >= P->ring) return fail(FFTUP_E_INVALID_ARG, "slot out of range");
    return FFTUP_OK;
}

int fftup_upload_rgb8_slot(fftup_plan* P, uint32_t slot, const uint8_t* rgb, size_t row_stride_bytes)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!rgb || row_stride_bytes < (size_t)3 * P->W) return fail(FFTUP_E_INVALID_ARG, "bad rgb pointer/stride");
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipMemcpy2DAsync(P->in_u8[slot], (size_t)3 * P->W, rgb, row_stride_bytes, (size_t)3 * P->W, P->H,
                             hipMemcpyHostToDevice, P->stream));
    if (fuse_u8(P)) {
        P->in_kind[slot] = 2;
    } else {
        launch_unpack(P, slot, P->stream);
        HIP_TRY(hipGetLastError());
        P->in_kind[slot] = 1;
    }
    HIP_TRY(hipStreamSynchronize(P->stream));   // blocking, like transferDataFromCPU (VkResample.cpp:385-429)
    return FFTUP_OK;
}

int fftup_upload_rgb8(fftup_plan* P, const uint8_t* rgb, size_t row_stride_bytes)
{
    return fftup_upload_rgb8_slot(P, 0, rgb, row_stride_bytes);
}

int fftup_upload_planar(fftup_plan* P, uint32_t slot, const void* planes, size_t row_stride, size_t plane_stride)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!planes || row_stride < P->W || plane_stride < row_stride * (P->H - 1) + P->W)
        return fail(FFTUP_E_INVALID_ARG, "bad planes pointer/strides");
    HIP_TRY(hipSetDevice(P->device));
    const size_t esz = P->esz;
    for (int c = 0; c < 3; c++)
        HIP_TRY(hipMemcpy2DAsync((char*)P->in_planar[slot] + c * P->in_plane_stride * esz, P->W * esz,
                                 (const char*)planes + c * plane_stride * esz, row_stride * esz, P->W * esz, P->H,
                                 hipMemcpyHostToDevice, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    P->in_kind[slot] = 1;
    return FFTUP_OK;
}

// ------------------------------------------------------------------------------------------------
// one frame: 4 launches on the plan's stream.  `which` < 0 launches all, otherwise only that one.
}  // extern "C"

template <int W> static void launch_r2c_t(fftup_plan* P, const RowR2CTParams& p, int mode)
{
    dim3 grid(P->H / 2, 3), block(W / 8);
    switch (mode) {
    case IN_F32: hipLaunchKernelGGL((k_row_r2c_t<W, IN_F32, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    case IN_F16: hipLaunchKernelGGL((k_row_r2c_t<W, IN_F16, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    case IN_U8_F32: hipLaunchKernelGGL((k_row_r2c_t<W, IN_U8_F32, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    default: hipLaunchKernelGGL((k_row_r2c_t<W, IN_U8_F16, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    }
}
template <int UW> static void launch_c2r_t(fftup_plan* P, const RowC2RTParams& p)
{
    dim3 grid(P->uH / 2, 3), block(UW / 8);
    if (P->half) hipLaunchKernelGGL((k_row_c2r_t<UW, true, TUNED_TK, true>), grid, block, 0, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_row_c2r_t<UW, false, TUNED_TK, true>), grid, block, 0, P->lanes[P->cur].stream, p);
}

// workgroups of the fused C2R+sharpen kernel.  Planes: strips in linear order over the 3 uH/2 row pairs.  Fused 8-bit store:
// strips per plane, the three planes' strips of the same rows 8 workgroups apart, rows of 8 strips (k_c2r_sharpen_g, OUT_U8)
static unsigned fused_grid(const fftup_plan* P, int pairs_per_strip)
{
    const int ppp = (int)P->uH / 2;
    if (P->u8out) return (unsigned)(((ppp + pairs_per_strip - 1) / pairs_per_strip + 7) / 8 * 24);
    return (unsigned)((3 * ppp + pairs_per_strip - 1) / pairs_per_strip);
}
template <class PL> static void launch_fused_t(fftup_plan* P, const FusedParams& p)
{
    dim3 grid(fused_grid(P, p.pairs_per_strip)), block(PL::T);
    hipStream_t st = P->lanes[P->cur].stream;
    if (P->u8out) {
        if (P->half) hipLaunchKernelGGL((k_c2r_sharpen_g<PL, true, TUNED_TK, 2, 4, true>), grid, block, FusedGLds<PL>::TOTAL, st, p);
        else hipLaunchKernelGGL((k_c2r_sharpen_g<PL, false, TUNED_TK, 2, 4, true>), grid, block, FusedGLds<PL>::TOTAL, st, p);
    }
    else if (P->half) hipLaunchKernelGGL((k_c2r_sharpen_g<PL, true, TUNED_TK>), grid, block, FusedGLds<PL>::TOTAL, st, p);
    else hipLaunchKernelGGL((k_c2r_sharpen_g<PL, false, TUNED_TK>), grid, block, FusedGLds<PL>::TOTAL, st, p);
}
static FusedParams fused_params(fftup_plan* P, uint32_t out_slot)
{
    FusedParams p{};
    p.S1 = P->lanes[P->cur].S1; p.odd_delta = (unsigned)(P->lanes[P->cur].S2 - P->lanes[P->cur].S1);
    if (P->U == 1) { p.S1 = P->lanes[P->cur].S2; p.odd_delta = 0; }      // half-integer factor: one buffer with all rows (k_col_pad)
    p.out = P->out[out_slot]; p.tw = P->twUW; p.uH = (int)P->uH; p.NT = P->NT;
    p.pairs_per_strip = P->pairs_per_strip; p.upsq = P->upsq; p.coef = P->coef;
    p.u8_wrap = (P->cfg.flags & FFTUP_FLAG_U8_WRAP) ? 1 : 0;
    return p;
}

static bool fast_sharpen_ok(const fftup_plan* P) { return !P->dbl && P->uW % 256 == 0 && P->uH % 16 == 0; }

static int launch_frame_tuned(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    const int kind = P->in_kind[in_slot];
    if (which < 0 || which == 0) {
        RowR2CTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.tw = P->twW; p.H = (int)P->H; p.NT = P->NT;
        int mode;
        if (kind == 2) { p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0; mode = P->half ? IN_U8_F16 : IN_U8_F32; }
        else { p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride; mode = P->half ? IN_F16 : IN_F32; }
        switch (P->W) {
        case 512: launch_r2c_t<512>(P, p, mode); break;
        case 1024: launch_r2c_t<1024>(P, p, mode); break;
        default: launch_r2c_t<2048>(P, p, mode); break;
        }
    }
    if (which < 0 || which == 1) {
        ColTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.twH = P->twH; p.twUH = P->twUH; p.W = (int)P->W; p.NT = P->NT;
        switch (P->H) {
        case 256: hipLaunchKernelGGL((k_col_v<TUNED_TK, 256>), dim3(P->NT, 3), dim3(128), 8192, P->lanes[P->cur].stream, p); break;
        case 512: hipLaunchKernelGGL((k_col_v<TUNED_TK, 512>), dim3(P->NT, 3), dim3(256), 16384, P->lanes[P->cur].stream, p); break;
        default: hipLaunchKernelGGL((k_col_v<TUNED_TK, 1024>), dim3(P->NT, 3), dim3(512), 32768, P->lanes[P->cur].stream, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fused) {
        const FusedParams p = fused_params(P, out_slot);
        switch (P->uW) {
        case 1024: launch_fused_t<FusedPlanPow2<1024>>(P, p); break;
        case 2048: launch_fused_t<FusedPlanPow2<2048>>(P, p); break;
        default: launch_fused_t<FusedPlanPow2<4096>>(P, p); break;
        }
        P->R_valid = false;
    } else if (which < 0 || which == 2 || which == 22) {   // 22: pre-sharpen tap requested for a fused plan
        RowC2RTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = P->twUW; p.uH = (int)P->uH; p.NT = P->NT;
        switch (P->uW) {
        case 1024: launch_c2r_t<1024>(P, p); break;
        case 2048: launch_c2r_t<2048>(P, p); break;
        default: launch_c2r_t<4096>(P, p); break;
        }
        P->R_valid = true;
    }
    return FFTUP_OK;
}

static void launch_sharpen_fast(fftup_plan* P, uint32_t out_slot)
{
    SharpenTParams p{};
    p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
    dim3 grid(P->uW / 256, P->uH / 16, 3), block(64, 4);
    if (P->half) hipLaunchKernelGGL((k_sharpen_t<true, 4>), grid, block, 0, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_sharpen_t<false, 4>), grid, block, 0, P->lanes[P->cur].stream, p);
}

// -p 1: the size-generic kernels instantiated on double2 + the double sharpen
static int launch_frame_f64(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    hipStream_t st = P->lanes[P->cur].stream;
    if (which < 0 || which == 0) {
        RowR2CParamsT<double2> p{};
        p.S1 = (double2*)P->lanes[P->cur].S1; p.tw = (const double2*)P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
        hipLaunchKernelGGL((k_row_r2c<IN_F64, double2>), dim3(P->H / 2, 3), dim3(P->thrW), P->ldsRowF, st, p);
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<double2, false>(P, st);
    else if (which < 0 || which == 1) {
        ColParamsT<double2> p{};
        p.S1 = (const double2*)P->lanes[P->cur].S1; p.S2 = (double2*)P->lanes[P->cur].S2;
        p.twH = (const double2*)P->twH; p.twUH = (const double2*)P->twUH; p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = 1.0 / (double)P->uH;
        dim3 grid(P->NT, 3), block(P->thrCol);
        switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col<8, double2>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col<4, double2>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col<2, double2>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col<1, double2>), grid, block, P->ldsCol, st, p); break;
        }
    }
    if (which < 0 || which == 2) {
        RowC2RParamsT<double2> p{};
        p.S2 = (const double2*)P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = (const double2*)P->twUW; p.plan = P->planUW;
        p.W = (int)P->W; p.uW = (int)P->uW; p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = 1.0 / (double)P->uW;
        hipLaunchKernelGGL((k_row_c2r<false, double2>), dim3(P->uH / 2, 3), dim3(P->thrUW), P->ldsRowI, st, p);
    }
    if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        hipLaunchKernelGGL(k_sharpen_f64, dim3((P->uW + 255) / 256, P->uH, 3), dim3(256), 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

// non-R2C path: four launches of size-generic kernels on complex data
template <typename C, int MODE> static void launch_c2c_fwd(fftup_plan* P, const RowR2CParamsT<C>& p, hipStream_t st)
{
    const dim3 grid(P->H, 3);
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->inplaceF) { hipLaunchKernelGGL((k_row_c2c_fwd<MODE, C, true>), grid, dim3(1024), P->ldsRowF, st, p); return; }
    }
    hipLaunchKernelGGL((k_row_c2c_fwd<MODE, C, false>), grid, dim3(P->thrW), P->ldsRowF, st, p);
}
template <typename C, bool HALF_OUT> static void launch_c2c_inv(fftup_plan* P, const RowC2RParamsT<C>& p, hipStream_t st)
{
    const dim3 grid(P->uH, 3);
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->inplaceI) { hipLaunchKernelGGL((k_row_c2c_inv<C, HALF_OUT, true>), grid, dim3(1024), P->ldsRowI, st, p); return; }
    }
    hipLaunchKernelGGL((k_row_c2c_inv<C, HALF_OUT, false>), grid, dim3(P->thrUW), P->ldsRowI, st, p);
}
template <typename C> static int launch_frame_cplx(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    hipStream_t st = P->lanes[P->cur].stream;
    const int kind = P->in_kind[in_slot];
    using S = scalar_t<C>;
    if ((which < 0 || which == 0) && P->fourF.on) {             // rows beyond one LDS buffer: four steps through HBM
        Row4Params<C> q{};
        const fftup_plan::Four& f = P->fourF;
        q.T = (C*)P->lanes[P->cur].T4; q.S1 = (C*)P->lanes[P->cur].S1; q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twW;
        q.plan1 = f.p1; q.plan2 = f.p2; q.N = (int)P->W; q.N1 = f.n1; q.N2 = f.n2; q.rows = (int)P->H; q.W = (int)P->W; q.TK = P->TK; q.NT = P->NT;
        if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; }
        else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; }
        (void)four_forward<C, false>(P, q, kind, st);
    } else if (which < 0 || which == 0) {
        RowR2CParamsT<C> p{};
        p.S1 = (C*)P->lanes[P->cur].S1; p.tw = (const C*)P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        if (kind == 2) {
            p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0;
            if constexpr (sizeof(S) == 4) {
                if (P->half) launch_c2c_fwd<C, IN_U8_F16>(P, p, st);
                else launch_c2c_fwd<C, IN_U8_F32>(P, p, st);
            }
        } else {
            p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
            if constexpr (sizeof(S) == 8) launch_c2c_fwd<C, IN_F64>(P, p, st);
            else if (P->half) launch_c2c_fwd<C, IN_F16>(P, p, st);
            else launch_c2c_fwd<C, IN_F32>(P, p, st);
        }
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<C, false>(P, st);
    else if (which < 0 || which == 1) {
        ColParamsT<C> p{};
        p.S1 = (const C*)P->lanes[P->cur].S1; p.S2 = (C*)P->lanes[P->cur].S2; p.twH = (const C*)P->twH; p.twUH = (const C*)P->twUH;
        p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = (S)(1.0 / (double)P->uH);
        dim3 grid(P->NT, 3), block(P->thrCol);
        switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col<8, C>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col<4, C>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col<2, C>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col<1, C>), grid, block, P->ldsCol, st, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fourI.on) {
        Row4Params<C> q{};
        const fftup_plan::Four& f = P->fourI;
        q.spec = (const C*)P->lanes[P->cur].S2; q.T = (C*)P->lanes[P->cur].T4; q.R = P->lanes[P->cur].R;
        q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twUW; q.plan1 = f.p1; q.plan2 = f.p2;
        q.N = (int)P->uW; q.N1 = f.n1; q.N2 = f.n2; q.rows = (int)P->uH; q.W = (int)P->W; q.TK = P->TK; q.NT = P->NT; q.zlx = P->zlx; q.zrx = P->zrx;
        q.inv_norm = (S)(1.0 / (double)P->uW);
        (void)four_inverse<C, false>(P, q, st);
        P->R_valid = true;
    } else if (which < 0 || which == 2) {
        RowC2RParamsT<C> p{};
        p.S2 = (const C*)P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = (const C*)P->twUW; p.plan = P->planUW;
        p.W = (int)P->W; p.uW = (int)P->uW; p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = (S)(1.0 / (double)P->uW);
        bool done = false;
        if constexpr (sizeof(S) == 4) {
            if (P->half) { launch_c2c_inv<C, true>(P, p, st); done = true; }
        }
        if (!done) launch_c2c_inv<C, false>(P, p, st);
        P->R_valid = true;
    }
    if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        bool done = false;
        if constexpr (sizeof(S) == 4) {
            if (P->half) { hipLaunchKernelGGL((k_sharpen_c<C, true>), dim3((P->uW + 255) / 256, P->uH, 3), dim3(256), 0, st, p); done = true; }
        }
        if (!done) hipLaunchKernelGGL((k_sharpen_c<C>), dim3((P->uW + 255) / 256, P->uH, 3), dim3(256), 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

// row R2C / stand-alone C2R launches of the register-resident mixed-radix plans (kernels_mixed.hpp)
template <class CFG> static void launch_row_mixed(fftup_plan* P, uint32_t in_slot, int kind)
{
    RowR2CTParams q{};
    q.S1 = P->lanes[P->cur].S1; q.tw = P->twW; q.H = (int)P->H; q.NT = P->NT;
    int mode;
    if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; mode = P->half ? IN_U8_F16 : IN_U8_F32; }
    else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; mode = P->half ? IN_F16 : IN_F32; }
    hipStream_t st = P->lanes[P->cur].stream;
    const dim3 grid(P->H / 2, 3), block(CFG::ROW_T);
    switch (mode) {
    case IN_F32: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_F32>), grid, block, 0, st, q); break;
    case IN_F16: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_F16>), grid, block, 0, st, q); break;
    case IN_U8_F32: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_U8_F32>), grid, block, 0, st, q); break;
    default: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_U8_F16>), grid, block, 0, st, q); break;
    }
}
template <class CT> static void launch_c2r_ct(fftup_plan* P, dim3 grid, const RowC2RParams& p)
{
    if (P->half) hipLaunchKernelGGL((k_row_c2r_ct<CT, true>), grid, dim3(CT::T), P->ldsRowI, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_row_c2r_ct<CT, false>), grid, dim3(CT::T), P->ldsRowI, P->lanes[P->cur].stream, p);
}

// (a frame's later launches must not mask the failure of an earlier one)
static void keep_first(hipError_t& first, hipError_t e) { if (first == hipSuccess) first = e; }

static int launch_frame(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    const int kind = P->in_kind[in_slot];
    if (kind == 0 && which != 22) return fail(FFTUP_E_NO_INPUT, "no input uploaded for this slot");     // (22, the pre-sharpen tap, reads spectra only)
    if (P->cplx) return P->dbl ? launch_frame_cplx<double2>(P, in_slot, out_slot, which) : launch_frame_cplx<float2>(P, in_slot, out_slot, which);
    if (P->dbl) return launch_frame_f64(P, in_slot, out_slot, which);
    if (P->tuned) {
        launch_frame_tuned(P, in_slot, out_slot, which);
        if ((which < 0 || which == 3) && !P->fused) launch_sharpen_fast(P, out_slot);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        return FFTUP_OK;
    }
    hipError_t jerr = hipSuccess;       // launches from a run-time specialised code object report their errors directly
    if (which < 0 || which == 0) {
        RowR2CParams p{};
        p.S1 = P->lanes[P->cur].S1; p.tw = P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        dim3 grid(P->H / 2, 3), block(P->thrW);
        if (P->mixed == 3 && P->jit->choice.row_kind != 2) {
            RowR2CTParams q{};
            q.S1 = P->lanes[P->cur].S1; q.tw = P->twW; q.H = (int)P->H; q.NT = P->NT;
            if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; }
            else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; }
            keep_first(jerr, fftup_jit::launch(P->jit->fn[kind == 2 ? fftup_jit::K_ROW_U8 : fftup_jit::K_ROW_PLANAR], grid, dim3(P->jit->choice.row_block), 0,
                                     P->lanes[P->cur].stream, q));
        } else if (P->mixed == 1 || P->mixed == 2) {
            if (P->mixed == 1) launch_row_mixed<MixedCfg1080>(P, in_slot, kind); else launch_row_mixed<MixedCfg720>(P, in_slot, kind);
        } else if (kind == 2) {
            p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0;
            if (P->half) hipLaunchKernelGGL(k_row_r2c<IN_U8_F16>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
            else hipLaunchKernelGGL(k_row_r2c<IN_U8_F32>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
        } else {
            p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
            if (P->half) hipLaunchKernelGGL(k_row_r2c<IN_F16>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
            else hipLaunchKernelGGL(k_row_r2c<IN_F32>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
        }
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<float2, false>(P, P->lanes[P->cur].stream);
    else if (which < 0 || which == 1) {
        ColParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.twH = P->twH; p.twUH = P->twUH; p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = 1.0f / (float)P->uH;
        dim3 grid(P->NT, 3), block(P->thrCol);
        if (P->mixed) {
            ColTParams q{};
            q.S1 = P->lanes[P->cur].S1; q.S2 = P->lanes[P->cur].S2; q.twH = P->twH; q.twUH = P->twUH; q.W = (int)P->W; q.NT = P->NT;
            if (P->mixed == 3) {
                const auto& ch = P->jit->choice;
                const dim3 jgrid(P->NT * (ch.col_kind >= 3 ? 4 / ch.col_cols : 1), 3);        // (long columns: two per workgroup)
                keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_COL], jgrid, dim3(ch.col_block), P->ldsCol, P->lanes[P->cur].stream, q));
            }
            else if (P->mixed == 1) hipLaunchKernelGGL(k_col_m<MixedCfg1080>, grid, dim3(4 * MixedCfg1080::COL_TPC), P->ldsCol, P->lanes[P->cur].stream, q);
            else hipLaunchKernelGGL(k_col_m<MixedCfg720>, grid, dim3(4 * MixedCfg720::COL_TPC), P->ldsCol, P->lanes[P->cur].stream, q);
        } else switch (P->TK) {
        case 8: hipLaunchKernelGGL(k_col<8>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 4: hipLaunchKernelGGL(k_col<4>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 2: hipLaunchKernelGGL(k_col<2>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        default: hipLaunchKernelGGL(k_col<1>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fused) {
        if (P->mixed == 3) {
            const FusedParams fp = fused_params(P, out_slot);
            keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_FUSED], dim3(fused_grid(P, fp.pairs_per_strip)),
                                     dim3(P->jit->choice.fused_t), P->jit->choice.fused_lds, P->lanes[P->cur].stream, fp));
        } else if (P->mixed == 2) launch_fused_t<MixedCfg720::FUSED>(P, fused_params(P, out_slot));
        else launch_fused_t<MixedCfg1080::FUSED>(P, fused_params(P, out_slot));       // (only the mixed plans are fused on this path)
        P->R_valid = false;
    } else if (which < 0 || which == 2 || which == 22) {                 // 22: pre-sharpen tap requested for a fused plan
        RowC2RParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = P->twUW; p.plan = P->planUW; p.W = (int)P->W; p.uW = (int)P->uW;
        p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = 1.0f / (float)P->uW;
        dim3 grid(P->uH / 2, 3), block(P->thrUW);
        if (P->mixed) {
            if (P->mixed == 3) {
                if (P->U == 1) p.S1 = p.S2;                              // half-integer factor: all rows in S2
                keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_C2R_CT], grid, dim3(P->jit->choice.ct_t), P->ldsRowI, P->lanes[P->cur].stream, p));
            }
            else if (P->mixed == 1) launch_c2r_ct<MixedCfg1080::CT>(P, grid, p);
            else launch_c2r_ct<MixedCfg720::CT>(P, grid, p);
        } else if (P->half) hipLaunchKernelGGL(k_row_c2r<true>, grid, block, P->ldsRowI, P->lanes[P->cur].stream, p);
        else hipLaunchKernelGGL(k_row_c2r<false>, grid, block, P->ldsRowI, P->lanes[P->cur].stream, p);
        P->R_valid = true;
    }
    if (P->fused) {
        // sharpen is part of launch 2
    } else if ((which < 0 || which == 3) && fast_sharpen_ok(P)) {
        launch_sharpen_fast(P, out_slot);
    } else if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        dim3 grid((P->uW / 4 + 255) / 256, P->uH, 3), block(256);
        if (P->half) hipLaunchKernelGGL(k_sharpen<true>, grid, block, 0, P->lanes[P->cur].stream, p);
        else hipLaunchKernelGGL(k_sharpen<false>, grid, block, 0, P->lanes[P->cur].stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = jerr;
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

extern "C" {

// shared body of fftup_execute_ring / fftup_execute_ring_timed.  With kernel_ms != NULL a HIP event is recorded
// before and after every kernel launch ON THE STREAM THAT RUNS IT (one event chain per lane) and the average
// duration of each of the plan's kernels over this very batch is returned.
static int execute_ring_impl(fftup_plan* P, uint32_t n_frames, uint32_t first_slot, double* ms_total, double* kernel_ms,
                             uint32_t stride)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (n_frames == 0) return fail(FFTUP_E_INVALID_ARG, "n_frames must be > 0");
    HIP_TRY(hipSetDevice(P->device));
    const int nk = P->fused ? 3 : 4;
    // every `stride`-th frame is bracketed with events (an event record costs ~1 us of queue time each)
    if (stride == 0) stride = 1;
    const uint32_t n_timed = kernel_ms ? (n_frames + stride - 1) / stride : 0;
    EventList ev;
    if (kernel_ms) {
        int erc = ev.create((size_t)n_timed * (nk + 1));
        if (erc) return erc;
    }
    // consecutive frames go to distinct lanes; they must then also write distinct output slots
    const int nl = std::max(1, std::min(P->nlanes, (int)P->ring));
    HIP_TRY(hipEventRecord(P->ev0, P->stream));
    for (int l = 1; l < nl; l++) HIP_TRY(hipStreamWaitEvent(P->lanes[l].stream, P->ev0, 0));
    int rc = FFTUP_OK;
    for (uint32_t i = 0; i < n_frames && !rc; i++) {
        uint32_t s = (first_slot + i) % P->ring;
        P->cur = (int)(i % (uint32_t)nl);
        if (kernel_ms && i % stride == 0) {
            hipEvent_t* e = &ev[(size_t)(i / stride) * (nk + 1)];
            (void)hipEventRecord(e[0], P->lanes[P->cur].stream);
            for (int k = 0; k < nk && !rc; k++) {
                rc = launch_frame(P, s, s, k);
                (void)hipEventRecord(e[k + 1], P->lanes[P->cur].stream);
            }
        } else {
            rc = launch_frame(P, s, s, -1);
        }
        P->last_lane = P->cur;
        P->cur = 0;
    }
    for (int l = 1; l < nl; l++) {
        (void)hipEventRecord(P->lanes[l].done, P->lanes[l].stream);
        (void)hipStreamWaitEvent(P->stream, P->lanes[l].done, 0);
    }
    hipError_t e1 = hipEventRecord(P->ev1, P->stream);
    hipError_t e2 = hipEventSynchronize(P->ev1);
    if (!rc && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(FFTUP_E_HIP, std::string("sync: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    if (!rc) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, P->ev0, P->ev1);
        if (ms_total) *ms_total = ms;
        if (kernel_ms) {
            for (int k = 0; k < FFTUP_NUM_KERNELS; k++) kernel_ms[k] = 0;
            for (uint32_t i = 0; i < n_timed; i++)
                for (int k = 0; k < nk; k++) {
                    float d = 0;
                    (void)hipEventElapsedTime(&d, ev[(size_t)i * (nk + 1) + k], ev[(size_t)i * (nk + 1) + k + 1]);
                    kernel_ms[k] += d;
                }
            for (int k = 0; k < nk; k++) kernel_ms[k] /= n_timed;
        }
        P->executed = 1;
    }
    return rc;
}

int fftup_execute_ring(fftup_plan* P, uint32_t n_frames, uint32_t first_slot, double* ms_total)
{
    return execute_ring_impl(P, n_frames, first_slot, ms_total, nullptr, 1);
}

int fftup_execute_ring_timed(fftup_plan* P, uint32_t n_frames, uint32_t first_slot, uint32_t stride, double* ms_total,
                             double* ms_per_kernel)
{
    if (!ms_per_kernel) return fail(FFTUP_E_INVALID_ARG, "null argument");
    return execute_ring_impl(P, n_frames, first_slot, ms_total, ms_per_kernel, stride);
}

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
        if (fftup_jit::choose(base.W, base.H, base.D, base.half, stage_radices(P->planUW), d, "", false) &&
            fftup_jit::fused_value(d) != fftup_jit::fused_value(base)) {
            d.u8out = base.u8out;
            cands.push_back(d);
        }
    }
    for (const auto& cand : fftup_jit::fused_candidates(base.UW, base.D, 5)) {
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

extern "C" {

// output buffers of the lanes beyond the first for the pipelined fftup_execute (appended to P->out behind the ring's slots,
// which is all the other entry points can name): created on the first call that needs them
static int ensure_execute_outputs(fftup_plan* P, int nl)
{
    while (P->out.size() < (size_t)P->ring + (size_t)nl - 1) {
        void* o = nullptr;
        int rc = dev_alloc(P, &o, (size_t)3 * P->uW * P->uH * (P->u8out ? 1 : P->esz) + 8);
        if (rc) return rc;
        P->out.push_back(o);
    }
    return FFTUP_OK;
}

int fftup_execute(fftup_plan* P, uint32_t n_iter, double* ms_per_iter)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (n_iter == 0) return fail(FFTUP_E_INVALID_ARG, "n_iter must be > 0");
    HIP_TRY(hipSetDevice(P->device));
    // The reference records n_iter identical pipelines in ONE command buffer, submits it once and reports wall time / n_iter
    // (VR:1260-1278).  The iterations are identical -- same input slot 0, same result -- so nothing orders them: iteration i
    // runs on stream i mod nl with that stream's own spectra and, beyond stream 0, its own output buffer (stream 0 writes
    // output slot 0, which is what fftup_download_* read).  A kernel of one iteration then overlaps other kernels of its
    // neighbours exactly as consecutive frames of fftup_execute_ring do; every iteration computes the bits a lone one computes.
    // FFTUP_FLAG_SEQUENTIAL_EXECUTE (or FFTUP_STREAMS=1) keeps the strict single-queue form: one stream, nothing overlaps.
    const bool sequential = (P->cfg.flags & FFTUP_FLAG_SEQUENTIAL_EXECUTE) != 0;
    const int nl = sequential ? 1 : (int)std::min<uint32_t>((uint32_t)P->nlanes, n_iter);
    int rc = ensure_execute_outputs(P, nl);
    if (rc) return rc;
    P->last_lane = 0;
    HIP_TRY(hipEventRecord(P->ev0, P->stream));
    for (int l = 1; l < nl; l++) HIP_TRY(hipStreamWaitEvent(P->lanes[l].stream, P->ev0, 0));
    for (uint32_t i = 0; i < n_iter && !rc; i++) {
        P->cur = (int)(i % (uint32_t)nl);
        rc = launch_frame(P, 0, P->cur == 0 ? 0 : P->ring + (uint32_t)P->cur - 1, -1);
    }
    P->cur = 0;
    for (int l = 1; l < nl; l++) {
        (void)hipEventRecord(P->lanes[l].done, P->lanes[l].stream);
        (void)hipStreamWaitEvent(P->stream, P->lanes[l].done, 0);
    }
    const hipError_t e1 = hipEventRecord(P->ev1, P->stream), e2 = hipEventSynchronize(P->ev1);
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(FFTUP_E_HIP, std::string("sync: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, P->ev0, P->ev1));
    if (ms_per_iter) *ms_per_iter = (double)ms / n_iter;
    P->executed = 1;
    return FFTUP_OK;
}

int fftup_profile_kernels(fftup_plan* P, uint32_t n_iter, double* ms_per_kernel)
{
    if (!P || !ms_per_kernel) return fail(FFTUP_E_INVALID_ARG, "null argument");
    if (n_iter == 0) return fail(FFTUP_E_INVALID_ARG, "n_iter must be > 0");
    HIP_TRY(hipSetDevice(P->device));
    // per iteration: e[0] .. e[NK] around the NK launch slots, then an EMPTY pair e[NK+1], e[NK+2]: what an event pair
    // costs on this stream with nothing between (4-5 us); it is subtracted, so that the figures are kernel durations
    // as rocprofv3 --kernel-trace reports them
    constexpr int NE = FFTUP_NUM_KERNELS + 3;
    EventList ev;
    int rc = ev.create((size_t)n_iter * NE);
    if (rc) return rc;
    for (uint32_t i = 0; i < n_iter && !rc; i++) {
        hipEvent_t* e = &ev.ev[(size_t)i * NE];
        (void)hipEventRecord(e[0], P->stream);
        for (int k = 0; k < FFTUP_NUM_KERNELS && !rc; k++) {
            rc = launch_frame(P, i % P->ring, i % P->ring, k);
            (void)hipEventRecord(e[k + 1], P->stream);
        }
        (void)hipEventRecord(e[FFTUP_NUM_KERNELS + 1], P->stream);
        (void)hipEventRecord(e[FFTUP_NUM_KERNELS + 2], P->stream);
    }
    hipError_t se = hipStreamSynchronize(P->stream);
    if (!rc && se != hipSuccess) rc = fail(FFTUP_E_HIP, std::string("sync: ") + hipGetErrorString(se));
    if (!rc) {
        double empty = 0;
        for (int k = 0; k < FFTUP_NUM_KERNELS; k++) ms_per_kernel[k] = 0;
        for (uint32_t i = 0; i < n_iter; i++) {
            hipEvent_t* e = &ev.ev[(size_t)i * NE];
            float ms = 0;
            for (int k = 0; k < FFTUP_NUM_KERNELS; k++) {
                (void)hipEventElapsedTime(&ms, e[k], e[k + 1]);
                ms_per_kernel[k] += ms;
            }
            (void)hipEventElapsedTime(&ms, e[FFTUP_NUM_KERNELS + 1], e[FFTUP_NUM_KERNELS + 2]);
            empty += ms;
        }
        empty /= n_iter;
        for (int k = 0; k < FFTUP_NUM_KERNELS; k++) ms_per_kernel[k] = std::max(0.0, ms_per_kernel[k] / n_iter - empty);
        P->executed = 1;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
int fftup_download_planar(fftup_plan* P, uint32_t slot, void* planes)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!planes) return fail(FFTUP_E_INVALID_ARG, "null destination");
    if (!P->executed) return fail(FFTUP_E_NO_INPUT, "nothing executed yet");
    if (P->u8out) return fail(FFTUP_E_INVALID_ARG, "the plan stores 8-bit RGB only (FFTUP_FLAG_FUSE_U8_STORE): use fftup_download_rgb8");
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipMemcpyAsync(planes, P->out[slot], (size_t)3 * P->uW * P->uH * P->esz, hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return FFTUP_OK;
}

int fftup_download_presharpen(fftup_plan* P, void* planes)
{
    if (!P || !planes) return fail(FFTUP_E_INVALID_ARG, "null argument");
    if (!P->executed) return fail(FFTUP_E_NO_INPUT, "nothing executed yet");
    HIP_TRY(hipSetDevice(P->device));
    if (P->fused && !P->R_valid) {
        // the fused kernel never writes the pre-sharpen image; rebuild it from the spectrum of the last
        // frame (still in S2) with the stand-alone C2R kernel
        if (!P->lanes[P->last_lane].R) {
            int rc = dev_alloc(P, &P->lanes[P->last_lane].R, P->r_bytes);
            if (rc) return rc;
            if (P->last_lane == 0) P->R = P->lanes[0].R;
        }
        P->cur = P->last_lane;
        const int rc = launch_frame(P, 0, 0, 22);
        P->cur = 0;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(P->lanes[P->last_lane].stream));
    }
    if (P->cplx)        // non-R2C path: the pre-sharpen image is complex; this tap returns its real parts
        HIP_TRY(hipMemcpy2DAsync(planes, P->esz, P->lanes[P->last_lane].R, 2 * P->esz, P->esz, (size_t)3 * P->uW * P->uH,
                                 hipMemcpyDeviceToHost, P->stream));
    else
        HIP_TRY(hipMemcpyAsync(planes, P->lanes[P->last_lane].R, (size_t)3 * P->uW * P->uH * P->esz, hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return FFTUP_OK;
}

}  // extern "C"

// 64-bit wrapping sum of 32-bit words (fftup_output_checksum): per-thread partial sums, wave reduction, one atomic per wave
__global__ void __launch_bounds__(256) k_checksum(const uint32_t* __restrict__ w, size_t n, unsigned long long* sum)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += w[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

extern "C" {

int fftup_output_checksum(fftup_plan* P, uint32_t slot, uint64_t* sum)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!sum) return fail(FFTUP_E_INVALID_ARG, "null destination");
    if (!P->executed) return fail(FFTUP_E_NO_INPUT, "nothing executed yet");
    HIP_TRY(hipSetDevice(P->device));
    if (!P->d_sum) {
        rc = dev_alloc(P, (void**)&P->d_sum, sizeof(uint64_t));
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(P->d_sum, 0, sizeof(uint64_t), P->stream));
    const size_t nwords = (size_t)3 * P->uW * P->uH * (P->u8out ? 1 : P->esz) / 4;       // (uW, uH even: whole words for binary16 and bytes too)
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, P->stream, (const uint32_t*)P->out[slot], nwords, (unsigned long long*)P->d_sum);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sum, P->d_sum, sizeof(uint64_t), hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return FFTUP_OK;
}

int fftup_download_input_planar(fftup_plan* P, uint32_t slot, void* planes)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!planes) return fail(FFTUP_E_INVALID_ARG, "null destination");
    if (P->in_kind[slot] != 1) return fail(FFTUP_E_NO_INPUT, "slot holds no planar input");
    HIP_TRY(hipSetDevice(P->device));
    const size_t esz = P->esz;
    for (int c = 0; c < 3; c++)
        HIP_TRY(hipMemcpyAsync((char*)planes + (size_t)c * P->W * P->H * esz, (char*)P->in_planar[slot] + c * P->in_plane_stride * esz,
                               (size_t)P->W * P->H * esz, hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return FFTUP_OK;
}

int fftup_download_rgb8(fftup_plan* P, uint32_t slot, uint8_t* rgb, size_t row_stride_bytes)
{
    int rc = check_slot(P, slot);
    if (rc) return rc;
    if (!rgb || row_stride_bytes < (size_t)3 * P->uW) return fail(FFTUP_E_INVALID_ARG, "bad rgb pointer/stride");
    if (!P->executed) return fail(FFTUP_E_NO_INPUT, "nothing executed yet");
    HIP_TRY(hipSetDevice(P->device));
    const uint8_t* src = (const uint8_t*)P->out[slot];                      // (FFTUP_FLAG_FUSE_U8_STORE: the slot holds the bytes already)
    if (!P->u8out) {
        launch_pack(P, slot, P->out_u8, P->stream);
        HIP_TRY(hipGetLastError());
        src = P->out_u8;
    }
    HIP_TRY(hipMemcpy2DAsync(rgb, row_stride_bytes, src, (size_t)3 * P->uW, (size_t)3 * P->uW, P->uH, hipMemcpyDeviceToHost, P->stream));
    HIP_TRY(hipStreamSynchronize(P->stream));
    return FFTUP_OK;
}

// ------------------------------------------------------------------------------------------------
// host-streamed frames: H2D | convert + frame kernels + convert | D2H on three kinds of streams,
// `ring` frames in flight
void* fftup_host_alloc(size_t bytes)
{
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);   // page-locked for every device (-alldevices)
    if (e != hipSuccess) {
        fail(FFTUP_E_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void fftup_host_free(void* ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

static int queue_init(fftup_plan* P)
{
    if (!P->q.empty()) return FFTUP_OK;
    // built aside and published only when complete: a failure half way leaves the plan without a queue
    std::vector<fftup_plan::QSlot> q(P->ring);
    int rc = FFTUP_OK;
    for (uint32_t s = 0; s < P->ring && !rc; s++) {
        if (P->u8out) q[s].out_u8 = (uint8_t*)P->out[s];                            // the output slot holds the bytes
        else if (s == 0) q[s].out_u8 = P->out_u8;
        else rc = dev_alloc(P, (void**)&q[s].out_u8, (size_t)3 * P->uW * P->uH + 8); // owned by P->allocs either way
        if (!rc) {
            hipError_t e = hipEventCreateWithFlags(&q[s].done, hipEventDisableTiming);
            if (e != hipSuccess) { q[s].done = nullptr; rc = fail(FFTUP_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
        }
    }
    if (rc) {
        for (auto& qs : q)
            if (qs.done) (void)hipEventDestroy(qs.done);
        return rc;
    }
    P->q.swap(q);
    return FFTUP_OK;
}

// ---- device-side PNG encoding (csrc/kernels_png.hpp) ----
static size_t png_stream_bound(size_t raw_bytes, int nblocks)
{
    // a Huffman code for 257 symbols spends at most ~8.1 bits per symbol on average; nine and an eighth are allowed for, 236
    // bytes of header per block, and the words the last atomicOr may touch
    return (raw_bytes + raw_bytes / 8 + raw_bytes / 64 + ((size_t)nblocks + 2) * 512 + 64 + 15) / 16 * 16;
}
static void png_geometry(fftup_plan* P)
{
    if (P->png_rpb) return;
    const size_t L = (size_t)3 * P->uW + 1;
    P->png_rpb = (int)std::max<size_t>(1, (192 * 1024) / L);           // ~192 KB of residuals per deflate block, whole rows
    P->png_nblocks = (int)((P->uH + (uint32_t)P->png_rpb - 1) / (uint32_t)P->png_rpb);
    P->png_stream_bytes = png_stream_bound(L * P->uH, P->png_nblocks);
}
static int png_slot_init(fftup_plan* P, fftup_plan::QSlot& Q)
{
    fftup_plan::PngSlot& G = Q.png;
    if (G.p.stream) return FFTUP_OK;
    png_geometry(P);
    const size_t L = (size_t)3 * P->uW + 1, nb = (size_t)P->png_nblocks, uH = P->uH;
    PngParams p{};
    int rc = dev_alloc(P, (void**)&p.raw, L * uH + 8);            // (+ 8: k_png_pack reads whole words)
    if (!rc) rc = dev_alloc(P, (void**)&p.rowhist, uH * 257 * sizeof(uint32_t));
    if (!rc) rc = dev_alloc(P, (void**)&p.rowsum, uH * 2 * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(P, (void**)&p.tab, nb * 257 * sizeof(uint32_t));
    if (!rc) rc = dev_alloc(P, (void**)&p.hdr, nb * 64 * sizeof(uint32_t));
    if (!rc) rc = dev_alloc(P, (void**)&p.hdr_bits, nb * sizeof(uint32_t));
    if (!rc) rc = dev_alloc(P, (void**)&p.block_bits, nb * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(P, (void**)&p.block_start, nb * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(P, (void**)&p.row_off, uH * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(P, (void**)&p.meta, 2 * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(P, (void**)&p.crc_parts, (P->png_stream_bytes / 4096 + 1) * sizeof(uint32_t));
    if (!rc) rc = dev_alloc(P, (void**)&p.stream, P->png_stream_bytes);
    if (rc) return rc;                                            // (what was allocated stays owned by the plan)
    p.uW = (int)P->uW; p.uH = (int)P->uH; p.rows_per_block = P->png_rpb; p.nblocks = P->png_nblocks;
    if (!G.meta_host) HIP_TRY(hipHostMalloc((void**)&G.meta_host, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    if (!G.parts_host) HIP_TRY(hipHostMalloc((void**)&G.parts_host, (P->png_stream_bytes / 4096 + 1) * sizeof(uint32_t), hipHostMallocDefault));
    if (!G.copied) HIP_TRY(hipEventCreateWithFlags(&G.copied, hipEventDisableTiming));
    if (!P->png_copy) HIP_TRY(hipStreamCreateWithFlags(&P->png_copy, hipStreamNonBlocking));
    G.p = p;
    return FFTUP_OK;
}

// shared body of fftup_submit_rgb8 / fftup_submit_png: one whole frame on one of the plan's streams
static int submit_frame(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* rgb_out, size_t out_stride, bool png,
                        uint64_t* ticket)
{
    // (png: rgb_out / out_stride are the optional destination of the finished file and its capacity)
    uint8_t* png_dest = png ? rgb_out : nullptr;
    const size_t png_cap = png ? out_stride : 0;
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (!rgb_in || in_stride < (size_t)3 * P->W) return fail(FFTUP_E_INVALID_ARG, "bad input pointer/stride");
    if (!png && (!rgb_out || out_stride < (size_t)3 * P->uW)) return fail(FFTUP_E_INVALID_ARG, "bad output pointer/stride");
    if (png && P->dbl) return fail(FFTUP_E_UNSUPPORTED_PRECISION, "device-side PNG encoding: -p 0 and -p 2 plans");
    if (png_dest) {
        png_geometry(P);
        hipPointerAttribute_t at{};
        if (((uintptr_t)png_dest & 15) || png_cap < P->png_stream_bytes + 57 || hipPointerGetAttributes(&at, png_dest) != hipSuccess ||
            at.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            return fail(FFTUP_E_INVALID_ARG, "png_out of fftup_submit_png: fftup_png_bound() bytes from fftup_host_alloc (the GPU writes into it)");
        }
    }
    HIP_TRY(hipSetDevice(P->device));
    // one submission at a time: slot choice, the lane's launches and the ticket are one critical section (a few tens of
    // microseconds; the wait below is for the frame that used this slot `ring` submissions ago)
    std::unique_lock<std::mutex> lock(P->q_mu);
    int rc = queue_init(P);
    if (rc) return rc;
    uint64_t t;
    for (;;) {                                                        // (the lock is released while waiting: the ticket is re-read)
        t = P->q_next.load(std::memory_order_relaxed);
        if (P->q[t % P->ring].png.state == 0) break;
        P->q_cv.wait(lock);                                           // a PNG stream of this slot is still to be collected
    }
    const uint32_t s = (uint32_t)(t % P->ring);
    fftup_plan::QSlot& Q = P->q[s];
    if (t >= P->ring) HIP_TRY(hipEventSynchronize(Q.done));          // the slot's previous frame has left the device
    if (png && (rc = png_slot_init(P, Q)) != FFTUP_OK) return rc;
    // The whole frame -- H2D, conversion, kernels, conversion, D2H -- goes to ONE stream (lane t % nlanes), so no
    // cross-stream dependency exists and nothing can stall behind a neighbour's wait when streams share a hardware
    // queue; the copies of one lane overlap the kernels and the opposite-direction copies of the other lanes.
    // (two lanes: with the copies in the streams a third one only adds contention, 0.56-0.75 ms/frame instead of 0.51)
    // (a PNG frame's chain is long -- nine more launches, most of them a handful of workgroups: all lanes take turns)
    const int lane = (int)(t % (uint64_t)(png ? P->nlanes : std::min(P->nlanes, 2)));
    hipStream_t cs = P->lanes[lane].stream;
    const size_t in_row = (size_t)3 * P->W, out_row = (size_t)3 * P->uW;
    if (in_stride == in_row) HIP_TRY(hipMemcpyAsync(P->in_u8[s], rgb_in, in_row * P->H, hipMemcpyHostToDevice, cs));
    else HIP_TRY(hipMemcpy2DAsync(P->in_u8[s], in_row, rgb_in, in_stride, in_row, P->H, hipMemcpyHostToDevice, cs));
    if (fuse_u8(P)) {
        P->in_kind[s] = 2;
    } else {
        launch_unpack(P, s, cs);
        P->in_kind[s] = 1;
    }
    P->cur = lane;
    rc = launch_frame(P, s, s, -1);
    P->last_lane = lane;
    P->cur = 0;
    if (rc) return rc;
    if (!P->u8out) launch_pack(P, s, Q.out_u8, cs);
    HIP_TRY(hipGetLastError());
    if (png) {
        // the 8-bit image stays on the device: filter rows, code them, pack the bits; only the stream's size comes back now,
        // the stream itself when fftup_wait_png knows how many bytes to ask for
        PngParams pp = Q.png.p;
        pp.rgb = Q.out_u8;
        HIP_TRY(hipMemsetAsync(pp.stream, 0, P->png_stream_bytes, cs));
        const size_t rb = (size_t)3 * P->uW;
        pp.row_in_lds = rb <= 48 * 1024 ? 1 : 0;                   // one row of residuals in LDS
        const size_t lds_pack = pp.row_in_lds ? (rb + 12) / 4 * 4 : 0;
        hipLaunchKernelGGL(k_png_filter, dim3(P->uH), dim3(256), 0, cs, pp);
        hipLaunchKernelGGL(k_png_codes, dim3(P->png_nblocks), dim3(256), 0, cs, pp);
        hipLaunchKernelGGL(k_png_layout, dim3(1), dim3(256), 0, cs, pp);
        hipLaunchKernelGGL(k_png_pack, dim3(P->uH), dim3(256), lds_pack, cs, pp);
        const size_t max_pieces = P->png_stream_bytes / 4096;
        if (max_pieces) hipLaunchKernelGGL(k_png_crc, dim3((unsigned)((max_pieces + 255) / 256)), dim3(256), 0, cs, pp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(Q.png.meta_host, pp.meta, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
        if (max_pieces) HIP_TRY(hipMemcpyAsync(Q.png.parts_host, pp.crc_parts, max_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
        if (png_dest) {                                            // the device knows the size: it delivers the stream itself
            void* dev_view = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&dev_view, png_dest, 0));
            hipLaunchKernelGGL(k_png_deliver, dim3(512), dim3(256), 0, cs, pp, (uint32_t*)dev_view);
            HIP_TRY(hipGetLastError());
        }
        Q.png.state = 1;
        Q.png.ticket = t;
        Q.png.dest = png_dest;
        Q.png.dest_cap = png_cap;
    } else if (out_stride == out_row) HIP_TRY(hipMemcpyAsync(rgb_out, Q.out_u8, out_row * P->uH, hipMemcpyDeviceToHost, cs));
    else HIP_TRY(hipMemcpy2DAsync(rgb_out, out_stride, Q.out_u8, out_row, out_row, P->uH, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipEventRecord(Q.done, cs));
    P->q_next.store(t + 1, std::memory_order_release);
    P->executed = 1;
    if (ticket) *ticket = t;
    return FFTUP_OK;
}

int fftup_submit_rgb8(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* rgb_out, size_t out_stride,
                      uint64_t* ticket)
{
    return submit_frame(P, rgb_in, in_stride, rgb_out, out_stride, false, ticket);
}

int fftup_submit_png(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* png_out, size_t capacity, uint64_t* ticket)
{
    return submit_frame(P, rgb_in, in_stride, png_out, png_out ? capacity : 0, true, ticket);
}

size_t fftup_png_bound(fftup_plan* P)
{
    if (!P) return 0;
    png_geometry(P);
    return P->png_stream_bytes + 57;                      // signature 8, IHDR 25, IDAT framing 12, IEND 12
}

// CRC-32 of the PNG chunks (ISO 3309, the zlib polynomial), eight bytes per step
static uint32_t crc32_png(uint32_t crc, const uint8_t* p, size_t n)
{
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) T[k][i] = (T[k - 1][i] >> 8) ^ T[0][T[k - 1][i] & 255];
    });
    crc = ~crc;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = T[7][a & 255] ^ T[6][(a >> 8) & 255] ^ T[5][(a >> 16) & 255] ^ T[4][a >> 24] ^
              T[3][b & 255] ^ T[2][(b >> 8) & 255] ^ T[1][(b >> 16) & 255] ^ T[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 255] ^ (crc >> 8);
    return ~crc;
}
// crc(A || B) from crc(A) and crc(B) for |B| = 4096: the operator "append 4096 zero bytes" is linear over GF(2) -- the matrix of
// one zero bit (the polynomial and a shift), squared fifteen times
static uint32_t crc32_shift_4096(uint32_t crc)
{
    static uint32_t M[32];
    static std::once_flag once;
    std::call_once(once, [] {
        uint32_t a[32], b[32];
        a[0] = 0xEDB88320u;
        for (int n = 1; n < 32; n++) a[n] = 1u << (n - 1);
        auto times = [](const uint32_t* m, uint32_t v) { uint32_t s = 0; for (int i = 0; v; v >>= 1, i++) if (v & 1) s ^= m[i]; return s; };
        for (int k = 0; k < 15; k++) {                          // 2^15 bits = 4096 bytes
            for (int n = 0; n < 32; n++) b[n] = times(a, a[n]);
            memcpy(a, b, sizeof a);
        }
        memcpy(M, a, sizeof M);
    });
    uint32_t s = 0;
    for (int i = 0; crc; crc >>= 1, i++)
        if (crc & 1) s ^= M[i];
    return s;
}
static void be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

int fftup_wait_png(fftup_plan* P, uint64_t ticket, uint8_t* png_out, size_t capacity, size_t* png_bytes)
{
    if (!P || !png_out || !png_bytes) return fail(FFTUP_E_INVALID_ARG, "null argument");
    const uint64_t next = P->q_next.load(std::memory_order_acquire);
    if (ticket >= next) return fail(FFTUP_E_INVALID_ARG, "ticket was never issued");
    fftup_plan::QSlot& Q = P->q[ticket % P->ring];
    if (ticket + P->ring < next || Q.png.state != 1 || Q.png.ticket != ticket)
        return fail(FFTUP_E_INVALID_ARG, "no PNG stream is waiting under this ticket");
    HIP_TRY(hipSetDevice(P->device));
    auto release = [&] {
        { std::lock_guard<std::mutex> lock(P->q_mu); Q.png.state = 0; }
        P->q_cv.notify_all();
    };
    hipError_t e = hipEventSynchronize(Q.done);
    const size_t zbytes = e == hipSuccess ? (size_t)Q.png.meta_host[0] : 0;
    if (e == hipSuccess && (zbytes < 6 || zbytes > P->png_stream_bytes || zbytes + 57 > capacity)) {
        release();
        return fail(FFTUP_E_INVALID_ARG, "PNG buffer too small: " + std::to_string(zbytes + 57) + " bytes needed (fftup_png_bound)");
    }
    if (e == hipSuccess && Q.png.dest != png_out) {        // (delivered by the device already when the buffer was named at submission)
        if (Q.png.dest) { release(); return fail(FFTUP_E_INVALID_ARG, "fftup_wait_png: the buffer named by fftup_submit_png holds this file"); }
        e = hipMemcpyAsync(png_out + 41, Q.png.p.stream, zbytes, hipMemcpyDeviceToHost, P->png_copy);
        if (e == hipSuccess) e = hipEventRecord(Q.png.copied, P->png_copy);
        if (e == hipSuccess) e = hipEventSynchronize(Q.png.copied);
    }
    uint32_t crc = 0;
    if (e == hipSuccess) {                                 // "IDAT", then the stream: whole 4 KB pieces from the device, the tail here
        crc = crc32_png(0, (const uint8_t*)"IDAT", 4);
        const size_t pieces = zbytes / 4096;
        for (size_t k = 0; k < pieces; k++) crc = crc32_shift_4096(crc) ^ Q.png.parts_host[k];
        crc = crc32_png(crc, png_out + 41 + pieces * 4096, zbytes - pieces * 4096);
    }
    release();                                             // (the slot's host mailboxes are not read below)
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("fftup_wait_png: ") + hipGetErrorString(e));
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    memcpy(png_out, sig, 8);
    uint8_t* q = png_out + 8;                              // IHDR: 8-bit RGB, no interlace
    be32(q, 13); memcpy(q + 4, "IHDR", 4); be32(q + 8, P->uW); be32(q + 12, P->uH);
    q[16] = 8; q[17] = 2; q[18] = 0; q[19] = 0; q[20] = 0;
    be32(q + 21, crc32_png(0, q + 4, 17));
    q = png_out + 33;
    be32(q, (uint32_t)zbytes); memcpy(q + 4, "IDAT", 4);
    be32(png_out + 41 + zbytes, crc);
    q = png_out + 41 + zbytes + 4;
    be32(q, 0); memcpy(q + 4, "IEND", 4); be32(q + 8, crc32_png(0, q + 4, 4));
    *png_bytes = zbytes + 57;
    return FFTUP_OK;
}

int fftup_wait(fftup_plan* P, uint64_t ticket)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    const uint64_t next = P->q_next.load(std::memory_order_acquire);
    if (ticket >= next) return fail(FFTUP_E_INVALID_ARG, "ticket was never issued");
    if (ticket + P->ring < next) return FFTUP_OK;             // its slot has been reused: submit already waited for it
    {
        const fftup_plan::QSlot& Q = P->q[ticket % P->ring];
        if (Q.png.state == 1 && Q.png.ticket == ticket) return fail(FFTUP_E_INVALID_ARG, "a ticket of fftup_submit_png is collected by fftup_wait_png");
    }
    // (a submission of another thread may re-record this slot's event right now: the wait then covers the later frame too)
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipEventSynchronize(P->q[ticket % P->ring].done));
    return FFTUP_OK;
}

int fftup_drain(fftup_plan* P)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    {
        std::lock_guard<std::mutex> lock(P->q_mu);
        if (P->q.empty()) return FFTUP_OK;
    }
    HIP_TRY(hipSetDevice(P->device));
    for (int l = 0; l < P->nlanes; l++) HIP_TRY(hipStreamSynchronize(P->lanes[l].stream));
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
    default: return "unknown error";
    }
}

const char* fftup_last_error(void) { return g_last_error.c_str(); }
const char* fftup_version(void) { return "fftup 0.4.0 (gfx950, ABI 2)"; }

}  // extern "C"
