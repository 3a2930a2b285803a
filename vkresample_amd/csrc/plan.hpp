// plan.hpp -- what the translation units of libfftup.so share: the plan object behind the opaque fftup_plan of include/fftup.h,
// error reporting, and the internal entry points between the units
//   fftup_plan.hip     plan construction (launchResample's plan semantics, VkResample.cpp:1409-1617), info, the plan-time tuner
//   fftup_launch.hip   the frame's kernel launches -- the only unit that instantiates the frame kernels
//   fftup_execute.hip  upload / execute / download (performVulkanUpscale, VkResample.cpp:1249-1279, and the transfers)
//   fftup_queue.hip    host-streamed frames: fftup_submit_rgb8 / fftup_wait / fftup_drain
//   fftup_png.hip      the device-side PNG encoder's host side
//   jit.cpp            the plan-time compiler
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fftup.h"
#include "fft_engine.hpp"
#include "jit.hpp"
#include "png_params.hpp"

using fftup::StagePlan;
using fftup::PngParams;

static constexpr int TUNED_TK = 4;     // column tile width of the size-specialised kernels

// ---- errors: code + thread-local detail (fftup_last_error)
int fail(int code, const std::string& msg);
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
    bool poly = false;                         // size-generic u = 2 plan: polyphase column kernel (k_col_poly), the C2R kernel reads the even rows from S1
    bool inplaceC = false;                     // -p 1 R2C plans: the column kernel's two transforms in one LDS buffer (k_col<TK, double2, true>)
    bool inplaceF = false, inplaceI = false;   // ... whose forward / inverse rows are too long for two LDS buffers: fft_lds_inplace
    // ... and rows too long for ONE buffer: four steps through HBM (k_row4_a / k_row4_b), row length = n1 * n2
    struct Four { bool on = false; int n1 = 0, n2 = 0, tka = 1, tkb = 1;    // N = n1 * n2; sequences per workgroup of pass A / pass B
                  StagePlan p1{}, p2{}; float2 *tw1 = nullptr, *tw2 = nullptr; size_t ldsA = 0, ldsB = 0; int thrA = 64, thrB = 64; };
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
    // its fftup_wait_png, 2 = being collected -- submissions skip such a slot, and wait on q_cv when every slot is held)
    struct PngSlot { PngParams p{}; bool ready = false; unsigned long long* meta_host = nullptr; uint32_t* parts_host = nullptr; hipEvent_t copied = nullptr;
                     int state = 0; uint64_t ticket = 0; std::thread::id owner{}; uint8_t* dest = nullptr; size_t dest_cap = 0; };
    struct QSlot { uint8_t* out_u8 = nullptr; hipEvent_t done = nullptr; PngSlot png;
                   bool used = false; uint64_t ticket = 0; };        // the latest submission that went through this slot
    std::vector<QSlot> q;
    std::atomic<uint64_t> q_next{0};   // next ticket (tickets count the plan's submissions); written under q_mu
    uint32_t q_cursor = 0;             // where the search for a free slot starts (slots are taken in turn, skipping uncollected PNG streams)
    std::mutex q_mu;                   // fftup_submit_rgb8 may be called by several host threads (codec workers sharing a plan)
    std::condition_variable q_cv;
    hipStream_t png_copy = nullptr;    // the sized D2H copies of fftup_wait_png
    int png_rpb = 0, png_nblocks = 0;  // rows per deflate block, blocks per frame
    size_t png_stream_bytes = 0;       // capacity of a slot's stream buffer
    uint32_t* png_crc_shift = nullptr; // device table of k_png_crc (created with the first PNG slot)
    bool png_foreign_collector = false; // some thread has collected (fftup_wait_png) a ticket another thread submitted; under q_mu
    float2 *twW = nullptr, *twH = nullptr, *twUW = nullptr, *twUH = nullptr;
    uint64_t device_bytes = 0;
    size_t r_bytes = 0;               // bytes of one pre-sharpen image
    uint64_t* d_sum = nullptr;        // fftup_output_checksum accumulator (created on first use)
    size_t in_plane_stride = 0;
    int executed = 0;

    std::vector<void*> allocs;
};

// ---- fftup_plan.hip
int dev_alloc(fftup_plan* P, void** ptr, size_t bytes);           // hipMalloc owned by the plan
int lane_count();                                                 // FFTUP_STREAMS
void set_strip_length(fftup_plan* P);
// the row kernel reads uint8 RGB directly (fp32 / fp16 plans only)
inline bool fuse_u8(const fftup_plan* P) { return (P->cfg.flags & FFTUP_FLAG_FUSE_U8_LOAD) && !P->dbl; }
inline int check_slot(fftup_plan* P, uint32_t slot)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (slot >= P->ring) return fail(FFTUP_E_INVALID_ARG, "slot out of range");
    return FFTUP_OK;
}

// ---- fftup_launch.hip: everything that names a kernel
// facts about the kernels the planner needs (defined next to the kernels)
int kernels_generic_max_threads(bool dbl);                        // threads per workgroup of the size-generic kernels
int kernels_aot_mixed_plan(uint32_t W, uint32_t H);               // 1: 1920x1080, 2: 1280x720 (ahead-of-time mixed-radix plans), 0: none
size_t kernels_tuned_col_lds(uint32_t H);                         // LDS bytes of the power-of-two column kernel
int kernels_set_attributes(fftup_plan* P);                        // dynamic LDS sizes above 64 KB, for the kernels THIS plan launches
// one frame on lane P->cur: `which` < 0 launches all of its kernels, 0..3 only that one, 22 = the pre-sharpen tap of a fused plan
int launch_frame(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which);
void launch_unpack(fftup_plan* P, uint32_t slot, hipStream_t st);                  // the host loop of VR:1636-1685 as a kernel
void launch_pack(fftup_plan* P, uint32_t slot, uint8_t* dst, hipStream_t st);      // ... and of VR:1708-1748
void launch_checksum(fftup_plan* P, uint32_t slot, hipStream_t st);                // 64-bit sum of the slot's words -> P->d_sum

// ---- fftup_execute.hip
int execute_ring_impl(fftup_plan* P, uint32_t n_frames, uint32_t first_slot, double* ms_total, double* kernel_ms, uint32_t stride);

// ---- fftup_queue.hip
int submit_frame(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* rgb_out, size_t out_stride, bool png, uint64_t* ticket);

// ---- fftup_png.hip
void png_geometry(fftup_plan* P);                                 // rows per deflate block, stream capacity (fixed per plan)
int png_slot_init(fftup_plan* P, fftup_plan::QSlot& Q);           // the encoder's buffers of a ring slot, on first use
int png_enqueue(fftup_plan* P, fftup_plan::QSlot& Q, hipStream_t cs, uint8_t* png_dest);    // the encoder's launches behind a frame
