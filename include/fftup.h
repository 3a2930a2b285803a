/*
 * fftup.h -- C ABI of the MI355X-native FFT upscaler (drop-in for VkResample's upscale path).
 *
 * The reference (DTolm/VkResample, one translation unit, no plugin layer) exposes the hot path as
 * the call sequence inside launchResample() (VkResample.cpp, "VR"); vkFFT/vkFFT.h is "VF".
 * Each entry point below replaces the reference calls cited next to it.  Plain pointers and
 * sizes only; no C++/torch types.  0 = success, non-zero = FFTUP_E_* (text: fftup_strerror).
 * The library never exits the process and never falls back to a CPU path: without a usable HIP
 * device every compute entry point fails with FFTUP_E_NO_DEVICE / FFTUP_E_HIP.
 *
 * Threading (mirrors VR:1282-1320, 1959-1969): one plan per host thread; a plan is used by one
 * thread at a time; distinct plans are independent (own stream, own device buffers).
 */
#ifndef FFTUP_H
#define FFTUP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define FFTUP_API __attribute__((visibility("default")))
#else
#define FFTUP_API
#endif

/* ---- error codes (reference: VkResult ints, 0 = VK_SUCCESS; VR:1286-1320, 1364-1367) ---- */
enum {
    FFTUP_OK = 0,
    FFTUP_E_INVALID_ARG = 1,    /* null pointer, bad slot, odd size, channels != 3 (VR:1368)            */
    FFTUP_E_UNSUPPORTED_SIZE = 2, /* a dimension is not 2,3,5,7-smooth: VF:4719-4726
                                     (VK_ERROR_FORMAT_NOT_SUPPORTED)                                   */
    FFTUP_E_UNSUPPORTED_PRECISION = 3, /* -p must be 0, 1 or 2                                         */
    FFTUP_E_NO_DEVICE = 4,      /* no HIP device / bad device id (VR:1292-1296)                         */
    FFTUP_E_HIP = 5,            /* a HIP runtime call failed (message in fftup_last_error)              */
    FFTUP_E_OUT_OF_MEMORY = 6,  /* device allocation failed (allocateFFTBuffer VR:361-384)              */
    FFTUP_E_NO_INPUT = 7,       /* execute/download before any upload                                   */
    FFTUP_E_INCOMPLETE = 8,     /* "Image not found" class of errors in the host mirror (VR:1366)       */
    FFTUP_E_WOULD_BLOCK = 9,    /* fftup_submit_png / fftup_submit_rgb8: every ring slot holds an uncollected PNG ticket of the
                                   CALLING thread and no other thread has collected anything on this plan (after a bounded
                                   wait, FFTUP_SELF_WAIT_MS, default 2000) -- waiting would never end            */
    FFTUP_E_OVERFLOW = 10       /* fftup_wait_png: the frame's deflate stream did not fit the encoder's buffer; nothing was
                                   written beyond it, the frame is not encoded                                   */
};

/* ---- flags ---- */
enum {
    FFTUP_FLAG_U8_WRAP = 1u,       /* u8 store wraps like the x86 C cast of VR:1715 instead of saturating */
    FFTUP_FLAG_FUSE_U8_LOAD = 2u,  /* row-FFT kernel reads the uint8 RGB image directly (README.md:31
                                      roadmap item); otherwise upload converts to the reference's planar
                                      float/half inputBuffer first (VR:1636-1688 semantics)               */
    FFTUP_FLAG_GENERIC_KERNELS = 4u, /* force the size-generic kernels even where a tuned plan exists    */
    FFTUP_FLAG_UNFUSED_SHARPEN = 8u, /* keep C2R and sharpen as two launches with the pre-sharpen image in
                                        HBM, like the reference (tempBuffer); default fuses them          */
    FFTUP_FLAG_TUNE_PLAN = 16u,      /* plans specialised at plan time (below): compile and time the alternatives for the fused
                                        kernel's factorization on the device, keep the fastest, remember it in
                                        <cache dir>/wisdom.txt (a few seconds, once per row length and device)            */
    FFTUP_FLAG_FUSE_U8_STORE = 32u,  /* 8-bit pipelines (the reference's own: PNG in, PNG out): the fused C2R+sharpen kernel stores
                                        the interleaved 8-bit RGB image itself (the conversion of VR:1708-1748 in registers); the
                                        float / half planes are never written, fftup_download_rgb8 / fftup_submit_rgb8 need no
                                        conversion launch, fftup_download_planar fails with FFTUP_E_INVALID_ARG.  Plans without a
                                        fused kernel (size-generic, -p 1, non-R2C, FFTUP_FLAG_UNFUSED_SHARPEN) ignore the flag:
                                        fftup_info.u8_store says which it is                                               */
    FFTUP_FLAG_SEQUENTIAL_EXECUTE = 64u, /* accepted and ignored: ordered iterations are what fftup_execute does (0.6 and earlier
                                        needed this flag for it)                                                               */
    FFTUP_FLAG_OVERLAP_ITERATIONS = 128u /* EXTENSION, not the reference's semantics: the n_iter identical iterations of one
                                        fftup_execute call alternate on the plan's streams and overlap like the distinct frames
                                        of fftup_execute_ring (a throughput figure; same bits).  A plan without a ring is then
                                        laid out for overlapping frames (one strip of the last kernel per compute unit)         */
};

/* Replaces VkResampleConfiguration (VR:45-59) + the part of VkFFTConfiguration (VF:22-94) that
 * launchResample() derives from it (VR:1409-1503). */
typedef struct fftup_config {
    uint32_t width, height;   /* input image size; both even                                         */
    uint32_t channels;        /* must be 3 (stbi_load(...,3) VR:1362, channels = 3 VR:1368)           */
    float    upscale;         /* -u; output = (uint32_t)(upscale*size) (VR:1417-1418)                */
    uint32_t precision;       /* -p: 0 single, 1 double (VR:1422), 2 half-memory/fp32-math (VR:1420-1421) */
    float    sharpen;         /* -s sharpening constant (VR:1616)                                    */
    int32_t  device;          /* -d HIP device ordinal                                               */
    uint32_t flags;           /* FFTUP_FLAG_*                                                        */
    uint32_t ring;            /* resident input/output frame slots (0 or 1 = one, like the reference; <= 1024) */
} fftup_config;

/* Environment read by fftup_plan_create (operational knobs, not part of the reference's surface):
 *   FFTUP_STREAMS=n     HIP streams consecutive frames of fftup_execute_ring / fftup_submit_rgb8 (and the iterations of
 *                       fftup_execute under FFTUP_FLAG_OVERLAP_ITERATIONS) alternate on (default 3, 1..4)
 *   FFTUP_JIT=0|1       run-time specialised plans (default 1); FFTUP_JIT_VERBOSE=1 prints why one fell back
 *   FFTUP_CACHE_DIR     code-object cache and wisdom file of those plans (default ~/.cache/fftup);
 *   FFTUP_KERNEL_DIR    kernel headers, when not the ones embedded in the library; FFTUP_HIPRTC_LIB: the run-time compiler's
 *                       shared object (default: libhiprtc.so of the ROCm install) */

typedef struct fftup_plan fftup_plan;   /* opaque; replaces VkGPU + 2x VkFFTApplication +
                                           2x VkShiftApplication + the three device buffers      */

enum { FFTUP_NUM_KERNELS = 4 };          /* row R2C, column fwd+pad+inv, row C2R, sharpen        */

/* ABI: fields are only ever APPENDED to this struct; `abi_version` (== FFTUP_ABI_VERSION of the library that filled it) says
 * how far it is valid: 1 = up to kernel_names, 2 = + kernel_min_bytes, tuned may be 2, abi_version itself.
 * (0.2.x builds had kernel_min_bytes in front of device_bytes: callers built against those must be rebuilt.) */
enum { FFTUP_ABI_VERSION = 2 };
typedef struct fftup_info {
    uint32_t out_width, out_height;      /* uW, uH                                                */
    uint32_t num_kernels;                /* launches per frame                                    */
    uint32_t tuned;                      /* 0: size-generic kernels, non-zero: size-specialised (1: ahead-of-time, 2: at plan time through hipRTC) */
    double   alg_bytes_per_frame;        /* B_alg of SURVEY 8(d) for this plan's I/O types        */
    double   kernel_alg_bytes[FFTUP_NUM_KERNELS]; /* algorithmic bytes of each kernel: a fused C2R+sharpen
                                                     launch keeps S2 + 2R + out although R never reaches HBM          */
    uint64_t device_bytes;               /* device memory owned by the plan ("VRAM per thread")   */
    char     device_name[256];
    char     kernel_names[FFTUP_NUM_KERNELS][64];
    /* ---- appended in ABI version 2 ---- */
    double   kernel_min_bytes[FFTUP_NUM_KERNELS]; /* bytes each kernel has to move through HBM as implemented
                                                     (e.g. fused: spectrum rows incl. strip halos + out)               */
    uint32_t abi_version;                /* FFTUP_ABI_VERSION of the library                      */
    uint32_t u8_store;                   /* 1: the plan's output slots hold 8-bit RGB (FFTUP_FLAG_FUSE_U8_STORE in effect)     */
} fftup_info;

/* devices_list() VR:239-268 */
FFTUP_API int fftup_device_count(void);
FFTUP_API int fftup_device_name(int device, char* buf, size_t buflen);
/* PCI bus id ("0000:c1:00.0") of a device: job accounting of multi-GPU runs (one process / thread per GPU: the ids of a
 * job's ranks must all differ) */
FFTUP_API int fftup_device_pci_bus_id(int device, char* buf, size_t buflen);

/* initializeVulkanFFT x2 + createShiftApp + createSharpenApp + 3x allocateFFTBuffer
 * (VR:1437-1448, 1506-1509, 1562, 1617).  Every even 2,3,5,7-smooth width and height up to 65536 whose upscaled sizes are
 * even and smooth is a valid plan, as in the reference: upscaled widths beyond 8192 (4096 for -p 1) take the reference's
 * non-R2C path (VR:1424); rows and columns too long for the compute unit's local memory run as two-launch "four-step"
 * transforms through device memory (the reference's multi-upload plans, VF:4773-4992).  fftup_plan_describe says which. */
FFTUP_API int fftup_plan_create(fftup_plan** out, const fftup_config* cfg);
/* deleteVulkanFFT x2, deleteShiftApp x2, buffer frees (VR:1759-1771) */
FFTUP_API void fftup_plan_destroy(fftup_plan* plan);
FFTUP_API int fftup_plan_info(const fftup_plan* plan, fftup_info* info);
/* one line saying which kernels the plan runs (for a plan specialised at plan time: the chosen factorizations) */
FFTUP_API int fftup_plan_describe(const fftup_plan* plan, char* buf, size_t buflen);

/* Run-time specialised plans (csrc/jit.hpp; the counterpart of VkFFT generating and compiling its shaders for the
 * requested size at plan time, VF:4707-5189 + the GLSL generator, glslang in VkResample's link line).  A plan with an
 * integer, half-, quarter- or eighth-integer upscale factor or a ratio over 3, 5 or 7 whose sizes come out exact (-u 1.125, 1.2, 1.25, 4/3, 1.4, 1.5, 1.6, 5/3, 1.75, 1.875, 2, 2.25, 2.5, 8/3, 3, 3.5, 4, 5, 6, 7, 8; -p 0 / -p 2; H <= 8192,
 * u*W <= 8192, 4 | u*W, u*H even)
 * whose size has no ahead-of-time
 * kernels gets its row, column and fused C2R+sharpen kernels
 * instantiated for exactly that size through hipRTC inside fftup_plan_create; fftup_info.tuned is then 2.  FFTUP_JIT=0
 * or FFTUP_FLAG_GENERIC_KERNELS keeps such plans on the size-generic kernels (as does a missing libhiprtc.so).
 * fftup_jit_check does the same WITHOUT a device: picks the factorizations, compiles for `arch` (NULL: gfx950; "": does not
 * compile) and writes a one-line description.  FFTUP_E_UNSUPPORTED_SIZE: no specialised factorization (the generic kernels run that
 * size); FFTUP_E_HIP: hipRTC unavailable or compilation failed (fftup_last_error has the log). */
FFTUP_API int fftup_jit_check(uint32_t width, uint32_t height, float upscale, uint32_t precision, const char* arch,
                              char* desc, size_t desclen);

/* host pack loop + transferDataFromCPU (VR:1636-1688).  rgb: interleaved 8-bit RGB, H rows of
 * row_stride_bytes (>= 3*W).  Blocking, like the reference. */
FFTUP_API int fftup_upload_rgb8(fftup_plan* plan, const uint8_t* rgb, size_t row_stride_bytes);
FFTUP_API int fftup_upload_rgb8_slot(fftup_plan* plan, uint32_t slot, const uint8_t* rgb, size_t row_stride_bytes);
/* transferDataFromCPU of an already packed planar buffer: float (precision 0), double (precision 1) or IEEE
 * half (precision 2) planes, 3 planes of H rows; strides in elements (the reference's inputBuffer uses
 * row stride W and plane stride (W+2)*H, VR:1644). */
FFTUP_API int fftup_upload_planar(fftup_plan* plan, uint32_t slot, const void* planes,
                                  size_t row_stride_elems, size_t plane_stride_elems);

/* performVulkanUpscale (VR:1249-1279): enqueue n_iter full pipelines, one synchronisation.
 * *ms_per_iter = device time from before the first to after the last launch, divided by n_iter -- the
 * reference's "Time: X ms" (VR:1270-1278).  Every iteration reads input slot 0 and computes the same frame; output slot 0
 * holds it afterwards.  The reference records the n_iter pipelines into ONE command buffer on ONE queue, and every stage ends
 * in a compute-to-compute pipeline barrier (vkFFT.h:7678, VR:1217): iteration i + 1 cannot start before the sharpen pass of
 * iteration i has finished.  So here: one stream, the iterations in order, nothing overlaps -- single-frame latency, the figure
 * that is comparable with the reference's.  FFTUP_FLAG_OVERLAP_ITERATIONS (an extension) lets the identical iterations of a
 * call alternate on the plan's FFTUP_STREAMS streams instead (own spectra, own scratch output per stream beyond the first,
 * allocated on the first such call): the same bits, the throughput figure of fftup_execute_ring. */
FFTUP_API int fftup_execute(fftup_plan* plan, uint32_t n_iter, double* ms_per_iter);
/* batched mode: n_frames pipelines, frame i reads input slot (first_slot+i) % ring and writes
 * output slot (first_slot+i) % ring; returns total device milliseconds. */
FFTUP_API int fftup_execute_ring(fftup_plan* plan, uint32_t n_frames, uint32_t first_slot, double* ms_total);
/* the same batch with a HIP event before and after every kernel launch of every `stride`-th frame, recorded on
 * the stream that runs the kernel: also returns the average duration (ms) of each of the plan's kernels.  Consecutive
 * frames run on two streams, so a kernel's duration includes the time it shares the GPU with the other
 * stream's kernels (fftup_profile_kernels gives the isolated durations). */
FFTUP_API int fftup_execute_ring_timed(fftup_plan* plan, uint32_t n_frames, uint32_t first_slot, uint32_t stride,
                                       double* ms_total, double* ms_per_kernel);
/* measurement aid: runs n_iter frames with a HIP event pair around every kernel launch on the
 * plan's stream and returns the average duration (ms) of each of the FFTUP_NUM_KERNELS kernels, net of the
 * duration of an empty event pair measured in the same loop (unused slots read 0). */
FFTUP_API int fftup_profile_kernels(fftup_plan* plan, uint32_t n_iter, double* ms_per_kernel);

/* transferDataToCPU + unpack loop (VR:1697-1748).  rgb8: u8 = trunc(255*x), saturating unless
 * FFTUP_FLAG_U8_WRAP.  planar: dense [3][uH][uW] float (precision 0), double (1) or half (2). */
FFTUP_API int fftup_download_rgb8(fftup_plan* plan, uint32_t slot, uint8_t* rgb, size_t row_stride_bytes);
FFTUP_API int fftup_download_planar(fftup_plan* plan, uint32_t slot, void* planes);
/* parity-test taps: the C2R output before sharpening (the reference's tempBuffer contents,
 * dense [3][uH][uW]) of the last executed frame -- on the non-R2C path (uW > 8192, VR:1424) the real
 * part of the complex image -- and the converted input planes [3][H][W]. */
FFTUP_API int fftup_download_presharpen(fftup_plan* plan, void* planes);
FFTUP_API int fftup_download_input_planar(fftup_plan* plan, uint32_t slot, void* planes);
/* job accounting (batched / multi-GPU runs): 64-bit wrapping sum of the 32-bit words of WHATEVER output slot `slot` holds --
 * the dense [3][uH][uW] float / half planes, or the interleaved 8-bit image [uH][uW][3] of a plan with
 * fftup_info.u8_store == 1 -- computed on the device: a frame's fingerprint without moving the frame over PCIe.  Sums over
 * the frames of a job do not depend on which rank or thread processed which frame; they are comparable only between plans
 * of the same precision and the same u8_store. */
FFTUP_API int fftup_output_checksum(fftup_plan* plan, uint32_t slot, uint64_t* sum);

/* Host-streamed batches (replaces the blocking transferDataFromCPU / transferDataToCPU + the two CPU
 * conversion loops of VR:1636-1748 for the batched mode, VR:1621-1760).  fftup_submit_rgb8 enqueues one whole
 * frame -- H2D copy, uint8 -> float conversion, the frame's kernels, float -> uint8 conversion, D2H copy -- and
 * returns at once with a ticket; up to `ring` frames are in flight, copies of one frame overlap the kernels of
 * its neighbours (each frame runs start to end on one of the plan's streams, consecutive frames on different ones).  A submit that finds its ring slot
 * still busy first waits for that slot's frame.  fftup_wait(ticket) returns when rgb_out of that submission is
 * complete; fftup_drain waits for everything submitted.  For the copies to be asynchronous both host buffers
 * must be page-locked: allocate them with fftup_host_alloc (pageable memory works, the copies then block).
 * Threads: fftup_submit_rgb8 / fftup_wait / fftup_drain of ONE plan may be called from several host threads at once (a
 * pool of PNG decode/encode workers feeding one plan per GPU -- plan creation is serialised by the HIP runtime, ~20 ms per
 * plan, so sixty-four plans of sixty-four threads cost seconds where one shared plan costs none); tickets are global to the plan.
 * Every other entry point needs one thread per plan at a time, as in the reference (one application per host thread). */
FFTUP_API void* fftup_host_alloc(size_t bytes);      /* NULL on failure (fftup_last_error) */
FFTUP_API void fftup_host_free(void* ptr);
FFTUP_API int fftup_submit_rgb8(fftup_plan* plan, const uint8_t* rgb_in, size_t in_stride_bytes, uint8_t* rgb_out,
                                size_t out_stride_bytes, uint64_t* ticket);
FFTUP_API int fftup_wait(fftup_plan* plan, uint64_t ticket);
FFTUP_API int fftup_drain(fftup_plan* plan);

/* The frame leaves the GPU as a finished PNG (round 4; replaces stbi_write_png's work, VR:1754, for the batched mode): like
 * fftup_submit_rgb8, but the 8-bit image stays on the device, where the PNG row filters (the reference writer's minimum-sum
 * heuristic), a Huffman-only deflate stream and its Adler-32 are computed (csrc/kernels_png.hpp); fftup_wait_png copies the
 * stream -- 14 MB instead of 25 MB of pixels for a 4096x2048 frame -- into png_out, frames it (signature, IHDR, IDAT, IEND,
 * CRCs) and returns the file's size.  png_out needs fftup_png_bound(plan) bytes (page-locked for a fast copy).  With png_out
 * named at submission already (a 16-byte aligned buffer of fftup_host_alloc; NULL: not yet) the GPU writes the stream into it
 * itself, sized by the count it knows, and fftup_wait_png(.., the same buffer, ..) only waits, adds the framing and the CRC --
 * no size round trip through the host: several frames of one thread stream back to back.  A ticket of
 * fftup_submit_png must be collected by fftup_wait_png: its ring slot stays with it until then.  Submissions (fftup_submit_png
 * or fftup_submit_rgb8, any thread) take the next slot that holds no uncollected stream -- so threads that keep tickets open
 * while they submit cannot wait for each other in a circle as long as fewer than `ring` streams are uncollected in total; with
 * every slot held a submission waits for a collector -- another thread's fftup_wait_png (one thread submitting, another
 * collecting, more frames than slots: the submission waits) -- unless every uncollected stream is the submitting thread's own
 * AND no other thread has ever collected on this plan, which would wait forever: after a bounded wait that call fails with
 * FFTUP_E_WOULD_BLOCK (with the default ring of 1 and one thread: collect each ticket before the next submission).
 * fftup_wait_png: a device error or FFTUP_E_OVERFLOW voids the ticket and frees its slot; a caller's error (capacity too
 * small, a buffer other than the one named at submission) leaves the stream on the device and the ticket collectable.
 * -p 0 and -p 2 plans whose stream bound fits one
 * IDAT chunk (2^31 - 1 bytes: FFTUP_E_UNSUPPORTED_SIZE beyond); thread-safe like fftup_submit_rgb8 / fftup_wait. */
FFTUP_API size_t fftup_png_bound(fftup_plan* plan);
FFTUP_API int fftup_submit_png(fftup_plan* plan, const uint8_t* rgb_in, size_t in_stride_bytes, uint8_t* png_out, size_t capacity,
                               uint64_t* ticket);
FFTUP_API int fftup_wait_png(fftup_plan* plan, uint64_t ticket, uint8_t* png_out, size_t capacity, size_t* png_bytes);

FFTUP_API const char* fftup_strerror(int code);
FFTUP_API const char* fftup_last_error(void);   /* thread-local detail of the last failure */
FFTUP_API const char* fftup_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FFTUP_H */
