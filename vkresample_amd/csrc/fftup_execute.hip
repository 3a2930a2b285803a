// fftup_execute.hip -- transfers and execution behind the C ABI: pack loop + transferDataFromCPU (VkResample.cpp:1636-1688),
// performVulkanUpscale (VR:1249-1279), transferDataToCPU + unpack loop (VR:1697-1748), the batched ring, kernel timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"

extern "C" {

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


}  // extern "C"

// shared body of fftup_execute_ring / fftup_execute_ring_timed (and of the plan-time tuner's timing, fftup_plan.hip).  With kernel_ms != NULL a HIP event is recorded
// before and after every kernel launch ON THE STREAM THAT RUNS IT (one event chain per lane) and the average
// duration of each of the plan's kernels over this very batch is returned.
int execute_ring_impl(fftup_plan* P, uint32_t n_frames, uint32_t first_slot, double* ms_total, double* kernel_ms,
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

extern "C" {

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
    // The reference records n_iter identical pipelines in ONE command buffer on one queue, every stage behind a pipeline barrier
    // (VR:1260-1278, vkFFT.h:7678, VR:1217), and reports wall time / n_iter: the iterations run in order, here on ONE stream.
    // FFTUP_FLAG_OVERLAP_ITERATIONS (extension): iteration i runs on stream i mod nl with that stream's own spectra and, beyond
    // stream 0, its own output buffer (stream 0 writes output slot 0, which is what fftup_download_* read); a kernel of one
    // iteration then overlaps other kernels of its neighbours exactly as consecutive frames of fftup_execute_ring do; every
    // iteration computes the bits a lone one computes.
    const bool sequential = !(P->cfg.flags & FFTUP_FLAG_OVERLAP_ITERATIONS);
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
    // test builds, FFTUP_EXPERIMENT evict_mb=N: N MB are written in front of EVERY kernel launch, so that each kernel starts with
    // nothing of its predecessor's output in the L2s or the Infinity Cache -- counter calibration runs (the FETCH_SIZE correction
    // of a kernel is measured on that kernel reading a known number of bytes from HBM); the durations then include the fill
    void* evict = nullptr;
    size_t evict_bytes = 0;
    if (const char* e = fftup_jit::experiment("evict_mb")) {
        evict_bytes = (size_t)std::max(0, atoi(e)) << 20;
        if (evict_bytes) HIP_TRY(hipMalloc(&evict, evict_bytes));
    }
    struct FreeEvict { void* p; ~FreeEvict() { if (p) (void)hipFree(p); } } free_evict{evict};
    for (uint32_t i = 0; i < n_iter && !rc; i++) {
        hipEvent_t* e = &ev.ev[(size_t)i * NE];
        (void)hipEventRecord(e[0], P->stream);
        for (int k = 0; k < FFTUP_NUM_KERNELS && !rc; k++) {
            if (evict) (void)hipMemsetAsync(evict, (int)(i + k) & 255, evict_bytes, P->stream);
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
    launch_checksum(P, slot, P->stream);
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

}  // extern "C"
