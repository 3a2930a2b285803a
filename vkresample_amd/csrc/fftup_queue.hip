// fftup_queue.hip -- host-streamed frames: fftup_submit_rgb8 / fftup_wait / fftup_drain replace the blocking
// transfers and the two CPU conversion loops of the reference's batched mode (VkResample.cpp:1621-1760): each frame's H2D copy,
// conversion, kernels, conversion and D2H copy go to ONE of the plan's streams, `ring` frames in flight.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"

extern "C" {

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


}  // extern "C"

// shared body of fftup_submit_rgb8 / fftup_submit_png: one whole frame on one of the plan's streams
int submit_frame(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* rgb_out, size_t out_stride, bool png, uint64_t* ticket)
{
    // (png: rgb_out / out_stride are the optional destination of the finished file and its capacity)
    uint8_t* png_dest = png ? rgb_out : nullptr;
    const size_t png_cap = png ? out_stride : 0;
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    if (!rgb_in || in_stride < (size_t)3 * P->W) return fail(FFTUP_E_INVALID_ARG, "bad input pointer/stride");
    if (!png && (!rgb_out || out_stride < (size_t)3 * P->uW)) return fail(FFTUP_E_INVALID_ARG, "bad output pointer/stride");
    if (png && P->dbl) return fail(FFTUP_E_UNSUPPORTED_PRECISION, "device-side PNG encoding: -p 0 and -p 2 plans");
    // a PNG chunk holds at most 2^31 - 1 bytes and the stream goes out as ONE IDAT chunk
    if (png && P->png_stream_bytes > 0x7fffffffu) return fail(FFTUP_E_UNSUPPORTED_SIZE, "device-side PNG encoding: image too large for one IDAT chunk");
    if (png_dest) {
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
    // The slot: the next one in turn that holds no uncollected PNG stream (a slot is a set of device buffers; which one a frame
    // goes through is the queue's business -- tickets count submissions).  Taking slots strictly in turn, as round 4 did, lets
    // two threads that each keep a ticket open while submitting wait for each other's slots in a circle; skipping held slots
    // cannot: while fewer than `ring` streams are uncollected a slot is free.  Every slot held: wait for a collector.  A slot
    // that is being collected right now (state 2) is about to be free whoever submitted it.  Slots this thread filled itself
    // are freed by ANOTHER thread in the producer / consumer pattern (one thread submits, one collects) and by nobody when
    // the caller does both: the submitter's id alone cannot tell, so once some other thread has collected on this plan the
    // wait is unconditional, and before that it is bounded -- a consumer that has not started yet gets two seconds to show
    // up -- and ends in FFTUP_E_WOULD_BLOCK (round 5 returned the error at once: spurious for a producer ahead of its consumer).
    uint32_t s = 0;
    bool waited_for_self = false;
    for (;;) {                                                        // (the lock is released while waiting: the search starts over)
        bool found = false, others = false;
        for (uint32_t i = 0; i < P->ring && !found; i++) {
            s = (P->q_cursor + i) % P->ring;
            if (P->q[s].png.state == 0) found = true;
            else if (P->q[s].png.state == 2 || P->q[s].png.owner != std::this_thread::get_id()) others = true;
        }
        if (found) break;
        if (others || P->png_foreign_collector) { P->q_cv.wait(lock); continue; }
        // (FFTUP_SELF_WAIT_MS: the bound, for callers that know better and for tests)
        const char* sw = getenv("FFTUP_SELF_WAIT_MS");
        const std::chrono::milliseconds bound(sw ? std::max(0, atoi(sw)) : 2000);
        if (waited_for_self || P->q_cv.wait_for(lock, bound) == std::cv_status::timeout) {
            bool still = true;                                        // (a wake-up without a free slot and a time-out both end here)
            for (uint32_t i = 0; i < P->ring; i++) still = still && P->q[i].png.state == 1 && P->q[i].png.owner == std::this_thread::get_id();
            if (still && !P->png_foreign_collector)
                return fail(FFTUP_E_WOULD_BLOCK, "all " + std::to_string(P->ring) + " ring slot(s) hold PNG tickets of this thread and no other thread "
                                                 "collects on this plan: collect one with fftup_wait_png before submitting again");
            waited_for_self = true;
        }
    }
    const uint64_t t = P->q_next.load(std::memory_order_relaxed);
    P->q_cursor = (s + 1) % P->ring;
    fftup_plan::QSlot& Q = P->q[s];
    if (Q.used) HIP_TRY(hipEventSynchronize(Q.done));                // the slot's previous frame has left the device
    if (png && (rc = png_slot_init(P, Q)) != FFTUP_OK) return rc;
    // The whole frame -- H2D, conversion, kernels, conversion, D2H -- goes to ONE stream (lane t % nlanes), so no
    // cross-stream dependency exists and nothing can stall behind a neighbour's wait when streams share a hardware
    // queue; the copies of one lane overlap the kernels and the opposite-direction copies of the other lanes.
    // (two lanes: with the copies in the streams a third one only adds contention, 0.56-0.75 ms/frame instead of 0.51)
    // (a PNG frame's chain is long -- nine more launches, most of them a handful of workgroups: all lanes take turns)
    const int lane = (int)(t % (uint64_t)(png ? P->nlanes : std::min(P->nlanes, 2)));
    hipStream_t cs = P->lanes[lane].stream;
    const size_t in_row = (size_t)3 * P->W, out_row = (size_t)3 * P->uW;
    // From the first enqueue on, a failure must leave the slot consistent: whatever was enqueued may still be running when the
    // next submission comes round to this slot, so its `done` event is recorded behind it on every path out (enqueue()).
    auto enqueue = [&]() -> int {
        if (in_stride == in_row) HIP_TRY(hipMemcpyAsync(P->in_u8[s], rgb_in, in_row * P->H, hipMemcpyHostToDevice, cs));
        else HIP_TRY(hipMemcpy2DAsync(P->in_u8[s], in_row, rgb_in, in_stride, in_row, P->H, hipMemcpyHostToDevice, cs));
        if (fuse_u8(P)) {
            P->in_kind[s] = 2;
        } else {
            launch_unpack(P, s, cs);
            P->in_kind[s] = 1;
        }
        P->cur = lane;
        const int frc = launch_frame(P, s, s, -1);
        P->last_lane = lane;
        P->cur = 0;
        if (frc) return frc;
        if (!P->u8out) launch_pack(P, s, Q.out_u8, cs);
        HIP_TRY(hipGetLastError());
        if (png) return png_enqueue(P, Q, cs, png_dest);
        if (out_stride == out_row) HIP_TRY(hipMemcpyAsync(rgb_out, Q.out_u8, out_row * P->uH, hipMemcpyDeviceToHost, cs));
        else HIP_TRY(hipMemcpy2DAsync(rgb_out, out_stride, Q.out_u8, out_row, out_row, P->uH, hipMemcpyDeviceToHost, cs));
        return FFTUP_OK;
    };
    rc = enqueue();
    const hipError_t rec = hipEventRecord(Q.done, cs);
    if (rc || rec != hipSuccess) {
        // the frame is void (no ticket); what reached the stream is waited for here, so that the slot's buffers are free for
        // whoever submits next (the error paths are not the fast paths)
        (void)hipStreamSynchronize(cs);
        return rc ? rc : fail(FFTUP_E_HIP, std::string("hipEventRecord: ") + hipGetErrorString(rec));
    }
    Q.used = true;
    Q.ticket = t;
    if (png) {
        Q.png.state = 1;
        Q.png.ticket = t;
        Q.png.owner = std::this_thread::get_id();
        Q.png.dest = png_dest;
        Q.png.dest_cap = png_cap;
    }
    P->q_next.store(t + 1, std::memory_order_release);
    P->executed = 1;
    if (ticket) *ticket = t;
    return FFTUP_OK;
}

extern "C" {

int fftup_submit_rgb8(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* rgb_out, size_t out_stride,
                      uint64_t* ticket)
{
    return submit_frame(P, rgb_in, in_stride, rgb_out, out_stride, false, ticket);
}

// the slot a ticket went through, or nullptr when that slot has carried a later frame since (the ticket's frame is then long done);
// q_mu held
static fftup_plan::QSlot* slot_of(fftup_plan* P, uint64_t ticket)
{
    for (auto& Q : P->q)
        if (Q.used && Q.ticket == ticket) return &Q;
    return nullptr;
}

int fftup_wait(fftup_plan* P, uint64_t ticket)
{
    if (!P) return fail(FFTUP_E_INVALID_ARG, "null plan");
    hipEvent_t done = nullptr;
    {
        std::lock_guard<std::mutex> lock(P->q_mu);                // (slot state is written by submit_frame under the lock)
        if (ticket >= P->q_next.load(std::memory_order_relaxed)) return fail(FFTUP_E_INVALID_ARG, "ticket was never issued");
        const fftup_plan::QSlot* Q = slot_of(P, ticket);
        if (!Q) return FFTUP_OK;                                  // its slot has been reused: submit already waited for it
        if (Q->png.state == 1 && Q->png.ticket == ticket) return fail(FFTUP_E_INVALID_ARG, "a ticket of fftup_submit_png is collected by fftup_wait_png");
        done = Q->done;
    }
    // (a submission of another thread may re-record this slot's event right now: the wait then covers the later frame too)
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipEventSynchronize(done));
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

}  // extern "C"
