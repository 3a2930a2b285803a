// fftup_png.hip -- host side of the device-side PNG encoder (csrc/kernels_png.hpp): the frame leaves the GPU as a finished PNG
// data stream (replaces stbi_write_png's work, VkResample.cpp:1754, for the batched mode); framing and chunk CRCs on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"
#include "crc32.hpp"
#include "kernels_png.hpp"

using namespace fftup;

// ---- geometry: fixed per plan
static size_t png_stream_bound(size_t raw_bytes, int nblocks)
{
    // a Huffman code for 257 symbols spends at most ~8.1 bits per symbol on average; nine and an eighth are allowed for, 236
    // bytes of header per block, and the words the last atomicOr may touch.  (An argument about a heuristic length limit, not a
    // guarantee: k_png_layout checks the real size against the buffer before a single bit is packed.)
    return (raw_bytes + raw_bytes / 8 + raw_bytes / 64 + ((size_t)nblocks + 2) * 512 + 64 + 15) / 16 * 16;
}
void png_geometry(fftup_plan* P)
{
    if (P->png_rpb) return;
    const size_t L = (size_t)3 * P->uW + 1;
    P->png_rpb = (int)std::max<size_t>(1, (192 * 1024) / L);           // ~192 KB of residuals per deflate block, whole rows
    P->png_nblocks = (int)((P->uH + (uint32_t)P->png_rpb - 1) / (uint32_t)P->png_rpb);
    P->png_stream_bytes = png_stream_bound(L * P->uH, P->png_nblocks);
}

// the encoder's buffers of one ring slot, created on the slot's first PNG frame.  Pointers go into the slot as they are
// allocated: a failure half way leaves them there (owned by the plan), and a later attempt continues instead of allocating again
int png_slot_init(fftup_plan* P, fftup_plan::QSlot& Q)
{
    fftup_plan::PngSlot& G = Q.png;
    if (G.ready) return FFTUP_OK;
    png_geometry(P);
    const size_t L = (size_t)3 * P->uW + 1, nb = (size_t)P->png_nblocks, uH = P->uH;
    PngParams& p = G.p;
    auto need = [&](auto** ptr, size_t bytes) -> int { return *ptr ? FFTUP_OK : dev_alloc(P, (void**)ptr, bytes); };
    int rc = need(&p.raw, L * uH + 8);                            // (+ 8: k_png_pack reads whole words)
    if (!rc) rc = need(&p.rowhist, uH * 257 * sizeof(uint32_t));
    if (!rc) rc = need(&p.rowsum, uH * 2 * sizeof(unsigned long long));
    if (!rc) rc = need(&p.tab, nb * 257 * sizeof(uint32_t));
    if (!rc) rc = need(&p.hdr, nb * 64 * sizeof(uint32_t));
    if (!rc) rc = need(&p.hdr_bits, nb * sizeof(uint32_t));
    if (!rc) rc = need(&p.block_bits, nb * sizeof(unsigned long long));
    if (!rc) rc = need(&p.block_start, nb * sizeof(unsigned long long));
    if (!rc) rc = need(&p.row_off, uH * sizeof(unsigned long long));
    if (!rc) rc = need(&p.meta, 3 * sizeof(unsigned long long));
    if (!rc) rc = need(&p.crc_parts, (P->png_stream_bytes / 4096 + 1) * sizeof(uint32_t));
    if (!rc) rc = need(&p.stream, P->png_stream_bytes);
    if (rc) return rc;
    if (!P->png_crc_shift) {                                      // one table per plan: shift_256^0..15 (crc32.hpp), 2 KB
        uint32_t m[16][32];
        fftup_crc::crc32_shift_256_powers(m);
        uint32_t* tab = nullptr;                                  // published only once it holds the table: a failed copy must not
        rc = dev_alloc(P, (void**)&tab, sizeof m);                // leave a pointer behind that the next attempt takes for a filled one
        if (rc) return rc;
        HIP_TRY(hipMemcpy(tab, m, sizeof m, hipMemcpyHostToDevice));
        P->png_crc_shift = tab;
    }
    p.crc_shift = P->png_crc_shift;
    p.capacity = P->png_stream_bytes;
    // (test knob, libfftup_knobs.so only: a smaller capacity than the buffer has, so that the overflow path can be exercised)
    if (const char* e = fftup_jit::experiment("png_capacity")) p.capacity = std::min<unsigned long long>(p.capacity, strtoull(e, nullptr, 10));
    p.uW = (int)P->uW; p.uH = (int)P->uH; p.rows_per_block = P->png_rpb; p.nblocks = P->png_nblocks;
    if (!G.meta_host) HIP_TRY(hipHostMalloc((void**)&G.meta_host, 3 * sizeof(unsigned long long), hipHostMallocDefault));
    if (!G.parts_host) HIP_TRY(hipHostMalloc((void**)&G.parts_host, (P->png_stream_bytes / 4096 + 1) * sizeof(uint32_t), hipHostMallocDefault));
    if (!G.copied) HIP_TRY(hipEventCreateWithFlags(&G.copied, hipEventDisableTiming));
    if (!P->png_copy) HIP_TRY(hipStreamCreateWithFlags(&P->png_copy, hipStreamNonBlocking));
    G.ready = true;
    return FFTUP_OK;
}

// The encoder's launches behind a frame on stream `cs` (called by submit_frame with the queue locked): the 8-bit image Q.out_u8
// stays on the device -- filter rows, code them, pack the bits; the stream's size and the per-piece CRCs come back now, the stream
// itself when fftup_wait_png knows how many bytes to ask for, or at once into `png_dest` (mapped page-locked memory) by the
// device's own stores.
int png_enqueue(fftup_plan* P, fftup_plan::QSlot& Q, hipStream_t cs, uint8_t* png_dest)
{
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
    if (max_pieces) hipLaunchKernelGGL(k_png_crc, dim3((unsigned)((max_pieces * 16 + 255) / 256)), dim3(256), 0, cs, pp);      // sixteen threads per 4 KB piece
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Q.png.meta_host, pp.meta, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    if (max_pieces) HIP_TRY(hipMemcpyAsync(Q.png.parts_host, pp.crc_parts, max_pieces * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    if (png_dest) {                                            // the device knows the size: it delivers the stream itself
        void* dev_view = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dev_view, png_dest, 0));
        hipLaunchKernelGGL(k_png_deliver, dim3(512), dim3(256), 0, cs, pp, (uint32_t*)dev_view);
        HIP_TRY(hipGetLastError());
    }
    return FFTUP_OK;
}

extern "C" {

int fftup_submit_png(fftup_plan* P, const uint8_t* rgb_in, size_t in_stride, uint8_t* png_out, size_t capacity, uint64_t* ticket)
{
    return submit_frame(P, rgb_in, in_stride, png_out, png_out ? capacity : 0, true, ticket);
}

size_t fftup_png_bound(fftup_plan* P)
{
    if (!P) return 0;
    return P->png_stream_bytes + 57;                      // signature 8, IHDR 25, IDAT framing 12, IEND 12
}

static void be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

int fftup_wait_png(fftup_plan* P, uint64_t ticket, uint8_t* png_out, size_t capacity, size_t* png_bytes)
{
    using fftup_crc::crc32_update;
    if (!P || !png_out || !png_bytes) return fail(FFTUP_E_INVALID_ARG, "null argument");
    fftup_plan::QSlot* Qp = nullptr;
    {
        // (slot state is written by submit_frame under the queue's lock: read it under it)
        std::lock_guard<std::mutex> lock(P->q_mu);
        if (ticket >= P->q_next.load(std::memory_order_relaxed)) return fail(FFTUP_E_INVALID_ARG, "ticket was never issued");
        for (auto& c : P->q)
            if (c.used && c.ticket == ticket && c.png.state == 1 && c.png.ticket == ticket) Qp = &c;
        if (!Qp) return fail(FFTUP_E_INVALID_ARG, "no PNG stream is waiting under this ticket");
        Qp->png.state = 2;                                            // being collected: a second collector of the same ticket finds nothing
        if (Qp->png.owner != std::this_thread::get_id()) P->png_foreign_collector = true;      // (producer / consumer: submit_frame may wait for us)
    }
    fftup_plan::QSlot& Q = *Qp;
    // From here on the ticket is this caller's.  A device error or an overflowed stream voids it and hands the slot back (a slot
    // left in state 1 would block every later submission that comes round to it); a CALLER's error -- a buffer too small, or not
    // the one named at submission -- leaves the encoded stream where it is and the ticket collectable again (keep = true).
    struct Release {
        fftup_plan* P; fftup_plan::QSlot& Q; bool keep = false;
        ~Release() { { std::lock_guard<std::mutex> lock(P->q_mu); Q.png.state = keep ? 1 : 0; } P->q_cv.notify_all(); }
    } release{P, Q};
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipEventSynchronize(Q.done));
    if (Q.png.meta_host[2])
        return fail(FFTUP_E_OVERFLOW, "PNG stream of " + std::to_string((size_t)Q.png.meta_host[2]) + " bytes exceeds the encoder's buffer of " +
                                          std::to_string(P->png_stream_bytes) + ": frame not encoded (use fftup_submit_rgb8 and encode on the host)");
    const size_t zbytes = (size_t)Q.png.meta_host[0];
    if (zbytes < 6 || zbytes > P->png_stream_bytes) return fail(FFTUP_E_HIP, "PNG encoder reported an impossible stream size");
    if (zbytes + 57 > capacity) {
        release.keep = true;
        return fail(FFTUP_E_INVALID_ARG, "PNG buffer too small: " + std::to_string(zbytes + 57) + " bytes needed (fftup_png_bound); the ticket stays collectable");
    }
    if (Q.png.dest != png_out) {                           // (delivered by the device already when the buffer was named at submission)
        if (Q.png.dest) {
            release.keep = true;
            return fail(FFTUP_E_INVALID_ARG, "fftup_wait_png: the buffer named by fftup_submit_png holds this file; the ticket stays collectable");
        }
        HIP_TRY(hipMemcpyAsync(png_out + 41, Q.png.p.stream, zbytes, hipMemcpyDeviceToHost, P->png_copy));
        HIP_TRY(hipEventRecord(Q.png.copied, P->png_copy));
        HIP_TRY(hipEventSynchronize(Q.png.copied));
    }
    // "IDAT", then the stream: whole 4 KB pieces from the device (k_png_crc), the tail here
    uint32_t crc = crc32_update(0, (const uint8_t*)"IDAT", 4);
    const size_t pieces = zbytes / 4096;
    for (size_t k = 0; k < pieces; k++) crc = fftup_crc::crc32_shift_4096(crc) ^ Q.png.parts_host[k];
    crc = crc32_update(crc, png_out + 41 + pieces * 4096, zbytes - pieces * 4096);
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    memcpy(png_out, sig, 8);
    uint8_t* q = png_out + 8;                              // IHDR: 8-bit RGB, no interlace
    be32(q, 13); memcpy(q + 4, "IHDR", 4); be32(q + 8, P->uW); be32(q + 12, P->uH);
    q[16] = 8; q[17] = 2; q[18] = 0; q[19] = 0; q[20] = 0;
    be32(q + 21, crc32_update(0, q + 4, 17));
    q = png_out + 33;
    be32(q, (uint32_t)zbytes); memcpy(q + 4, "IDAT", 4);   // (zbytes <= 2^31 - 1: fftup_submit_png refuses plans whose bound is larger)
    be32(png_out + 41 + zbytes, crc);
    q = png_out + 41 + zbytes + 4;
    be32(q, 0); memcpy(q + 4, "IEND", 4); be32(q + 8, crc32_update(0, q + 4, 4));
    *png_bytes = zbytes + 57;
    return FFTUP_OK;
}

}  // extern "C"
