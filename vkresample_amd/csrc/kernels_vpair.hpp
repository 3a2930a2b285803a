// kernels_vpair.hpp -- fused C2R + sharpen for output rows of 4096 points (2048-wide inputs, u = 2): k_c2r_sharpen_v.
//
// Same job, same strips, same prefetch and the same quirk handling as k_c2r_sharpen_g<FusedPlanPow2<4096>>
// (kernels_pow2.hpp, which documents them: pairing and the DC leak B3, row-end wrap B5, deferred pixel, corner sample);
// what differs is how the data moves inside the compute unit.  k_c2r_sharpen_g is a constant-geometry Stockham
// transform: three workgroup-wide LDS exchanges with a barrier each, L rows written as 16 single floats per thread and
// read back as 12 sixteen-byte taps.  Round 3 measured that kernel's time as (vector issue) + (LDS operand traffic) +
// (barrier skew), none hiding the others (profiles/r03_a_*), so this kernel removes LDS traffic and barriers:
//
//   * the transform is a digit-swap Cooley-Tukey: 4096 = 8^4, thread (wave w, lane l = a + 8 b) starts with the input
//     digits (n3 | n2, n1, n0) = (register | w, a, b) and each exchange swaps the register digit with ONE thread digit:
//       A: register <-> wave       the only workgroup-wide exchange: LDS, one barrier; a lane's elements of one
//                                  instruction are contiguous -- no swizzle, no address arithmetic, immediates only;
//                                  its twiddles exp(-2 pi i n2 k0 / 64) are wave-uniform: seven scalar register pairs
//       B: register <-> lane bits 0-2   inside a wave: LDS block of the wave itself (the 4 KB it has just read in A),
//                                  program order instead of a barrier
//       C: register <-> lane bits 3-5   no LDS at all: v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4) --
//                                  new in gfx950, one instruction per register pair -- and three row_ror:8 DPP moves
//                                  per pair for bit 3: 40 vector instructions for the whole exchange
//     and ends with X[w + 8 l + 512 q] in register q of thread (w, l);
//   * both rows of a pair travel together: the L values (|u^2 g| clamped) of rows a, a+1 at one x are ONE 8-byte element
//     (4 bytes for -p 2: a binary16 pair).  Row layout in LDS: element of x at (x & 7) * 512 + (x >> 3), so that wave w
//     -- which owns x = w mod 8 -- writes one contiguous run per register (8 stores, conflict-free) and thread c reads
//     the eight pixels 8c .. 8c+7 of both rows with four ds_read2st64 (plus two halo elements);
//   * the sharpen filter runs on VERTICAL pairs (output rows a-1 and a at the same x): the previous pair's elements
//     stay in registers (P0), centre taps are (P0.y, P1.x), and for -p 2 one v_pk_minimum3_f16 gives the column minimum
//     of both output rows, with no shifted-pair building at all.
//
// Per wave and row pair: 43 LDS instructions (70), 2 barriers (4).  Numerics: the same butterflies (bfly8_pk), table
// twiddles with powers by multiplication, and the sharpen arithmetic of k_c2r_sharpen_g, operation for operation.
#pragma once
#include "kernels_pow2.hpp"

namespace fftup {

// z * w, w in a scalar register pair (wave-uniform twiddle): the two packed instructions of cmul_tw
__device__ __forceinline__ float2 cmul_tw_s(float2 z, float2 w)
{
    typedef float cf2 __attribute__((ext_vector_type(2)));
    const cf2 zv = {z.x, z.y}, wv = {w.x, w.y};
    cf2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(zv), "s"(wv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(zv), "s"(wv), "v"(t));
    return make_float2(r.x, r.y);
}

// Exchange C: transpose (register index bits 2,1,0) with (lane bits 5,4,3) of eight complex registers.
// v_permlane32_swap a, b swaps a[lanes 32..63] with b[lanes 0..31]; v_permlane16_swap swaps the odd 16-lane rows of a with
// the even rows of b: element (register bit = 0, lane bit = 1) <-> (register bit = 1, lane bit = 0), which is the
// transposition of that bit pair.  (Written as asm: the builtins of this compiler return the first register twice.  A
// vector instruction's result needs two wait states before a permlane swap or a DPP move reads it: the s_nop in front;
// inside the blocks dependent instructions are at least eight apart.)
__device__ __forceinline__ void lane_transpose_hi3(float2 (&v)[8])
{
#define FFTUP_SWAP8(OP, A0, B0, A1, B1, A2, B2, A3, B3)                                                                         \
    asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %8, %9\n\t" OP         \
                 " %10, %11\n\t" OP " %12, %13\n\t" OP " %14, %15"                                                             \
                 : "+v"(v[A0].x), "+v"(v[B0].x), "+v"(v[A0].y), "+v"(v[B0].y), "+v"(v[A1].x), "+v"(v[B1].x), "+v"(v[A1].y),     \
                   "+v"(v[B1].y), "+v"(v[A2].x), "+v"(v[B2].x), "+v"(v[A2].y), "+v"(v[B2].y), "+v"(v[A3].x), "+v"(v[B3].x),     \
                   "+v"(v[A3].y), "+v"(v[B3].y))
    FFTUP_SWAP8("v_permlane32_swap_b32", 0, 4, 1, 5, 2, 6, 3, 7);
    FFTUP_SWAP8("v_permlane16_swap_b32", 0, 2, 1, 3, 4, 6, 5, 7);
#undef FFTUP_SWAP8
    // lane bit 3: T = A(lane ^ 8) everywhere; A(lanes 8-15 of a row) = B(lane ^ 8); B(lanes 0-7) = T
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %16, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %17, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %18, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %19, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %20, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %21, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %22, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %23, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %8, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %10, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %12, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %14, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_mov_b32_dpp %1, %16 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %3, %17 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %5, %18 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %7, %19 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %9, %20 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %11, %21 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %13, %22 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b32_dpp %15, %23 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3"
        : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[0].y), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[2].y), "+v"(v[3].y),
          "+v"(v[4].x), "+v"(v[5].x), "+v"(v[4].y), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[7].x), "+v"(v[6].y), "+v"(v[7].y),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7));
}

struct VPlan4096 {
    static constexpr int UW = 4096, T = 512;
    static constexpr size_t XB = 32768;          // exchange buffer: 4096 complex elements
    // [0, LB): L pairs of the current row pair (8 bytes per x, 4 for -p 2); [LB, LB + XB): exchanges A and B
    static constexpr size_t lds(bool half) { return (half ? 16384 : 32768) + XB; }
};
struct VTwid {
    float2 a[7];        // exchange A: exp(-2 pi i m w / 64), m = 1..7 (wave-uniform)
    float2 b, c;        // base twiddles of stages 2 and 3: exp(-2 pi i (w + 8 a) / 512), exp(-2 pi i (w + 8 l) / 4096)
};
__device__ __forceinline__ void vfft_load_tw(VTwid& t, const float2* __restrict__ tw, int lt)
{
    const int wu = __builtin_amdgcn_readfirstlane(lt >> 6);
#pragma unroll
    for (int m = 1; m < 8; m++) t.a[m - 1] = twid<-1>(tw[(64 * m * wu) & 4095]);
    t.b = twid<-1>(tw[8 * ((lt >> 6) + 8 * (lt & 7))]);
    t.c = twid<-1>(tw[(lt >> 6) + 8 * (lt & 63)]);
}

// Inverse transform of length 4096 on 512 threads.  On entry v[m] = Z[jj + 512 m], jj = (l >> 3) + 8 (l & 7) + 64 w;
// on return v[q] = X[w + 8 l + 512 q] (not scaled).  zb: 32 KB, 128-byte aligned, free on entry; one __syncthreads inside.
// hook(k), k = 0..3: called at four points spread over the transform (the caller's deferred output stores)
template <class Hook>
__device__ __forceinline__ void vfft4096(float2 (&v)[8], char* zb, int lt, const VTwid& tw, Hook hook)
{
    const unsigned w = (unsigned)lt >> 6, l = (unsigned)lt & 63u, a = l & 7u, b = l >> 3;
    const unsigned zbase = lds_addr(zb);
    bfly8_pk<-1>(v);
    {   // exchange A: element (k0, n2 = w, lane) at ((k0 * 8 + w) * 64 + l)
        const unsigned aw = zbase + 8u * (w * 64u + l);
#pragma unroll
        for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)(aw + 4096u * k) = r; }
        hook(0);
        __syncthreads();
        const unsigned ar = zbase + 8u * (w * 512u + l);
#pragma unroll
        for (int m = 0; m < 8; m++) { const lds_f2raw r = *(const lds_f2*)(size_t)(ar + 512u * m); v[m] = make_float2(r.x, r.y); }
#pragma unroll
        for (int i = 0; i < 8; i += 2) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y));
    }
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmul_tw_s(v[m], tw.a[m - 1]);
    bfly8_pk<-1>(v);
    {   // exchange B, inside the wave's own 4 KB block: element (n1, b, k1) at n1 * 64 + 8 b + (n1 ^ k1)
        const unsigned blk = zbase + w * 4096u;
        const unsigned awr = blk + 512u * a + 64u * b + 8u * a;
#pragma unroll
        for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)(awr ^ (8u * k)) = r; }
        hook(1);
        lds_sync<true>();
        const unsigned ard = blk + 64u * b + 8u * a;
#pragma unroll
        for (int m = 0; m < 8; m++) { const lds_f2raw r = *(const lds_f2*)(size_t)((ard ^ (8u * m)) + 512u * m); v[m] = make_float2(r.x, r.y); }
#pragma unroll
        for (int i = 0; i < 8; i += 2) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y));
        lds_sync<true>();
    }
    twiddle_powers<8>(v, tw.b);
    bfly8_pk<-1>(v);
    hook(2);
    lane_transpose_hi3(v);
    hook(3);
    twiddle_powers<8>(v, tw.c);
    bfly8_pk<-1>(v);
}

// =================================================================================== column pass, H = 1024, digit-swap form
// k_col_t<1024, 4> (kernels_pow2.hpp) with the exchanges of the digit-swap transform: same tiles, same thread <-> element
// map at load and store (thread pp = tid / 4 of column col = tid % 4 owns rows pp + 128 i), same polyphase arithmetic
// (forward transform of length H, phase t[k], inverse transform of length H -> the odd rows).  Its two transforms of
// 1024 = 8 * 8 * 8 * 2 points needed three workgroup-wide LDS exchanges each -- 12 barriers per workgroup, and in a frame
// whose streams overlap every barrier is a place where a wave waits for seven others that compete with the fused kernel's
// waves for issue slots: with the barriers taken out (results invalid) the column kernel alone is no faster, the FRAME
// 8.6 % (profiles/r03_x_column_kernel.txt).  Here the forward transform runs decimation-in-time
//       registers <-> wave (LDS, barrier) | registers <-> lane bits 5-3 (permlane swaps) | register bit 2 <-> lane bit 2 (DPP) | radix 2
// and leaves F[k] at k = wave + 8 (lane bits 5-3) + 256 (lane bit 2) + 64 r + 512 k3 in register r + 4 k3; the phase is
// applied there, and the inverse runs the mirror image, decimation-in-frequency, from exactly that layout
//       radix 2 | register bit 2 <-> lane bit 2 | registers <-> lane bits 5-3 | registers <-> wave (LDS, barrier)
// ending in the load layout: 2 exchanges through LDS and 3 barriers per workgroup (6 and 12).
__device__ __forceinline__ void lane_swap_bit2(float2 (&v)[8])      // element (register r + 4, lane bit 2 = 0) <-> (register r, lane bit 2 = 1)
{
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32 %16, %0\n\tv_mov_b32 %17, %2\n\tv_mov_b32 %18, %4\n\tv_mov_b32 %19, %6\n\t"
        "v_mov_b32 %20, %8\n\tv_mov_b32 %21, %10\n\tv_mov_b32 %22, %12\n\tv_mov_b32 %23, %14\n\t"
        "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %2, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %6, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %8, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %10, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %12, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %14, %15 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_mov_b32_dpp %1, %16 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %3, %17 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %5, %18 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %7, %19 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %9, %20 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %11, %21 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %13, %22 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_mov_b32_dpp %15, %23 row_shl:4 row_mask:0xf bank_mask:0x5"
        : "+v"(v[0].x), "+v"(v[4].x), "+v"(v[0].y), "+v"(v[4].y), "+v"(v[1].x), "+v"(v[5].x), "+v"(v[1].y), "+v"(v[5].y),
          "+v"(v[2].x), "+v"(v[6].x), "+v"(v[2].y), "+v"(v[6].y), "+v"(v[3].x), "+v"(v[7].x), "+v"(v[3].y), "+v"(v[7].y),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7));
}
// registers <-> wave through LDS: element (register k, wave w, lane l) at ((k * 8 + w) * 64 + l); afterwards register m of
// wave w is what wave m held in register w.  One barrier inside; `buf` (32 KB) must be free on entry.
__device__ __forceinline__ void wave_exchange(float2 (&v)[8], unsigned zbase, unsigned w, unsigned l)
{
    const unsigned aw = zbase + 8u * (w * 64u + l);
#pragma unroll
    for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)(aw + 4096u * k) = r; }
    __syncthreads();
    const unsigned ar = zbase + 8u * (w * 512u + l);
#pragma unroll
    for (int m = 0; m < 8; m++) { const lds_f2raw r = *(const lds_f2*)(size_t)(ar + 512u * m); v[m] = make_float2(r.x, r.y); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y));
}

template <int TK>
__global__ void __launch_bounds__(512, FFTUP_COL_WAVES) k_col_v(ColTParams p)
{
    static_assert(TK == 4, "four columns of 128 threads: lane bits 0-1 = column, bits 2-5 and the wave = pp");
    constexpr int H = 1024;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const unsigned zbase = lds_addr(smem);
    const int tid = threadIdx.x;
    const unsigned w = (unsigned)tid >> 6, l = (unsigned)tid & 63u;
    const int col = tid & 3, pp = tid >> 2;                   // pp = 16 w + h, h = l >> 2
    const int tile = blockIdx.x, c = blockIdx.y;
    const bool valid = tile * TK + col <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    const int wu = __builtin_amdgcn_readfirstlane((int)w);
    const int k1 = (int)(l >> 3), lb2 = (int)((l >> 2) & 1u), hh = (int)(l >> 2);
    const int kt = wu + 8 * k1 + 256 * lb2;                   // the thread's part of k after the forward transform
    float2 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = valid ? src[tid + 512 * i] : make_float2(0.f, 0.f);      // row pp + 128 i
    // ---- forward, exp(+2 pi i n k / H), decimation in time
    float2 ta[7];
#pragma unroll
    for (int m = 1; m < 8; m++) ta[m - 1] = p.twH[(16 * m * wu) & (H - 1)];                     // wave-uniform: scalar registers
    const float2 tb = p.twH[2 * (wu + 8 * k1)], tc = p.twH[kt];
    const float2 tph = twid<-1>(p.twUH[kt]);                                                   // exp(-2 pi i kt / 2H)
    bfly8_pk<+1>(v);
    wave_exchange(v, zbase, w, l);
#pragma unroll
    for (int m = 1; m < 8; m++) v[m] = cmul_tw_s(v[m], ta[m - 1]);
    bfly8_pk<+1>(v);
    lane_transpose_hi3(v);
    twiddle_powers<8>(v, tb);
    bfly8_pk<+1>(v);
    lane_swap_bit2(v);
    {   // radix 2 over the bit that came out of the lane: twiddle exp(2 pi i (kt + 64 r) / H) on the upper element
        v[4] = cmul_tw(v[4], tc);
        v[5] = cmul_tw(v[5], cmul_tw(tc, rot16<1>()));
        v[6] = cmul_tw(v[6], cmul_tw(tc, rot16<2>()));
        v[7] = cmul_tw(v[7], cmul_tw(tc, rot16<3>()));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const pk2 a = {v[r].x, v[r].y}, b = {v[r + 4].x, v[r + 4].y};
            const pk2 s0 = pk_add(a, b), s1 = pk_sub(a, b);
            v[r] = make_float2(s0.x, s0.y); v[r + 4] = make_float2(s1.x, s1.y);
        }
    }
    // ---- phase: register r + 4 k3 holds F[k], k = kt + 64 r + 512 k3;  t[k] = exp(-2 pi i k / 2H) * (k < H/2 ? 1 : -1),
    // i.e. exp(-2 pi i kt / 2H) * exp(-2 pi i r / 32) * (k3 ? +i : 1)
    {
        const float2 t1 = cmul_tw(tph, twid<-1>(rot32<1>())), t2 = cmul_tw(tph, twid<-1>(rot32<2>())), t3 = cmul_tw(tph, twid<-1>(rot32<3>()));
        const float2 tt[4] = {tph, t1, t2, t3};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = cmul_tw(v[r], tt[r]);
            const float2 u = cmul_tw(v[r + 4], tt[r]);
            const pk2 iu = pk_muli<1>(pk2{u.x, u.y});
            v[r + 4] = make_float2(iu.x, iu.y);
        }
    }
    // ---- inverse, exp(-2 pi i k m / H), decimation in frequency
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const pk2 a = {v[r].x, v[r].y}, b = {v[r + 4].x, v[r + 4].y};
        const pk2 s0 = pk_add(a, b), s1 = pk_sub(a, b);
        v[r] = make_float2(s0.x, s0.y); v[r + 4] = make_float2(s1.x, s1.y);
    }
    lane_swap_bit2(v);                                       // lane bit 2 = h0 now, registers = k2
    {
        const float2 e = twid<-1>(rot16<1>());               // exp(-2 pi i / 16)
        const float2 base = lb2 ? e : make_float2(1.f, 0.f); // exp(-2 pi i k2 h0 / 16)
        twiddle_powers<8>(v, base);
    }
    bfly8_pk<-1>(v);
    lane_transpose_hi3(v);                                   // lane bits 5-3 = g, registers = k1
    twiddle_powers<8>(v, twid<-1>(p.twH[8 * hh]));           // exp(-2 pi i k1 h / 128)
    bfly8_pk<-1>(v);
    __syncthreads();                                         // everybody has read the forward exchange
    wave_exchange(v, zbase, w, l);                           // wave = 16s digit of pp, registers = k0
    twiddle_powers<8>(v, twid<-1>(p.twH[pp]));               // exp(-2 pi i k0 pp / H)
    bfly8_pk<-1>(v);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    constexpr float inv = 1.0f / (float)H;
    if (valid) {
#pragma unroll
        for (int i = 0; i < 8; i++) dst[tid + 512 * i] = cscale(v[i], inv);
    }
}

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <bool HALF, int TK, bool OUT_U8 = false>
__global__ void __launch_bounds__(512, 4) k_c2r_sharpen_v(FusedParams p)
{
    constexpr int UW = 4096, T = 512, NI = 2, KH = 1024, NB0 = 512, U = 2;
    constexpr float inv = 0.5f / (float)UW;      // the spectrum rows carry twice the reference's scale (k_col_t)
    using LP = typename std::conditional<HALF, h2v, f2v>::type;        // one L element: rows (a, a+1) at one x
    constexpr unsigned ES = sizeof(LP);
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* zb = smem + 4096 * ES;
    float* red = (float*)zb;                     // [0..15] corner partial sums, [16] corner DC term (strip start only)
    int lt = threadIdx.x;
    const int uH = p.uH;
    const int pairs_per_plane = uH / 2;
    const long plane = (long)UW * uH;
    VTwid tws;
    vfft_load_tw(tws, p.tw, lt);

    int f0 = blockIdx.x * p.pairs_per_strip;
    const int f1 = min(f0 + p.pairs_per_strip, 3 * pairs_per_plane);
    while (f0 < f1) {
        const int c = f0 / pairs_per_plane;
        const int j0 = f0 - c * pairs_per_plane;
        const int j1 = min(j0 + (f1 - f0), pairs_per_plane);
        f0 += j1 - j0;
        const int y0 = 2 * j0, y1 = 2 * j1;
        const bool top = (y0 == 0);
        const int a0 = top ? 0 : y0 - 1;
        const int npairs = (j1 - j0) + 1;
        const unsigned tile_stride32 = (unsigned)(uH / U) * TK;
        const float2* base = p.S1 + (long)c * p.NT * (long)tile_stride32;
        auto koff = [&](int k) -> unsigned {
            return (__umul24((unsigned)k / TK, tile_stride32) + ((unsigned)k % TK)) * (unsigned)sizeof(float2);
        };
        typedef const __attribute__((address_space(1))) char* gptr_t;
        auto rowbase = [&](int row) -> gptr_t {
            const unsigned off = (((unsigned)row / U) * TK + ((unsigned)row % U) * p.odd_delta) * (unsigned)sizeof(float2);
            gptr_t r = (gptr_t)base + __builtin_amdgcn_readfirstlane(off);
            asm("" : "+s"(r));
            return r;
        };
        auto gload = [](gptr_t r, unsigned off) -> float2 {
            asm("" : "+v"(off));
            const lds_f2raw t = *(const __attribute__((address_space(1))) lds_f2raw*)(r + off);
            return make_float2(t.x, t.y);
        };
        auto S2at = [&](int k, int row) -> float2 { return gload(rowbase(row), koff(k)); };
        auto dc_im = [&](int row) -> float { return *(const __attribute__((address_space(1))) float*)(rowbase(row) + 4); };
        const bool need_corner = !top && (y1 + 1 < uH);
        const int rs = y1 + 1;

        struct In { float2 a[NI], am[NI], b[NI], bm[NI]; float lka, lkb; };
        unsigned ko[NI], kom[NI];
        {
            const int jj = ((lt & 63) >> 3) + 8 * (lt & 7) + 64 * (lt >> 6);      // first-stage butterfly of this thread
#pragma unroll
            for (int m = 0; m < NI; m++) { ko[m] = koff(jj + NB0 * m); kom[m] = koff(KH - jj - NB0 * m); }
        }
        auto load_pair = [&](int i) -> In {
            In in;
            const int a = a0 + 2 * i;
            const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
            const gptr_t ra = rowbase(ya), rb = rowbase(yb);
#pragma unroll
            for (int m = 0; m < NI; m++) {
                in.a[m] = gload(ra, ko[m]); in.am[m] = gload(ra, kom[m]);
                in.b[m] = gload(rb, ko[m]); in.bm[m] = gload(rb, kom[m]);
            }
            in.lka = dc_im(ya ^ 1);
            in.lkb = dc_im(yb ^ 1);
            return in;
        };
        // nst = output stores issued behind the prefetch (the counter is in order: the loads are done when at most nst
        // operations are outstanding)
        auto settle = [](In& in, int nst) {
            switch (nst) {
            case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
            default: __builtin_amdgcn_s_waitcnt(0x0F70); break;             // vmcnt(0), nothing else
            }
#pragma unroll
            for (int m = 0; m < NI; m++)
                asm volatile("" : "+v"(in.a[m].x), "+v"(in.a[m].y), "+v"(in.am[m].x), "+v"(in.am[m].y), "+v"(in.b[m].x), "+v"(in.b[m].y),
                                  "+v"(in.bm[m].x), "+v"(in.bm[m].y));
            asm volatile("" : "+v"(in.lka), "+v"(in.lkb));
        };

        if (need_corner) {
            float part = 0.f;
            for (int kk = lt + 1; kk <= KH; kk += T) part += S2at(kk, rs).x;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
            if ((lt & 63) == 0) red[lt >> 6] = part;
            if (lt == T - 1) {
                float2 d = S2at(0, rs), dp = S2at(0, rs ^ 1);
                red[16] = (rs & 1) ? d.x + dp.y : d.x - dp.y;
            }
        }
        // first-stage inputs of pair i from its prefetched elements: A + i B at k = jj + 512 m, the conjugate combination at
        // the mirror partners (vkFFT.h:2096-2131); thread 0: the Nyquist element and the DC terms incl. the pair leak (B3)
        auto form = [&](const In& in, int i, float2 (&v)[8]) __attribute__((always_inline)) {
            const int a = a0 + 2 * i;
#pragma unroll
            for (int m = 0; m < 8; m++) v[m] = make_float2(0.f, 0.f);
#pragma unroll
            for (int m = 0; m < NI; m++) {
                v[m] = cadd_i(in.a[m], in.b[m]);
                v[8 - NI + m] = cadd_conj_i(in.am[m], in.bm[m]);
            }
            if (lt == 0) {
                v[NI] = make_float2(in.am[0].x - in.bm[0].y, in.am[0].y + in.bm[0].x);       // k = KH = W/2
                const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
                v[0] = make_float2(in.a[0].x + ((ya & 1) ? in.lka : -in.lka), in.b[0].x + ((yb & 1) ? in.lkb : -in.lkb));
            }
        };
        // (Requesting the elements of pair s+2 in step s right behind the transform -- before that step's output stores, the
        // formed inputs of pair s+1 waiting in registers -- moves the wait for the memory pipeline from the start of the step
        // to the middle of it and costs 5 us; the ten loads of a wave are 160 requests of 32 bytes, section 4 of DESIGN.md.)
        In in = load_pair(0);
        settle(in, 0);
        __syncthreads();            // red[] published; the previous segment's last reads of the L pairs are over
        float corner = 0.f;
        if (need_corner) {
            float sum = 0.f;
            for (int w2 = 0; w2 < T / 64; w2++) sum += red[w2];
            corner = (red[16] + 2.0f * sum) * inv;
        }
        __syncthreads();            // red[] lives in the exchange buffer: all reads before exchange A writes it
        float pn0 = 0.f, pn1 = 0.f;                         // thread T-1: L(a-3, UW-2 / UW-1), taps of the deferred pixel
        float lprev0 = 0.f;                                 // thread T-1: L(a-2, 0)
        LP P0[10];                                          // rows (a-2, a-1) at x = 8 lt - 1 .. 8 lt + 8
#pragma unroll
        for (int i = 0; i < 10; i++) P0[i] = LP{};
        // Output stores and the memory system (measured in round 3, profiles/r03_v_*): HBM takes ~14 bytes per clock and
        // compute unit when all units write, so a kilobyte store of every wave of a unit is a batch of ~1200 cycles.  Issued
        // together where the pixels are produced the batches make the wave wait to issue them and, behind them, the next
        // prefetch (fp32 planes: 2000 of a step's 10500 cycles; without the stores the kernel needs 52 instead of 64 us);
        // issued at four points of the next step's transform they stall the transform instead (70 us); and the two 16-byte
        // halves of a thread's 32 bytes of an fp32 row must go out back to back -- thousands of cycles apart every 128-byte
        // line is written twice, half each time (87 us).  So: binary16 planes (16 bytes per thread and row = whole kilobytes
        // per instruction) store row a-1 at the end of the sharpen phase and row a, from registers, in the next step's
        // transform; fp32 planes store both rows where they are produced.
        constexpr int NSV = (HALF && !OUT_U8) ? 1 : 0;      // deferred 16-byte stores per step
        f4t sv_val[NSV > 0 ? NSV : 1];
        long sv_off[1] = {0};                               // wave-uniform part of its address (bytes)
        bool sv_on[1] = {false};
        sv_val[0] = (f4t)(0.f);
        auto issue_store = [&](int k) __attribute__((always_inline)) {
            if constexpr (NSV == 1) {
                if (k == 1 && sv_on[0]) __builtin_nontemporal_store(sv_val[0], (f4t*)((char*)p.out + sv_off[0] + (unsigned)lt * 16u));
            }
        };

        for (int s = 0; s < npairs; s++) {
            const int a = a0 + 2 * s;
            // ================= transform of pair s
            asm volatile("" : "+v"(lt));
            float2 v[8];
            form(in, s, v);
            in = load_pair(min(s + 1, npairs - 1));                                 // lands during this step (last step: a harmless re-read)
            const int nst = (NSV == 1 && sv_on[0]) ? 1 : 0;                         // the deferred store is issued behind the prefetch
            vfft4096(v, zb, lt, tws, issue_store);
            settle(in, nst);
            {   // L pairs of x = w + 8 l + 512 q at element 512 w + l + 64 q
                const unsigned lw = lds_addr(smem) + ES * (512u * ((unsigned)lt >> 6) + ((unsigned)lt & 63u));
                if constexpr (HALF) {
                    const h2v up2 = h2_splat(p.upsq), one2 = h2_splat(1.0f);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const f2v sv = mk2(v[q].x, v[q].y) * mk2(inv, inv);
                        const h2v g = {(_Float16)sv.x, (_Float16)sv.y};
                        const h2v Lv = __builtin_elementwise_min(h2_bits(bits_h2(up2 * g) & 0x7fff7fffu), one2);
                        *(__attribute__((address_space(3))) h2v*)(size_t)(lw + ES * 64u * q) = Lv;
                    }
                } else {
                    const float ks = inv * p.upsq;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const f2v sv = mk2(v[q].x, v[q].y) * mk2(ks, ks);
                        const lds_f2raw r = {absmin1(sv.x), absmin1(sv.y)};
                        *(lds_f2*)(size_t)(lw + ES * 64u * q) = r;
                    }
                }
            }
            __syncthreads();                                                        // L pairs of rows a, a+1 visible
            // ================= sharpen rows a-1 and a: thread lt owns x = 8 lt .. 8 lt + 7
            LP P1[10];
            {
                const unsigned cb = lds_addr(smem) + ES * (unsigned)lt;
                typedef __attribute__((address_space(3))) LP lds_lp;
#pragma unroll
                for (int m = 0; m < 8; m++) P1[m + 1] = *(const lds_lp*)(size_t)(cb + ES * 512u * m);
                P1[0] = *(const lds_lp*)(size_t)(cb + ES * (7u * 512u - 1u));          // x = 8 lt - 1 (lt = 0: replaced below)
                P1[9] = *(const lds_lp*)(size_t)(cb + ES);                              // x = 8 lt + 8 (lt = T-1: replaced below)
            }
            const LP la = *(const __attribute__((address_space(3))) LP*)(size_t)lds_addr(smem);     // L(a, 0), L(a+1, 0): broadcast
            const float la0 = (float)la.x, la1 = (float)la.y;
            float lse = la1;
            {
                const int r2 = min(a + 2, uH - 1) - a;
                if (r2 == 0) lse = la0;
                else if (r2 > 1 && s == npairs - 1) lse = to_L<HALF>(corner, p.upsq);
            }
            if (lt == 0) P1[0] = P1[1];                                                 // id_x_m clamp (VkResample.cpp:889)
            if (lt == T - 1) {                                                          // x = UW wraps into the next row (quirk B5)
                P1[9].x = la.y;
                if constexpr (HALF) P1[9].y = (_Float16)lse; else P1[9].y = lse;
                if (a != 0) P0[9].y = la.x;                                             // L(a-1, UW) = L(a, 0), known now
            }
            if (a == 0) {                                                               // row -1 clamps to row 0 (top strip, first step)
                asm volatile("");
#pragma unroll
                for (int i = 0; i < 10; i++) { P0[i].x = 0; P0[i].y = P1[i].x; }
            }
            const bool out0 = (a - 1) >= y0 && (a - 1) < y1;                        // row y = a-1
            const bool out1 = a >= y0 && a < y1;                                    // row y = a
            // the pixel deferred by the previous step, (a-2, UW-1), from registers
            if (lt == T - 1) {
                const float m0 = (float)P0[7].x, m1 = (float)P0[8].x, me = (float)P0[9].x;      // row a-2: UW-2, UW-1 | L(a-1, 0)
                const float s0 = (float)P0[7].y, s1 = (float)P0[8].y;                           // row a-1
                if (s > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1)
                    deferred_pixel<HALF, OUT_U8>(p, OUT_U8 ? ((long)(a - 2) * UW + (UW - 1)) * 3 + c : c * plane + (long)(a - 2) * UW + (UW - 1),
                                                 pn0, pn1, (a - 2 == 0) ? me : lprev0, m0, m1, me, s0, s1, la0);
                if (a == 0) { pn0 = (float)P1[7].x; pn1 = (float)P1[8].x; }
                else { pn0 = s0; pn1 = s1; }
                lprev0 = la0;
            }
            if constexpr (HALF) {
#pragma clang fp contract(off)
                const h2v ncoef = h2_splat(-p.coef);
                h2v C[10], vmn[10], vmx[10];
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    C[i] = h2_bits(__builtin_amdgcn_alignbit(bits_h2(P1[i]), bits_h2(P0[i]), 16));      // rows (a-1, a)
                    vmn[i] = pk_min3(P0[i], C[i], P1[i]);
                    vmx[i] = pk_max3(P0[i], C[i], P1[i]);
                }
                h2v o[8];
#pragma unroll
                for (int i = 1; i <= 8; i++) {
                    const h2v mn1 = pk_min3(vmn[i - 1], vmn[i], vmn[i + 1]), mx1 = pk_max3(vmx[i - 1], vmx[i], vmx[i + 1]);
                    const h2v mn0 = pk_min3(vmn[i], C[i - 1], C[i + 1]), mx0 = pk_max3(vmx[i], C[i - 1], C[i + 1]);
                    o[i - 1] = sharpen_eval_pair_half(P0[i], P1[i], C[i - 1], C[i + 1], C[i], mn0, mn1, mx0, mx1, ncoef);
                }
#pragma unroll
                for (int wr = 0; wr < 2; wr++) {
                    const bool on = wr == 0 ? out0 : out1;
                    if constexpr (OUT_U8) {
                        if (!on) continue;
                        uint8_t* d8 = (uint8_t*)p.out + ((long)(a - 1 + wr) * UW * 3 + c) + (unsigned)lt * 24u;
                        uint8_t q0[4], q1[4];
                        cvt4_f_u8((float)o[0][wr], (float)o[1][wr], (float)o[2][wr], (float)o[3][wr], p.u8_wrap, q0);
                        cvt4_f_u8((float)o[4][wr], (float)o[5][wr], (float)o[6][wr], (float)o[7][wr], p.u8_wrap, q1);
#pragma unroll
                        for (int i = 0; i < 4; i++) { d8[3 * i] = q0[i]; d8[12 + 3 * i] = q1[i]; }
                    } else {
                        const long off = (c * plane + (long)(a - 1 + wr) * UW) * 2;       // wave-uniform, bytes
                        h2v r01 = {o[0][wr], o[1][wr]}, r23 = {o[2][wr], o[3][wr]}, r45 = {o[4][wr], o[5][wr]}, r67 = {o[6][wr], o[7][wr]};
                        const f4t val = {__builtin_bit_cast(float, r01), __builtin_bit_cast(float, r23), __builtin_bit_cast(float, r45), __builtin_bit_cast(float, r67)};
                        if (wr == 0) { if (on) __builtin_nontemporal_store(val, (f4t*)((char*)p.out + off + (unsigned)lt * 16u)); }
                        else { sv_val[0] = val; sv_off[0] = off; sv_on[0] = on; }
                    }
                }
            } else {
                f2v C[10], vmn[10], vmx[10];
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    C[i] = mk2(P0[i].y, P1[i].x);                                                       // rows (a-1, a)
                    vmn[i] = mk2(min3f(P0[i].x, P0[i].y, P1[i].x), min3f(P0[i].y, P1[i].x, P1[i].y));
                    vmx[i] = mk2(max3f(P0[i].x, P0[i].y, P1[i].x), max3f(P0[i].y, P1[i].x, P1[i].y));
                }
                f2v o[8];
                auto eval = [&](int i) __attribute__((always_inline)) {
                    const f2v mn1 = mk2(min3f(vmn[i - 1].x, vmn[i].x, vmn[i + 1].x), min3f(vmn[i - 1].y, vmn[i].y, vmn[i + 1].y));
                    const f2v mx1 = mk2(max3f(vmx[i - 1].x, vmx[i].x, vmx[i + 1].x), max3f(vmx[i - 1].y, vmx[i].y, vmx[i + 1].y));
                    const f2v mn0 = mk2(min3f(vmn[i].x, C[i - 1].x, C[i + 1].x), min3f(vmn[i].y, C[i - 1].y, C[i + 1].y));
                    const f2v mx0 = mk2(max3f(vmx[i].x, C[i - 1].x, C[i + 1].x), max3f(vmx[i].y, C[i - 1].y, C[i + 1].y));
                    o[i - 1] = sharpen_eval_pair(P0[i], P1[i], C[i - 1] + C[i + 1], C[i], mn0, mn1, mx0, mx1, p.coef);
                };
#pragma unroll
                for (int i = 1; i <= 8; i++) eval(i);
#pragma unroll
                for (int wr = 0; wr < 2; wr++) {
                    if (wr == 0 ? !out0 : !out1) continue;
                    if constexpr (OUT_U8) {
                        uint8_t* d8 = (uint8_t*)p.out + ((long)(a - 1 + wr) * UW * 3 + c) + (unsigned)lt * 24u;
                        uint8_t q0[4], q1[4];
                        cvt4_f_u8(o[0][wr], o[1][wr], o[2][wr], o[3][wr], p.u8_wrap, q0);
                        cvt4_f_u8(o[4][wr], o[5][wr], o[6][wr], o[7][wr], p.u8_wrap, q1);
#pragma unroll
                        for (int i = 0; i < 4; i++) { d8[3 * i] = q0[i]; d8[12 + 3 * i] = q1[i]; }
                    } else {
                        char* dst = (char*)p.out + (c * plane + (long)(a - 1 + wr) * UW) * 4 + (unsigned)lt * 32u;
                        const f4t lo = {o[0][wr], o[1][wr], o[2][wr], o[3][wr]}, hi = {o[4][wr], o[5][wr], o[6][wr], o[7][wr]};
                        __builtin_nontemporal_store(lo, (f4t*)dst);
                        __builtin_nontemporal_store(hi, (f4t*)(dst + 16));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 10; i++) P0[i] = P1[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) issue_store(k);         // the last step's rows
    }
}

// =================================================================================== -p 2, 8-bit RGB out: all three planes per strip
// FFTUP_FLAG_FUSE_U8_STORE with k_c2r_sharpen_g / k_c2r_sharpen_v stores one colour plane per workgroup: byte stores at stride
// 3, the three bytes of a pixel written by three workgroups at three times -- every 64-byte piece of the image is written
// three times, partially (WRITE_SIZE 77 MB for a 25 MB image, profiles/r03_z_pmc_summary_fp16u8_u8store.txt), and the kernel
// is slower than the one that writes 50 MB of binary16 planes.  Here a strip owns its row pairs in ALL THREE planes: per
// step three transforms + three sharpen passes (the vertical-pair kernel needs ten saved registers per plane), the 8-bit
// values of the three planes meet in registers, are interleaved with v_perm_b32 and leave as 24 contiguous bytes per thread
// and row: the image is written once.  (Cost: a strip has four row pairs instead of twelve behind its one halo pair.)
__device__ __forceinline__ constexpr unsigned rgb_sel_rg(int j)      // selector of perm(G, R): output dword j of 12 interleaved bytes
{
    unsigned s = 0;
    for (int i = 0; i < 4; i++) {
        const int pos = 4 * j + i, px = pos / 3, ch = pos % 3;
        s |= (unsigned)(ch == 0 ? px : ch == 1 ? 4 + px : 0) << (8 * i);
    }
    return s;
}
__device__ __forceinline__ constexpr unsigned rgb_sel_b(int j)       // selector of perm(B, t): B bytes into their places, the rest passes
{
    unsigned s = 0;
    for (int i = 0; i < 4; i++) {
        const int pos = 4 * j + i, px = pos / 3, ch = pos % 3;
        s |= (unsigned)(ch == 2 ? 4 + px : i) << (8 * i);
    }
    return s;
}
// four pixels: r, g, b = four bytes each (pixel k in byte k) -> 12 interleaved bytes
__device__ __forceinline__ void rgb_interleave4(unsigned r, unsigned g, unsigned b, unsigned (&o)[3])
{
#pragma unroll
    for (int j = 0; j < 3; j++) o[j] = __builtin_amdgcn_perm(b, __builtin_amdgcn_perm(g, r, rgb_sel_rg(j)), rgb_sel_b(j));
}

template <int TK>
__global__ void __launch_bounds__(512, 3) k_c2r_sharpen_v_rgb8(FusedParams p)
{
    constexpr int UW = 4096, T = 512, NI = 2, KH = 1024, NB0 = 512, U = 2;
    constexpr float inv = 0.5f / (float)UW;
    constexpr unsigned ES = sizeof(h2v);
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* zb = smem + 4096 * ES;
    float* red = (float*)zb;                     // [c][0..15] corner partial sums, [c][16] corner DC term (strip start only)
    int lt = threadIdx.x;
    const int uH = p.uH;
    const int pairs_per_plane = uH / 2;
    VTwid tws;
    vfft_load_tw(tws, p.tw, lt);
    // the strip: row pairs [j0, j1) of every plane
    const int j0 = blockIdx.x * p.pairs_per_strip;
    const int j1 = min(j0 + p.pairs_per_strip, pairs_per_plane);
    if (j0 >= j1) return;
    const int y0 = 2 * j0, y1 = 2 * j1;
    const bool top = (y0 == 0);
    const int a0 = top ? 0 : y0 - 1;
    const int npairs = (j1 - j0) + 1;
    const unsigned tile_stride32 = (unsigned)(uH / U) * TK;
    auto koff = [&](int k) -> unsigned {
        return (__umul24((unsigned)k / TK, tile_stride32) + ((unsigned)k % TK)) * (unsigned)sizeof(float2);
    };
    typedef const __attribute__((address_space(1))) char* gptr_t;
    auto rowbase = [&](int c, int row) -> gptr_t {
        const float2* base = p.S1 + (long)c * p.NT * (long)tile_stride32;
        const unsigned off = (((unsigned)row / U) * TK + ((unsigned)row % U) * p.odd_delta) * (unsigned)sizeof(float2);
        gptr_t r = (gptr_t)base + __builtin_amdgcn_readfirstlane(off);
        asm("" : "+s"(r));
        return r;
    };
    auto gload = [](gptr_t r, unsigned off) -> float2 {
        asm("" : "+v"(off));
        const lds_f2raw t = *(const __attribute__((address_space(1))) lds_f2raw*)(r + off);
        return make_float2(t.x, t.y);
    };
    const bool need_corner = !top && (y1 + 1 < uH);
    const int rs = y1 + 1;
    struct In { float2 a[NI], am[NI], b[NI], bm[NI]; float lka, lkb; };
    unsigned ko[NI], kom[NI];
    {
        const int jj = ((lt & 63) >> 3) + 8 * (lt & 7) + 64 * (lt >> 6);
#pragma unroll
        for (int m = 0; m < NI; m++) { ko[m] = koff(jj + NB0 * m); kom[m] = koff(KH - jj - NB0 * m); }
    }
    auto load_pair = [&](int i, int c) -> In {
        In in;
        const int a = a0 + 2 * i;
        const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
        const gptr_t ra = rowbase(c, ya), rb = rowbase(c, yb);
#pragma unroll
        for (int m = 0; m < NI; m++) {
            in.a[m] = gload(ra, ko[m]); in.am[m] = gload(ra, kom[m]);
            in.b[m] = gload(rb, ko[m]); in.bm[m] = gload(rb, kom[m]);
        }
        in.lka = *(const __attribute__((address_space(1))) float*)(rowbase(c, ya ^ 1) + 4);
        in.lkb = *(const __attribute__((address_space(1))) float*)(rowbase(c, yb ^ 1) + 4);
        return in;
    };
    auto settle = [](In& in) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int m = 0; m < NI; m++)
            asm volatile("" : "+v"(in.a[m].x), "+v"(in.a[m].y), "+v"(in.am[m].x), "+v"(in.am[m].y), "+v"(in.b[m].x), "+v"(in.b[m].y),
                              "+v"(in.bm[m].x), "+v"(in.bm[m].y));
        asm volatile("" : "+v"(in.lka), "+v"(in.lkb));
    };
    // corner samples L(y1 + 1, 0) of the three planes (SE tap of the strip's last pixel), as k_c2r_sharpen_g forms them
    float corner[3] = {0.f, 0.f, 0.f};
    if (need_corner) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float part = 0.f;
            for (int kk = lt + 1; kk <= KH; kk += T) part += gload(rowbase(c, rs), koff(kk)).x;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
            if ((lt & 63) == 0) red[32 * c + (lt >> 6)] = part;
            if (lt == T - 1) {
                const float2 d = gload(rowbase(c, rs), koff(0)), dp = gload(rowbase(c, rs ^ 1), koff(0));
                red[32 * c + 16] = (rs & 1) ? d.x + dp.y : d.x - dp.y;
            }
        }
    }
    In in = load_pair(0, 0);
    settle(in);
    __syncthreads();
    if (need_corner) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float sum = 0.f;
            for (int w2 = 0; w2 < T / 64; w2++) sum += red[32 * c + w2];
            corner[c] = (red[32 * c + 16] + 2.0f * sum) * inv;
        }
    }
    __syncthreads();            // red[] lives in the exchange buffer
    float pn0[3] = {0.f, 0.f, 0.f}, pn1[3] = {0.f, 0.f, 0.f}, lprev0[3] = {0.f, 0.f, 0.f};      // thread T-1: taps of the deferred pixels
    h2v P0[3][10];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int i = 0; i < 10; i++) P0[c][i] = h2v{};
    auto nohook = [](int) {};

    for (int s = 0; s < npairs; s++) {
        const int a = a0 + 2 * s;
        const bool out0 = (a - 1) >= y0 && (a - 1) < y1, out1 = a >= y0 && a < y1;
        unsigned ob[3][2][2];                    // 8-bit values: [plane][row a-1 / a][pixels 0-3 / 4-7], pixel k in byte k
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // ================= transform of pair s, plane c
            asm volatile("" : "+v"(lt));
            float2 v[8];
#pragma unroll
            for (int m = 0; m < 8; m++) v[m] = make_float2(0.f, 0.f);
#pragma unroll
            for (int m = 0; m < NI; m++) {
                v[m] = cadd_i(in.a[m], in.b[m]);
                v[8 - NI + m] = cadd_conj_i(in.am[m], in.bm[m]);
            }
            if (lt == 0) {
                v[NI] = make_float2(in.am[0].x - in.bm[0].y, in.am[0].y + in.bm[0].x);
                const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
                v[0] = make_float2(in.a[0].x + ((ya & 1) ? in.lka : -in.lka), in.b[0].x + ((yb & 1) ? in.lkb : -in.lkb));
            }
            in = (c < 2) ? load_pair(s, c + 1) : load_pair(min(s + 1, npairs - 1), 0);
            vfft4096(v, zb, lt, tws, nohook);
            settle(in);
            {
                const unsigned lw = lds_addr(smem) + ES * (512u * ((unsigned)lt >> 6) + ((unsigned)lt & 63u));
                const h2v up2 = h2_splat(p.upsq), one2 = h2_splat(1.0f);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const f2v sv = mk2(v[q].x, v[q].y) * mk2(inv, inv);
                    const h2v g = {(_Float16)sv.x, (_Float16)sv.y};
                    const h2v Lv = __builtin_elementwise_min(h2_bits(bits_h2(up2 * g) & 0x7fff7fffu), one2);
                    *(__attribute__((address_space(3))) h2v*)(size_t)(lw + ES * 64u * q) = Lv;
                }
            }
            __syncthreads();
            // ================= sharpen rows a-1 and a of plane c
            h2v P1[10];
            {
                const unsigned cb = lds_addr(smem) + ES * (unsigned)lt;
                typedef __attribute__((address_space(3))) h2v lds_lp;
#pragma unroll
                for (int m = 0; m < 8; m++) P1[m + 1] = *(const lds_lp*)(size_t)(cb + ES * 512u * m);
                P1[0] = *(const lds_lp*)(size_t)(cb + ES * (7u * 512u - 1u));
                P1[9] = *(const lds_lp*)(size_t)(cb + ES);
            }
            const h2v la = *(const __attribute__((address_space(3))) h2v*)(size_t)lds_addr(smem);
            const float la0 = (float)la.x, la1 = (float)la.y;
            float lse = la1;
            {
                const int r2 = min(a + 2, uH - 1) - a;
                if (r2 == 0) lse = la0;
                else if (r2 > 1 && s == npairs - 1) lse = to_L<true>(corner[c], p.upsq);
            }
            if (lt == 0) P1[0] = P1[1];
            if (lt == T - 1) {
                P1[9].x = la.y;
                P1[9].y = (_Float16)lse;
                if (a != 0) P0[c][9].y = la.x;
            }
            if (a == 0) {
                asm volatile("");
#pragma unroll
                for (int i = 0; i < 10; i++) { P0[c][i].x = 0; P0[c][i].y = P1[i].x; }
            }
            if (lt == T - 1) {
                const float m0 = (float)P0[c][7].x, m1 = (float)P0[c][8].x, me = (float)P0[c][9].x;
                const float s0 = (float)P0[c][7].y, s1 = (float)P0[c][8].y;
                if (s > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1)
                    deferred_pixel<true, true>(p, ((long)(a - 2) * UW + (UW - 1)) * 3 + c, pn0[c], pn1[c], (a - 2 == 0) ? me : lprev0[c], m0, m1, me, s0, s1, la0);
                if (a == 0) { pn0[c] = (float)P1[7].x; pn1[c] = (float)P1[8].x; }
                else { pn0[c] = s0; pn1[c] = s1; }
                lprev0[c] = la0;
            }
            ob[c][0][0] = ob[c][0][1] = ob[c][1][0] = ob[c][1][1] = 0u;
            if (out0 || out1) {                  // (the halo step only fills the saved rows)
#pragma clang fp contract(off)
                const h2v ncoef = h2_splat(-p.coef);
                h2v C[10], vmn[10], vmx[10];
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    C[i] = h2_bits(__builtin_amdgcn_alignbit(bits_h2(P1[i]), bits_h2(P0[c][i]), 16));
                    vmn[i] = pk_min3(P0[c][i], C[i], P1[i]);
                    vmx[i] = pk_max3(P0[c][i], C[i], P1[i]);
                }
                h2v o[8];
#pragma unroll
                for (int i = 1; i <= 8; i++) {
                    const h2v mn1 = pk_min3(vmn[i - 1], vmn[i], vmn[i + 1]), mx1 = pk_max3(vmx[i - 1], vmx[i], vmx[i + 1]);
                    const h2v mn0 = pk_min3(vmn[i], C[i - 1], C[i + 1]), mx0 = pk_max3(vmx[i], C[i - 1], C[i + 1]);
                    o[i - 1] = sharpen_eval_pair_half(P0[c][i], P1[i], C[i - 1], C[i + 1], C[i], mn0, mn1, mx0, mx1, ncoef);
                }
#pragma unroll
                for (int wr = 0; wr < 2; wr++) {
                    uint8_t q0[4], q1[4];
                    cvt4_f_u8((float)o[0][wr], (float)o[1][wr], (float)o[2][wr], (float)o[3][wr], p.u8_wrap, q0);
                    cvt4_f_u8((float)o[4][wr], (float)o[5][wr], (float)o[6][wr], (float)o[7][wr], p.u8_wrap, q1);
                    ob[c][wr][0] = q0[0] | (q0[1] << 8) | (q0[2] << 16) | ((unsigned)q0[3] << 24);
                    ob[c][wr][1] = q1[0] | (q1[1] << 8) | (q1[2] << 16) | ((unsigned)q1[3] << 24);
                }
            }
#pragma unroll
            for (int i = 0; i < 10; i++) P0[c][i] = P1[i];
        }
        // ================= the three planes meet: 24 interleaved bytes per thread and row
#pragma unroll
        for (int wr = 0; wr < 2; wr++) {
            if (wr == 0 ? !out0 : !out1) continue;
            unsigned lo[3], hi[3];
            rgb_interleave4(ob[0][wr][0], ob[1][wr][0], ob[2][wr][0], lo);
            rgb_interleave4(ob[0][wr][1], ob[1][wr][1], ob[2][wr][1], hi);
            char* dst = (char*)p.out + (long)(a - 1 + wr) * UW * 3 + (unsigned)lt * 24u;
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            typedef u4v __attribute__((aligned(4))) u4v_u;
            typedef u2v __attribute__((aligned(4))) u2v_u;
            *(u4v_u*)dst = u4v{lo[0], lo[1], lo[2], hi[0]};
            *(u2v_u*)(dst + 16) = u2v{hi[1], hi[2]};
        }
    }
}

}  // namespace fftup
