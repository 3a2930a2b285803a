// kernels_dswap.hpp -- digit-swap transforms: the exchanges of a transform go through registers (v_permlane*_swap, DPP)
// wherever the two digits that trade places live in one wave, and through LDS once, where they do not.
//
// The Stockham kernels of kernels_pow2.hpp keep a constant geometry (thread p owns x[p + Tc i] before every stage), so every
// exchange moves data between all waves of the workgroup: LDS plus s_barrier, three times per 1024-point transform.  Here the
// index is a tuple of digits -- (register | wave, lane bits 5-3, lane bits 2-0) -- and each exchange swaps the register digit
// with ONE thread digit:
//       registers <-> wave            the only workgroup-wide exchange: LDS, one barrier; a lane's elements of one instruction
//                                     are contiguous -- no swizzle, immediates only; its twiddles are wave-uniform (scalar registers)
//       registers <-> lane bits 5-3   no LDS: v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4) -- new in gfx950, one
//                                     instruction per register pair -- and three row_ror:8 DPP moves per pair for bit 3
//       register bit <-> lane bit 2   DPP row_shr:4 / row_shl:4 with bank masks
// numpy replays of the index algebra: tests/test_digit_swap_models.py.
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

// registers <-> wave for WV < 8 waves (H = 512, 256): the wave digit has WV values against 8 registers, so a thread comes out of
// the exchange with the WV values of the wave digit x Q = 8 / WV of the eight first-stage outputs: element (k0, wave n_w, lane)
// at (k0 * WV + n_w) * 64 + lane; thread of wave w' reads register m = WV b + n_w from (8 w' + m) * 64 + lane, i.e. k0 = Q w' + b.
// FWD: registers k0 -> registers WV b + n_w; !FWD: the way back.  One barrier inside; the buffer must be free on entry.
template <int WV, bool FWD>
__device__ __forceinline__ void wave_exchange_q(float2 (&v)[8], unsigned zbase, unsigned w, unsigned l)
{
    const unsigned a_k0 = zbase + 8u * (w * 64u + l);         // + 512 WV k0: element (k0, wave w)
    const unsigned a_m = zbase + 8u * (w * 512u + l);         // + 512 m:     register m of wave w
#pragma unroll
    for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)((FWD ? a_k0 + 512u * WV * k : a_m + 512u * k)) = r; }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) { const lds_f2raw r = *(const lds_f2*)(size_t)((FWD ? a_m + 512u * m : a_k0 + 512u * WV * m)); v[m] = make_float2(r.x, r.y); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y));
}
// Q butterflies of radix WV on registers WV b .. WV b + WV - 1
template <int WV, int DIR> __device__ __forceinline__ void bfly_groups(float2 (&v)[8])
{
    if constexpr (WV == 8) bfly8_pk<DIR>(v);
    else if constexpr (WV == 4) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            pk2 a0 = {v[4 * b].x, v[4 * b].y}, a1 = {v[4 * b + 1].x, v[4 * b + 1].y}, a2 = {v[4 * b + 2].x, v[4 * b + 2].y}, a3 = {v[4 * b + 3].x, v[4 * b + 3].y};
            bfly4_pk<DIR>(a0, a1, a2, a3);
            v[4 * b] = make_float2(a0.x, a0.y); v[4 * b + 1] = make_float2(a1.x, a1.y);
            v[4 * b + 2] = make_float2(a2.x, a2.y); v[4 * b + 3] = make_float2(a3.x, a3.y);
        }
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const pk2 a = {v[2 * b].x, v[2 * b].y}, c = {v[2 * b + 1].x, v[2 * b + 1].y};
            const pk2 s0 = pk_add(a, c), s1 = pk_sub(a, c);
            v[2 * b] = make_float2(s0.x, s0.y); v[2 * b + 1] = make_float2(s1.x, s1.y);
        }
    }
}

// grid (NT, 3), one column tile per workgroup: all 771 workgroups of a 2048x1024 frame are resident at once.  (Round 4 measured one
// workgroup per tile column running the three planes' tiles one after the other, the next tile's loads issued before this tile's
// transform: 19.9 instead of 14.6 us -- two waves per SIMD do not hide their own latencies; profiles/r04_g_column_pipelined.txt.)
// H = 128 WV rows on 64 WV threads, WV = 8, 4, 2 waves (H = 1024, 512, 256): n = (H/8) r + 16 w + 2 g + h0 -- registers r, wave w, lane
// bits 5-3 g, lane bit 2 h0 (lane bits 1-0: the tile's column) -- and k = k0 + 8 k1 + 8 WV k2 + 64 WV k3 (k1: WV values).
template <int TK, int H = 1024>
__global__ void __launch_bounds__(H / 2, kColWaves) k_col_v(ColTParams p)
{
    static_assert(TK == 4, "four columns of H/8 threads: lane bits 0-1 = column, bits 2-5 and the wave = pp");
    static_assert(H == 1024 || H == 512 || H == 256, "8, 4 or 2 waves");
    constexpr int WV = H / 128, Q = 8 / WV, T = 64 * WV;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const unsigned zbase = lds_addr(smem);
    const int tid = threadIdx.x;
    const unsigned w = (unsigned)tid >> 6, l = (unsigned)tid & 63u;
    const int col = tid & 3, pp = tid >> 2;                   // pp = 16 w + h, h = l >> 2
    const int tile = blockIdx.x, c = blockIdx.y;
    const bool valid = tile * TK + col <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    const int wu = __builtin_amdgcn_readfirstlane((int)w);
    const int s1 = (int)(l >> 3), lb2 = (int)((l >> 2) & 1u), hh = (int)(l >> 2);
    const int kk = Q * wu + s1 / WV + 8 * (s1 % WV);          // k0 + 8 k1 of the thread once lane bits 5-3 hold s = WV b + k1
    const int kt = kk + 32 * WV * lb2;                        // the thread's part of k after the forward transform
    float2 v[8];
    stamp_begin(1, c);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = valid ? src[tid + T * i] : make_float2(0.f, 0.f);        // row pp + (H/8) i
    // ---- forward, exp(+2 pi i n k / H), decimation in time
    float2 ta[Q][WV - 1];                                     // exp(2 pi i n_w k0 / 8 WV), k0 = Q wave + b: wave-uniform, scalar registers
#pragma unroll
    for (int b = 0; b < Q; b++)
#pragma unroll
        for (int m = 1; m < WV; m++) ta[b][m - 1] = p.twH[(16 * m * (Q * wu + b)) & (H - 1)];
    const float2 tb = p.twH[2 * kk], tc = p.twH[kt];
    const float2 tph = twid<-1>(p.twUH[kt]);                                                   // exp(-2 pi i kt / 2H)
    bfly8_pk<+1>(v);
    if constexpr (WV == 8) wave_exchange(v, zbase, w, l);
    else wave_exchange_q<WV, true>(v, zbase, w, l);
#pragma unroll
    for (int b = 0; b < Q; b++)
#pragma unroll
        for (int m = 1; m < WV; m++) v[WV * b + m] = cmul_tw_s(v[WV * b + m], ta[b][m - 1]);
    bfly_groups<WV, +1>(v);
    lane_transpose_hi3(v);
    twiddle_powers<8>(v, tb);
    bfly8_pk<+1>(v);
    lane_swap_bit2(v);
    {   // radix 2 over the bit that came out of the lane: twiddle exp(2 pi i (kt + 8 WV r) / H) on the upper element
        v[4] = cmul_tw(v[4], tc);
        v[5] = cmul_tw(v[5], cmul_tw(tc, rot16<1>()));
        v[6] = cmul_tw(v[6], cmul_tw(tc, rot16<2>()));
        v[7] = cmul_tw(v[7], cmul_tw(tc, rot16<3>()));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const pk2 a = {v[r].x, v[r].y}, b = {v[r + 4].x, v[r + 4].y};
            const pk2 s0 = pk_add(a, b), s1_ = pk_sub(a, b);
            v[r] = make_float2(s0.x, s0.y); v[r + 4] = make_float2(s1_.x, s1_.y);
        }
    }
    // ---- phase: register r + 4 k3 holds F[k], k = kt + 8 WV r + 64 WV k3;  t[k] = exp(-2 pi i k / 2H) * (k < H/2 ? 1 : -1),
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
        const pk2 s0 = pk_add(a, b), s1_ = pk_sub(a, b);
        v[r] = make_float2(s0.x, s0.y); v[r + 4] = make_float2(s1_.x, s1_.y);
    }
    lane_swap_bit2(v);                                       // lane bit 2 = h0 now, registers = k2
    {
        const float2 e = twid<-1>(rot16<1>());               // exp(-2 pi i / 16)
        const float2 base = lb2 ? e : make_float2(1.f, 0.f); // exp(-2 pi i k2 h0 / 16)
        twiddle_powers<8>(v, base);
    }
    bfly8_pk<-1>(v);
    lane_transpose_hi3(v);                                   // lane bits 5-3 = g, registers = WV b + k1
    {
        const float2 w1 = twid<-1>(p.twH[8 * hh]);           // exp(-2 pi i k1 h / 16 WV)
        if constexpr (WV == 8) twiddle_powers<8>(v, w1);
        else {
#pragma unroll
            for (int b = 0; b < Q; b++) twiddle_powers<WV>(&v[WV * b], w1);
        }
    }
    bfly_groups<WV, -1>(v);                                  // registers = WV b + n_w
    __syncthreads();                                         // everybody has read the forward exchange
    if constexpr (WV == 8) wave_exchange(v, zbase, w, l);    // wave = 16s digit of pp, registers = k0
    else wave_exchange_q<WV, false>(v, zbase, w, l);
    twiddle_powers<8>(v, twid<-1>(p.twH[pp]));               // exp(-2 pi i k0 pp / H)
    bfly8_pk<-1>(v);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    constexpr float inv = 1.0f / (float)H;
    if (valid) {
        // (8-byte write-through stores, 512 contiguous bytes per wave and instruction.  Lane pairs trading elements for 16-byte stores --
        // DPP quad_perm, 32 more vector instructions per thread -- measured no faster: profiles/r06_c_wt_default.txt)
#pragma unroll
        for (int i = 0; i < 8; i++) spec_store8(dst + tid + T * i, cscale(v[i], inv));
    }
    stamp_end(1, c);
}

// (A digit-swap row kernel -- 2048 = 8 (registers) x 4 (waves) x 8 x 8 (lane bits), 3 barriers instead of 7, parity-green --
// was measured in round 4 and runs exactly as fast as k_row_r2c_t<2048>: that kernel sits at the memory system's rate for a
// 10-us launch; profiles/r04_f_row_digit_swap.txt.  Its index algebra stays in tests/test_digit_swap_models.py.)

}  // namespace fftup
