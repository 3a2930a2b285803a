// kernels_pow2.hpp -- size-specialised kernels (compile-time plans) for power-of-two images, u = 2.
//
// Same math as kernels_generic.hpp, restructured for gfx950:
//   * every thread keeps its E points of a transform in registers across all Stockham stages; LDS is
//     only the exchange medium (one padded buffer, in place), first-stage inputs come straight from
//     HBM and last-stage outputs go straight back;
//   * thread (p, col) of a transform of length N with Tc = N/E threads per sequence always owns
//     x[p + Tc*i], i < E, whatever the radix of the stage (radix R uses the E/R butterflies
//     {v[b + m*E/R]}), so gathers are one stride-Tc read per stage;
//   * column kernel: forward FFT(H), centred zero-pad/shift and inverse FFT(2H) fused; the zero rows
//     are never materialised (vkFFT.h:1670-1695 read guard, VkResample.cpp:514-526 shift).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "fft_engine.hpp"
#include "kernels_generic.hpp"




namespace fftup {

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n / 2); }
// radix of the stage that starts at sub-transform length Ns: the largest allowed one (RMAX = 8, or 16 for
// threads that own 16 points) the remaining length still holds
constexpr int stage_radix(int N, int Ns, int RMAX = 8) { return (N / Ns >= RMAX) ? RMAX : (N / Ns); }

// LDS element index of point idx of sequence col (TK interleaved sequences)
template <int TK> __device__ __forceinline__ int lidx(int idx, int col) { return lpad(idx * TK + col); }

// ---- twiddles.  Stage with Ns > 1 of butterfly j needs exp(DIR*2 pi i*m*k/(Ns*R)), k = j % Ns,
// m < R.  Every thread fetches ONE base twiddle per stage from the table, all of them up front
// (TwSet::load, issued next to the first-stage input loads so that no stage waits on memory), and
// forms the powers by multiplication (<= 3 roundings, ~2e-7).  A thread's butterflies b > 0 of one
// stage differ from b = 0 by a compile-time rotation (only in the last stage, where Ns > Tc).
constexpr int num_stages(int N, int RMAX = 8, int Ns = 1) { return Ns >= N ? 0 : 1 + num_stages(N, RMAX, Ns * stage_radix(N, Ns, RMAX)); }
constexpr int stage_ns(int N, int s, int RMAX = 8) { return s == 0 ? 1 : stage_ns(N, s - 1, RMAX) * stage_radix(N, stage_ns(N, s - 1, RMAX), RMAX); }

template <int N, int E, int RMAX = 8> struct TwSet {
    static constexpr int S = num_stages(N, RMAX);
    float2 w[S > 1 ? S - 1 : 1];                 // base twiddle of stages 1..S-1 (table sign: exp(+i..))
    template <int s> __device__ __forceinline__ void load_stage(const float2* __restrict__ tw, int p)
    {
        if constexpr (s < S) {
            constexpr int Ns = stage_ns(N, s, RMAX);
            constexpr int R = stage_radix(N, Ns, RMAX);
            constexpr int tstep = N / (Ns * R);
            w[s - 1] = tw[(p & (Ns - 1)) * tstep];
            load_stage<s + 1>(tw, p);
        }
    }
    __device__ __forceinline__ void load(const float2* __restrict__ tw, int p) { load_stage<1>(tw, p); }
};

// exp(2 pi i q/16), q = 0..15, as compile-time constants
constexpr float kCos16[16] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                              -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f,
                              -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.f,
                              0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
template <int Q> __device__ __forceinline__ float2 rot16()
{
    constexpr float cr = kCos16[Q & 15], ci = kCos16[(Q + 12) & 15];
    return make_float2(cr, ci);
}

template <int R> __device__ __forceinline__ void twiddle_powers(float2* v, float2 w1)
{
    if constexpr (R == 2) {
        v[1] = cmul(v[1], w1);
    } else if constexpr (R == 4) {
        float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
    } else {
        float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        float2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
        if constexpr (R == 16) {
            float2 w8 = cmul(w4, w4);
            v[8] = cmul(v[8], w8); v[9] = cmul(v[9], cmul(w8, w1)); v[10] = cmul(v[10], cmul(w5, w5));
            v[11] = cmul(v[11], cmul(w8, w3)); v[12] = cmul(v[12], cmul(w6, w6)); v[13] = cmul(v[13], cmul(w8, w5));
            v[14] = cmul(v[14], cmul(w7, w7)); v[15] = cmul(v[15], cmul(w8, w7));
        }
    }
}

// ---- one stage on registers: E/R butterflies of radix R (compile-time recursion over b)
template <int N, int E, int R, int Ns, int DIR, int B>
__device__ __forceinline__ void butterfly_b(float2 (&v)[E], float2 wbase)
{
    constexpr int Tc = N / E;
    constexpr int NB = E / R;
    if constexpr (B < NB) {
        float2 w[R];
#pragma unroll
        for (int m = 0; m < R; m++) w[m] = v[B + m * NB];
        if constexpr (Ns > 1) {
            float2 w1 = twid<DIR>(wbase);
            // last stage (Ns > Tc): k_b = p + b*Tc, i.e. an extra b/E of a revolution
            if constexpr (Ns > Tc && B > 0) w1 = cmul(w1, twid<DIR>(rot16<B * (16 / E)>()));
            twiddle_powers<R>(w, w1);
        }
        bfly<R, DIR>(w);
#pragma unroll
        for (int m = 0; m < R; m++) v[B + m * NB] = w[m];
        butterfly_b<N, E, R, Ns, DIR, B + 1>(v, wbase);
    }
}

template <int N, int E, int R, int Ns, int DIR>
__device__ __forceinline__ void reg_butterflies(float2 (&v)[E], float2 wbase)
{
    constexpr int Tc = N / E;
    static_assert(Ns <= Tc || Ns * R == N, "per-butterfly twiddle offsets are only derived for the last stage");
    butterfly_b<N, E, R, Ns, DIR, 0>(v, wbase);
}

// ---- Stockham autosort scatter of a stage's outputs into LDS
template <int N, int E, int R, int Ns, int TK>
__device__ __forceinline__ void reg_scatter(const float2 (&v)[E], float2* __restrict__ buf, int p, int col)
{
    constexpr int Tc = N / E;
    constexpr int NB = E / R;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int j = p + b * Tc;
        const int k = j & (Ns - 1);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; q++) buf[lidx<TK>(j0 + q * Ns, col)] = v[b + q * NB];
    }
}

template <int N, int E, int TK>
__device__ __forceinline__ void reg_gather(float2 (&v)[E], const float2* __restrict__ buf, int p, int col)
{
    constexpr int Tc = N / E;
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = buf[lidx<TK>(p + Tc * i, col)];
}

// ---- all stages.  On entry v[i] = x[p + Tc*i].  If FINAL_TO_LDS the result X is left in LDS in
// natural order (valid after the trailing barrier); otherwise v[i] = X[p + Tc*i] on return.
// `buf` must not be in use by anyone on entry (callers barrier before re-using it).
template <int N, int E, int DIR, int TK, bool FINAL_TO_LDS, int S = 0, int RMAX = 8>
__device__ __forceinline__ void reg_fft(float2 (&v)[E], float2* __restrict__ buf, int p, int col,
                                        const TwSet<N, E, RMAX>& tws)
{
    constexpr int Ns = stage_ns(N, S, RMAX);
    constexpr int R = stage_radix(N, Ns, RMAX);
    static_assert(E % R == 0, "radix must divide the per-thread point count");
    reg_butterflies<N, E, R, Ns, DIR>(v, tws.w[S > 0 ? S - 1 : 0]);
    constexpr bool last = (Ns * R == N);
    if constexpr (!last || FINAL_TO_LDS) {
        reg_scatter<N, E, R, Ns, TK>(v, buf, p, col);
        __syncthreads();
    }
    if constexpr (!last) {
        reg_gather<N, E, TK>(v, buf, p, col);
        __syncthreads();
        reg_fft<N, E, DIR, TK, FINAL_TO_LDS, S + 1, RMAX>(v, buf, p, col, tws);
    }
}

// =================================================================================== row R2C
struct RowR2CTParams {
    const void* in;
    float2* S1;
    const float2* tw;
    long in_row_stride, in_plane_stride;
    int H, NT;
};

template <int MODE> __device__ __forceinline__ float load_px_t(const RowR2CTParams& p, int c, int y, int x)
{
    if constexpr (MODE == IN_F32) return ((const float*)p.in)[c * p.in_plane_stride + y * p.in_row_stride + x];
    else if constexpr (MODE == IN_F16) return __half2float(((const __half*)p.in)[c * p.in_plane_stride + y * p.in_row_stride + x]);
    else if constexpr (MODE == IN_U8_F32) return cvt_u8_f32(((const uint8_t*)p.in)[y * p.in_row_stride + 3l * x + c]);
    else return cvt_u8_f16(((const uint8_t*)p.in)[y * p.in_row_stride + 3l * x + c]);
}

// grid (H/2, 3), block W/8.  LDS: lpad_size(W) float2.
template <int W, int MODE, int TK>
__global__ void __launch_bounds__(W / 8) k_row_r2c_t(RowR2CTParams p)
{
    constexpr int E = 8, T = W / E;
    __shared__ float2 buf[lpad_size(W)];
    const int tid = threadIdx.x, c = blockIdx.y;
    const int j = blockIdx.x;      // (an XCD-aware pair order -- pairs 2i, 2i+1 on one XCD -- measured no gain)
    float2 v[E];
    TwSet<W, E> tws;
    tws.load(p.tw, tid);
#pragma unroll
    for (int i = 0; i < E; i++)
        v[i] = make_float2(load_px_t<MODE>(p, c, 2 * j, tid + T * i), load_px_t<MODE>(p, c, 2 * j + 1, tid + T * i));
    reg_fft<W, E, +1, 1, true>(v, buf, tid, 0, tws);
    // unpack (vkFFT.h:4292-4323).  8 consecutive lanes cover one tile segment [A(TK)|B(TK)] of
    // 2*TK float2; each lane stores 16 bytes (two complex values).
    static_assert(TK == 4 || TK == 8, "tile width");
    constexpr int LPT = TK;                         // lanes per tile segment (2*TK complex / 2 per lane)
    const long tile_stride = (long)p.H * TK;
    float2* base = p.S1 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    constexpr int NTILE = (W / 2 + 1 + TK - 1) / TK;
    for (int g = tid; g < NTILE * LPT; g += T) {
        const int tile = g / LPT, l = g % LPT;
        const bool isB = l >= LPT / 2;
        const int kk = (l % (LPT / 2)) * 2;          // 0,2,.. within the tile
        float2 o[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int k = tile * TK + kk + e;
            float2 r = make_float2(0.f, 0.f);
            if (k <= W / 2) {
                float2 zk = buf[lpad(k)];
                float2 zn = buf[lpad((W - k) & (W - 1))];
                r = isB ? make_float2(0.5f * (zk.y + zn.y), 0.5f * (-zk.x + zn.x))
                        : make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            }
            o[e] = r;
        }
        float4* dst = (float4*)(base + (long)tile * tile_stride + (isB ? TK : 0) + kk);
        *dst = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
    }
}

// =================================================================================== column
struct ColTParams {
    const float2* S1;
    float2* S2;
    const float2 *twH, *twUH;
    int W, NT;
};

// grid (NT, 3), block TK*H/8.  Forward length H (E=8), inverse length 2H (E=16), both with H/8
// threads per column.  LDS: lpad_size(2H*TK) float2.
template <int H, int TK>
__global__ void __launch_bounds__(TK* H / 8) k_col_t(ColTParams p)
{
    constexpr int UH = 2 * H;
    constexpr int Tc = H / 8;                        // threads per column
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x;
    const int col = tid % TK, pp = tid / TK;
    const int tile = blockIdx.x, c = blockIdx.y;
    const bool valid = tile * TK + col <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    float2 v[8];
    TwSet<H, 8> twsF;
    TwSet<UH, 16, 16> twsI;                          // inverse: radix-16 stages (2H = 16*16*8 for H = 1024)
    twsF.load(p.twH, pp);
    twsI.load(p.twUH, pp);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = valid ? src[(pp + Tc * i) * TK + col] : make_float2(0.f, 0.f);
    reg_fft<H, 8, +1, TK, true>(v, buf, pp, col, twsF);           // F[ky] natural order in LDS
    // inverse input (shift VkResample.cpp:514-526 + zero-pad guard vkFFT.h:1670-1695, u = 2):
    //   G[ky'] = F[ky'] (ky' < H/2), F[ky' - H] (ky' >= 3H/2), 0 otherwise.
    // Thread owns G[pp + Tc*i], i < 16 (UH/16 = Tc): i<4 -> F[pp+Tc*i]; i>=12 -> F[pp+Tc*(i-8)].
    float2 g[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < 4) g[i] = buf[lidx<TK>(pp + Tc * i, col)];
        else if (i >= 12) g[i] = buf[lidx<TK>(pp + Tc * (i - 8), col)];
        else g[i] = make_float2(0.f, 0.f);
    }
    __syncthreads();
    reg_fft<UH, 16, -1, TK, false, 0, 16>(g, buf, pp, col, twsI);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * UH * TK;
    constexpr float inv = 1.0f / (float)UH;
    if (valid) {
#pragma unroll
        for (int i = 0; i < 16; i++) dst[(pp + Tc * i) * TK + col] = cscale(g[i], inv);
    }
}

// =================================================================================== row C2R
struct RowC2RTParams {
    const float2* S2;
    void* R;
    const float2* tw;
    int uH, NT;
};

// grid (uH/2, 3), block UW/8.  u = 2: kx = 0..UW/4 non-zero.  LDS: lpad_size(UW) float2.
template <int UW, bool HALF_OUT, int TK, bool WIDE>
__global__ void __launch_bounds__(UW / 8) k_row_c2r_t(RowC2RTParams p)
{
    constexpr int E = 8, T = UW / E;                 // T = UW/8; W/2 = UW/4 = 2T
    __shared__ float2 buf[lpad_size(UW)];
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    const long tile_stride = (long)p.uH * TK;
    const float2* base = p.S2 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    auto ldAB = [&](int k, float2& A, float2& B) {
        const float2* s = base + (long)(k / TK) * tile_stride + (k % TK);
        A = s[0];
        B = s[TK];
    };
    // thread owns Z[tid + T*i]: i=0,1 direct (k = tid, tid+T); i=2: k = 2T = W/2 only for tid 0;
    // i=3..5 zero; i=6: mirror of k' = 2T - tid; i=7: mirror of k' = T - tid  (vkFFT.h:2096-2106)
    float2 v[E];
    float2 A, B;
    TwSet<UW, E> tws;
    tws.load(p.tw, tid);
    ldAB(tid, A, B);
    v[0] = make_float2(A.x - B.y, A.y + B.x);        // tid 0: DC element, same formula (vkFFT.h:2110-2131)
    ldAB(tid + T, A, B);
    v[1] = make_float2(A.x - B.y, A.y + B.x);
    ldAB(2 * T - tid, A, B);                          // k' in (T, 2T]
    v[6] = make_float2(A.x + B.y, -A.y + B.x);
    v[2] = (tid == 0) ? make_float2(A.x - B.y, A.y + B.x) : make_float2(0.f, 0.f);
    ldAB(T - tid, A, B);                              // k' in (0, T]
    v[7] = make_float2(A.x + B.y, -A.y + B.x);
    v[3] = v[4] = v[5] = make_float2(0.f, 0.f);
    const long plane = (long)UW * p.uH;
    constexpr float inv = 1.0f / (float)UW;
    if constexpr (!WIDE) {
        reg_fft<UW, E, -1, 1, false>(v, buf, tid, 0, tws);
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int n = tid + T * i;
            if constexpr (HALF_OUT) {
                __half* R = (__half*)p.R + c * plane + (long)(2 * j) * UW;
                R[n] = __float2half_rn(v[i].x * inv);
                R[UW + n] = __float2half_rn(v[i].y * inv);
            } else {
                float* R = (float*)p.R + c * plane + (long)(2 * j) * UW;
                R[n] = v[i].x * inv;
                R[UW + n] = v[i].y * inv;
            }
        }
    } else {
        reg_fft<UW, E, -1, 1, true>(v, buf, tid, 0, tws);
        // natural order in LDS: each thread takes 4 consecutive points twice -> 16-byte stores
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int n0 = (tid + T * h) * 4;
            float2 z[4];
#pragma unroll
            for (int e = 0; e < 4; e++) z[e] = buf[lpad(n0 + e)];
            if constexpr (HALF_OUT) {
                __half* R = (__half*)p.R + c * plane + (long)(2 * j) * UW + n0;
                __half2 r0 = __floats2half2_rn(z[0].x * inv, z[1].x * inv), r1 = __floats2half2_rn(z[2].x * inv, z[3].x * inv);
                __half2 i0 = __floats2half2_rn(z[0].y * inv, z[1].y * inv), i1 = __floats2half2_rn(z[2].y * inv, z[3].y * inv);
                *(float2*)R = make_float2(*(float*)&r0, *(float*)&r1);
                *(float2*)(R + UW) = make_float2(*(float*)&i0, *(float*)&i1);
            } else {
                float* R = (float*)p.R + c * plane + (long)(2 * j) * UW + n0;
                *(float4*)R = make_float4(z[0].x * inv, z[1].x * inv, z[2].x * inv, z[3].x * inv);
                *(float4*)(R + UW) = make_float4(z[0].y * inv, z[1].y * inv, z[2].y * inv, z[3].y * inv);
            }
        }
    }
}

// lane i <- lane i-1 / lane i+1 of the wave (gfx9 DPP wave shifts); lane 0 / 63 keep `edge`
__device__ __forceinline__ float lane_from_below(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_from_above(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// =================================================================================== sharpen
// One thread = 4 consecutive pixels x RPT rows; a wave covers 256 pixels of a row; left/right
// neighbours come from the adjacent lanes (ds_bpermute), wave-edge lanes load them.
// Block (64, 4); grid (uW/256, uH/(4*RPT), 3).  Requires uW % 256 == 0 and uH % (4*RPT) == 0.
template <bool HALF> struct PxRow {
    float L[6];     // L[0] = left neighbour, L[1..4] = own pixels, L[5] = right neighbour
};

template <bool HALF>
__device__ __forceinline__ void sharpen_load_row(PxRow<HALF>& r, const void* Rp, long plane_off, long plane,
                                                 int uW, int row, int x0, int lane, float upsq)
{
    using A = Arith<HALF>;
    // rows past the end: same column of the last written row (see oracle); in-row part
    long f = (long)row * uW + x0;
    while (f >= plane) f -= uW;
    float t[4];
    if constexpr (HALF) {
        float2 raw = *(const float2*)((const __half*)Rp + plane_off + f);
        __half2 h0 = *(__half2*)&raw.x, h1 = *(__half2*)&raw.y;
        t[0] = __low2float(h0); t[1] = __high2float(h0); t[2] = __low2float(h1); t[3] = __high2float(h1);
    } else {
        float4 raw = *(const float4*)((const float*)Rp + plane_off + f);
        t[0] = raw.x; t[1] = raw.y; t[2] = raw.z; t[3] = raw.w;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) r.L[1 + i] = fminf(fmaxf(fabsf(A::r(upsq * t[i])), 0.0f), 1.0f);
    float left = lane_from_below(r.L[4], 0.f);
    float right = lane_from_above(r.L[1], 0.f);
    if (lane == 0) {
        if (x0 == 0) left = r.L[1];                   // id_x_m clamp (VkResample.cpp:889)
        else {
            float tv;
            if constexpr (HALF) tv = __half2float(((const __half*)Rp)[plane_off + f - 1]);
            else tv = ((const float*)Rp)[plane_off + f - 1];
            left = fminf(fmaxf(fabsf(A::r(upsq * tv)), 0.0f), 1.0f);
        }
    }
    if (lane == 63) {
        long fr = (long)row * uW + x0 + 4;            // x == uW wraps into the next row (quirk B5)
        while (fr >= plane) fr -= uW;
        float tv;
        if constexpr (HALF) tv = __half2float(((const __half*)Rp)[plane_off + fr]);
        else tv = ((const float*)Rp)[plane_off + fr];
        right = fminf(fmaxf(fabsf(A::r(upsq * tv)), 0.0f), 1.0f);
    }
    r.L[0] = left;
    r.L[5] = right;
}

// fp32: fast reciprocal/sqrt (<= 1 ulp each); half: exact per-operation rounding (bit-exact vs oracle)
template <bool HALF>
__device__ __forceinline__ float sharpen_eval(float N, float S, float Wv, float E, float C,
                                              float mn1, float mx1, float mn0, float mx0, float coef)
{
    if constexpr (HALF) {
        using A = Arith<true>;
        float minlen = A::r(0.5f * A::r(mn0 + mn1));
        float maxlen = A::r(0.5f * A::r(mx0 + mx1));
        minlen = A::r(__fdiv_rn(minlen, A::r(1.0f - minlen)));
        maxlen = A::r(__fdiv_rn(A::r(1.0f - maxlen), maxlen));
        float scale = (minlen < maxlen) ? minlen : maxlen;
        scale = A::r(-coef * A::r(__fsqrt_rn(scale)));
        float s4 = A::r(A::r(A::r(N + Wv) + E) + S);
        float num = A::r(C + A::r(scale * s4));
        float den = A::r(1.0f + A::r(scale * 4.0f));
        return A::r(__fdiv_rn(num, den));
    } else {
        float minlen = 0.5f * (mn0 + mn1);
        float maxlen = 0.5f * (mx0 + mx1);
        float a = minlen * __builtin_amdgcn_rcpf(1.0f - minlen);
        float b = (1.0f - maxlen) * __builtin_amdgcn_rcpf(maxlen);
        float scale = (a < b) ? a : b;
        scale = -coef * __builtin_amdgcn_sqrtf(scale);
        float s4 = ((N + Wv) + E) + S;
        return (C + scale * s4) * __builtin_amdgcn_rcpf(1.0f + scale * 4.0f);
    }
}

// fp32 fast form of VkResample.cpp:909-922.  a < b  <=>  mn + mx < 1 (both denominators positive), so
// one quotient n/d with d in [0.5,1] is formed; sqrt(n/d) = n * rsq(n*d).  2 transcendental ops/pixel.
__device__ __forceinline__ float sharpen_eval_fast(float s4, float C, float mn0, float mn1, float mx0, float mx1, float coef)
{
    const float mn = 0.5f * (mn0 + mn1), mx = 0.5f * (mx0 + mx1);
    const bool lo = (mn + mx) < 1.0f;
    const float n = lo ? mn : 1.0f - mx;
    const float d = lo ? 1.0f - mn : mx;
    const float scale = -coef * n * __builtin_amdgcn_rsqf(fmaxf(n * d, 1e-30f));
    return fmaf(scale, s4, C) * __builtin_amdgcn_rcpf(fmaf(scale, 4.0f, 1.0f));
}

struct SharpenTParams {
    const void* R;
    void* out;
    int uW, uH;
    float upsq, coef;
};

template <bool HALF, int RPT>
__global__ void __launch_bounds__(256) k_sharpen_t(SharpenTParams p)
{
    const int lane = threadIdx.x;
    const int x0 = (blockIdx.x * 64 + lane) * 4;
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * RPT;
    const int c = blockIdx.z;
    const int uW = p.uW;
    const long plane = (long)uW * p.uH;
    const long poff = c * plane;
    PxRow<HALF> ra, rb, rc;
    sharpen_load_row<HALF>(ra, p.R, poff, plane, uW, y0 > 0 ? y0 - 1 : 0, x0, lane, p.upsq);
    sharpen_load_row<HALF>(rb, p.R, poff, plane, uW, y0, x0, lane, p.upsq);
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const int y = y0 + r;
        sharpen_load_row<HALF>(rc, p.R, poff, plane, uW, y + 1, x0, lane, p.upsq);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float N = ra.L[1 + i], S = rc.L[1 + i], Wv = rb.L[i], C = rb.L[1 + i], E = rb.L[2 + i];
            // same values as the reference's nested min/max chains (min/max are exact and associative)
            float mn0 = fminf(fminf(N, S), fminf(fminf(Wv, C), E));
            float mx0 = fmaxf(fmaxf(N, S), fmaxf(fmaxf(Wv, C), E));
            float mn1 = fminf(mn0, fminf(fminf(ra.L[i], ra.L[2 + i]), fminf(rc.L[i], rc.L[2 + i])));
            float mx1 = fmaxf(mx0, fmaxf(fmaxf(ra.L[i], ra.L[2 + i]), fmaxf(rc.L[i], rc.L[2 + i])));
            // -p 2 keeps the exactly rounded binary16 sequence here (bit-exact against the oracle)
            if constexpr (HALF) o[i] = sharpen_eval<true>(N, S, Wv, E, C, mn1, mx1, mn0, mx0, p.coef);
            else o[i] = sharpen_eval_fast(((N + Wv) + E) + S, C, mn0, mn1, mx0, mx1, p.coef);
        }
        const long of = poff + (long)y * uW + x0;
        if constexpr (HALF) {
            __half2 h0 = __floats2half2_rn(o[0], o[1]), h1 = __floats2half2_rn(o[2], o[3]);
            typedef float f2v __attribute__((ext_vector_type(2)));
            f2v val = {*(float*)&h0, *(float*)&h1};
            __builtin_nontemporal_store(val, (f2v*)((__half*)p.out + of));
        } else {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v val = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(val, (f4v*)((float*)p.out + of));
        }
        ra = rb;
        rb = rc;
    }
}


// =================================================================================== fused C2R + sharpen
// A strip = output rows [y0,y1) of one plane (a strip that crosses a plane boundary is processed as two
// segments).  Its row pairs are transformed one after the other; the clamped |u^2 g| rows ("L rows") of a
// pair are parked in the LDS exchange buffer that just produced them, three such buffers rotate, and after
// each pair the two rows whose 3x3 neighbourhood is complete are sharpened and stored.  The pre-sharpen
// image never goes to HBM (the reference writes and re-reads it: tempBuffer, 2 x 100 MB per frame).
//
// Pairing: a strip needs rows y0-1 .. y1, so it pairs (y0-1,y0),(y0+1,y0+2),.. -- one pair more than
// it outputs.  The reference pairs (2j,2j+1) and its C2R leaks Im(DC column) between the two rows of a
// pair (vkFFT.h:2110-2131); with any other pairing the same result is obtained by
// adding that leak explicitly: row y gets -Im D[y+1] (y even) or +Im D[y-1] (y odd) on its DC term.
//
// Quirk B5 (VkResample.cpp:891-892): the right neighbour of x = uW-1 is x = 0 of the NEXT row, so pixel
// (y, uW-1) needs L(y+2, 0).  For the newest sharpened row that value arrives with the next pair: the
// pixel is finished one step later (placeholder first, then the final value, ordered by the step barriers);
// for the last row of a strip the one
// missing sample g[y1+1][0] = (sum over k of Z[k])/uW is evaluated directly from the spectrum row.
struct FusedParams {
    const float2* S2;
    void* out;               // dense [3][uH][uW] float / half
    const float2* tw;
    int uH, NT;
    int pairs_per_strip;
    float upsq, coef;
};

__host__ __device__ constexpr size_t fused_buf_bytes(int uw) { return (sizeof(float2) * lpad_size(uw) + 15) & ~(size_t)15; }

template <bool HALF> __device__ __forceinline__ float to_L(float g, float upsq)
{
    using A = Arith<HALF>;
    // C2R output is stored as half for -p 2 (vkFFT.h:7289-7290) before the sharpen shader scales it
    if constexpr (HALF) g = __half2float(__float2half_rn(g));
    return fminf(fmaxf(fabsf(A::r(upsq * g)), 0.0f), 1.0f);
}

// -p 2 form for the fused kernel: the reference evaluates this shader in float16_t (VkResample.cpp:823-826).
// Additions and multiplications use the native binary16 instructions (correctly rounded, identical to the
// oracle's per-operation rounding); the quotient is selected first (a < b <=> mn + mx < 1, as above) and formed
// from v_rcp_f32 plus one Newton step, the root from v_rsq_f32 plus one step, each rounded once to binary16:
// within one fp16 ulp of the exactly rounded sequence (which k_sharpen_t keeps, bit for bit) -- the Vulkan
// spec allows the reference's own fp16 division 2.5 ulp.
__device__ __forceinline__ float div_f32_newton(float a, float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return fmaf(fmaf(-q, b, a), r, q);
}
__device__ __forceinline__ float sharpen_eval_half_fast(float N, float S, float Wv, float E, float C,
                                                        float mn0, float mn1, float mx0, float mx1, float coef)
{
    const __half one = __float2half_rn(1.0f), hlf = __float2half_rn(0.5f);
    const __half mn = __hmul(hlf, __hadd(__float2half_rn(mn0), __float2half_rn(mn1)));
    const __half mx = __hmul(hlf, __hadd(__float2half_rn(mx0), __float2half_rn(mx1)));
    const bool lo = (__half2float(mn) + __half2float(mx)) < 1.0f;           // exact in fp32
    const float n = __half2float(lo ? mn : __hsub(one, mx));
    const float d = __half2float(lo ? __hsub(one, mn) : mx);                // in [0.5, 1]
    const float q = __half2float(__float2half_rn(div_f32_newton(n, d)));
    float rt = 0.f;
    if (q > 0.f) {
        const float rs = __builtin_amdgcn_rsqf(q);
        const float s0 = q * rs;
        rt = fmaf(fmaf(-s0, s0, q), 0.5f * rs, s0);
    }
    const __half scale = __hmul(__float2half_rn(-coef), __float2half_rn(rt));
    const __half s4 = __hadd(__hadd(__hadd(__float2half_rn(N), __float2half_rn(Wv)), __float2half_rn(E)), __float2half_rn(S));
    // the product must round on its own: keep the compiler from contracting it with the add into v_fma_f16
    __half prod = __hmul(scale, s4);
    asm volatile("" : "+v"(prod));
    const __half num = __hadd(__float2half_rn(C), prod);
    const __half den = __hadd(one, __hmul(scale, __float2half_rn(4.0f)));
    return __half2float(__float2half_rn(div_f32_newton(__half2float(num), __half2float(den))));
}

// 4 output pixels from 3 tap rows of 6 values each (t[r][0] = left neighbour .. t[r][5] = right)
template <bool HALF>
__device__ __forceinline__ void sharpen_quad(const float (&t)[3][6], float coef, float (&o)[4])
{
    float hmn[3][4], hmx[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hmn[r][i] = fminf(fminf(t[r][i], t[r][i + 1]), t[r][i + 2]);
            hmx[r][i] = fmaxf(fmaxf(t[r][i], t[r][i + 1]), t[r][i + 2]);
        }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float N = t[0][i + 1], S = t[2][i + 1], Wv = t[1][i], C = t[1][i + 1], E = t[1][i + 2];
        const float mn0 = fminf(fminf(N, S), hmn[1][i]);           // cross = N, S and the centre row triple
        const float mx0 = fmaxf(fmaxf(N, S), hmx[1][i]);
        const float mn1 = fminf(fminf(hmn[0][i], hmn[2][i]), mn0);  // full 3x3 (min/max are exact: any order)
        const float mx1 = fmaxf(fmaxf(hmx[0][i], hmx[2][i]), mx0);
        if constexpr (HALF) o[i] = sharpen_eval_half_fast(N, S, Wv, E, C, mn0, mn1, mx0, mx1, coef);
        else o[i] = sharpen_eval_fast(((N + Wv) + E) + S, C, mn0, mn1, mx0, mx1, coef);
    }
}

// ---------------------------------------------------------------------------------------------------
// The kernel: ONE workgroup of 2*T threads per compute unit in which the two halves swap roles every step: while one half transforms row pair s (LDS exchange buffer X[s%3]), the
// other half sharpens the rows of pair s-1 (X[(s-1)%3] and X[(s-2)%3]) and stages the spectrum rows of
// pair s+1 into LDS.  Both halves pass the same 8 workgroup barriers per step, so in every barrier interval
// each SIMD holds latency-bound FFT waves next to arithmetic-bound sharpen waves.  One strip per CU keeps
// the halo overhead at one pair in thirteen for the headline size.
template <int UW> struct Fused2Lds {
    static constexpr int KH = UW / 4;                                        // highest non-zero kx (= W/2)
    static constexpr size_t XB = fused_buf_bytes(UW);                        // exchange buffer / two L rows
    static constexpr size_t SB = (sizeof(float2) * (2 * (KH + 1) + 2) + 15) & ~(size_t)15;   // A row, B row, leak terms
    static constexpr size_t RED = 3 * XB + 2 * SB;
    static constexpr size_t TOTAL = RED + 32 * sizeof(float);
};

template <int UW, bool HALF, int TK>
__global__ void __launch_bounds__(UW / 4) k_c2r_sharpen_t(FusedParams p)
{
    constexpr int E = 8, T = UW / E;
    constexpr int KH = UW / 4;
    constexpr float inv = 1.0f / (float)UW;
    using L = Fused2Lds<UW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)(smem + L::RED);      // [0..15] corner partial sums, [16] corner DC term, [20..21] deferred-pixel taps
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / T);      // 0 / 1, wave-uniform
    const int lt = tid - grp * T;
    const int uH = p.uH;
    const int pairs_per_plane = uH / 2;
    const long tile_stride = (long)uH * TK;
    const long plane = (long)UW * uH;
    // Loads and stores retire through ONE in-order counter (vmcnt): a wave that waits for a load also waits
    // for every output store it issued before it.  Hence all loads belong to the FFT role (which stores
    // nothing, and whose previous stores are a whole step old): twiddles first (L1/L2 hits, needed at stage
    // 1), then the staging loads (a whole step to land); the sharpen role never waits on memory at all.

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
        const float2* base = p.S2 + (long)c * p.NT * tile_stride;
        // 32-bit element offsets from the (wave-uniform) plane base: one plane of S2 is < 2^31 elements
        const unsigned tile_stride32 = (unsigned)uH * TK;
        auto S2at = [&](int k, int row) -> float2 {
            // byte offset kept in 32 bits so that the load takes the "SGPR base + VGPR offset" form
            const unsigned off = ((unsigned)(k / TK) * tile_stride32 + (unsigned)row * TK + (unsigned)(k % TK)) * (unsigned)sizeof(float2);
            return *(const float2*)((const char*)base + off);
        };
        const bool need_corner = !top && (y1 + 1 < uH);
        const int rs = y1 + 1;

        // ---- staging of the spectrum rows of pair i: issue (registers) and commit (LDS slot i&1)
        struct Stage { float2 a0, a1, b0, b1, x0, x1; };
        auto stage_issue = [&](int i) -> Stage {
            Stage st;
            const int a = a0 + 2 * i;
            const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);       // rows past the plane: duplicate of the last row
            st.a0 = S2at(lt, ya); st.a1 = S2at(lt + T, ya);
            st.b0 = S2at(lt, yb); st.b1 = S2at(lt + T, yb);
            // lane 0: k = W/2; lane 1: DC column of the reference partners (leak); others: a harmless re-read
            const int kx = (lt == 0) ? KH : 0;
            st.x0 = S2at(kx, (lt == 1) ? (ya ^ 1) : ya);
            st.x1 = S2at(kx, (lt == 1) ? (yb ^ 1) : yb);
            return st;
        };
        auto stage_commit = [&](int i, const Stage& st) {
            float2* SA = (float2*)(smem + 3 * L::XB + (i & 1) * L::SB);
            float2* SBp = SA + (KH + 1);
            SA[lt] = st.a0; SA[lt + T] = st.a1;
            SBp[lt] = st.b0; SBp[lt + T] = st.b1;
            if (lt == 0) { SA[KH] = st.x0; SBp[KH] = st.x1; }
            if (lt == 1) {
                const int a = a0 + 2 * i;
                const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
                // leak: row y gets -Im D[y+1] (y even) / +Im D[y-1] (y odd)
                SBp[KH + 1] = make_float2((ya & 1) ? st.x0.y : -st.x0.y, (yb & 1) ? st.x1.y : -st.x1.y);
            }
        };

        // ---- prologue: each half stages the first pair it will transform (pair 0 by half 0, pair 1 by
        // half 1); half 1 also evaluates the corner sums; one barrier publishes everything
        if (grp < npairs) {
            const Stage st = stage_issue(grp);
            stage_commit(grp, st);
        }
        if (grp == 1 && need_corner) {
            float part = S2at(lt + 1, rs).x + S2at(lt + 1 + T, rs).x;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
            if ((lt & 63) == 0) red[lt >> 6] = part;
            if (lt == T - 1) {
                float2 d = S2at(0, rs), dp = S2at(0, rs ^ 1);
                red[16] = (rs & 1) ? d.x + dp.y : d.x - dp.y;
            }
        }
        __syncthreads();

        for (int s = 0; s <= npairs; s++) {
            if (grp == (s & 1)) {
                // ================= FFT role: pair s
                if (s < npairs) {
                    const float2* SA = (const float2*)(smem + 3 * L::XB + (s & 1) * L::SB);
                    const float2* SBp = SA + (KH + 1);
                    float2* buf = (float2*)(smem + (s % 3) * L::XB);
                    float* cur = (float*)buf;
                    TwSet<UW, E> tws;
                    tws.load(p.tw, lt);
                    float2 v[E];
                    {
                        float2 A = SA[lt], B = SBp[lt];
                        v[0] = make_float2(A.x - B.y, A.y + B.x);
                        A = SA[lt + T]; B = SBp[lt + T];
                        v[1] = make_float2(A.x - B.y, A.y + B.x);
                        A = SA[2 * T - lt]; B = SBp[2 * T - lt];
                        v[6] = make_float2(A.x + B.y, -A.y + B.x);
                        v[2] = make_float2(0.f, 0.f);
                        if (lt == 0) {
                            v[2] = make_float2(A.x - B.y, A.y + B.x);                       // k = 2T = W/2
                            const float2 lk = SBp[KH + 1];
                            v[0] = make_float2(SA[0].x + lk.x, SBp[0].x + lk.y);              // DC terms incl. the pair leak
                        }
                        A = SA[T - lt]; B = SBp[T - lt];
                        v[7] = make_float2(A.x + B.y, -A.y + B.x);
                        v[3] = v[4] = v[5] = make_float2(0.f, 0.f);
                    }
                    // this half's next transform is pair s+2 (clamped re-read when there is none)
                    const bool do_stage = (s + 2 < npairs);
                    const Stage st = stage_issue(do_stage ? s + 2 : s);
                    reg_fft<UW, E, -1, 1, false>(v, buf, lt, 0, tws);                           // 6 barriers
#pragma unroll
                    for (int i = 0; i < E; i++) {
                        cur[lt + T * i] = to_L<HALF>(v[i].x * inv, p.upsq);
                        cur[UW + lt + T * i] = to_L<HALF>(v[i].y * inv, p.upsq);
                    }
                    if (do_stage) stage_commit(s + 2, st);        // same slot (s&1) that was consumed above
                    __syncthreads();                                                            // 7
                    __syncthreads();                                                            // 8
                } else {
#pragma unroll
                    for (int b = 0; b < 8; b++) __syncthreads();
                }
            } else {
                // ================= sharpen role: rows of pair i = s-1; stage pair s+1
                const int i = s - 1;
                const int a = a0 + 2 * i;
                const float* cur = (const float*)(smem + ((i + 3) % 3) * L::XB);               // rows a, a+1
                const float* ring = (const float*)(smem + ((i + 2) % 3) * L::XB);              // rows a-2, a-1
                auto rowp = [&](int r) -> const float* { return r < 0 ? ring + (r + 2) * UW : cur + r * UW; };
                // both output rows of a 4-pixel column group in one pass: the four L rows a-2 .. a+1 are read
                // once and their horizontal minima/maxima are shared by the two 3x3 windows
                if constexpr (HALF) {
                    // -p 2: one undivided pass per column half (the binary16 evaluation needs the registers), its three
                    // barriers after it
    #pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        const bool out0 = i >= 0 && (a - 1) >= y0 && (a - 1) < y1;      // row y = a-1
                        const bool out1 = i >= 0 && a >= y0 && a < y1;                  // row y = a
                        if (out0 || out1) {
                            const int x0 = 4 * (lt + T * h);
                            // row -1 clamps to row 0: for a == 0 the "a-1" slot aliases row a
                            const float* rows[4] = {rowp(-2), (a == 0) ? rowp(0) : rowp(-1), rowp(0), rowp(1)};
                            float t[4][6];
    #pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (r == 0 && !out0) continue;
                                float4 q = *(const float4*)(rows[r] + x0);
                                t[r][1] = q.x; t[r][2] = q.y; t[r][3] = q.z; t[r][4] = q.w;
                                float el = q.x, er = 0.f;
                                if ((lt & 63) == 0 && x0 != 0) el = rows[r][x0 - 1];
                                if ((lt & 63) == 63) {
                                    // x = UW wraps to x = 0 of the next row (rows past a+1: see below)
                                    const float* nx = (r < 3) ? ((a == 0 && r == 1) ? rowp(1) : rows[r + 1]) : rows[3];
                                    er = (x0 + 4 == UW) ? nx[0] : rows[r][x0 + 4];
                                }
                                t[r][0] = lane_from_below(q.w, el);
                                t[r][5] = lane_from_above(q.x, er);
                            }
                            const bool last_chunk = (x0 + 4 == UW);
                            if (last_chunk) {
                                // SE tap of pixel (a, UW-1) is L(a+2, 0): past the plane it clamps to row uH-1; in the
                                // last step it is the corner sample; otherwise the pixel is finished next step
                                const int r2 = min(a + 2, uH - 1) - a;
                                if (r2 <= 1) t[3][5] = rowp(r2)[0];
                                else if (i == npairs - 1) {
                                    float sum = 0.f;
                                    for (int w2 = 0; w2 < T / 64; w2++) sum += red[w2];
                                    t[3][5] = to_L<HALF>((red[16] + 2.0f * sum) * inv, p.upsq);
                                }
                            }
                            float hmn[4][4], hmx[4][4];
    #pragma unroll
                            for (int r = 0; r < 4; r++)
    #pragma unroll
                                for (int k = 0; k < 4; k++) {
                                    hmn[r][k] = fminf(fminf(t[r][k], t[r][k + 1]), t[r][k + 2]);
                                    hmx[r][k] = fmaxf(fmaxf(t[r][k], t[r][k + 1]), t[r][k + 2]);
                                }
    #pragma unroll
                            for (int w = 0; w < 2; w++) {
                                if (w == 0 ? !out0 : !out1) continue;
                                float o[4];
    #pragma unroll
                                for (int k = 0; k < 4; k++) {
                                    const float N = t[w][k + 1], S = t[w + 2][k + 1], Wv = t[w + 1][k], C = t[w + 1][k + 1], E = t[w + 1][k + 2];
                                    const float mn0 = fminf(fminf(N, S), hmn[w + 1][k]);
                                    const float mx0 = fmaxf(fmaxf(N, S), hmx[w + 1][k]);
                                    const float mn1 = fminf(fminf(hmn[w][k], hmn[w + 2][k]), mn0);
                                    const float mx1 = fmaxf(fmaxf(hmx[w][k], hmx[w + 2][k]), mx0);
                                    if constexpr (HALF) o[k] = sharpen_eval_half_fast(N, S, Wv, E, C, mn0, mn1, mx0, mx1, p.coef);
                                    else o[k] = sharpen_eval_fast(((N + Wv) + E) + S, C, mn0, mn1, mx0, mx1, p.coef);
                                }
                                const long row_of = c * plane + (long)(a - 1 + w) * UW;      // wave-uniform
                                if constexpr (HALF) {
                                    __half2 h0 = __floats2half2_rn(o[0], o[1]), h1 = __floats2half2_rn(o[2], o[3]);
                                    typedef float f2v __attribute__((ext_vector_type(2)));
                                    f2v val = {*(float*)&h0, *(float*)&h1};
                                    __builtin_nontemporal_store(val, (f2v*)((char*)((__half*)p.out + row_of) + (unsigned)x0 * 2u));
                                } else {
                                    typedef float f4v __attribute__((ext_vector_type(4)));
                                    f4v val = {o[0], o[1], o[2], o[3]};
                                    __builtin_nontemporal_store(val, (f4v*)((char*)((float*)p.out + row_of) + (unsigned)x0 * 4u));
                                }
                            }
                        }
                        __syncthreads();
                        __syncthreads();
                        __syncthreads();                                                           // 6 in total
                    }
                } else {
    #pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        const bool out0 = i >= 0 && (a - 1) >= y0 && (a - 1) < y1;      // row y = a-1
                        const bool out1 = i >= 0 && a >= y0 && a < y1;                  // row y = a
                        const bool act = out0 || out1;
                        const int x0 = 4 * (lt + T * h);
                        float t[4][6];
                        // the pass is cut in three by the barriers it has to take part in anyway, so that its arithmetic
                        // is spread over the other half's transform stages instead of running beside only one of them
                        if (act) {
                            // row -1 clamps to row 0: for a == 0 the "a-1" slot aliases row a
                            const float* rows[4] = {rowp(-2), (a == 0) ? rowp(0) : rowp(-1), rowp(0), rowp(1)};
    #pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (r == 0 && !out0) {
    #pragma unroll
                                    for (int k = 0; k < 6; k++) t[0][k] = 0.f;
                                    continue;
                                }
                                float4 q = *(const float4*)(rows[r] + x0);
                                t[r][1] = q.x; t[r][2] = q.y; t[r][3] = q.z; t[r][4] = q.w;
                                float el = q.x, er = 0.f;
                                if ((lt & 63) == 0 && x0 != 0) el = rows[r][x0 - 1];
                                if ((lt & 63) == 63) {
                                    // x = UW wraps to x = 0 of the next row (rows past a+1: see below)
                                    const float* nx = (r < 3) ? ((a == 0 && r == 1) ? rowp(1) : rows[r + 1]) : rows[3];
                                    er = (x0 + 4 == UW) ? nx[0] : rows[r][x0 + 4];
                                }
                                t[r][0] = lane_from_below(q.w, el);
                                t[r][5] = lane_from_above(q.x, er);
                            }
                            const bool last_chunk = (x0 + 4 == UW);
                            if (last_chunk) {
                                // SE tap of pixel (a, UW-1) is L(a+2, 0): past the plane it clamps to row uH-1; in the
                                // last step it is the corner sample; otherwise the pixel is finished next step
                                const int r2 = min(a + 2, uH - 1) - a;
                                if (r2 <= 1) t[3][5] = rowp(r2)[0];
                                else if (i == npairs - 1) {
                                    float sum = 0.f;
                                    for (int w2 = 0; w2 < T / 64; w2++) sum += red[w2];
                                    t[3][5] = to_L<HALF>((red[16] + 2.0f * sum) * inv, p.upsq);
                                }
                            }
                        }
    #pragma unroll
                        for (int w = 0; w < 2; w++) {
                            __syncthreads();
                            if (!act || (w == 0 ? !out0 : !out1)) continue;
                            float o[4];
    #pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const float N = t[w][k + 1], S = t[w + 2][k + 1], Wv = t[w + 1][k], C = t[w + 1][k + 1], E = t[w + 1][k + 2];
                                // only t[][] lives across the barriers; min/max are exact in any order
                                const float mn0 = fminf(fminf(N, S), fminf(fminf(Wv, C), E));
                                const float mx0 = fmaxf(fmaxf(N, S), fmaxf(fmaxf(Wv, C), E));
                                const float mn1 = fminf(fminf(fminf(fminf(t[w][k], N), t[w][k + 2]), fminf(fminf(t[w + 2][k], S), t[w + 2][k + 2])), mn0);
                                const float mx1 = fmaxf(fmaxf(fmaxf(fmaxf(t[w][k], N), t[w][k + 2]), fmaxf(fmaxf(t[w + 2][k], S), t[w + 2][k + 2])), mx0);
                                if constexpr (HALF) o[k] = sharpen_eval_half_fast(N, S, Wv, E, C, mn0, mn1, mx0, mx1, p.coef);
                                else o[k] = sharpen_eval_fast(((N + Wv) + E) + S, C, mn0, mn1, mx0, mx1, p.coef);
                            }
                            const long row_of = c * plane + (long)(a - 1 + w) * UW;      // wave-uniform
                            if constexpr (HALF) {
                                __half2 h0 = __floats2half2_rn(o[0], o[1]), h1 = __floats2half2_rn(o[2], o[3]);
                                typedef float f2v __attribute__((ext_vector_type(2)));
                                f2v val = {*(float*)&h0, *(float*)&h1};
                                __builtin_nontemporal_store(val, (f2v*)((char*)((__half*)p.out + row_of) + (unsigned)x0 * 2u));
                            } else {
                                typedef float f4v __attribute__((ext_vector_type(4)));
                                f4v val = {o[0], o[1], o[2], o[3]};
                                __builtin_nontemporal_store(val, (f4v*)((char*)((float*)p.out + row_of) + (unsigned)x0 * 4u));
                            }
                        }
                        __syncthreads();                                                           // 3 per pass, 6 in total
                    }
                }
                if (i >= 0 && lt == T - 1) {
                    // finish the pixel deferred by the previous pair: (a-2, UW-1); L(a,0) is known now
                    if (i > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1) {
                        const float* r2 = rowp(-2);
                        const float* r1 = rowp(-1);
                        const float* r0 = rowp(0);
                        const float ne = (a - 2 == 0) ? r1[0] : r2[0];
                        const float pn0 = red[20], pn1 = red[21];
                        const float t[3][6] = {{pn0, pn0, pn1, ne, ne, ne},
                                               {r2[UW - 2], r2[UW - 2], r2[UW - 1], r1[0], r1[0], r1[0]},
                                               {r1[UW - 2], r1[UW - 2], r1[UW - 1], r0[0], r0[0], r0[0]}};
                        float o[4];
                        sharpen_quad<HALF>(t, p.coef, o);
                        const long of = c * plane + (long)(a - 2) * UW + (UW - 1);
                        if constexpr (HALF) ((__half*)p.out)[of] = __float2half_rn(o[1]);
                        else ((float*)p.out)[of] = o[1];
                    }
                    const float* rn = (a == 0) ? rowp(0) : rowp(-1);
                    red[20] = rn[UW - 2];
                    red[21] = rn[UW - 1];
                }
                __syncthreads();                                                               // 7
                __syncthreads();                                                               // 8
            }
        }
    }
}

}  // namespace fftup
