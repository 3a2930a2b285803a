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
constexpr int stage_radix(int N, int Ns) { return (N / Ns >= 8) ? 8 : (N / Ns); }

// LDS element index of point idx of sequence col (TK interleaved sequences)
template <int TK> __device__ __forceinline__ int lidx(int idx, int col) { return lpad(idx * TK + col); }

// ---- one stage on registers: E/R butterflies of radix R, twiddles for Ns > 1
template <int N, int E, int R, int Ns, int DIR>
__device__ __forceinline__ void reg_butterflies(float2 (&v)[E], int p, const float2* __restrict__ tw)
{
    constexpr int Tc = N / E;
    constexpr int NB = E / R;
    constexpr int tstep = N / (Ns * R);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float2 w[R];
#pragma unroll
        for (int m = 0; m < R; m++) w[m] = v[b + m * NB];
        if constexpr (Ns > 1) {
            const int k = (p + b * Tc) & (Ns - 1);
            apply_twiddles<R, DIR>(w, tw, k * tstep);
        }
        bfly<R, DIR>(w);
#pragma unroll
        for (int m = 0; m < R; m++) v[b + m * NB] = w[m];
    }
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
template <int N, int E, int DIR, int TK, bool FINAL_TO_LDS, int Ns = 1>
__device__ __forceinline__ void reg_fft(float2 (&v)[E], float2* __restrict__ buf, int p, int col,
                                        const float2* __restrict__ tw)
{
    constexpr int R = stage_radix(N, Ns);
    static_assert(E % R == 0, "radix must divide the per-thread point count");
    reg_butterflies<N, E, R, Ns, DIR>(v, p, tw);
    constexpr bool last = (Ns * R == N);
    if constexpr (!last || FINAL_TO_LDS) {
        reg_scatter<N, E, R, Ns, TK>(v, buf, p, col);
        __syncthreads();
    }
    if constexpr (!last) {
        reg_gather<N, E, TK>(v, buf, p, col);
        __syncthreads();
        reg_fft<N, E, DIR, TK, FINAL_TO_LDS, Ns * R>(v, buf, p, col, tw);
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
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    float2 v[E];
#pragma unroll
    for (int i = 0; i < E; i++)
        v[i] = make_float2(load_px_t<MODE>(p, c, 2 * j, tid + T * i), load_px_t<MODE>(p, c, 2 * j + 1, tid + T * i));
    reg_fft<W, E, +1, 1, true>(v, buf, tid, 0, p.tw);
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
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = valid ? src[(pp + Tc * i) * TK + col] : make_float2(0.f, 0.f);
    reg_fft<H, 8, +1, TK, true>(v, buf, pp, col, p.twH);          // F[ky] natural order in LDS
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
    reg_fft<UH, 16, -1, TK, false>(g, buf, pp, col, p.twUH);
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
        reg_fft<UW, E, -1, 1, false>(v, buf, tid, 0, p.tw);
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
        reg_fft<UW, E, -1, 1, true>(v, buf, tid, 0, p.tw);
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
    float left = __shfl_up(r.L[4], 1);
    float right = __shfl_down(r.L[1], 1);
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
            o[i] = sharpen_eval<HALF>(N, S, Wv, E, C, mn1, mx1, mn0, mx0, p.coef);
        }
        const long of = poff + (long)y * uW + x0;
        if constexpr (HALF) {
            __half2 h0 = __floats2half2_rn(o[0], o[1]), h1 = __floats2half2_rn(o[2], o[3]);
            *(float2*)((__half*)p.out + of) = make_float2(*(float*)&h0, *(float*)&h1);
        } else {
            *(float4*)((float*)p.out + of) = make_float4(o[0], o[1], o[2], o[3]);
        }
        ra = rb;
        rb = rc;
    }
}

}  // namespace fftup
