// kernels_mixed.hpp -- compile-time plans for the mixed-radix sizes of BASELINE config 4
// (1920x1080 -> 3840x2160: radix 2/3/4/5/8 Stockham stages).  Same algorithm and LDS ping-pong layout as
// kernels_generic.hpp; length, thread count and radix sequence are template constants, so the stage loops
// unroll and all index arithmetic (including the k = j % Ns of the odd stages) folds at compile time.
#pragma once
#include "fft_engine.hpp"
#include "kernels_generic.hpp"

namespace fftup {

template <int N_, int T_, int... R> struct CtPlan {
    static constexpr int N = N_, T = T_;
};

// all stages: data in `a`, ping-pong with `b`; returns the buffer holding the result (barrier executed)
template <int N, int DIR, int TK, int T, int Ns>
__device__ __forceinline__ float2* fft_lds_ct(float2* a, float2* b, const float2* __restrict__ tw, int tid)
{
    (void)b; (void)tw; (void)tid;
    return a;
}
template <int N, int DIR, int TK, int T, int Ns, int R, int... Rest>
__device__ __forceinline__ float2* fft_lds_ct(float2* a, float2* b, const float2* __restrict__ tw, int tid)
{
    stage_lds<R, DIR, TK>(a, b, N, Ns, tw, tid, T);
    __syncthreads();
    return fft_lds_ct<N, DIR, TK, T, Ns * R, Rest...>(b, a, tw, tid);
}
template <int DIR, int TK, int N, int T, int... R>
__device__ __forceinline__ float2* run_plan(CtPlan<N, T, R...>, float2* a, float2* b, const float2* __restrict__ tw, int tid)
{
    static_assert((R * ... * 1) == N, "radices must multiply to N");
    return fft_lds_ct<N, DIR, TK, T, 1, R...>(a, b, tw, tid);
}

// ---- row R2C (see k_row_r2c).  grid (H/2, 3), block PW::T, dynamic LDS 2*lpad_size(W) float2
template <class PW, int MODE>
__global__ void __launch_bounds__(PW::T) k_row_r2c_ct(RowR2CParams p)
{
    constexpr int W = PW::N, T = PW::T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* a = (float2*)smem;
    float2* b = a + lpad_size(W);
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    for (int n = tid; n < W; n += T)
        a[lpad(n)] = make_float2(load_px<MODE>(p, c, 2 * j, n), load_px<MODE>(p, c, 2 * j + 1, n));
    __syncthreads();
    const float2* Z = run_plan<+1, 1>(PW{}, a, b, p.tw, tid);
    const int TK = p.TK;
    const long tile_stride = (long)p.H * TK;
    float2* base = p.S1 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    for (int k = tid; k <= W / 2; k += T) {
        float2 zk = Z[lpad(k)];
        float2 zn = Z[lpad(k == 0 ? 0 : W - k)];
        float2 A = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        float2 B = make_float2(0.5f * (zk.y + zn.y), 0.5f * (-zk.x + zn.x));
        float2* dst = base + (long)(k / TK) * tile_stride + (k % TK);
        dst[0] = A;
        dst[TK] = B;
    }
}

// ---- column (see k_col), u = 2.  grid (NT, 3), block PUH::T, dynamic LDS 2*lpad_size(UH*TK) float2
template <class PH, class PUH, int TK>
__global__ void __launch_bounds__(PUH::T) k_col_ct(ColParams p)
{
    constexpr int H = PH::N, UH = PUH::N, T = PUH::T;
    static_assert(PH::T == PUH::T && UH == 2 * H, "one block size for both transforms, u = 2");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* a = (float2*)smem;
    float2* b = a + lpad_size(UH * TK);
    const int tid = threadIdx.x, tile = blockIdx.x, c = blockIdx.y;
    const int ncol_valid = min(TK, p.W / 2 + 1 - tile * TK);
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    for (int e = tid; e < H * TK; e += T) {
        float2 v = make_float2(0.f, 0.f);
        if ((e % TK) < ncol_valid) v = src[e];
        a[lpad(e)] = v;
    }
    __syncthreads();
    float2* F = run_plan<+1, TK>(PH{}, a, b, p.twH, tid);
    float2* G = (F == a) ? b : a;
    // shift + zero-pad guard for u = 2: G[ky] = F[ky] (ky < H/2), F[ky-H] (ky >= 3H/2), else 0
    for (int e = tid; e < UH * TK; e += T) {
        const int ky = e / TK, col = e % TK;
        float2 v = make_float2(0.f, 0.f);
        if (ky < H / 2) v = F[lpad(e)];
        else if (ky >= UH - H / 2) v = F[lpad((ky - H) * TK + col)];
        G[lpad(e)] = v;
    }
    __syncthreads();
    const float2* D = run_plan<-1, TK>(PUH{}, G, F, p.twUH, tid);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * UH * TK;
    constexpr float inv = 1.0f / (float)UH;
    for (int e = tid; e < UH * TK; e += T)
        if ((e % TK) < ncol_valid) dst[e] = cscale(D[lpad(e)], inv);
}

// ---- row C2R (see k_row_c2r), u = 2.  grid (uH/2, 3), block PUW::T, dynamic LDS 2*lpad_size(UW) float2
template <class PUW, bool HALF_OUT>
__global__ void __launch_bounds__(PUW::T) k_row_c2r_ct(RowC2RParams p)
{
    constexpr int UW = PUW::N, T = PUW::T, KH = UW / 4;       // kx = 0..W/2 = UW/4 non-zero
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* a = (float2*)smem;
    float2* b = a + lpad_size(UW);
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    const int TK = p.TK;
    const long tile_stride = (long)p.uH * TK;
    const float2* base = p.S2 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    for (int k = tid + 1; k <= UW / 2; k += T) {
        float2 A = make_float2(0.f, 0.f), B = A;
        if (k <= KH) {
            const float2* s = base + (long)(k / TK) * tile_stride + (k % TK);
            A = s[0];
            B = s[TK];
        }
        a[lpad(k)] = make_float2(A.x - B.y, A.y + B.x);
        a[lpad(UW - k)] = make_float2(A.x + B.y, -A.y + B.x);
    }
    if (tid == 0) {
        float2 A = base[0], B = base[TK];
        a[lpad(0)] = make_float2(A.x - B.y, A.y + B.x);
    }
    __syncthreads();
    const float2* z = run_plan<-1, 1>(PUW{}, a, b, p.tw, tid);
    const long plane = (long)UW * p.uH;
    constexpr float inv = 1.0f / (float)UW;
    // 4 consecutive points per thread: 16-byte (8-byte for half) stores
    for (int n0 = tid * 4; n0 < UW; n0 += T * 4) {
        float2 q[4];
#pragma unroll
        for (int e = 0; e < 4; e++) q[e] = z[lpad(n0 + e)];
        if constexpr (HALF_OUT) {
            __half* R = (__half*)p.R + c * plane + (long)(2 * j) * UW + n0;
            __half2 r0 = __floats2half2_rn(q[0].x * inv, q[1].x * inv), r1 = __floats2half2_rn(q[2].x * inv, q[3].x * inv);
            __half2 i0 = __floats2half2_rn(q[0].y * inv, q[1].y * inv), i1 = __floats2half2_rn(q[2].y * inv, q[3].y * inv);
            *(float2*)R = make_float2(*(float*)&r0, *(float*)&r1);
            *(float2*)(R + UW) = make_float2(*(float*)&i0, *(float*)&i1);
        } else {
            float* R = (float*)p.R + c * plane + (long)(2 * j) * UW + n0;
            *(float4*)R = make_float4(q[0].x * inv, q[1].x * inv, q[2].x * inv, q[3].x * inv);
            *(float4*)(R + UW) = make_float4(q[0].y * inv, q[1].y * inv, q[2].y * inv, q[3].y * inv);
        }
    }
}

#ifndef FFTUP_COLT
#define FFTUP_COLT 1024
#endif
constexpr int COLT = FFTUP_COLT;
// plans of the 1080p -> 2160p configuration
using Plan1920 = CtPlan<1920, 256, 8, 8, 2, 3, 5>;
using Plan3840 = CtPlan<3840, 512, 8, 8, 4, 3, 5>;
using Plan1080 = CtPlan<1080, COLT, 8, 3, 3, 3, 5>;      // x TK = 4 columns
using Plan2160 = CtPlan<2160, COLT, 8, 2, 3, 3, 3, 5>;

}  // namespace fftup
