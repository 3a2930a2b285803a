// kernels_mixed.hpp -- kernels for the mixed-radix sizes (1920x1080 -> 3840x2160 = BASELINE config 4, 1280x720 -> 2560x1440):
//   * register-resident row R2C (1920 = 15*8*16, 1280 = 5*16*16) and polyphase column (1080 = 9*10*12, 720 = 9*8*10) kernels
//     on the three-stage engine MrFftT of kernels_pow2.hpp; the fused C2R+sharpen kernel is k_c2r_sharpen_g<FusedPlanMr16<..>>;
//   * a stand-alone C2R with compile-time radix LDS ping-pong stages (two-launch path, pre-sharpen tap).
#pragma once
#include "fft_engine.hpp"
#include "kernels_generic.hpp"
#include "kernels_pow2.hpp"

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

// ---- row C2R (see k_row_c2r), u = 2.  grid (uH/2, 3), block PUW::T, dynamic LDS 2*lpad_size(UW) float2
// (U: integer upscale factor, see k_c2r_sharpen_g; output row y = row y/U of spectrum buffer y%U, buffers p.S2 - p.S1 apart)
// (the upscale factor is D / (2 DD): DD = 1 for integer and half-integer factors, 2 for quarter-integer ones -- -u 1.25 = 5/4: D = 5, DD = 2)
template <class PUW, bool HALF_OUT, int U = 2, int D = 2 * U, int DD = 1>
__global__ void __launch_bounds__(PUW::T) k_row_c2r_ct(RowC2RParams p)
{
    constexpr int UW = PUW::N, T = PUW::T, KH = UW * DD / D;  // kx = 0..W/2 = UW/2u non-zero
    static_assert((UW * DD) % D == 0, "the output width is the upscale factor times an even input width");
    static_assert(UW % 4 == 0, "four consecutive points per thread in the store loop");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* a = (float2*)smem;
    float2* b = a + lpad_size(UW);
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    const int TK = p.TK;
    // polyphase column pass: row 2j is row j of S1, row 2j+1 is row j of the odd-row buffer (both at twice the scale);
    // in general row y is row y/U of buffer y%U
    const long tile_stride = (long)(p.uH / U) * TK;
    const long delta = p.S2 - p.S1;
    const int ya = 2 * j, yb = 2 * j + 1;
    const float2* rowA = p.S1 + (ya % U) * delta + (long)c * p.NT * tile_stride + (long)(ya / U) * TK;
    const float2* rowB = p.S1 + (yb % U) * delta + (long)c * p.NT * tile_stride + (long)(yb / U) * TK;
    for (int k = tid + 1; k <= UW / 2; k += T) {
        float2 A = make_float2(0.f, 0.f), B = A;
        if (k <= KH) {
            const long o = (long)(k / TK) * tile_stride + (k % TK);
            A = rowA[o];
            B = rowB[o];
        }
        a[lpad(k)] = make_float2(A.x - B.y, A.y + B.x);
        a[lpad(UW - k)] = make_float2(A.x + B.y, -A.y + B.x);
    }
    if (tid == 0) {
        float2 A = rowA[0], B = rowB[0];
        a[lpad(0)] = make_float2(A.x - B.y, A.y + B.x);
    }
    __syncthreads();
    const float2* z = run_plan<-1, 1>(PUW{}, a, b, p.tw, tid);
    const long plane = (long)UW * p.uH;
    constexpr float inv = (1.0f / (float)U) / (float)UW;
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

// =================================================================================== register-resident mixed-radix kernels
// One butterfly per thread and stage, the points in registers, one in-place LDS buffer (MrFftT in kernels_pow2.hpp).
// A size is described by a configuration struct (MixedCfg1080, MixedCfg720 below): row radices (first one odd: the
// stage-0 scatter then spreads over all LDS slots without an index map), column radices, threads per column.

// ---- row R2C, W = RR0 * RR1 * RR2.  grid (H/2, 3), block CFG::ROW_T, LDS W float2.
template <class CFG, int MODE>
__global__ void __launch_bounds__(CFG::ROW_T) k_row_r2c_m(RowR2CTParams p)
{
    constexpr int W = CFG::W, TK = 4, T = CFG::ROW_T, R0 = CFG::RR0;
    using F = MrFftT<W, +1, 1, CFG::RR0, CFG::RR1, CFG::RR2, true>;
    static_assert(T >= F::NB0 && T >= F::NB1 && T >= F::NB2, "one butterfly per thread and stage");
    __shared__ float2 buf[W];
    const int tid = threadIdx.x, c = blockIdx.y, j = blockIdx.x;
    typename F::Tw tw;
    F::load_tw(tw, p.tw, tid);
    float2 v[F::VN];
    if (tid < F::NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++)
            v[m] = make_float2(load_px_t<MODE>(p, c, 2 * j, tid + F::NB0 * m), load_px_t<MODE>(p, c, 2 * j + 1, tid + F::NB0 * m));
    }
    F::run(v, buf, tid, 0, tw);
    // unpack (vkFFT.h:4292-4323), as k_row_r2c_t: 4 consecutive lanes cover one tile segment [A(4)|B(4)], 16 bytes per lane
    const long tile_stride = (long)p.H * TK;
    float2* base = p.S1 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    constexpr int NTILE = (W / 2 + 1 + TK - 1) / TK;
    for (int g = tid; g < NTILE * TK; g += T) {
        const int tile = g / TK, l = g % TK;
        const bool isB = l >= TK / 2;
        const int kk = (l % (TK / 2)) * 2;
        float2 o[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int k = tile * TK + kk + e;
            float2 r = make_float2(0.f, 0.f);
            if (k <= W / 2) {
                const float2 zk = buf[k], zn = buf[k == 0 ? 0 : W - k];
                r = isB ? make_float2(0.5f * (zk.y + zn.y), 0.5f * (-zk.x + zn.x)) : make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            }
            o[e] = r;
        }
        spec_store16(base + (long)tile * tile_stride + (isB ? TK : 0) + kk, o[0], o[1]);
    }
}

// ---- column, H = CR0 * CR1 * CR2, polyphase form (k_col_t): forward, phase, inverse, odd rows out.
// grid (NT, 3), block 4 * CFG::COL_TPC (4 columns x threads per column), LDS H*4 float2.
// 1080 = 9 * 10 * 12: balanced radices, 120 / 108 / 90 butterflies per column and stage on 120 threads (9 * 8 * 15 would
// need 135 threads per column and leave half of them idle in its radix-15 stage), 8 waves, which fit beside a strip.
template <class CFG>
__global__ void __launch_bounds__(4 * CFG::COL_TPC) k_col_m(ColTParams p)
{
    constexpr int H = CFG::H, TK = 4, R0 = CFG::CR0, R2 = CFG::CR2;
    using FF = MrFftT<H, +1, TK, CFG::CR0, CFG::CR1, CFG::CR2, true>;
    using FI = MrFftT<H, -1, TK, CFG::CR0, CFG::CR1, CFG::CR2, false>;
    static_assert(CFG::COL_TPC >= FF::NB0 && CFG::COL_TPC >= FF::NB1 && CFG::COL_TPC >= FF::NB2, "one butterfly per thread and stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x, col = tid % TK, j = tid / TK;
    const int tile = blockIdx.x, c = blockIdx.y;
    const bool valid = tile * TK + col <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    typename FF::Tw tw;
    FF::load_tw(tw, p.twH, j);
    const float2 ph = p.twUH[j < FF::NB0 ? j : 0];           // exp(+2 pi i j/2H)
    float2 v[FF::VN];
    if (j < FF::NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++) v[m] = valid ? src[(j + FF::NB0 * m) * TK + col] : make_float2(0.f, 0.f);
    }
    FF::run(v, buf, j, col, tw);                             // F[k] in natural order in LDS
    if (j < FI::NB0) {
        // t[k] = exp(-2 pi i k/2H) * (k < H/2 ? 1 : -1), k = j + NB0 m: exp(-2 pi i j/2H) times the (2 R0)-th roots of
        // unity exp(-2 pi i m/2R0) (NB0/2H = 1/2R0), taken from the same table: twUH[NB0 m]
        const float2 w = twid<-1>(ph);
#pragma unroll
        for (int m = 0; m < R0; m++) {
            const float2 f = buf[(j + FI::NB0 * m) * TK + col];
            float2 t = (m == 0) ? w : cmul(w, twid<-1>(p.twUH[FI::NB0 * m]));
            if (j + FI::NB0 * m >= H / 2) t = make_float2(-t.x, -t.y);
            v[m] = cmul(f, t);
        }
    }
    __syncthreads();
    FI::run(v, buf, j, col, tw);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    constexpr float inv = 1.0f / (float)H;
    if (j < FI::NB2 && valid) {
#pragma unroll
        for (int m = 0; m < R2; m++) spec_store8(dst + (j + FI::NB2 * m) * TK + col, cscale(v[m], inv));
    }
}

// =================================================================================== N-stage register-resident kernels
// The same two kernels on the N-stage engine MrFftNT (any radix list, several butterflies per thread in the middle
// stages, XOR index map): what a run-time specialised plan (jit.hpp) uses for a length without a three-stage
// factorization (2000 = 8*5*5*10, 3584 = 8*8*8*7, ...).  CFG::RowN = MrFftNT<W, +1, ROW_T, 1, radices...>,
// CFG::ColF / CFG::ColI = MrFftNT<H, +1 / -1, COL_TPC, 4, radices...>.

// ---- row R2C.  grid (H/2, 3), block CFG::ROW_T, LDS lswz_size(W) float2.
template <class CFG, int MODE>
__global__ void __launch_bounds__(CFG::ROW_T) k_row_r2c_n(RowR2CTParams p)
{
    using F = typename CFG::RowN;
    constexpr int W = CFG::W, TK = 4, T = CFG::ROW_T, R0 = F::rs(0), NB0 = W / R0, RL = F::rs(F::NST - 1), NBL = W / RL;
    static_assert(T >= NB0 && T >= NBL, "one butterfly per thread in the first and the last stage");
    __shared__ __attribute__((aligned(128))) float2 buf[lswz_size(W)];
    const int tid = threadIdx.x, c = blockIdx.y, j = blockIdx.x;
    typename F::Tw tw;
    F::load_tw(tw, p.tw, tid);
    float2 v[F::VN];
    if (tid < NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++)
            v[m] = make_float2(load_px_t<MODE>(p, c, 2 * j, tid + NB0 * m), load_px_t<MODE>(p, c, 2 * j + 1, tid + NB0 * m));
    }
    F::template run<true>(v, buf, buf, tid, tw);            // (in place: the last gather is behind a barrier)
    if (tid < NBL) {
#pragma unroll
        for (int m = 0; m < RL; m++) buf[lswz(tid + NBL * m)] = v[m];
    }
    __syncthreads();
    // unpack (vkFFT.h:4292-4323), as k_row_r2c_m
    const long tile_stride = (long)p.H * TK;
    float2* base = p.S1 + (long)c * p.NT * tile_stride + (long)(2 * j) * TK;
    constexpr int NTILE = (W / 2 + 1 + TK - 1) / TK;
    for (int g = tid; g < NTILE * TK; g += T) {
        const int tile = g / TK, l = g % TK;
        const bool isB = l >= TK / 2;
        const int kk = (l % (TK / 2)) * 2;
        float2 o[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int k = tile * TK + kk + e;
            float2 r = make_float2(0.f, 0.f);
            if (k <= W / 2) {
                const float2 zk = buf[lswz(k)], zn = buf[lswz(k == 0 ? 0 : W - k)];
                r = isB ? make_float2(0.5f * (zk.y + zn.y), 0.5f * (-zk.x + zn.x)) : make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            }
            o[e] = r;
        }
        spec_store16(base + (long)tile * tile_stride + (isB ? TK : 0) + kk, o[0], o[1]);
    }
}

// ---- column, polyphase form (k_col_t).  A workgroup transforms CC = CFG::COL_COLS columns of a 4-wide spectrum tile (4, or 2
// for lengths whose first and last stage need more than 256 threads per column).  grid (NT * 4/CC, 3), block CC * CFG::COL_TPC,
// LDS lswz_size(CC H) float2.
template <class CFG>
__global__ void __launch_bounds__(CFG::COL_COLS * CFG::COL_TPC) k_col_n(ColTParams p)
{
    using FF = typename CFG::ColF;
    using FI = typename CFG::ColI;
    constexpr int H = CFG::H, TK = 4, CC = CFG::COL_COLS, TC = CFG::COL_TPC, R0 = FF::rs(0), NB0 = H / R0, RL = FF::rs(FF::NST - 1), NBL = H / RL;
    static_assert(TC >= NB0 && TC >= NBL, "one butterfly per thread in the first and the last stage");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x, col = tid % CC, j = tid / CC;
    const int tile = blockIdx.x / (TK / CC), gcol = (blockIdx.x % (TK / CC)) * CC + col, c = blockIdx.y;
    const bool valid = tile * TK + gcol <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    typename FF::Tw twf;
    typename FI::Tw twi;
    FF::load_tw(twf, p.twH, j);
    FI::load_tw(twi, p.twH, j);
    float2 v[FF::VN];
    if (j < NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++) v[m] = valid ? src[(j + NB0 * m) * TK + gcol] : make_float2(0.f, 0.f);
    }
    FF::template run<true>(v, buf, buf, j, twf, col);
    if (j < NBL) {
#pragma unroll
        for (int m = 0; m < RL; m++) buf[lidx<CC>(j + NBL * m, col)] = v[m];       // F[k] in natural order
    }
    __syncthreads();
    if (j < NB0) {
        // t[k] = exp(-2 pi i k/2H) * (k < H/2 ? 1 : -1), straight from the table of 2H-th roots
#pragma unroll
        for (int m = 0; m < R0; m++) {
            const int k = j + NB0 * m;
            float2 t = twid<-1>(p.twUH[k]);
            if (k >= H / 2) t = make_float2(-t.x, -t.y);
            v[m] = cmul(buf[lidx<CC>(k, col)], t);
        }
    }
    __syncthreads();
    FI::template run<true>(v, buf, buf, j, twi, col);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    constexpr float inv = 1.0f / (float)H;
    if (j < NBL && valid) {
#pragma unroll
        for (int m = 0; m < RL; m++) spec_store8(dst + (j + NBL * m) * TK + gcol, cscale(v[m], inv));
    }
}

// ---- column for an integer upscale factor U > 2: one forward transform, U-1 residue transforms.
// Output row n = U m + r of the zero-padded inverse (length U H) is, for r = 0, H times the column itself (never
// written: the row kernels read S1) and for r = 1..U-1 the length-H transform of F[k] t_r[k],
//   t_r[k] = exp(-2 pi i r k / UH) * (k < H/2 ? 1 : exp(+2 pi i r / U))
// (the upper half of the spectrum sits (U-1) H rows higher).  U = 2 is the odd-row transform of k_col_t.  Residue buffer
// r lives (r-1) * buf_stride elements behind p.S2, all at U times the reference's normalisation (D / H).
template <class CFG, int U>
__global__ void __launch_bounds__(CFG::COL_COLS * CFG::COL_TPC) k_col_u(ColTParams p)
{
    using FF = typename CFG::ColF;
    using FI = typename CFG::ColI;
    constexpr int H = CFG::H, TK = 4, CC = CFG::COL_COLS, TC = CFG::COL_TPC, R0 = FF::rs(0), NB0 = H / R0, RL = FF::rs(FF::NST - 1), NBL = H / RL;
    static_assert(TC >= NB0 && TC >= NBL, "one butterfly per thread in the first and the last stage");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x, col = tid % CC, j = tid / CC;
    const int tile = blockIdx.x / (TK / CC), gcol = (blockIdx.x % (TK / CC)) * CC + col, c = blockIdx.y;
    const bool valid = tile * TK + gcol <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    typename FF::Tw twf;
    typename FI::Tw twi;
    FF::load_tw(twf, p.twH, j);
    FI::load_tw(twi, p.twH, j);
    float2 v[FF::VN], f[R0];
    if (j < NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++) v[m] = valid ? src[(j + NB0 * m) * TK + gcol] : make_float2(0.f, 0.f);
    }
    FF::template run<true>(v, buf, buf, j, twf, col);
    if (j < NBL) {
#pragma unroll
        for (int m = 0; m < RL; m++) buf[lidx<CC>(j + NBL * m, col)] = v[m];       // F[k] in natural order
    }
    __syncthreads();
    if (j < NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++) f[m] = buf[lidx<CC>(j + NB0 * m, col)];       // kept in registers over the residues
    }
    __syncthreads();
    const long buf_stride = (long)3 * p.NT * H * TK;
    constexpr float inv = 1.0f / (float)H;
#pragma unroll 1
    for (int r = 1; r < U; r++) {
        if (j < NB0) {
            const float2 hi = twid<+1>(p.twUH[r * H]);                              // exp(+2 pi i r/U) from the table of UH-th roots
#pragma unroll
            for (int m = 0; m < R0; m++) {
                const int k = j + NB0 * m;
                float2 t = twid<-1>(p.twUH[r * k]);                                 // r k < U H
                if (k >= H / 2) t = cmul(t, hi);
                v[m] = cmul(f[m], t);
            }
        }
        FI::template run<true>(v, buf, buf, j, twi, col);
        float2* dst = p.S2 + (r - 1) * buf_stride + ((long)c * p.NT + tile) * H * TK;
        if (j < NBL && valid) {
#pragma unroll
            for (int m = 0; m < RL; m++) spec_store8(dst + (j + NBL * m) * TK + gcol, cscale(v[m], inv));
        }
        __syncthreads();                                                            // the buffer is free for the next residue
    }
}

// ---- column for a half-integer upscale factor (-u 1.5, 2.5): forward transform of length H, the reference's shift and
// zero-padding (VkResample.cpp:514-526, read guard vkFFT.h:1670-1695) as the gather of the inverse's first stage, inverse
// transform of length UH = u H, ALL rows written to one buffer at the reference's normalisation (no residue split: the
// rows of the zero-padded inverse are not subsequences of equal length here).  The guard [p.zly, p.zry) is the reference's, as its
// float arithmetic puts it: [H/2, UH - H/2) for every binary-fraction factor, sometimes a row off that for the others (-u 1.2 ...).
// CFG::ColF = MrFftNT<H, +1, COL_TPC, 4, ...>, CFG::ColIU = MrFftNT<UH, -1, COL_TPC, 4, ...>.  LDS lswz_size(4 UH) float2.
template <class CFG>
__global__ void __launch_bounds__(CFG::COL_COLS * CFG::COL_TPC) k_col_pad(ColTParams p)
{
    using FF = typename CFG::ColF;
    using FI = typename CFG::ColIU;
    constexpr int H = CFG::H, UH = CFG::UH, TK = 4, CC = CFG::COL_COLS, TC = CFG::COL_TPC;
    constexpr int R0 = FF::rs(0), NB0 = H / R0, RL = FF::rs(FF::NST - 1), NBL = H / RL;
    constexpr int Q0 = FI::rs(0), MB0 = UH / Q0, QL = FI::rs(FI::NST - 1), MBL = UH / QL;
    static_assert(TC >= NB0 && TC >= NBL && TC >= MB0 && TC >= MBL, "one butterfly per thread in the first and the last stages");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x, col = tid % CC, j = tid / CC;
    const int tile = blockIdx.x / (TK / CC), gcol = (blockIdx.x % (TK / CC)) * CC + col, c = blockIdx.y;
    const bool valid = tile * TK + gcol <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    typename FF::Tw twf;
    typename FI::Tw twi;
    FF::load_tw(twf, p.twH, j);
    FI::load_tw(twi, p.twUH, j);
    float2 v[FF::VN], w[FI::VN];
    if (j < NB0) {
#pragma unroll
        for (int m = 0; m < R0; m++) v[m] = valid ? src[(j + NB0 * m) * TK + gcol] : make_float2(0.f, 0.f);
    }
    FF::template run<true>(v, buf, buf, j, twf, col);
    if (j < NBL) {
#pragma unroll
        for (int m = 0; m < RL; m++) buf[lidx<CC>(j + NBL * m, col)] = v[m];       // F[k] in natural order
    }
    __syncthreads();
    if (j < MB0) {
#pragma unroll
        for (int m = 0; m < Q0; m++) {
            const int ky = j + MB0 * m;                                             // row of the zero-padded buffer
            float2 g = make_float2(0.f, 0.f);
            // (as k_col: the guard first, then the shifted upper half, then the un-shifted rows below H -- with the symmetric guard
            // [H/2, UH - H/2) that is rows below H/2 and the upper half; a guard that starts a row early zeroes that row)
            if (ky >= p.zly && ky < p.zry) {}
            else if (ky >= UH - H / 2) g = buf[lidx<CC>(ky - (UH - H), col)];
            else if (ky < H) g = buf[lidx<CC>(ky, col)];
            w[m] = g;
        }
    }
    __syncthreads();
    FI::template run<true>(w, buf, buf, j, twi, col);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * UH * TK;
    constexpr float inv = 1.0f / (float)UH;
    if (j < MBL && valid) {
#pragma unroll
        for (int m = 0; m < QL; m++) spec_store8(dst + (j + MBL * m) * TK + gcol, cscale(w[m], inv));
    }
}

// ---- the sizes.  CT = the stand-alone C2R plan (two-launch path and pre-sharpen tap), FUSED = the default fused plan.
using Plan3840 = CtPlan<3840, 512, 8, 8, 4, 3, 5>;
using Plan2560 = CtPlan<2560, 512, 8, 8, 8, 5>;
struct MixedCfg1080 {                 // 1920 x 1080 -> 3840 x 2160 (BASELINE config 4)
    static constexpr int W = 1920, H = 1080;
    static constexpr int RR0 = 15, RR1 = 8, RR2 = 16, ROW_T = 256;
    static constexpr int CR0 = 9, CR1 = 10, CR2 = 12, COL_TPC = 120;
    using CT = Plan3840;
    using FUSED = FusedPlan3840x16;
};
struct MixedCfg720 {                  // 1280 x 720 -> 2560 x 1440
    static constexpr int W = 1280, H = 720;
    static constexpr int RR0 = 5, RR1 = 16, RR2 = 16, ROW_T = 256;
    static constexpr int CR0 = 9, CR1 = 8, CR2 = 10, COL_TPC = 90;
    using CT = Plan2560;
    using FUSED = FusedPlanMr16<2560, 10>;
};

}  // namespace fftup
