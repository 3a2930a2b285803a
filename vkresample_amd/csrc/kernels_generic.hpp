// kernels_generic.hpp -- size-generic HIP kernels of the upscale path (any 2,3,5,7-smooth size).
//
// Three FFT kernels + sharpen per frame (the reference records 20 dispatches):
//   k_row_r2c   replaces F0        (VkFFT type 5, vkFFT.h:1945-2058 / 4274-4377)
//   k_col       replaces F1,F2,S,I0,I1 (vkFFT.h:1656-1717, VkResample.cpp:514-526, vkFFT.h:1670-1695)
//   k_row_c2r   replaces I2        (VkFFT type 6, vkFFT.h:2059-2201 / 4378-4491)
//   k_sharpen   replaces C         (VkResample.cpp:819-925)
// Spectrum layout in HBM ("blocked half-spectrum"): S[c][tile][ky][TK] float2, tile = kx / TK,
// kx = 0..W/2.  A column tile is one contiguous block for k_col; a row pair (2j,2j+1) of a tile
// is one contiguous 2*TK*8-byte segment for the row kernels.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "fft_engine.hpp"

namespace fftup {

enum InMode { IN_F32 = 0, IN_F16 = 1, IN_U8_F32 = 2, IN_U8_F16 = 3, IN_F64 = 4 };   // IN_F64: -p 1 plans (C = double2)

// VkResample.cpp:1644  x = float(double(v)/255.0)   -- fp32 division is correctly rounded here
// and agrees with the double-rounded reference expression for all 256 inputs (tests check it).
// RN(v / 255) in three instructions instead of the ten of an IEEE division: q0 = v c, c = RN(1/255); the residual
// v - 255 q0 is exact in one fma; one correction step.  Equal to the division for all 256 codes (tests/test_gpu_parity.py
// compares every code with the oracle; the same three operations in C: 0 mismatches, the bare product has 126).
__device__ __forceinline__ float div255(float x)
{
    const float c = 1.0f / 255.0f;
    const float q0 = x * c;
    return fmaf(fmaf(-255.0f, q0, x), c, q0);
}
__device__ __forceinline__ float cvt_u8_f32(uint8_t v) { return div255((float)v); }
// VkResample.cpp:1676  x = half((float)half(v)/255.0)  (round to nearest even)
__device__ __forceinline__ float cvt_u8_f16(uint8_t v) { return __half2float(__float2half_rn(div255((float)v))); }

// Workgroup size limit of the size-generic kernels: 1024 threads (128 VGPRs) for float2, 512 (256 VGPRs) for double2 --
// at 128 the double instantiations spill (fftup.hip clamps its thread counts accordingly).
template <typename C> struct GenericMaxThreads { static constexpr int value = sizeof(C) > 8 ? 512 : 1024; };

template <typename C> struct RowR2CParamsT {
    const void* in;          // planar float/half/double (row stride, plane stride in elements) or u8 RGB (row stride bytes)
    C* S1;                   // blocked half spectrum, H rows
    const C* tw;             // W-th roots
    StagePlan plan;          // n = W
    int W, H;
    long in_row_stride, in_plane_stride;
    int TK, NT;              // tile width (complex), number of tiles = ceil((W/2+1)/TK)
};
using RowR2CParams = RowR2CParamsT<float2>;

template <int MODE, typename P> __device__ __forceinline__ auto load_px(const P& p, int c, int y, int x)
{
    if constexpr (MODE == IN_F64) {
        return ((const double*)p.in)[c * p.in_plane_stride + y * p.in_row_stride + x];
    } else if constexpr (MODE == IN_F32) {
        return ((const float*)p.in)[c * p.in_plane_stride + y * p.in_row_stride + x];
    } else if constexpr (MODE == IN_F16) {
        return __half2float(((const __half*)p.in)[c * p.in_plane_stride + y * p.in_row_stride + x]);
    } else if constexpr (MODE == IN_U8_F32) {
        return cvt_u8_f32(((const uint8_t*)p.in)[y * p.in_row_stride + 3l * x + c]);
    } else {
        return cvt_u8_f16(((const uint8_t*)p.in)[y * p.in_row_stride + 3l * x + c]);
    }
}

// grid (H/2, 3); dynamic LDS = 2 * lpad_size(W) complex
// INPLACE (-p 1 plans whose stages allow it, stage_fits_inplace with 8 points per thread): ONE buffer, two workgroups per compute
// unit instead of one (a 4096-point double2 row pair: 70 instead of 139 KB) -- these kernels wait for LDS and twiddle loads
// most of the time, a second resident workgroup fills the gaps
template <int MODE, typename C = float2, bool INPLACE = false>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value, INPLACE ? 4 : 1) k_row_r2c(RowR2CParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.W);
    const int tid = threadIdx.x, T = blockDim.x;
    const int j = blockIdx.x, c = blockIdx.y;
    const int W = p.W;
    for (int n = tid; n < W; n += T)
        a[lpad(n)] = mk<C>(load_px<MODE>(p, c, 2 * j, n), load_px<MODE>(p, c, 2 * j + 1, n));
    __syncthreads();
    const C* Z = a;
    if constexpr (INPLACE) fft_lds_inplace<+1, 8>(a, p.plan, p.tw, tid, T);
    else Z = fft_lds<+1, 1>(a, b, p.plan, p.tw, tid, T);
    // unpack two real rows (vkFFT.h:4292-4323): A = (Z[k]+conj Z[W-k])/2, B = (Z[k]-conj Z[W-k])/(2i)
    const long tile_stride = (long)p.H * p.TK;
    C* base = p.S1 + (long)c * p.NT * tile_stride;
    for (int k = tid; k <= W / 2; k += T) {
        C zk = Z[lpad(k)];
        C zn = Z[lpad(k == 0 ? 0 : W - k)];
        C A = mk<C>(S(0.5) * (zk.x + zn.x), S(0.5) * (zk.y - zn.y));
        C B = mk<C>(S(0.5) * (zk.y + zn.y), S(0.5) * (-zk.x + zn.x));
        C* dst = base + (long)(k / p.TK) * tile_stride + (long)(2 * j) * p.TK + (k % p.TK);
        dst[0] = A;
        dst[p.TK] = B;
    }
}

template <typename C> struct ColParamsT {
    const C* S1;
    C* S2;
    const C *twH, *twUH;
    StagePlan planH, planUH;
    int W, H, uH;
    int NT;
    int ncols;               // kx columns present: W/2 + 1 (R2C plans) or W (non-R2C plans)
    int zly, zry;            // inverse read guard: rows [zly,zry) read as zero (VkResample.cpp:1494-1495)
    scalar_t<C> inv_norm;    // 1/uH
};
using ColParams = ColParamsT<float2>;

// grid (NT, 3); dynamic LDS = 2 * lpad_size(uH*TK) complex
// INPLACE (-p 1 plans whose stages allow it, stage_fits_inplace_tk with COL_INPLACE_PT points per thread): ONE buffer of uH * TK -- twice the tile
// width in the same LDS (64-byte instead of 32-byte pieces for the row kernels on either side), and the shift between the two
// transforms moves the upper half of the spectrum through registers instead of copying everything into the second buffer.
template <int TK, typename C = float2, bool INPLACE = false>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value, INPLACE ? 4 : 1) k_col(ColParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.uH * TK);
    const int tid = threadIdx.x, T = blockDim.x;
    const int tile = blockIdx.x, c = blockIdx.y;
    const int H = p.H, uH = p.uH;
    const int ncol_valid = min(TK, p.ncols - tile * TK);
    const C* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    for (int e = tid; e < H * TK; e += T) {
        C v = mk<C>(S(0), S(0));
        if ((e % TK) < ncol_valid) v = src[e];
        a[lpad(e)] = v;
    }
    __syncthreads();
    const C* D;
    if constexpr (INPLACE) {
        fft_lds_inplace_tk<+1, COL_INPLACE_PT, TK>(a, p.planH, p.twH, tid, T);
        // shift (VkResample.cpp:514-526) and the zero-padding read guard of the inverse plan, as below, within the one buffer:
        // rows [H/2, H) of F go to [uH - H/2, uH) -- through registers, the two ranges overlap for factors below 1.5 -- the
        // rows in [zly, zry) are read as zero, rows below H keep the un-shifted F
        constexpr int PT = COL_INPLACE_PT;
        C up[PT];
        const int nup = (H - H / 2) * TK, e0 = (H / 2) * TK;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const int e = tid + i * T;
            if (e < nup) up[i] = a[lpad(e0 + e)];
        }
        __syncthreads();
        for (int e = H * TK + tid; e < uH * TK; e += T) a[lpad(e)] = mk<C>(S(0), S(0));    // (rows of F never written: H <= ky < uH)
        __syncthreads();
        const int d0 = (uH - (H - H / 2)) * TK;                                             // row uH - H + H/2
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const int e = tid + i * T;
            if (e < nup) a[lpad(d0 + e)] = up[i];
        }
        __syncthreads();
        for (int e = p.zly * TK + tid; e < p.zry * TK; e += T) a[lpad(e)] = mk<C>(S(0), S(0));
        __syncthreads();
        fft_lds_inplace_tk<-1, COL_INPLACE_PT, TK>(a, p.planUH, p.twUH, tid, T);
        D = a;
    } else {
        C* F = fft_lds<+1, TK>(a, b, p.planH, p.twH, tid, T);
        C* G = (F == a) ? b : a;
        // shift (VkResample.cpp:514-526): buffer row ky' holds F[ky'-(uH-H)] for ky' >= uH-H/2, else the
        // un-shifted F[ky'] while ky' < H; then the zero-padding read guard of the inverse plan.
        for (int e = tid; e < uH * TK; e += T) {
            const int ky = e / TK, col = e % TK;
            C v = mk<C>(S(0), S(0));
            if (!(ky >= p.zly && ky < p.zry)) {
                if (ky >= uH - H / 2) v = F[lpad((ky - (uH - H)) * TK + col)];
                else if (ky < H) v = F[lpad(e)];
            }
            G[lpad(e)] = v;
        }
        __syncthreads();
        D = fft_lds<-1, TK>(G, F, p.planUH, p.twUH, tid, T);
    }
    C* dst = p.S2 + ((long)c * p.NT + tile) * uH * TK;
    for (int e = tid; e < uH * TK; e += T)
        if ((e % TK) < ncol_valid) dst[e] = cscale(D[lpad(e)], p.inv_norm);
}

// ---- the same column pass for u = 2 in polyphase form (round 5; what k_col_t / k_col_n are for the specialised plans): with the
// symmetric guard [H/2, uH - H/2) row 2j of the zero-padded inverse is row j of S1 over 2 -- never computed, never written: the
// C2R kernel reads S1 -- and row 2j+1 is the length-H inverse of F[k] t[k], t[k] = exp(-2 pi i k / 2H) (k < H/2), minus that above
// (the upper half of the spectrum sits uH - H rows higher: a half turn).  Half the inverse transform, half the writes, and a buffer
// of H TK instead of uH TK points: in place (fft_lds_inplace_tk), twice the tile width in the same LDS.  Odd rows at the
// reference's normalisation (1 / uH), H rows per tile.  grid (NT, 3), dynamic LDS lpad_size(H TK) complex.
template <int TK, typename C = float2>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value, 4) k_col_poly(ColParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    const int tid = threadIdx.x, T = blockDim.x;
    const int tile = blockIdx.x, c = blockIdx.y;
    const int H = p.H;
    const int ncol_valid = min(TK, p.ncols - tile * TK);
    const C* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    for (int e = tid; e < H * TK; e += T) {
        C v = mk<C>(S(0), S(0));
        if ((e % TK) < ncol_valid) v = src[e];
        a[lpad(e)] = v;
    }
    __syncthreads();
    fft_lds_inplace_tk<+1, COL_INPLACE_PT, TK>(a, p.planH, p.twH, tid, T);
    for (int e = tid; e < H * TK; e += T) {
        const int k = e / TK;
        C t = twid<-1>(p.twUH[k]);                            // exp(-2 pi i k / uH), uH = 2H
        if (k >= H / 2) t = mk<C>(-t.x, -t.y);
        a[lpad(e)] = cmul(a[lpad(e)], t);
    }
    __syncthreads();
    fft_lds_inplace_tk<-1, COL_INPLACE_PT, TK>(a, p.planH, p.twH, tid, T);
    C* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    for (int e = tid; e < H * TK; e += T)
        if ((e % TK) < ncol_valid) dst[e] = cscale(a[lpad(e)], p.inv_norm);
}

template <typename C> struct RowC2RParamsT {
    const C* S1;             // polyphase plans only (k_row_c2r_ct): even spectrum rows; S2 then holds the odd rows
    const C* S2;
    void* R;                 // dense [3][uH][uW] float, half or double
    const C* tw;             // uW-th roots
    StagePlan plan;          // n = uW
    int W, uW, uH;
    int TK, NT;
    int zlx, zrx;            // column-index read guard [zlx,zrx) (VkResample.cpp:1492-1493)
    scalar_t<C> inv_norm;    // 1/uW
    int poly;                // size-generic plans with u = 2 (k_col_poly): row 2j is row j of S1 (times 1/2), row 2j+1 row j of S2, H rows per tile each
};
using RowC2RParams = RowC2RParamsT<float2>;

// grid (uH/2, 3); dynamic LDS = 2 * lpad_size(uW) complex
template <bool HALF_OUT, typename C = float2, bool INPLACE = false>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value, INPLACE ? 4 : 1) k_row_c2r(RowC2RParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.uW);
    const int tid = threadIdx.x, T = blockDim.x;
    const int j = blockIdx.x, c = blockIdx.y;
    const int uW = p.uW;
    // rows 2j and 2j+1 of the spectrum after the column pass: consecutive rows of S2 -- or (poly: k_col_poly wrote the odd rows only)
    // row j of S1, which IS row 2j of the zero-padded inverse up to the factor 1/u = 1/2 (exact), and row j of the odd rows
    const long tile_stride = (long)(p.poly ? p.uH / 2 : p.uH) * p.TK;
    const C* baseA = (p.poly ? p.S1 + (long)j * p.TK : p.S2 + (long)(2 * j) * p.TK) + (long)c * p.NT * tile_stride;
    const C* baseB = (p.poly ? p.S2 + (long)j * p.TK : p.S2 + (long)(2 * j + 1) * p.TK) + (long)c * p.NT * tile_stride;
    const S sa = p.poly ? S(0.5) : S(1);
    // vkFFT.h:2059-2131: Z[k] = A + iB, Z[uW-k] = conj(A) + i conj(B); column index cidx = k-1
    for (int cidx = tid; cidx < uW / 2; cidx += T) {
        const int k = cidx + 1;
        C A = mk<C>(S(0), S(0)), B = A;
        if ((cidx < p.zlx || cidx >= p.zrx) && k <= p.W / 2) {
            const long o = (long)(k / p.TK) * tile_stride + (k % p.TK);
            A = cscale(baseA[o], sa);
            B = baseB[o];
        }
        a[lpad(k)] = mk<C>(A.x - B.y, A.y + B.x);
        a[lpad(uW - k)] = mk<C>(A.x + B.y, -A.y + B.x);
    }
    if (tid == 0) {
        C A = cscale(baseA[0], sa), B = baseB[0];
        a[lpad(0)] = mk<C>(A.x - B.y, A.y + B.x);
    }
    __syncthreads();
    const C* z = a;
    if constexpr (INPLACE) fft_lds_inplace<-1, 8>(a, p.plan, p.tw, tid, T);
    else z = fft_lds<-1, 1>(a, b, p.plan, p.tw, tid, T);
    const long plane = (long)uW * p.uH;
    for (int n = tid; n < uW; n += T) {
        C v = cscale(z[lpad(n)], p.inv_norm);
        if constexpr (HALF_OUT) {
            __half* R = (__half*)p.R + c * plane + (long)(2 * j) * uW;
            R[n] = __float2half_rn(v.x);
            R[uW + n] = __float2half_rn(v.y);
        } else {
            S* R = (S*)p.R + c * plane + (long)(2 * j) * uW;
            R[n] = v.x;
            R[uW + n] = v.y;
        }
    }
}

// ---------------------------------------------------------------- the non-R2C path
// VkResample.cpp:1424: beyond uW = 8192 (4096 for -p 1) the reference drops R2C/C2R and runs full complex transforms on
// a complex input whose imaginary parts it never initialises (VR:1620, 1647) -- defined as 0 here.  Same blocked
// spectrum layout with all W columns; the column kernel is k_col (its shift is the y half of the four-quadrant shift
// VR:527-546), the x half happens in the gather of the inverse row kernel together with the read guard
// [W/2, (2u-1) uW / 2u) of VR:1497-1498.
// grid (H, 3); dynamic LDS = 2 * lpad_size(W) complex
template <int MODE, typename C = float2, bool INPLACE = false>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value) k_row_c2c_fwd(RowR2CParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.W);
    const int tid = threadIdx.x, T = blockDim.x;
    const int y = blockIdx.x, c = blockIdx.y;
    const int W = p.W;
    for (int n = tid; n < W; n += T) a[lpad(n)] = mk<C>((S)load_px<MODE>(p, c, y, n), S(0));
    __syncthreads();
    const C* Z = a;
    if constexpr (INPLACE) fft_lds_inplace<+1, 16>(a, p.plan, p.tw, tid, T);       // rows beyond ~9600 points: one buffer (float plans; host-checked)
    else Z = fft_lds<+1, 1>(a, b, p.plan, p.tw, tid, T);
    const long tile_stride = (long)p.H * p.TK;
    C* base = p.S1 + (long)c * p.NT * tile_stride + (long)y * p.TK;
    for (int k = tid; k < W; k += T) base[(long)(k / p.TK) * tile_stride + (k % p.TK)] = Z[lpad(k)];
}

// grid (uH, 3); dynamic LDS = 2 * lpad_size(uW) complex.  R: complex [3][uH][uW]; HALF_OUT (-p 2 = half MEMORY only, VR:1420-1421
// set independently of performR2C): binary16 pairs -- the last write of the inverse is the one the reference's plan stores as
// half (VF:7282-7292: axis 0 of the inverse), the spectrum buffers stay float.
template <typename C = float2, bool HALF_OUT = false, bool INPLACE = false>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value) k_row_c2c_inv(RowC2RParamsT<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.uW);
    const int tid = threadIdx.x, T = blockDim.x;
    const int y = blockIdx.x, c = blockIdx.y;
    const int uW = p.uW, W = p.W;
    const long tile_stride = (long)p.uH * p.TK;
    const C* base = p.S2 + (long)c * p.NT * tile_stride + (long)y * p.TK;
    for (int kx = tid; kx < uW; kx += T) {
        C v = mk<C>(S(0), S(0));
        if (!(kx >= p.zlx && kx < p.zrx)) {
            // columns >= W/2 of the forward spectrum sit uW - W further on (race-free form of the in-place shift)
            int k = -1;
            if (kx >= uW - W / 2) k = kx - (uW - W);
            else if (kx < W) k = kx;
            if (k >= 0) v = base[(long)(k / p.TK) * tile_stride + (k % p.TK)];
        }
        a[lpad(kx)] = v;
    }
    __syncthreads();
    const C* z = a;
    if constexpr (INPLACE) fft_lds_inplace<-1, 16>(a, p.plan, p.tw, tid, T);
    else z = fft_lds<-1, 1>(a, b, p.plan, p.tw, tid, T);
    if constexpr (HALF_OUT) {
        __half2* R = (__half2*)p.R + ((long)c * p.uH + y) * uW;
        for (int n = tid; n < uW; n += T) {
            const C v = cscale(z[lpad(n)], p.inv_norm);
            R[n] = __floats2half2_rn((float)v.x, (float)v.y);
        }
    } else {
        C* R = (C*)p.R + ((long)c * p.uH + y) * uW;
        for (int n = tid; n < uW; n += T) R[n] = cscale(z[lpad(n)], p.inv_norm);
    }
}

// ---------------------------------------------------------------- non-R2C rows longer than an LDS buffer: four steps through HBM
// Beyond 16 384 points (about 4 800 for -p 1) a complex row does not fit the 160 KB of LDS even once; the reference
// switches such axes to multi-upload plans -- several dispatches with a transposition through a temporary buffer and the
// "four-step" twiddles between them (vkFFT.h:4773-4992, 2290-2388, 6562-6576).  Same decomposition here, in two launches per
// direction: N = N1 * N2, input index i = N2 i1 + i2, output index o = o1 + N1 o2,
//     X[o1 + N1 o2] = sum_{i2} w_N2^(i2 o2) * [ w_N^(i2 o1) * sum_{i1} x[N2 i1 + i2] w_N1^(i1 o1) ]          (w_n = exp(DIR 2 pi i / n))
//   pass A (k_row4_a): for TK consecutive i2 of one row: load (pixels, or the spectrum row with the x half of the shift and
//                      the read guard, as k_row_c2c_inv), transform over i1 (length N1), multiply by w_N^(i2 o1), write T[o1][i2];
//   pass B (k_row4_b): for TK consecutive o1 of one row: read T[o1][.] (contiguous), transform over i2 (length N2), store
//                      element o1 + N1 o2 (blocked spectrum, or the complex pre-sharpen image scaled by 1/uW).
// T: one complex row matrix per image row, [3][rows][N] in HBM (the reference's temporary buffer of those plans).
// COLUMNS longer than the LDS (uH beyond ~9 600) run through the same two kernels: such plans keep the spectrum with tiles of ONE
// column (TK = 1: a column is a dense sequence), forward in place (S1 -> T -> S1), inverse with the y half of the shift and the
// read guard in pass A's load (S1 -> T -> S2, scaled by 1/uH) -- what k_col does in one launch.
template <typename C> struct Row4Params {
    const void* in;          // pass A, pixel modes: planar float/half/double or u8 RGB
    const C* spec;           // pass A, IN4_TILES: blocked spectrum S2 (uH rows); IN4_DENSE / IN4_DENSE_SHIFT: dense [3][rows][W]
    C* T;                    // scratch [3][rows][N]
    C* S1;                   // pass B, OUT4_TILES: blocked spectrum (H rows)
    void* R;                 // pass B, OUT4_DENSE / _HALF: dense [3][rows][N], scaled by inv_norm (binary16 pairs for _HALF)
    const C *tw1, *tw2, *twN;   // N1-th, N2-th, N-th roots
    StagePlan plan1, plan2;
    int N, N1, N2, rows;     // sequence length = N1 * N2; sequences per plane
    long in_row_stride, in_plane_stride;
    int W, TK, NT;           // length of the un-padded sequence (shift / guard); spectrum tile width and count
    int zlx, zrx;            // read guard [zlx, zrx) (VkResample.cpp:1494-1498)
    scalar_t<C> inv_norm;    // pass B scale
};
// what pass A loads (beyond the pixel modes of InMode) and what pass B stores
enum { IN4_TILES = 8,        // inverse rows: spectrum row y from the blocked spectrum, x half of the shift + read guard (k_row_c2c_inv)
       IN4_DENSE = 9,        // forward columns: sequence y of a dense [3][rows][N] array
       IN4_DENSE_SHIFT = 10  // inverse columns: the same with the y half of the shift + read guard (k_col), source length W
};
enum { OUT4_TILES = 0, OUT4_DENSE = 1, OUT4_HALF = 2 };

// grid (rows, N2 / TKS, 3); dynamic LDS = 2 * lpad_size(N1 * TKS) complex
template <int DIR, int TKS, int MODE, typename C = float2>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value) k_row4_a(Row4Params<C> p)
{
    using S = scalar_t<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.N1 * TKS);
    const int tid = threadIdx.x, T = blockDim.x;
    const int i20 = blockIdx.y * TKS, y = blockIdx.x, c = blockIdx.z;
    const int N = p.N, N1 = p.N1, N2 = p.N2;
    const long tile_stride = (long)p.rows * p.TK;
    for (int e = tid; e < N1 * TKS; e += T) {
        const int i1 = e / TKS, col = e % TKS;
        const int i = N2 * i1 + i20 + col;
        C v = mk<C>(S(0), S(0));
        if constexpr (MODE == IN4_DENSE) {
            v = p.spec[((long)c * p.rows + y) * N + i];
        } else if constexpr (MODE == IN4_TILES || MODE == IN4_DENSE_SHIFT) {
            if (!(i >= p.zlx && i < p.zrx)) {
                int k = -1;                               // elements >= W/2 of the un-padded spectrum sit N - W further on
                if (i >= N - p.W / 2) k = i - (N - p.W);
                else if (i < p.W) k = i;
                if (k >= 0) {
                    if constexpr (MODE == IN4_TILES) v = p.spec[(long)c * p.NT * tile_stride + (long)(k / p.TK) * tile_stride + (long)y * p.TK + (k % p.TK)];
                    else v = p.spec[((long)c * p.rows + y) * p.W + k];
                }
            }
        } else {
            v.x = (S)load_px<MODE>(p, c, y, i);           // (imaginary input: defined as 0, see k_row_c2c_fwd)
        }
        a[lpad(e)] = v;
    }
    __syncthreads();
    const C* Z = fft_lds<DIR, TKS>(a, b, p.plan1, p.tw1, tid, T);
    C* dst = p.T + ((long)c * p.rows + y) * N;
    for (int e = tid; e < N1 * TKS; e += T) {
        const int o1 = e / TKS, col = e % TKS, i2 = i20 + col;
        dst[(long)o1 * N2 + i2] = cmul(Z[lpad(e)], twid<DIR>(p.twN[o1 * i2]));      // o1 i2 < N1 N2 = N
    }
}

// grid (rows, N1 / TKS, 3); dynamic LDS = 2 * lpad_size(N2 * TKS) complex
template <int DIR, int TKS, int OUT, typename C = float2>
__global__ void __launch_bounds__(GenericMaxThreads<C>::value) k_row4_b(Row4Params<C> p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* a = (C*)smem;
    C* b = a + lpad_size(p.N2 * TKS);
    const int tid = threadIdx.x, T = blockDim.x;
    const int o10 = blockIdx.y * TKS, y = blockIdx.x, c = blockIdx.z;
    const int N = p.N, N1 = p.N1, N2 = p.N2;
    const C* src = p.T + ((long)c * p.rows + y) * N + (long)o10 * N2;
    for (int e = tid; e < N2 * TKS; e += T) {
        const int col = e / N2, i2 = e % N2;             // (consecutive threads read consecutive i2 of one o1)
        a[lpad(i2 * TKS + col)] = src[(long)col * N2 + i2];
    }
    __syncthreads();
    const C* Z = fft_lds<DIR, TKS>(a, b, p.plan2, p.tw2, tid, T);
    const long tile_stride = (long)p.rows * p.TK;
    for (int e = tid; e < N2 * TKS; e += T) {
        const int o2 = e / TKS, col = e % TKS;
        const int o = o10 + col + N1 * o2;
        const C z = Z[lpad(e)];
        if constexpr (OUT == OUT4_TILES) {
            p.S1[(long)c * p.NT * tile_stride + (long)(o / p.TK) * tile_stride + (long)y * p.TK + (o % p.TK)] = z;
        } else {
            const C v = cscale(z, p.inv_norm);
            if constexpr (OUT == OUT4_HALF) ((__half2*)p.R)[((long)c * p.rows + y) * N + o] = __floats2half2_rn((float)v.x, (float)v.y);
            else ((C*)p.R)[((long)c * p.rows + y) * N + o] = v;
        }
    }
}

// ---------------------------------------------------------------- sharpen (VkResample.cpp:819-925)
struct SharpenParams {
    const void* R;           // dense [3][uH][uW]
    void* out;               // dense [3][uH][uW]
    int uW, uH;
    float upsq, coef;        // constants as the shader sees them ("%f" text, VkResample.cpp:893-920)
};

// arithmetic policy: fp32, or "every operation rounded to binary16" (GLSL float16_t, -p 2)
template <bool HALF> struct Arith {
    static __device__ __forceinline__ float r(float x)
    {
        if constexpr (HALF) return __half2float(__float2half_rn(x));
        else return x;
    }
};

template <bool HALF>
__device__ __forceinline__ float sharpen_px(const float* len, float coef)
{
    using A = Arith<HALF>;
    float mn0 = fminf(len[1], fminf(len[3], fminf(len[4], fminf(len[5], len[7]))));
    float mn1 = fminf(mn0, fminf(len[0], fminf(len[2], fminf(len[6], len[8]))));
    float mx0 = fmaxf(len[1], fmaxf(len[3], fmaxf(len[4], fmaxf(len[5], len[7]))));
    float mx1 = fmaxf(mx0, fmaxf(len[0], fmaxf(len[2], fmaxf(len[6], len[8]))));
    float minlen = A::r(0.5f * A::r(mn0 + mn1));
    float maxlen = A::r(0.5f * A::r(mx0 + mx1));
    minlen = A::r(__fdiv_rn(minlen, A::r(1.0f - minlen)));
    maxlen = A::r(__fdiv_rn(A::r(1.0f - maxlen), maxlen));
    float scale = (minlen < maxlen) ? minlen : maxlen;
    scale = A::r(-coef * A::r(__fsqrt_rn(scale)));
    float s4 = A::r(A::r(A::r(len[1] + len[3]) + len[5]) + len[7]);
    float num = A::r(len[4] + A::r(scale * s4));
    float den = A::r(1.0f + A::r(scale * 4.0f));
    return A::r(__fdiv_rn(num, den));
}

// one thread = 4 consecutive pixels of one row; grid (ceil(uW/4/256), uH, 3)
template <bool HALF>
__global__ void __launch_bounds__(256) k_sharpen(SharpenParams p)
{
    using A = Arith<HALF>;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y, c = blockIdx.z;
    const int uW = p.uW, uH = p.uH;
    if (x0 >= uW) return;
    const long plane = (long)uW * uH;
    const int ym = y > 0 ? y - 1 : y;
    const int rows[3] = {ym, y, y + 1};          // no upper clamp (VkResample.cpp:891-892)
    float L[3][6];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            int x = x0 - 1 + i;
            if (x < 0) x = 0;                    // id_x_m clamp (VkResample.cpp:889)
            long f = (long)rows[r] * uW + x;     // x == uW wraps to the next row (quirk B5)
            while (f >= plane) f -= uW;          // reads past the plane: same column, last row (see oracle)
            float t;
            if constexpr (HALF) t = __half2float(((const __half*)p.R)[c * plane + f]);
            else t = ((const float*)p.R)[c * plane + f];
            t = fabsf(A::r(p.upsq * t));
            L[r][i] = fminf(fmaxf(t, 0.0f), 1.0f);
        }
    }
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float len[9] = {L[0][i], L[0][i + 1], L[0][i + 2], L[1][i], L[1][i + 1], L[1][i + 2],
                        L[2][i], L[2][i + 1], L[2][i + 2]};
        o[i] = sharpen_px<HALF>(len, p.coef);
    }
    const long of = c * plane + (long)y * uW + x0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (x0 + i < uW) {
            if constexpr (HALF) ((__half*)p.out)[of + i] = __float2half_rn(o[i]);
            else ((float*)p.out)[of + i] = o[i];
        }
    }
}

// ---------------------------------------------------------------- host-loop replacements
// VkResample.cpp:1636-1685: u8 interleaved -> planar float/half (row stride W, plane stride ps)
template <bool HALF>
__global__ void __launch_bounds__(256) k_unpack_u8(const uint8_t* rgb, long row_stride_bytes, void* planes,
                                                    int W, int H, long plane_stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* s = rgb + (long)y * row_stride_bytes + 3l * x;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if constexpr (HALF) ((__half*)planes)[c * plane_stride + (long)y * W + x] = __float2half_rn(div255((float)s[c]));
        else ((float*)planes)[c * plane_stride + (long)y * W + x] = cvt_u8_f32(s[c]);
    }
}

// u8 = (unsigned char)(255.0 * x) of VkResample.cpp:1715: the product in double (exact for a float x), the C cast = truncation;
// out of range (undefined in the reference): saturating, or wrapping like the x86 cast (FFTUP_FLAG_U8_WRAP)
__device__ __forceinline__ uint8_t cvt_f_u8(float v, int wrap)
{
    const double d = 255.0 * (double)v;
    if (wrap) return (d > -2147483648.0 && d < 2147483648.0) ? (uint8_t)(((int)d) & 0xFF) : 0;
    return !(d > 0.0) ? 0 : (d >= 255.0 ? 255 : (uint8_t)d);
}

// The same for four pixels in registers (the fused kernel's 8-bit store): saturating form without double arithmetic.
// trunc(255 x) of the EXACT product = trunc of the product rounded TOWARD ZERO (an integer n <= 255 x is representable, so the
// rounded product cannot fall below it); and v_cvt_pk_u8_f32 -- float to a saturated byte, negative and NaN -> 0 -- follows the
// rounding mode of the MODE register (measured: tools/ub/cvt_pk_u8.hip -- to nearest even by default, truncating under
// round-toward-zero).  So: four v_mul_f32 and four v_cvt_pk_u8_f32 between two changes of the rounding mode -- one asm statement,
// the compiler cannot move anything affected in between.  Bit for bit cvt_f_u8 (tests: the fused store against planes +
// k_pack_u8 on whole frames).
__device__ __forceinline__ void cvt4_f_u8(float a, float b, float c, float d, int wrap, uint8_t (&o)[4])
{
    if (wrap) { o[0] = cvt_f_u8(a, 1); o[1] = cvt_f_u8(b, 1); o[2] = cvt_f_u8(c, 1); o[3] = cvt_f_u8(d, 1); return; }      // (wave-uniform)
    unsigned ta, tb, tc, td;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_mul_f32 %0, 0x437f0000, %4\n\tv_mul_f32 %1, 0x437f0000, %5\n\tv_mul_f32 %2, 0x437f0000, %6\n\tv_mul_f32 %3, 0x437f0000, %7\n\t"
                 "v_cvt_pk_u8_f32 %0, %0, 0, 0\n\tv_cvt_pk_u8_f32 %1, %1, 0, 0\n\tv_cvt_pk_u8_f32 %2, %2, 0, 0\n\tv_cvt_pk_u8_f32 %3, %3, 0, 0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(td) : "v"(a), "v"(b), "v"(c), "v"(d));
    o[0] = (uint8_t)ta; o[1] = (uint8_t)tb; o[2] = (uint8_t)tc; o[3] = (uint8_t)td;
}
// ... and from two binary16 pairs (-p 2): v_fma_mix_f32 converts the half and multiplies it in one instruction (255 x of a
// binary16 x is exact in fp32), v_cvt_pk_u8_f32 truncates under round-toward-zero: 8 instead of 16 vector instructions.
typedef _Float16 cvt_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cvt4_h_u8(cvt_h2 ab, cvt_h2 cd, int wrap, uint8_t (&o)[4])
{
    if (wrap) { o[0] = cvt_f_u8((float)ab.x, 1); o[1] = cvt_f_u8((float)ab.y, 1); o[2] = cvt_f_u8((float)cd.x, 1); o[3] = cvt_f_u8((float)cd.y, 1); return; }
    unsigned ta, tb, tc, td;
    const float k255 = 255.0f;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_fma_mix_f32 %0, %4, %6, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, %6, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, %5, %6, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %5, %6, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_u8_f32 %0, %0, 0, 0\n\tv_cvt_pk_u8_f32 %1, %1, 0, 0\n\tv_cvt_pk_u8_f32 %2, %2, 0, 0\n\tv_cvt_pk_u8_f32 %3, %3, 0, 0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(td) : "v"(ab), "v"(cd), "s"(k255));
    o[0] = (uint8_t)ta; o[1] = (uint8_t)tb; o[2] = (uint8_t)tc; o[3] = (uint8_t)td;
}

// VkResample.cpp:1708-1748: planar float/half -> u8 interleaved, u8 = (unsigned char)(255.0*x).  One thread = four consecutive
// pixels of a row: three 16-byte (8-byte) plane loads, twelve bytes out as three dwords -- a wave writes 768 contiguous bytes.
// (One thread per pixel with three byte stores each took 20 us for the 4096x2048 image; the stores, not the bytes, were the cost.)
// grid (ceil(uW / 1024), uH), block 256.
template <bool HALF>
__global__ void __launch_bounds__(256) k_pack_u8(const void* planes, uint8_t* rgb, int uW, int uH, int wrap)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= uW) return;
    const long plane = (long)uW * uH, at = (long)y * uW + x;
    uint8_t* dst = rgb + at * 3;
    if (x + 4 <= uW && (uW & 3) == 0) {                     // whole, aligned quad (rows of a multiple of 4 pixels: every quad)
        uint8_t b[3][4];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if constexpr (HALF) {
                const uint2 r = *(const uint2*)((const __half*)planes + c * plane + at);
                cvt4_h_u8(__builtin_bit_cast(cvt_h2, r.x), __builtin_bit_cast(cvt_h2, r.y), wrap, b[c]);
            } else {
                const float4 r = *(const float4*)((const float*)planes + c * plane + at);
                cvt4_f_u8(r.x, r.y, r.z, r.w, wrap, b[c]);
            }
        }
        unsigned o[3] = {0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int k = 3 * i + c;
                o[k >> 2] |= (unsigned)b[c][i] << (8 * (k & 3));
            }
        unsigned* d = (unsigned*)dst;                       // (12 x: a multiple of 4 bytes into a 16-byte aligned image)
        d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
        return;
    }
    for (int i = 0; i < 4 && x + i < uW; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v;
            if constexpr (HALF) v = __half2float(((const __half*)planes)[c * plane + at + i]);
            else v = ((const float*)planes)[c * plane + at + i];
            dst[3 * i + c] = cvt_f_u8(v, wrap);
        }
}

// ---------------------------------------------------------------- -p 1 (double) variants
// The GLSL the reference generates for -p 1 declares double/dvec2 buffers but keeps its literals unsuffixed, i.e.
// float constants promoted to double (VkResample.cpp:893-920): upsq and coef arrive here as such floats.
__device__ __forceinline__ double sharpen_px_f64(const double* len, double coef)
{
    double mn0 = fmin(len[1], fmin(len[3], fmin(len[4], fmin(len[5], len[7]))));
    double mn1 = fmin(mn0, fmin(len[0], fmin(len[2], fmin(len[6], len[8]))));
    double mx0 = fmax(len[1], fmax(len[3], fmax(len[4], fmax(len[5], len[7]))));
    double mx1 = fmax(mx0, fmax(len[0], fmax(len[2], fmax(len[6], len[8]))));
    double minlen = 0.5 * (mn0 + mn1);
    double maxlen = 0.5 * (mx0 + mx1);
    minlen = minlen / (1.0 - minlen);
    maxlen = (1.0 - maxlen) / maxlen;
    double scale = (minlen < maxlen) ? minlen : maxlen;
    scale = -coef * sqrt(scale);
    return (len[4] + scale * (((len[1] + len[3]) + len[5]) + len[7])) / (1.0 + scale * 4.0);
}

// The filter's three divisions and its root without IEEE division sequences (3 x ~14 + ~20 instructions per pixel made
// k_sharpen_f64 a 170 us kernel for 402 MB of traffic).  As in the fp32 kernels (sharpen_eval_pair): a < b <=> mn + mx < 1, so ONE
// quotient n / d is formed, n = min(mn, 1 - mx), d = 1 - n in [0.5, 1], everything carried doubled; sqrt(n / d) = n rsq(n d).
// v_rsq_f64 / v_rcp_f64 deliver ~2^-23: two coupled Newton steps for the root (Goldschmidt form), one for the reciprocal plus the
// residual correction of the quotient: <= 2 ulp of the correctly rounded sequence (GLSL asks 2.5 ulp of one double division;
// tests: 1e-9 against the oracle after the filter's sqrt amplification near 0).
__device__ __forceinline__ double min_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double sharpen_eval_f64(double s4, double C, double mn0, double mn1, double mx0, double mx1, double m2coef)
{
    const double smn = mn0 + mn1, smx = mx0 + mx1;           // 2 mn, 2 mx
    const double n2 = min_f64(smn, 2.0 - smx);               // 2 n
    const double d2 = 2.0 - n2;                              // 2 d (exact: see sharpen_eval_pair)
    const double pr = fma(n2, d2, 1e-300);                   // 4 n d; n = 0 -> root 0, no NaN
    const double y = __builtin_amdgcn_rsq(pr);
    double g = pr * y, h = 0.5 * y;                          // g -> sqrt(pr), h -> 1 / (2 sqrt(pr))
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    h = fma(h, r, h);
    const double scale = m2coef * (n2 * h);                  // -coef sqrt(n / d) = -coef n2 / sqrt(4 n d) = (-2 coef) n2 h
    const double num = fma(scale, s4, C), den = fma(scale, 4.0, 1.0);
    double yd = __builtin_amdgcn_rcp(den);
    yd = fma(fma(-den, yd, 1.0), yd, yd);
    const double q = num * yd;
    return fma(fma(-den, q, num), yd, q);
}

// min(|upsq x|, 1) = |upsq| |x| clamped to [0, 1] by the output modifier: one instruction per tap (written with fmin / fmax the
// compiler adds a canonicalising v_max x, x in front of every chain -- it cannot know the loaded value is no signalling NaN)
__device__ __forceinline__ double tap_f64(double upsq, double x)
{
    double r;
    asm("v_mul_f64 %0, |%1|, |%2| clamp" : "=v"(r) : "v"(upsq), "v"(x));
    return r;
}
// one thread = 2 consecutive pixels x RPT rows (the three tap rows slide down); block (64, 4), grid (ceil(uW/512), ceil(uH/RPT), 3).
// Even widths: ONE 16-byte load and ONE 16-byte store per thread and row -- a wave reads and writes 1 KB runs -- the left / right
// taps come from the neighbouring lanes (DPP wave shifts, as the fp32 kernels do), only the lanes at the ends of a wave or a row
// load theirs.  (8-byte loads and stores at a 16-byte stride -- four loads per pair, every 64-byte line of the output written in
// two instructions -- held the kernel at 107 us whatever its arithmetic.)  Odd widths: element-wise taps.  Planes below 2^31 elements.
constexpr int SHARPEN_F64_RPT = 16;
__device__ __forceinline__ double dpp_f64(double v, double edge, bool from_below)
{
    const int2 a = __builtin_bit_cast(int2, v), e = __builtin_bit_cast(int2, edge);
    int2 r;
    if (from_below) { r.x = __builtin_amdgcn_update_dpp(e.x, a.x, 0x138, 0xf, 0xf, false); r.y = __builtin_amdgcn_update_dpp(e.y, a.y, 0x138, 0xf, 0xf, false); }
    else { r.x = __builtin_amdgcn_update_dpp(e.x, a.x, 0x130, 0xf, 0xf, false); r.y = __builtin_amdgcn_update_dpp(e.y, a.y, 0x130, 0xf, 0xf, false); }
    return __builtin_bit_cast(double, r);
}
template <bool EVEN>
__device__ __forceinline__ void sharpen_f64_row(double (&L)[4], const double* __restrict__ R, unsigned plane, unsigned uW, int uH, int row,
                                                const unsigned (&xo)[4], bool edge_l, bool edge_r, double upsq)
{
    // taps x0-1 (clamped at 0, VkResample.cpp:889) .. x0+2 (no upper clamp: x == uW is the next row's first pixel, quirk B5);
    // reads past the plane: same column, last written row (see oracle) -- row <= uH and x <= uW here, so clamping the row and one
    // conditional step back (only the right taps can need it) cover every case the oracle's loop does
    const unsigned base = (unsigned)(row < uH ? row : uH - 1) * uW;
    if constexpr (EVEN) {
        const double2 v = *(const double2*)(R + base + xo[1]);
        L[1] = tap_f64(upsq, v.x);
        L[2] = tap_f64(upsq, v.y);
        L[0] = dpp_f64(L[2], 0.0, true);
        L[3] = dpp_f64(L[1], 0.0, false);
        if (edge_l) L[0] = tap_f64(upsq, R[base + xo[0]]);
        if (edge_r) {
            unsigned f = base + xo[3];
            f = f >= plane ? f - uW : f;
            L[3] = tap_f64(upsq, R[f]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned f = base + xo[i];
            if (i >= 2) f = f >= plane ? f - uW : f;
            L[i] = tap_f64(upsq, R[f]);
        }
    }
}
// EXACT (test builds, FFTUP_EXPERIMENT f64_exact_sharpen=1): the filter as the shader writes it, IEEE divisions and root
// (sharpen_px_f64) -- what sharpen_eval_f64 is measured against in ulps (tests/test_gpu_parity.py)
template <bool EVEN, bool EXACT = false>
__global__ void __launch_bounds__(256) k_sharpen_f64(SharpenParams p)
{
    constexpr int RPT = SHARPEN_F64_RPT;
    const int lane = threadIdx.x;
    const int x0 = ((blockIdx.x * 4 + threadIdx.y) * 64 + lane) * 2;      // (a workgroup: 512 pixels = 4 KB of RPT rows)
    const int y0 = blockIdx.y * RPT;
    const int c = blockIdx.z;
    const int uW = p.uW, uH = p.uH;
    if (x0 >= uW) return;
    const unsigned plane = (unsigned)uW * (unsigned)uH;
    const double* R = (const double*)p.R + (long)c * plane;
    double* out = (double*)p.out + (long)c * plane;
    const double upsq = (double)p.upsq, m2coef = -2.0 * (double)p.coef;
    // (x0 + 2 > uW: the tap of a pixel beyond the row -- odd widths -- unused)
    const unsigned xo[4] = {(unsigned)(x0 > 0 ? x0 - 1 : 0), (unsigned)x0, (unsigned)(x0 + 1), (unsigned)(x0 + 2 > uW ? uW : x0 + 2)};
    const bool edge_l = lane == 0, edge_r = lane == 63 || x0 + 2 >= uW;      // (the lane above has left the kernel)
    double a[4], b[4], cc[4];
    sharpen_f64_row<EVEN>(a, R, plane, uW, uH, y0 > 0 ? y0 - 1 : 0, xo, edge_l, edge_r, upsq);
    sharpen_f64_row<EVEN>(b, R, plane, uW, uH, y0, xo, edge_l, edge_r, upsq);
#pragma unroll 2
    for (int r = 0; r < RPT; r++) {
        const int y = y0 + r;
        if (y >= uH) break;
        sharpen_f64_row<EVEN>(cc, R, plane, uW, uH, y + 1, xo, edge_l, edge_r, upsq);
        double vmn[4], vmx[4], o[2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            vmn[i] = min_f64(min_f64(a[i], b[i]), cc[i]);
            vmx[i] = max_f64(max_f64(a[i], b[i]), cc[i]);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const double mn1 = min_f64(min_f64(vmn[k], vmn[k + 1]), vmn[k + 2]), mx1 = max_f64(max_f64(vmx[k], vmx[k + 1]), vmx[k + 2]);
            const double mn0 = min_f64(min_f64(vmn[k + 1], b[k]), b[k + 2]), mx0 = max_f64(max_f64(vmx[k + 1], b[k]), b[k + 2]);
            // (len[1] + len[3]) + len[5]) + len[7] = ((N + W) + E) + S, left to right as the shader (VkResample.cpp:921)
            const double s4 = ((a[k + 1] + b[k]) + b[k + 2]) + cc[k + 1];
            if constexpr (EXACT) {
                const double len[9] = {a[k], a[k + 1], a[k + 2], b[k], b[k + 1], b[k + 2], cc[k], cc[k + 1], cc[k + 2]};
                o[k] = sharpen_px_f64(len, (double)p.coef);
            } else {
                o[k] = sharpen_eval_f64(s4, b[k + 1], mn0, mn1, mx0, mx1, m2coef);
            }
        }
        double* dst = out + ((unsigned)y * (unsigned)uW + (unsigned)x0);
        if constexpr (EVEN) {
            typedef double d2v __attribute__((ext_vector_type(2)));
            d2v val = {o[0], o[1]};
            __builtin_nontemporal_store(val, (d2v*)dst);
        } else {
            dst[0] = o[0];
            if (x0 + 1 < uW) dst[1] = o[1];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = b[i]; b[i] = cc[i]; }
    }
}

// VkResample.cpp:1650-1668: x = (double)v / 255.0
__global__ void __launch_bounds__(256) k_unpack_u8_f64(const uint8_t* rgb, long row_stride_bytes, double* planes, int W, int H,
                                                        long plane_stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* s = rgb + (long)y * row_stride_bytes + 3l * x;
#pragma unroll
    for (int c = 0; c < 3; c++) planes[c * plane_stride + (long)y * W + x] = __ddiv_rn((double)s[c], 255.0);
}

// VkResample.cpp:1722-1734: u8 = (unsigned char)(255.0 * x)
__global__ void __launch_bounds__(256) k_pack_u8_f64(const double* planes, uint8_t* rgb, int uW, int uH, int wrap)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= uW) return;
    const long plane = (long)uW * uH;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double d = 255.0 * planes[c * plane + (long)y * uW + x];
        uint8_t o;
        if (wrap) o = (d > -2147483648.0 && d < 2147483648.0) ? (uint8_t)(((int)d) & 0xFF) : 0;
        else o = !(d > 0.0) ? 0 : (d >= 255.0 ? 255 : (uint8_t)d);
        rgb[((long)y * uW + x) * 3 + c] = o;
    }
}

// sharpen on the complex image of the non-R2C path: len = length(u^2 z) (VkResample.cpp:865-907 with vec2 inputs);
// one thread = one pixel.  HALF (-p 2): f16vec2 inputs, float16_t arithmetic -- every operation of tex = u^2 * z,
// length() = sqrt(x*x + y*y) and of the filter rounded to binary16 like the oracle's -- binary16 output.
template <typename C = float2, bool HALF = false>
__global__ void __launch_bounds__(256) k_sharpen_c(SharpenParams p)
{
    using S = scalar_t<C>;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, c = blockIdx.z;
    const int uW = p.uW, uH = p.uH;
    if (x >= uW) return;
    const long plane = (long)uW * uH;
    const C* R = (const C*)p.R + c * plane;
    const int xs[3] = {x > 0 ? x - 1 : x, x, x + 1};
    const int ys[3] = {y > 0 ? y - 1 : y, y, y + 1};
    if constexpr (HALF) {
        using A = Arith<true>;
        const __half2* Rh = (const __half2*)p.R + c * plane;
        float len[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                long f = (long)ys[a] * uW + xs[b];
                while (f >= plane) f -= uW;
                const float2 z = __half22float2(Rh[f]);
                const float tr = A::r(p.upsq * z.x), ti = A::r(p.upsq * z.y);
                const float l = A::r(__fsqrt_rn(A::r(A::r(tr * tr) + A::r(ti * ti))));
                len[a * 3 + b] = l > 1.0f ? 1.0f : (l < 0.0f ? 0.0f : l);
            }
        ((__half*)p.out)[c * plane + (long)y * uW + x] = __float2half_rn(sharpen_px<true>(len, p.coef));
        return;
    }
    S len[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            long f = (long)ys[a] * uW + xs[b];       // x == uW wraps to the next row (quirk B5)
            while (f >= plane) f -= uW;              // reads past the plane: same column, last row (see oracle)
            const C z = R[f];
            const S tr = (S)p.upsq * z.x, ti = (S)p.upsq * z.y;
            S l;
            if constexpr (sizeof(S) == 8) l = sqrt(tr * tr + ti * ti);
            else l = __fsqrt_rn(tr * tr + ti * ti);
            len[a * 3 + b] = l > S(1) ? S(1) : (l < S(0) ? S(0) : l);
        }
    if constexpr (sizeof(S) == 8) ((double*)p.out)[c * plane + (long)y * uW + x] = sharpen_px_f64(len, (double)p.coef);
    else ((float*)p.out)[c * plane + (long)y * uW + x] = sharpen_px<false>(len, p.coef);
}

}  // namespace fftup
