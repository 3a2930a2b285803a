// fft_engine.hpp -- LDS-staged Stockham FFT building blocks for gfx950 (wave64).
//
// Radix-2/3/4/5/7/8 butterflies and one Stockham autosort stage that works on TK interleaved
// sequences held in LDS as complex [n][TK].  Sign convention follows the reference: DIR=+1 is
// exp(+2 pi i nk/N) (VkFFT "forward", vkFFT.h:4545/751), DIR=-1 the inverse kernel; the 1/N
// of the inverse (vkFFT.h:2921-2923) is applied by the caller when it stores the result.
// Twiddles come from a per-length table tw[k] = exp(+2 pi i k/N) computed on the host in double
// and rounded once to fp32 (the reference evaluates fp32 cos/sin in-shader, vkFFT.h:2417-2421).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fftup {

// Spectrum stores of the specialised row and column kernels: WRITE-THROUGH (sc1).  With the default policy the 25 MB of spectrum
// a row or column pass writes sit dirty in the XCDs' L2s when the kernel ends, and the end-of-kernel write-back is serial time
// in front of the next launch of the stream: row pass 9.0 -> 7.5 us, column pass 14.0 -> 11.5 us, the frame of ordered
// iterations 78.0 -> 75 us at 2048x1024 (profiles/r06_b_wt_stores.txt; the frame of overlapped iterations does not move).  The
// output image keeps its non-temporal stores: written through, its lines leave the L2s no earlier but the NEXT kernel pays (row
// pass behind it 9.9 -> 18.7 us, overlapped frame +2.3 us; same file).  -DFFTUP_SPECTRUM_WT=0 builds the default-policy stores.
#ifndef FFTUP_SPECTRUM_WT
#define FFTUP_SPECTRUM_WT 1
#endif
typedef float st_f4 __attribute__((ext_vector_type(4)));
typedef float st_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void spec_store16(float2* dst, float2 a, float2 b)       // dst 16-byte aligned
{
#if FFTUP_SPECTRUM_WT
    const st_f4 v = {a.x, a.y, b.x, b.y};
    // (s_nop 1: a store of more than 64 bits reads its data registers late -- the compiler, which does not know that this is a
    // store, may overwrite them in the very next instruction)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#else
    *(float4*)dst = make_float4(a.x, a.y, b.x, b.y);
#endif
}
__device__ __forceinline__ void spec_store8(float2* dst, float2 a)
{
#if FFTUP_SPECTRUM_WT
    const st_f2 v = {a.x, a.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#else
    *dst = a;
#endif
}
// (The size-generic kernels keep default-policy stores: written through, their frames do not move -- 174.5 against 173.6 us at
// 2048x1024, 261 against 259 us at -p 1; profiles/r06_k_wt_generic.txt -- they are not short of memory bandwidth.)

struct StagePlan {           // radix sequence of one 1-D transform (product = N)
    int32_t n;
    int32_t nstages;
    uint8_t radix[16];
};

// complex arithmetic on float2 (fp32 plans) or double2 (-p 1 plans); C is deduced at every call site
template <typename C> using scalar_t = decltype(C::x);
template <typename C> __device__ __forceinline__ C mk(scalar_t<C> x, scalar_t<C> y)
{
    C r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ float ffma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double ffma(double a, double b, double c) { return fma(a, b, c); }
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { return mk<C>(a.x + b.x, a.y + b.y); }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { return mk<C>(a.x - b.x, a.y - b.y); }
template <typename C> __device__ __forceinline__ C cmul(C a, C b)
{
    return mk<C>(ffma(a.x, b.x, -a.y * b.y), ffma(a.x, b.y, a.y * b.x));
}
template <typename C> __device__ __forceinline__ C cscale(C a, scalar_t<C> s) { return mk<C>(a.x * s, a.y * s); }
// multiply by DIR*i
template <int DIR, typename C> __device__ __forceinline__ C mul_i(C a)
{
    return DIR > 0 ? mk<C>(-a.y, a.x) : mk<C>(a.y, -a.x);
}
template <int DIR, typename C> __device__ __forceinline__ C twid(C w)   // table holds exp(+i..)
{
    return DIR > 0 ? w : mk<C>(w.x, -w.y);
}

// LDS index padding: one element of padding per 16 (keeps stride-R*TK scatter writes of the
// first Stockham stages off a single pair of banks; ds_write_b64 banks = (addr/4) % 32).
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int lpad_size(int n) { return n + (n >> 4) + 1; }

// ---------------------------------------------------------------- butterflies (in place on v[])
template <int DIR, typename C> __device__ __forceinline__ void bfly2(C* v)
{
    C a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <int DIR, typename C> __device__ __forceinline__ void bfly4(C* v)
{
    C t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    C t2 = cadd(v[1], v[3]), t3 = mul_i<DIR>(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = cadd(t1, t3);
    v[3] = csub(t1, t3);
}
template <int DIR, typename C> __device__ __forceinline__ void bfly8(C* v)
{
    using S = scalar_t<C>;
    const S h = S(0.70710678118654752440);
    C e[4] = {v[0], v[2], v[4], v[6]};
    C o[4] = {v[1], v[3], v[5], v[7]};
    bfly4<DIR>(e);
    bfly4<DIR>(o);
    // w^q, w = exp(DIR*2 pi i/8): o1 = h t1, t1 = (1 + DIR*i) o[1]; o3 = h t3, t3 = (-1 + DIR*i) o[3].  The scaling is
    // folded into the final additions as explicit fused multiply-adds (one rounding less than scale-then-add, and the
    // same instructions whatever the contraction mode: the library is built with -ffp-contract=on, so that results do
    // not depend on which fusions the optimiser happens to find in a particular instantiation).
    C t1 = cadd(o[1], mul_i<DIR>(o[1]));
    C o2 = mul_i<DIR>(o[2]);
    C t3 = csub(mul_i<DIR>(o[3]), o[3]);
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = mk<C>(ffma(h, t1.x, e[1].x), ffma(h, t1.y, e[1].y));
    v[5] = mk<C>(ffma(-h, t1.x, e[1].x), ffma(-h, t1.y, e[1].y));
    v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
    v[3] = mk<C>(ffma(h, t3.x, e[3].x), ffma(h, t3.y, e[3].y));
    v[7] = mk<C>(ffma(-h, t3.x, e[3].x), ffma(-h, t3.y, e[3].y));
}
template <int DIR, typename C> __device__ __forceinline__ void bfly3(C* v)
{
    using S = scalar_t<C>;
    const S s3 = S(0.86602540378443864676);
    C t1 = cadd(v[1], v[2]);
    C t2 = mk<C>(ffma(S(-0.5), t1.x, v[0].x), ffma(S(-0.5), t1.y, v[0].y));
    C t3 = cscale(mul_i<DIR>(csub(v[1], v[2])), s3);
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, t3);
    v[2] = csub(t2, t3);
}
template <int DIR, typename C> __device__ __forceinline__ void bfly5(C* v)
{
    using S = scalar_t<C>;
    const S c1 = S(0.30901699437494742410), c2 = S(-0.80901699437494742410);
    const S s1 = S(0.95105651629515357212), s2 = S(0.58778525229247312917);
    C t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    C t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    C a = v[0];
    C p1 = mk<C>(a.x + c1 * t1.x + c2 * t2.x, a.y + c1 * t1.y + c2 * t2.y);
    C p2 = mk<C>(a.x + c2 * t1.x + c1 * t2.x, a.y + c2 * t1.y + c1 * t2.y);
    C q1 = mul_i<DIR>(mk<C>(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    C q2 = mul_i<DIR>(mk<C>(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    v[0] = cadd(a, cadd(t1, t2));
    v[1] = cadd(p1, q1);
    v[4] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[3] = csub(p2, q2);
}
template <int DIR, typename C> __device__ __forceinline__ void bfly7(C* v)
{
    using S = scalar_t<C>;
    const S c1 = S(0.62348980185873353053), c2 = S(-0.22252093395631440429), c3 = S(-0.90096886790241912624);
    const S s1 = S(0.78183148246802980871), s2 = S(0.97492791218182360702), s3 = S(0.43388373911755812048);
    C t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
    C u1 = csub(v[1], v[6]), u2 = csub(v[2], v[5]), u3 = csub(v[3], v[4]);
    C a = v[0];
    C p1 = mk<C>(a.x + c1 * t1.x + c2 * t2.x + c3 * t3.x, a.y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
    C p2 = mk<C>(a.x + c2 * t1.x + c3 * t2.x + c1 * t3.x, a.y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
    C p3 = mk<C>(a.x + c3 * t1.x + c1 * t2.x + c2 * t3.x, a.y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
    C q1 = mul_i<DIR>(mk<C>(s1 * u1.x + s2 * u2.x + s3 * u3.x, s1 * u1.y + s2 * u2.y + s3 * u3.y));
    C q2 = mul_i<DIR>(mk<C>(s2 * u1.x - s3 * u2.x - s1 * u3.x, s2 * u1.y - s3 * u2.y - s1 * u3.y));
    C q3 = mul_i<DIR>(mk<C>(s3 * u1.x - s1 * u2.x + s2 * u3.x, s3 * u1.y - s1 * u2.y + s2 * u3.y));
    v[0] = cadd(a, cadd(t1, cadd(t2, t3)));
    v[1] = cadd(p1, q1); v[6] = csub(p1, q1);
    v[2] = cadd(p2, q2); v[5] = csub(p2, q2);
    v[3] = cadd(p3, q3); v[4] = csub(p3, q3);
}
// ---------------------------------------------------------------- the same butterflies on register PAIRS (fp32, gfx950)
// C = pk2: one complex number = one aligned VGPR pair, every complex addition ONE v_pk_add_f32 -- a +- i b included
// (op_sel swaps b's halves, neg_lo / neg_hi puts the sign) -- real constants come from a scalar register pair.  Same
// operations and roundings as the float2 forms above.  These overloads are picked by the composite butterflies below
// (bfly10 / bfly12 / bfly15) when they are instantiated with C = pk2.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_add(pk2 a, pk2 b) { return a + b; }
__device__ __forceinline__ pk2 pk_sub(pk2 a, pk2 b) { return a - b; }
template <int SGN> __device__ __forceinline__ pk2 pk_addi(pk2 a, pk2 b)      // a + SGN * i * b
{
    pk2 r;
    if constexpr (SGN > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ pk2 pk_mulc(pk2 t, float c)                         // t * c
{
    const pk2 cc = {c, c};
    pk2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "s"(cc));
    return r;
}
template <int NEG_E = 0> __device__ __forceinline__ pk2 pk_fmac(pk2 t, float c, pk2 e)     // t * c + e   (NEG_E: t * c - e)
{
    const pk2 cc = {c, c};
    pk2 r;
    if constexpr (NEG_E == 0) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(cc), "v"(e));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(t), "s"(cc), "v"(e));
    return r;
}
template <int DIR> __device__ __forceinline__ void bfly2(pk2* v)
{
    const pk2 a = v[0], b = v[1];
    v[0] = pk_add(a, b);
    v[1] = pk_sub(a, b);
}
template <int DIR> __device__ __forceinline__ void bfly4(pk2* v)
{
    const pk2 t0 = pk_add(v[0], v[2]), t1 = pk_sub(v[0], v[2]), t2 = pk_add(v[1], v[3]), d = pk_sub(v[1], v[3]);
    v[0] = pk_add(t0, t2);
    v[2] = pk_sub(t0, t2);
    v[1] = pk_addi<DIR>(t1, d);
    v[3] = pk_addi<-DIR>(t1, d);
}
template <int DIR> __device__ __forceinline__ void bfly3(pk2* v)
{
    const pk2 t1 = pk_add(v[1], v[2]);
    const pk2 t2 = pk_fmac(t1, -0.5f, v[0]);
    const pk2 ds = pk_mulc(pk_sub(v[1], v[2]), 0.86602540378443864676f);
    v[0] = pk_add(v[0], t1);
    v[1] = pk_addi<DIR>(t2, ds);
    v[2] = pk_addi<-DIR>(t2, ds);
}
template <int DIR> __device__ __forceinline__ void bfly5(pk2* v)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const pk2 a = v[0];
    const pk2 t1 = pk_add(v[1], v[4]), t2 = pk_add(v[2], v[3]), t3 = pk_sub(v[1], v[4]), t4 = pk_sub(v[2], v[3]);
    const pk2 p1 = pk_fmac(t2, c2, pk_fmac(t1, c1, a));
    const pk2 p2 = pk_fmac(t2, c1, pk_fmac(t1, c2, a));
    const pk2 q1 = pk_fmac(t3, s1, pk_mulc(t4, s2));               // s1 t3 + s2 t4
    const pk2 q2 = pk_fmac<1>(t3, s2, pk_mulc(t4, s1));            // s2 t3 - s1 t4
    v[0] = pk_add(a, pk_add(t1, t2));
    v[1] = pk_addi<DIR>(p1, q1);
    v[4] = pk_addi<-DIR>(p1, q1);
    v[2] = pk_addi<DIR>(p2, q2);
    v[3] = pk_addi<-DIR>(p2, q2);
}

// z * (c + i s), the constant in a scalar register pair: two packed instructions
__device__ __forceinline__ pk2 pk_cmulc(pk2 z, float c, float sn)
{
    const pk2 w = {c, sn};
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(z), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "s"(w), "v"(t));
    return r;
}
// radix 9 = 3 x 3 as bfly9 below, on register pairs
template <int DIR> __device__ __forceinline__ void bfly9(pk2* v)
{
    constexpr float c[5] = {1.0f, 0.76604444311897803520f, 0.17364817766693034885f, -0.5f, -0.93969262078590838405f};
    constexpr float sn[5] = {0.0f, 0.64278760968653932632f, 0.98480775301220805937f, 0.86602540378443864676f, 0.34202014332566873304f};
    pk2 y[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
        y[n2][0] = v[n2]; y[n2][1] = v[3 + n2]; y[n2][2] = v[6 + n2];
        bfly3<DIR>(y[n2]);
    }
    y[1][1] = pk_cmulc(y[1][1], c[1], DIR > 0 ? sn[1] : -sn[1]); y[1][2] = pk_cmulc(y[1][2], c[2], DIR > 0 ? sn[2] : -sn[2]);
    y[2][1] = pk_cmulc(y[2][1], c[2], DIR > 0 ? sn[2] : -sn[2]); y[2][2] = pk_cmulc(y[2][2], c[4], DIR > 0 ? sn[4] : -sn[4]);
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        pk2 z[3] = {y[0][k1], y[1][k1], y[2][k1]};
        bfly3<DIR>(z);
        v[k1] = z[0]; v[k1 + 3] = z[1]; v[k1 + 6] = z[2];
    }
}

// cos/sin of 2 pi q / 16 and 2 pi q / 9 (q as used by bfly16 / bfly9)
template <int DIR, int Q, typename C> __device__ __forceinline__ C rot16c(C a)
{
    using S = scalar_t<C>;
    constexpr double c[10] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173, 0.0,
                              -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613, -1.0,
                              -0.92387953251128675613};
    constexpr double sn[10] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613, 1.0,
                               0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173, 0.0,
                               -0.38268343236508977173};
    return cmul(a, mk<C>(S(c[Q]), S(DIR > 0 ? sn[Q] : -sn[Q])));
}
template <int DIR, int Q, typename C> __device__ __forceinline__ C rot9c(C a)
{
    using S = scalar_t<C>;
    constexpr double c[5] = {1.0, 0.76604444311897803520, 0.17364817766693034885, -0.5, -0.93969262078590838405};
    constexpr double sn[5] = {0.0, 0.64278760968653932632, 0.98480775301220805937, 0.86602540378443864676, 0.34202014332566873304};
    return cmul(a, mk<C>(S(c[Q]), S(DIR > 0 ? sn[Q] : -sn[Q])));
}

// radix 16 = 4 x 4 (n = 4 n1 + n2, k = k1 + 4 k2), internal twiddles exp(DIR 2 pi i n2 k1 / 16)
template <int DIR, typename C> __device__ __forceinline__ void bfly16(C* v)
{
    C y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
        y[n2][0] = v[n2]; y[n2][1] = v[4 + n2]; y[n2][2] = v[8 + n2]; y[n2][3] = v[12 + n2];
        bfly4<DIR>(y[n2]);
    }
    y[1][1] = rot16c<DIR, 1>(y[1][1]); y[1][2] = rot16c<DIR, 2>(y[1][2]); y[1][3] = rot16c<DIR, 3>(y[1][3]);
    y[2][1] = rot16c<DIR, 2>(y[2][1]); y[2][2] = mul_i<DIR>(y[2][2]);      y[2][3] = rot16c<DIR, 6>(y[2][3]);
    y[3][1] = rot16c<DIR, 3>(y[3][1]); y[3][2] = rot16c<DIR, 6>(y[3][2]); y[3][3] = rot16c<DIR, 9>(y[3][3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        C z[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
        bfly4<DIR>(z);
        v[k1] = z[0]; v[k1 + 4] = z[1]; v[k1 + 8] = z[2]; v[k1 + 12] = z[3];
    }
}
// radix 9 = 3 x 3 (n = 3 n1 + n2, k = k1 + 3 k2), internal twiddles exp(DIR 2 pi i n2 k1 / 9)
template <int DIR, typename C> __device__ __forceinline__ void bfly9(C* v)
{
    C y[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
        y[n2][0] = v[n2]; y[n2][1] = v[3 + n2]; y[n2][2] = v[6 + n2];
        bfly3<DIR>(y[n2]);
    }
    y[1][1] = rot9c<DIR, 1>(y[1][1]); y[1][2] = rot9c<DIR, 2>(y[1][2]);
    y[2][1] = rot9c<DIR, 2>(y[2][1]); y[2][2] = rot9c<DIR, 4>(y[2][2]);
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        C z[3] = {y[0][k1], y[1][k1], y[2][k1]};
        bfly3<DIR>(z);
        v[k1] = z[0]; v[k1 + 3] = z[1]; v[k1 + 6] = z[2];
    }
}
// radix 15 = 3 x 5 by the prime-factor mapping (no internal twiddles): n = (5 n1 + 3 n2) mod 15,
// k = (10 k1 + 6 k2) mod 15, so that w15^(nk) = w3^(n1 k1) w5^(n2 k2)
template <int DIR, typename C> __device__ __forceinline__ void bfly15(C* v)
{
    C y[5][3];
#pragma unroll
    for (int n2 = 0; n2 < 5; n2++) {
#pragma unroll
        for (int n1 = 0; n1 < 3; n1++) y[n2][n1] = v[(5 * n1 + 3 * n2) % 15];
        bfly3<DIR>(y[n2]);                          // -> y[n2][k1]
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        C z[5] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1], y[4][k1]};
        bfly5<DIR>(z);                              // -> z[k2]
#pragma unroll
        for (int k2 = 0; k2 < 5; k2++) v[(10 * k1 + 6 * k2) % 15] = z[k2];
    }
}

// radix 10 = 2 x 5 and radix 12 = 3 x 4 by the prime-factor mapping (coprime factors, no internal twiddles):
// n = (N2 n1 + N1 n2) mod N,  k = (N2 (N2^-1 mod N1) k1 + N1 (N1^-1 mod N2) k2) mod N
template <int DIR, typename C> __device__ __forceinline__ void bfly10(C* v)
{
    C y[5][2];
#pragma unroll
    for (int n2 = 0; n2 < 5; n2++) {
        y[n2][0] = v[(2 * n2) % 10];                // n1 = 0
        y[n2][1] = v[(5 + 2 * n2) % 10];            // n1 = 1
        bfly2<DIR>(y[n2]);                          // -> y[n2][k1]
    }
#pragma unroll
    for (int k1 = 0; k1 < 2; k1++) {
        C z[5] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1], y[4][k1]};
        bfly5<DIR>(z);                              // -> z[k2]
#pragma unroll
        for (int k2 = 0; k2 < 5; k2++) v[(5 * k1 + 6 * k2) % 10] = z[k2];
    }
}
// radix 14 = 2 x 7 (first radix of the fused kernel for -u 7: a multiple of 2u): n = (7 n1 + 2 n2) mod 14, k = (7 k1 + 8 k2) mod 14
template <int DIR, typename C> __device__ __forceinline__ void bfly14(C* v)
{
    C y[7][2];
#pragma unroll
    for (int n2 = 0; n2 < 7; n2++) {
        y[n2][0] = v[(2 * n2) % 14];                // n1 = 0
        y[n2][1] = v[(7 + 2 * n2) % 14];            // n1 = 1
        bfly2<DIR>(y[n2]);                          // -> y[n2][k1]
    }
#pragma unroll
    for (int k1 = 0; k1 < 2; k1++) {
        C z[7] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1], y[4][k1], y[5][k1], y[6][k1]};
        bfly7<DIR>(z);                              // -> z[k2]
#pragma unroll
        for (int k2 = 0; k2 < 7; k2++) v[(7 * k1 + 8 * k2) % 14] = z[k2];
    }
}
template <int DIR, typename C> __device__ __forceinline__ void bfly12(C* v)
{
    C y[4][3];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
#pragma unroll
        for (int n1 = 0; n1 < 3; n1++) y[n2][n1] = v[(4 * n1 + 3 * n2) % 12];
        bfly3<DIR>(y[n2]);                          // -> y[n2][k1]
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        C z[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
        bfly4<DIR>(z);                              // -> z[k2]
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) v[(4 * k1 + 9 * k2) % 12] = z[k2];
    }
}

template <int R, int DIR, typename C> __device__ __forceinline__ void bfly(C* v)
{
    if constexpr (R == 2) bfly2<DIR>(v);
    else if constexpr (R == 3) bfly3<DIR>(v);
    else if constexpr (R == 4) bfly4<DIR>(v);
    else if constexpr (R == 5) bfly5<DIR>(v);
    else if constexpr (R == 7) bfly7<DIR>(v);
    else if constexpr (R == 8) bfly8<DIR>(v);
    else if constexpr (R == 9) bfly9<DIR>(v);
    else if constexpr (R == 10) bfly10<DIR>(v);
    else if constexpr (R == 12) bfly12<DIR>(v);
    else if constexpr (R == 14) bfly14<DIR>(v);
    else if constexpr (R == 15) bfly15<DIR>(v);
    else if constexpr (R == 16) bfly16<DIR>(v);
}

// twiddle the R inputs of one butterfly: v[m] *= exp(DIR * 2 pi i * m * tidx / N), tidx = k*tstep.
// One table fetch for m=1; powers 2 and 4 are fetched too (cheap, L1/L2 resident), the rest are
// products -- two roundings instead of one, ~1e-7 relative.
template <int R, int DIR, typename C>
__device__ __forceinline__ void apply_twiddles(C* v, const C* __restrict__ tw, int tidx)
{
    if constexpr (R == 2) {
        v[1] = cmul(v[1], twid<DIR>(tw[tidx]));
    } else if constexpr (R == 3) {
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2);
    } else if constexpr (R == 4) {
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]);
        C w3 = cmul(w1, w2);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
    } else if constexpr (R == 5) {
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        C w3 = cmul(w1, w2);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
    } else if constexpr (R == 7) {
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        C w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
        v[4] = cmul(v[4], w4); v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6);
    } else if constexpr (R == 8) {
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        C w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
    } else {
        // R = 9, 15, 16: powers 1, 2, 4, 8 from the table, the others as products of two of them
        static_assert(R == 9 || R == 15 || R == 16, "radix");
        C w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]), w8 = twid<DIR>(tw[8 * tidx]);
        C w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7); v[8] = cmul(v[8], w8);
        if constexpr (R >= 15) {
            v[9] = cmul(v[9], cmul(w8, w1)); v[10] = cmul(v[10], cmul(w8, w2)); v[11] = cmul(v[11], cmul(w8, w3));
            v[12] = cmul(v[12], cmul(w8, w4)); v[13] = cmul(v[13], cmul(w8, w5)); v[14] = cmul(v[14], cmul(w8, w6));
        }
        if constexpr (R == 16) v[15] = cmul(v[15], cmul(w8, w7));
    }
}

// ---------------------------------------------------------------- one generic Stockham stage
// in/out: LDS, C [n][TK] with lpad() applied to the flattened element index.
// Butterfly j (0 <= j < N/R) of sequence col reads in[j + m*N/R], writes
// out[(j - k)*R + k + m*Ns], k = j % Ns (Stockham autosort, decimation in time).
template <int R, int DIR, int TK, typename C>
__device__ __forceinline__ void stage_lds(const C* __restrict__ in, C* __restrict__ out,
                                          int N, int Ns, const C* __restrict__ tw, int tid, int T)
{
    const int nb = N / R;
    const int tstep = nb / Ns;                 // N / (Ns*R)
    const bool ns_pow2 = (Ns & (Ns - 1)) == 0;
    for (int g = tid; g < nb * TK; g += T) {
        const int col = g % TK;                // TK is a compile-time power of two
        const int j = g / TK;
        const int k = ns_pow2 ? (j & (Ns - 1)) : (j % Ns);
        C v[R];
#pragma unroll
        for (int m = 0; m < R; m++) v[m] = in[lpad((j + m * nb) * TK + col)];
        if (Ns > 1) apply_twiddles<R, DIR>(v, tw, k * tstep);
        bfly<R, DIR>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int m = 0; m < R; m++) out[lpad((j0 + m * Ns) * TK + col)] = v[m];
    }
}

// Full transform of TK interleaved sequences.  Data in `a` (valid after a barrier executed by the
// caller); ping-pongs between a and b; returns the buffer holding the result (already synced).
template <int DIR, int TK, typename C>
__device__ __forceinline__ C* fft_lds(C* a, C* b, const StagePlan& P,
                                      const C* __restrict__ tw, int tid, int T)
{
    const int N = P.n;
    int Ns = 1;
    for (int s = 0; s < P.nstages; s++) {
        const int R = P.radix[s];
        switch (R) {
        case 8: stage_lds<8, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        case 4: stage_lds<4, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        case 2: stage_lds<2, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        case 3: stage_lds<3, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        case 5: stage_lds<5, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        default: stage_lds<7, DIR, TK>(a, b, N, Ns, tw, tid, T); break;
        }
        Ns *= R;
        __syncthreads();
        C* t = a; a = b; b = t;
    }
    return a;
}

// ---------------------------------------------------------------- the same IN PLACE, one LDS buffer
// For sequences too long for two buffers (complex rows of 10 240 .. 16 384 points on the non-R2C path: 139 KB of the
// 160 KB for ONE; the reference splits such axes into several uploads with a transposition through a temporary buffer,
// vkFFT.h:4773-4992, 2290-2388, 6562-6576).  A Stockham stage writes every output to another place than it read its inputs
// from, so a stage cannot run in place element by element -- but it can thread by thread: everybody takes the inputs of all
// his butterflies into registers (at most PT points), the workgroup synchronises, everybody writes his outputs to their
// autosort positions.  Needs N / R <= (PT / R) * T for every radix R of the plan (stage_fits_inplace: checked by the host).
template <int R, int DIR, int PT, typename C>
__device__ __forceinline__ void stage_lds_inplace(C* __restrict__ buf, int N, int Ns, const C* __restrict__ tw, int tid, int T)
{
    constexpr int NBT = PT / R;
    const int nb = N / R;
    const int tstep = nb / Ns;
    const bool ns_pow2 = (Ns & (Ns - 1)) == 0;
    C v[NBT][R];
#pragma unroll
    for (int b = 0; b < NBT; b++) {
        const int j = tid + b * T;
        if (j < nb) {
#pragma unroll
            for (int m = 0; m < R; m++) v[b][m] = buf[lpad(j + m * nb)];
        }
    }
    __syncthreads();                            // every input of the stage is in somebody's registers
#pragma unroll
    for (int b = 0; b < NBT; b++) {
        const int j = tid + b * T;
        if (j < nb) {
            const int k = ns_pow2 ? (j & (Ns - 1)) : (j % Ns);
            if (Ns > 1) apply_twiddles<R, DIR>(v[b], tw, k * tstep);
            bfly<R, DIR>(v[b]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int m = 0; m < R; m++) buf[lpad(j0 + m * Ns)] = v[b][m];
        }
    }
    __syncthreads();
}
__host__ __device__ constexpr bool stage_fits_inplace(int N, int R, int T, int PT) { return N / R <= (PT / R) * T; }

// ... and on TK interleaved sequences (element n of sequence col at lpad(n * TK + col)): the column kernel's form.  A thread runs
// up to PT / R butterflies of the N / R * TK of a stage: needs N / R * TK <= (PT / R) * T.
template <int R, int DIR, int PT, int TK, typename C>
__device__ __forceinline__ void stage_lds_inplace_tk(C* __restrict__ buf, int N, int Ns, const C* __restrict__ tw, int tid, int T)
{
    constexpr int NBT = PT / R;
    const int nb = N / R;
    const int tstep = nb / Ns;
    const bool ns_pow2 = (Ns & (Ns - 1)) == 0;
    C v[NBT][R];
#pragma unroll
    for (int b = 0; b < NBT; b++) {
        const int g = tid + b * T;
        if (g < nb * TK) {
            const int col = g % TK, j = g / TK;
#pragma unroll
            for (int m = 0; m < R; m++) v[b][m] = buf[lpad((j + m * nb) * TK + col)];
        }
    }
    __syncthreads();                            // every input of the stage is in somebody's registers
#pragma unroll
    for (int b = 0; b < NBT; b++) {
        const int g = tid + b * T;
        if (g < nb * TK) {
            const int col = g % TK, j = g / TK;
            const int k = ns_pow2 ? (j & (Ns - 1)) : (j % Ns);
            if (Ns > 1) apply_twiddles<R, DIR>(v[b], tw, k * tstep);
            bfly<R, DIR>(v[b]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int m = 0; m < R; m++) buf[lpad((j0 + m * Ns) * TK + col)] = v[b][m];
        }
    }
    __syncthreads();
}
constexpr int COL_INPLACE_PT = 8;        // points per thread of the in-place column kernel (k_col<TK, double2, true>; the plan checks with it)
__host__ __device__ constexpr bool stage_fits_inplace_tk(int N, int TK, int R, int T, int PT) { return (N / R) * TK <= (PT / R) * T; }
template <int DIR, int PT, int TK, typename C>
__device__ __forceinline__ void fft_lds_inplace_tk(C* a, const StagePlan& P, const C* __restrict__ tw, int tid, int T)
{
    const int N = P.n;
    int Ns = 1;
    for (int s = 0; s < P.nstages; s++) {
        const int R = P.radix[s];
        switch (R) {
        case 8: stage_lds_inplace_tk<8, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        case 4: stage_lds_inplace_tk<4, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        case 2: stage_lds_inplace_tk<2, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        case 3: stage_lds_inplace_tk<3, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        case 5: stage_lds_inplace_tk<5, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        default: stage_lds_inplace_tk<7, DIR, PT, TK>(a, N, Ns, tw, tid, T); break;
        }
        Ns *= R;
    }
}

// Data in `a` (valid after a barrier executed by the caller); the result is left in `a` (synced).
template <int DIR, int PT, typename C>
__device__ __forceinline__ void fft_lds_inplace(C* a, const StagePlan& P, const C* __restrict__ tw, int tid, int T)
{
    const int N = P.n;
    int Ns = 1;
    for (int s = 0; s < P.nstages; s++) {
        const int R = P.radix[s];
        switch (R) {
        case 8: stage_lds_inplace<8, DIR, PT>(a, N, Ns, tw, tid, T); break;
        case 4: stage_lds_inplace<4, DIR, PT>(a, N, Ns, tw, tid, T); break;
        case 2: stage_lds_inplace<2, DIR, PT>(a, N, Ns, tw, tid, T); break;
        case 3: stage_lds_inplace<3, DIR, PT>(a, N, Ns, tw, tid, T); break;
        case 5: stage_lds_inplace<5, DIR, PT>(a, N, Ns, tw, tid, T); break;
        default: stage_lds_inplace<7, DIR, PT>(a, N, Ns, tw, tid, T); break;
        }
        Ns *= R;
    }
}

}  // namespace fftup
