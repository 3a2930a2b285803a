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

#include <type_traits>

#include "fft_engine.hpp"
#include "kernels_generic.hpp"

namespace fftup {

// launch bound of the column kernels (k_col_t, k_col_v): 8 waves per SIMD = 64 VGPRs: four 512-thread workgroups per compute
// unit, all 771 of a frame resident at once
constexpr int kColWaves = 8;

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n / 2); }
// radix of the stage that starts at sub-transform length Ns: the largest allowed one (RMAX = 8, or 16 for
// threads that own 16 points) the remaining length still holds
constexpr int stage_radix(int N, int Ns, int RMAX = 8) { return (N / Ns >= RMAX) ? RMAX : (N / Ns); }

// LDS index map of the register-resident kernels.  A stage's scatter writes (ds_write_b64: groups of 16 lanes, 16
// eight-byte slots) hit elements 8 or more apart, its gather reads (ds_read_b64: groups of 32 lanes, 32 slots) hit
// consecutive elements.  XOR-ing index bits 3..6 into bits 0..3 keeps every aligned block of 16 elements in place (so
// consecutive reads stay conflict-free and no padding is needed) and spreads the strided writes over all slots:
// tools/lds_conflicts.py finds 4 / 2 LDS cycles per write / read (the conflict-free minimum) for every plan used here,
// against 5.3 / 4 with one padding element per 16 (the gathers paid double for the padding).
__device__ __forceinline__ int lswz(int i) { return i ^ ((i >> 3) & 15); }
__host__ __device__ constexpr int lswz_size(int n) { return (n + 15) & ~15; }
// LDS element index of point idx of sequence col (TK interleaved sequences)
template <int TK> __device__ __forceinline__ int lidx(int idx, int col) { return lswz(idx * TK + col); }
// The map is linear over GF(2), and the R elements a thread touches in one exchange are A + q*S with S a power of two
// and the bits of the q field clear in A, so lswz(A + q*S) = lswz(A) ^ lswz(q*S): one swizzled byte address per
// exchange, and per element an exclusive-or with a constant < 128 bytes (bits 0..3 of the index) plus a constant that
// goes into the instruction's offset field -- instead of a shift, an and/xor and a scaled add per element.  Needs the
// buffer 128-byte aligned (so that adding its base commutes with the exclusive-or); done on raw 32-bit LDS addresses.
constexpr int lswz_c(int i) { return i ^ ((i >> 3) & 15); }
typedef float lds_f2raw __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_f2raw lds_f2;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void lds_put(unsigned a0, int C, float2 v)
{
    lds_f2raw r = {v.x, v.y};
    *(lds_f2*)(size_t)((a0 ^ (unsigned)((C & 15) * 8)) + (unsigned)((C & ~15) * 8)) = r;
}
__device__ __forceinline__ float2 lds_get(unsigned a0, int C)
{
    const lds_f2raw r = *(const lds_f2*)(size_t)((a0 ^ (unsigned)((C & 15) * 8)) + (unsigned)((C & ~15) * 8));
    return make_float2(r.x, r.y);
}

// (spec_store16 / spec_store8 / spec_store: the write-through spectrum stores, fft_engine.hpp)
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

// exp(2 pi i q/32), q = 0..31
constexpr float kCos32[32] = {1.f, 0.98078528040323045f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                              0.55557023301960223f, 0.38268343236508977f, 0.19509032201612827f, 0.f,
                              -0.19509032201612827f, -0.38268343236508977f, -0.55557023301960223f, -0.70710678118654752f,
                              -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323045f, -1.f,
                              -0.98078528040323045f, -0.92387953251128674f, -0.83146961230254524f, -0.70710678118654752f,
                              -0.55557023301960223f, -0.38268343236508977f, -0.19509032201612827f, 0.f,
                              0.19509032201612827f, 0.38268343236508977f, 0.55557023301960223f, 0.70710678118654752f,
                              0.83146961230254524f, 0.92387953251128674f, 0.98078528040323045f};
template <int Q> __device__ __forceinline__ float2 rot32()
{
    constexpr float cr = kCos32[Q & 31], ci = kCos32[(Q + 24) & 31];
    return make_float2(cr, ci);
}

// LDS exchange synchronisation: the whole workgroup (s_barrier), or -- when every wave transforms its own sequences
// in its own LDS region -- nothing but program order: a wave's LDS instructions execute in issue order, so the
// fences only keep the compiler from moving the reads above the writes.
template <bool WAVE> __device__ __forceinline__ void lds_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// z * w in two packed instructions: t = (-z.y w.y, z.y w.x), then (z.x w.x + t.x, z.x w.y + t.y) -- the same three
// roundings as cmul().  (Left to itself the compiler spends four: it negates and swaps z into a fresh register pair
// first, or keeps a second, rotated copy of every twiddle.)
__device__ __forceinline__ float2 cmul_tw(float2 z, float2 w)
{
    typedef float cf2 __attribute__((ext_vector_type(2)));
    const cf2 zv = {z.x, z.y}, wv = {w.x, w.y};
    cf2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(zv), "v"(wv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(zv), "v"(wv), "v"(t));
    return make_float2(r.x, r.y);
}

// a + i b and conj(a) + i conj(b) in one packed addition each (the C2R kernels' input formation, vkFFT.h:2096-2131)
__device__ __forceinline__ float2 cadd_i(float2 a, float2 b)
{
    const pk2 r = pk_addi<1>(pk2{a.x, a.y}, pk2{b.x, b.y});
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cadd_conj_i(float2 a, float2 b)
{
    const pk2 av = {a.x, a.y}, bv = {b.x, b.y};
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
}
// ---- radix-4 / radix-8 butterflies written on register pairs: every complex addition is ONE v_pk_add_f32, a +- i b
// included (op_sel swaps b's halves, neg_lo / neg_hi puts the sign), the 1/sqrt2 rotations are two v_pk_fma_f32 each.
// Same operations and roundings as bfly4 / bfly8 of fft_engine.hpp; the compiler, given those, builds the operands of
// the rotations with register moves (~50 per transform of the fused kernel).
template <int SGN> __device__ __forceinline__ pk2 pk_fmah(pk2 t, pk2 e)         // e + SGN * t / sqrt2
{
    const pk2 h = {0.70710678118654752440f, 0.70710678118654752440f};
    pk2 r;
    if constexpr (SGN > 0) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(h), "v"(e));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(t), "s"(h), "v"(e));
    return r;
}
template <int DIR> __device__ __forceinline__ void bfly4_pk(pk2& a0, pk2& a1, pk2& a2, pk2& a3)
{
    const pk2 t0 = pk_add(a0, a2), t1 = pk_sub(a0, a2), t2 = pk_add(a1, a3), d = pk_sub(a1, a3);
    a0 = pk_add(t0, t2);
    a2 = pk_sub(t0, t2);
    a1 = pk_addi<DIR>(t1, d);
    a3 = pk_addi<-DIR>(t1, d);
}
template <int DIR> __device__ __forceinline__ void bfly8_pk(float2* v)
{
    pk2 e0 = {v[0].x, v[0].y}, e1 = {v[2].x, v[2].y}, e2 = {v[4].x, v[4].y}, e3 = {v[6].x, v[6].y};
    pk2 o0 = {v[1].x, v[1].y}, o1 = {v[3].x, v[3].y}, o2 = {v[5].x, v[5].y}, o3 = {v[7].x, v[7].y};
    bfly4_pk<DIR>(e0, e1, e2, e3);
    bfly4_pk<DIR>(o0, o1, o2, o3);
    const pk2 t1 = pk_addi<DIR>(o1, o1);                   // (1 + DIR i) o1
    const pk2 u3 = pk_addi<-DIR>(o3, o3);                  // (1 - DIR i) o3 = -((-1 + DIR i) o3)
    const pk2 r0 = pk_add(e0, o0), r4 = pk_sub(e0, o0);
    const pk2 r1 = pk_fmah<1>(t1, e1), r5 = pk_fmah<-1>(t1, e1);
    const pk2 r2 = pk_addi<DIR>(e2, o2), r6 = pk_addi<-DIR>(e2, o2);
    const pk2 r3 = pk_fmah<-1>(u3, e3), r7 = pk_fmah<1>(u3, e3);
    v[0] = make_float2(r0.x, r0.y); v[1] = make_float2(r1.x, r1.y); v[2] = make_float2(r2.x, r2.y); v[3] = make_float2(r3.x, r3.y);
    v[4] = make_float2(r4.x, r4.y); v[5] = make_float2(r5.x, r5.y); v[6] = make_float2(r6.x, r6.y); v[7] = make_float2(r7.x, r7.y);
}
// z * exp(DIR 2 pi i Q/16), the constant in a scalar register pair: two packed instructions (as cmul_tw)
template <int DIR, int Q> __device__ __forceinline__ pk2 pk_rot16(pk2 z)
{
    static_assert(Q >= 0 && Q < 10, "bfly16 needs Q = 1, 2, 3, 6, 9");
    constexpr float c[10] = {1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                             -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.0f,
                             -0.92387953251128675613f};
    constexpr float sn[10] = {0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f, 1.0f,
                              0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                              -0.38268343236508977173f};
    const pk2 w = {c[Q], DIR > 0 ? sn[Q] : -sn[Q]};
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(z), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "s"(w), "v"(t));
    return r;
}
template <int SGN> __device__ __forceinline__ pk2 pk_muli(pk2 b)               // SGN * i * b
{
    pk2 r;
    if constexpr (SGN > 0) asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(b));
    else asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(b));
    return r;
}
// radix 16 = 4 x 4 exactly as bfly16 of fft_engine.hpp (n = 4 n1 + n2, k = k1 + 4 k2), on register pairs
template <int DIR> __device__ __forceinline__ void bfly16_pk(float2* v)
{
    pk2 y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
#pragma unroll
        for (int n1 = 0; n1 < 4; n1++) y[n2][n1] = pk2{v[4 * n1 + n2].x, v[4 * n1 + n2].y};
        bfly4_pk<DIR>(y[n2][0], y[n2][1], y[n2][2], y[n2][3]);
    }
    y[1][1] = pk_rot16<DIR, 1>(y[1][1]); y[1][2] = pk_rot16<DIR, 2>(y[1][2]); y[1][3] = pk_rot16<DIR, 3>(y[1][3]);
    y[2][1] = pk_rot16<DIR, 2>(y[2][1]); y[2][2] = pk_muli<DIR>(y[2][2]);      y[2][3] = pk_rot16<DIR, 6>(y[2][3]);
    y[3][1] = pk_rot16<DIR, 3>(y[3][1]); y[3][2] = pk_rot16<DIR, 6>(y[3][2]); y[3][3] = pk_rot16<DIR, 9>(y[3][3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        bfly4_pk<DIR>(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
        v[k1] = make_float2(y[0][k1].x, y[0][k1].y); v[k1 + 4] = make_float2(y[1][k1].x, y[1][k1].y);
        v[k1 + 8] = make_float2(y[2][k1].x, y[2][k1].y); v[k1 + 12] = make_float2(y[3][k1].x, y[3][k1].y);
    }
}
// butterflies of the register-resident kernels
template <int R, int DIR> __device__ __forceinline__ void bfly_reg(float2* v)
{
    if constexpr (R == 8) bfly8_pk<DIR>(v);
    else if constexpr (R == 16) bfly16_pk<DIR>(v);
    else if constexpr (R == 2 || R == 3 || R == 4 || R == 5 || R == 9 || R == 10 || R == 12 || R == 15) {
        pk2 z[R];                                            // the generic composites on register pairs (fft_engine.hpp)
#pragma unroll
        for (int m = 0; m < R; m++) z[m] = pk2{v[m].x, v[m].y};
        bfly<R, DIR>(z);
#pragma unroll
        for (int m = 0; m < R; m++) v[m] = make_float2(z[m].x, z[m].y);
    } else bfly<R, DIR>(v);
}

template <int R> __device__ __forceinline__ void twiddle_powers(float2* v, float2 w1)
{
    if constexpr (R == 2) {
        v[1] = cmul_tw(v[1], w1);
    } else if constexpr (R == 4) {
        float2 w2 = cmul_tw(w1, w1), w3 = cmul_tw(w2, w1);
        v[1] = cmul_tw(v[1], w1); v[2] = cmul_tw(v[2], w2); v[3] = cmul_tw(v[3], w3);
    } else {
        float2 w2 = cmul_tw(w1, w1), w3 = cmul_tw(w2, w1), w4 = cmul_tw(w2, w2);
        float2 w5 = cmul_tw(w4, w1), w6 = cmul_tw(w3, w3), w7 = cmul_tw(w4, w3);
        v[1] = cmul_tw(v[1], w1); v[2] = cmul_tw(v[2], w2); v[3] = cmul_tw(v[3], w3); v[4] = cmul_tw(v[4], w4);
        v[5] = cmul_tw(v[5], w5); v[6] = cmul_tw(v[6], w6); v[7] = cmul_tw(v[7], w7);
        if constexpr (R == 16) {
            float2 w8 = cmul_tw(w4, w4);
            v[8] = cmul_tw(v[8], w8); v[9] = cmul_tw(v[9], cmul_tw(w8, w1)); v[10] = cmul_tw(v[10], cmul_tw(w5, w5));
            v[11] = cmul_tw(v[11], cmul_tw(w8, w3)); v[12] = cmul_tw(v[12], cmul_tw(w6, w6)); v[13] = cmul_tw(v[13], cmul_tw(w8, w5));
            v[14] = cmul_tw(v[14], cmul_tw(w7, w7)); v[15] = cmul_tw(v[15], cmul_tw(w8, w7));
        }
    }
}

// ---- one stage on registers: E/R butterflies of radix R (compile-time recursion over b)
template <int N, int E, int R, int Ns, int DIR, int B, bool PKB = true>
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
            if constexpr (Ns > Tc && B > 0) w1 = cmul(w1, twid<DIR>(rot32<B * (32 / E)>()));
            twiddle_powers<R>(w, w1);
        }
        // (PKB = false: the generic form, which the compiler can fold when inputs are known zeros; measured: no gain)
        if constexpr (PKB) bfly_reg<R, DIR>(w);
        else bfly<R, DIR>(w);
#pragma unroll
        for (int m = 0; m < R; m++) v[B + m * NB] = w[m];
        butterfly_b<N, E, R, Ns, DIR, B + 1, PKB>(v, wbase);
    }
}

template <int N, int E, int R, int Ns, int DIR, bool PKB = true>
__device__ __forceinline__ void reg_butterflies(float2 (&v)[E], float2 wbase)
{
    constexpr int Tc = N / E;
    static_assert(Ns <= Tc || Ns * R == N, "per-butterfly twiddle offsets are only derived for the last stage");
    butterfly_b<N, E, R, Ns, DIR, 0, PKB>(v, wbase);
}

// ---- Stockham autosort scatter of a stage's outputs into LDS
template <int N, int E, int R, int Ns, int TK>
__device__ __forceinline__ void reg_scatter(const float2 (&v)[E], float2* __restrict__ buf, int p, int col)
{
    constexpr int Tc = N / E;
    constexpr int NB = E / R;
    const unsigned base = lds_addr(buf);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int j = p + b * Tc;
        const int k = j & (Ns - 1);
        const int j0 = (j - k) * R + k;
        const unsigned a0 = base + 8u * (unsigned)lidx<TK>(j0, col);       // bits of q*Ns*TK are clear in j0*TK + col
#pragma unroll
        for (int q = 0; q < R; q++) lds_put(a0, lswz_c(q * Ns * TK), v[b + q * NB]);
    }
}

template <int N, int E, int TK>
__device__ __forceinline__ void reg_gather(float2 (&v)[E], const float2* __restrict__ buf, int p, int col)
{
    constexpr int Tc = N / E;
    const unsigned a0 = lds_addr(buf) + 8u * (unsigned)lidx<TK>(p, col);    // p < Tc: bits of i*Tc*TK are clear
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = lds_get(a0, lswz_c(i * Tc * TK));
    // All reads in flight together, ONE wait: left alone the scheduler issues a read, waits, multiplies by the twiddle, issues
    // the next read .. -- E/2 LDS round trips in a row (350 cycles per gather, s_memtime marks) where one suffices.
#pragma unroll
    for (int i = 0; i < E; i += 2) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y));
}

// ---- all stages.  On entry v[i] = x[p + Tc*i].  If FINAL_TO_LDS the result X is left in LDS in
// natural order (valid after the trailing barrier); otherwise v[i] = X[p + Tc*i] on return.
// `buf` must not be in use by anyone on entry (callers barrier before re-using it).
template <int N, int E, int DIR, int TK, bool FINAL_TO_LDS, int S = 0, int RMAX = 8, bool WAVE = false>
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
        lds_sync<WAVE>();
    }
    if constexpr (!last) {
        reg_gather<N, E, TK>(v, buf, p, col);
        lds_sync<WAVE>();
        reg_fft<N, E, DIR, TK, FINAL_TO_LDS, S + 1, RMAX, WAVE>(v, buf, p, col, tws);
    }
}

// ---- the same with TWO exchange buffers: the exchange of stage s goes through z when an even number of exchanges follows
// it, else through c.  A buffer is rewritten two exchanges after it was read, with the barrier of the exchange in
// between, so ONE barrier per exchange suffices (scatter, barrier, gather) -- and the last exchange always uses z, which
// leaves c free for the caller's output while slower waves still gather.  Returns with v[i] = X[p + Tc*i].
template <int N, int E, int DIR, int S = 0>
__device__ __forceinline__ void reg_fft_pp(float2 (&v)[E], float2* __restrict__ c, float2* __restrict__ z, int p, const TwSet<N, E, 8>& tws, int p0)
{
    constexpr int Ns = stage_ns(N, S, 8);
    constexpr int R = stage_radix(N, Ns, 8);
    constexpr int NE = num_stages(N, 8) - 1;             // exchanges
    static_assert(E % R == 0, "radix must divide the per-thread point count");
    reg_butterflies<N, E, R, Ns, DIR>(v, tws.w[S > 0 ? S - 1 : 0]);
    if constexpr (Ns * R != N) {
        float2* __restrict__ b = ((NE - 1 - S) % 2 == 0) ? z : c;
        reg_scatter<N, E, R, Ns, 1>(v, b, S == 0 ? p0 : p, 0);         // (p0: the first stage's butterfly index, which need not be p)
        __syncthreads();
        reg_gather<N, E, 1>(v, b, p, 0);
        reg_fft_pp<N, E, DIR, S + 1>(v, c, z, p, tws, p0);
    }
}

#ifdef FFTUP_PLANE_STAMPS
// measurement build (tools/plane_stamps.py): when does each workgroup of the row / column pass begin and end?
// [kernel 0..1][workgroup][begin, end, plane], 100 MHz wall clock, plain stores
__device__ unsigned long long g_plane_stamps[3][2048][3];
__device__ __forceinline__ unsigned stamp_wg() { return blockIdx.x + blockIdx.y * gridDim.x; }
__device__ __forceinline__ void stamp_begin(int k, int c) { if (threadIdx.x == 0) { g_plane_stamps[k][stamp_wg()][0] = wall_clock64(); g_plane_stamps[k][stamp_wg()][2] = (unsigned long long)c; } }
__device__ __forceinline__ void stamp_end(int k, int) { __syncthreads(); if (threadIdx.x == 0) g_plane_stamps[k][stamp_wg()][1] = wall_clock64(); }
#else
__device__ __forceinline__ void stamp_begin(int, int) {}
__device__ __forceinline__ void stamp_end(int, int) {}
#endif

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

// R2C unpack of a row pair transformed as ONE complex row (vkFFT.h:4292-4323): Z in natural order in `buf` -> rows 2j (A) and
// 2j + 1 (B) of the blocked half spectrum.  8 consecutive lanes cover one tile segment [A(TK) | B(TK)] of 2*TK float2; each
// lane stores 16 bytes (two complex values).
template <int W, int TK>
__device__ __forceinline__ void row_unpack_store(const float2* __restrict__ buf, const RowR2CTParams& p, int c, int j, int tid)
{
    constexpr int T = W / 8;
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
                float2 zk = buf[lswz(k)];
                float2 zn = buf[lswz((W - k) & (W - 1))];
                r = isB ? make_float2(0.5f * (zk.y + zn.y), 0.5f * (-zk.x + zn.x))
                        : make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            }
            o[e] = r;
        }
        spec_store16(base + (long)tile * tile_stride + (isB ? TK : 0) + kk, o[0], o[1]);
    }
}

// grid (H/2, 3), block W/8.  LDS: lswz_size(W) float2.
template <int W, int MODE, int TK>
__global__ void __launch_bounds__(W / 8) k_row_r2c_t(RowR2CTParams p)
{
    constexpr int E = 8, T = W / E;
    __shared__ __attribute__((aligned(128))) float2 buf[lswz_size(W)];      // (128: reg_scatter / reg_gather)
    const int tid = threadIdx.x, c = blockIdx.y;
    const int j = blockIdx.x;      // (an XCD-aware pair order -- pairs 2i, 2i+1 on one XCD -- measured no gain)
    float2 v[E];
    TwSet<W, E> tws;
    stamp_begin(0, c);
    tws.load(p.tw, tid);
#pragma unroll
    for (int i = 0; i < E; i++)
        v[i] = make_float2(load_px_t<MODE>(p, c, 2 * j, tid + T * i), load_px_t<MODE>(p, c, 2 * j + 1, tid + T * i));
    reg_fft<W, E, +1, 1, true>(v, buf, tid, 0, tws);
    row_unpack_store<W, TK>(buf, p, c, j, tid);
    stamp_end(0, c);
}

// (Round 4 measured "read the row pair's 2 x 3 W bytes once, 16 bytes per thread, stage them in LDS, transform the three planes
// in one workgroup" against this kernel's sixteen byte loads per thread and plane: 14.9 us with the planes one after the other
// in a 256-thread workgroup, 12.7 us side by side in 768 threads, 12.1 us as it is -- the input loads are not what the row
// kernel waits for; profiles/r04_c_row_u8_variants.txt.)

// =================================================================================== column
struct ColTParams {
    const float2* S1;
    float2* S2;
    const float2 *twH, *twUH;
    int W, NT;
    int zly, zry;            // k_col_pad: rows [zly, zry) of the zero-padded spectrum read as zero (the reference's guard as ITS float arithmetic
                             // puts it, VkResample.cpp:1494-1495: [H/2, UH - H/2) or a row off that for factors that are no binary fraction)
};

// v[i] = F[pp + Tc*i] * w * exp(-2 pi i * q/16), q = i (i < 4) or i + 8 (i >= 4: the extra half turn is the -1)
template <int TK, int Tc, int I>
__device__ __forceinline__ void col_phase(float2 (&v)[8], const float2* __restrict__ buf, int pp, int col, float2 w)
{
    if constexpr (I < 8) {
        constexpr int q = (I < 4) ? I : I + 8;
        const float2 f = buf[lidx<TK>(pp + Tc * I, col)];
        const float2 t = (q == 0) ? w : cmul(w, twid<-1>(rot16<q>()));
        v[I] = cmul(f, t);
        col_phase<TK, Tc, I + 1>(v, buf, pp, col, w);
    }
}

// grid (NT, 3), block TK*H/8, H/8 threads per column.  LDS: lswz_size(H*TK) float2.
//
// The reference transforms a column forward (length H), places it in a zero-padded buffer of 2H rows
// (G[ky'] = F[ky'] for ky' < H/2, F[ky'-H] for ky' >= 3H/2, zero between: shift VkResample.cpp:514-526, read guard
// vkFFT.h:1670-1695) and transforms back (length 2H).  With n = 2m / 2m+1 in D[n] = sum G[ky'] exp(-2 pi i n ky'/2H):
//   D[2m]   = sum_k F[k] exp(-2 pi i m k/H)            = H * (the column itself)  -> S2[2m] = S1[m] / 2, no arithmetic;
//   D[2m+1] = sum_k F[k] t[k] exp(-2 pi i m k/H),  t[k] = exp(-2 pi i k/2H) * (k < H/2 ? 1 : -1)
// (the second half sits H rows higher: exp(-i pi (2m+1)) = -1).  So the inverse is a transform of length H of the
// phase-shifted spectrum and yields the ODD rows only; the even rows are never written -- the row kernels read them
// from S1.  Both halves are kept at TWICE the reference's normalisation (S1 as it is, odd rows D/H instead of D/2H);
// the consumers fold the 1/2 into their final scale.  Exact for every column vector, Nyquist row included.
template <int H, int TK>
__global__ void __launch_bounds__(TK* H / 8, kColWaves) k_col_t(ColTParams p)
{
    constexpr int Tc = H / 8;                        // threads per column
    extern __shared__ __attribute__((aligned(128))) char smem[];
    float2* buf = (float2*)smem;
    const int tid = threadIdx.x;
    const int col = tid % TK, pp = tid / TK;
    const int tile = blockIdx.x, c = blockIdx.y;
    const bool valid = tile * TK + col <= p.W / 2;
    const float2* src = p.S1 + ((long)c * p.NT + tile) * H * TK;
    float2 v[8];
    TwSet<H, 8> tws;                                 // same base twiddles for both directions (conjugated inside)
    tws.load(p.twH, pp);
    const float2 ph = p.twUH[pp];                    // exp(+2 pi i pp/2H)
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = valid ? src[(pp + Tc * i) * TK + col] : make_float2(0.f, 0.f);
    reg_fft<H, 8, +1, TK, true>(v, buf, pp, col, tws);            // F[k] natural order in LDS
    // thread owns F[pp + Tc*i]: t = conj(ph) * exp(-2 pi i * i/16) * (i < 4 ? 1 : -1)      (Tc/2H = 1/16)
    col_phase<TK, Tc, 0>(v, buf, pp, col, twid<-1>(ph));
    __syncthreads();
    reg_fft<H, 8, -1, TK, false>(v, buf, pp, col, tws);
    float2* dst = p.S2 + ((long)c * p.NT + tile) * H * TK;
    constexpr float inv = 1.0f / (float)H;
    if (valid) {
#pragma unroll
        for (int i = 0; i < 8; i++) spec_store8(dst + (pp + Tc * i) * TK + col, cscale(v[i], inv));
    }
}

// =================================================================================== row C2R
struct RowC2RTParams {
    const float2* S1;        // even spectrum rows (the forward spectrum itself), see k_col_t
    const float2* S2;        // odd spectrum rows; both at twice the reference's normalisation
    void* R;
    const float2* tw;
    int uH, NT;
};

// grid (uH/2, 3), block UW/8.  u = 2: kx = 0..UW/4 non-zero.  LDS: lswz_size(UW) float2.
template <int UW, bool HALF_OUT, int TK, bool WIDE>
__global__ void __launch_bounds__(UW / 8) k_row_c2r_t(RowC2RTParams p)
{
    constexpr int E = 8, T = UW / E;                 // T = UW/8; W/2 = UW/4 = 2T
    __shared__ __attribute__((aligned(128))) float2 buf[lswz_size(UW)];
    const int tid = threadIdx.x, j = blockIdx.x, c = blockIdx.y;
    const long tile_stride = (long)(p.uH / 2) * TK;
    const long roff = (long)c * p.NT * tile_stride + (long)j * TK;
    auto ldAB = [&](int k, float2& A, float2& B) {
        const long o = roff + (long)(k / TK) * tile_stride + (k % TK);
        A = p.S1[o];                                 // row 2j
        B = p.S2[o];                                 // row 2j+1
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
    constexpr float inv = 0.5f / (float)UW;          // 1/2: the spectrum rows carry twice the reference's scale
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
            for (int e = 0; e < 4; e++) z[e] = buf[lswz(n0 + e)];
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

// ---- two pixels per operation.  The vector ALUs of gfx950 issue one instruction per wave every four cycles whatever it
// does; v_pk_add/mul/fma_f32 retire two lanes' worth of fp32 work in that slot, so the filter is written on pairs of
// pixels: 12 packed + 8 one-lane (min, W+E, v_rsq, v_rcp) operations per pair instead of 2 x 22.
// Same formula as sharpen_eval_fast with the two exact halvings dropped (everything is carried doubled:
// smn = 2 mn, smx = 2 mx) and the a < b selection written as n = min(mn, 1 - mx), d = max(1 - mn, mx)
// (a < b <=> mn + mx < 1 <=> mn < 1 - mx <=> 1 - mn > mx).
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v mk2(float a, float b) { f2v r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2v sharpen_eval_pair(f2v N, f2v S, f2v WE, f2v C, f2v mn0, f2v mn1, f2v mx0, f2v mx1, float coef)
{
    const f2v two = mk2(2.0f, 2.0f);
    const f2v smn = mn0 + mn1, smx = mx0 + mx1;
    const f2v u = two - smx;
    const f2v n2 = mk2(fminf(smn.x, u.x), fminf(smn.y, u.y));
    // d = max(1 - mn, mx) = 1 - min(mn, 1 - mx) = 1 - n, bit for bit: when the minimum is u = 2 - smx, smx >= 1 and both
    // subtractions are exact (Sterbenz); when it is smn, 2 - n2 IS 2 - smn
    const f2v d2 = two - n2;                                                            // in [1, 2]
    const f2v pr = __builtin_elementwise_fma(n2, d2, mk2(1e-30f, 1e-30f));               // n2 = 0 -> r = 0, no NaN
    const f2v r = n2 * mk2(__builtin_amdgcn_rsqf(pr.x), __builtin_amdgcn_rsqf(pr.y));   // sqrt(n/d)
    const f2v s4 = (WE + N) + S;
    const f2v num = __builtin_elementwise_fma(mk2(-coef, -coef), r * s4, C);
    const f2v den = __builtin_elementwise_fma(mk2(-4.0f * coef, -4.0f * coef), r, mk2(1.0f, 1.0f));
    return num * mk2(__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y));
}

// one-lane addition the vectoriser cannot re-pack
__device__ __forceinline__ float add1(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// One row of taps for four pixels: q = the pixels x0..x0+3 exactly as their 16-byte LDS read delivered them (a 4-aligned
// register tuple, so (q.x, q.y) and (q.z, q.w) ARE the aligned pairs the packed operations want), l / r = the pixels
// x0-1 / x0+4.  (A plain float[6] per row makes the compiler gather every row into six consecutive registers:
// ~10 moves per row, a tenth of the kernel's vector instructions.)
typedef float f4t __attribute__((ext_vector_type(4)));
struct TapRow {
    f4t q;
    float l, r;
    __device__ __forceinline__ float operator[](int k) const { return k == 0 ? l : k == 5 ? r : q[(k - 1) & 3]; }
    __device__ __forceinline__ f2v lo() const { return __builtin_shufflevector(q, q, 0, 1); }
    __device__ __forceinline__ f2v hi() const { return __builtin_shufflevector(q, q, 2, 3); }
};
// vertical 3-row minima / maxima of 6 columns (4 pixels + halo) for the window whose top row is t[w]
__device__ __forceinline__ void sharpen_vminmax(const TapRow (&t)[4], int w, float (&vmn)[6], float (&vmx)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        vmn[i] = fminf(fminf(t[w][i], t[w + 1][i]), t[w + 2][i]);
        vmx[i] = fmaxf(fmaxf(t[w][i], t[w + 1][i]), t[w + 2][i]);
    }
}
// 4 output pixels of the window rows t[w..w+2] (7 three-input min/max per pixel: min/max are exact in any order)
__device__ __forceinline__ f4t sharpen_quad_packed(const TapRow (&t)[4], int w, const float (&vmn)[6], const float (&vmx)[6], float coef)
{
    const TapRow& c = t[w + 1];
    float mn0[4], mn1[4], mx0[4], mx1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        mn1[k] = fminf(fminf(vmn[k], vmn[k + 1]), vmn[k + 2]);                          // full 3x3
        mx1[k] = fmaxf(fmaxf(vmx[k], vmx[k + 1]), vmx[k + 2]);
        mn0[k] = fminf(fminf(vmn[k + 1], c[k]), c[k + 2]);                              // cross: N, C, S, W, E
        mx0[k] = fmaxf(fmaxf(vmx[k + 1], c[k]), c[k + 2]);
    }
    // (W + E by one-lane additions: the horizontally shifted pairs (x-1, x) and (x+1, x+2) straddle the aligned register
    // pairs the packed operations need, and building them costs more moves than a packed add saves)
    const f2v r01 = sharpen_eval_pair(t[w].lo(), t[w + 2].lo(), mk2(add1(c[0], c[2]), add1(c[1], c[3])), c.lo(),
                                      mk2(mn0[0], mn0[1]), mk2(mn1[0], mn1[1]), mk2(mx0[0], mx0[1]), mk2(mx1[0], mx1[1]), coef);
    const f2v r23 = sharpen_eval_pair(t[w].hi(), t[w + 2].hi(), mk2(add1(c[2], c[4]), add1(c[3], c[5])), c.hi(),
                                      mk2(mn0[2], mn0[3]), mk2(mn1[2], mn1[3]), mk2(mx0[2], mx0[3]), mk2(mx1[2], mx1[3]), coef);
    return __builtin_shufflevector(r01, r23, 0, 1, 2, 3);
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
    const float2* S1;        // even spectrum rows = the forward spectrum; odd rows at S1 + odd_delta (one allocation)
    unsigned odd_delta;      // in float2 elements
    void* out;               // dense [3][uH][uW] float / half
    const float2* tw;
    int uH, NT;
    int pairs_per_strip;
    float upsq, coef;
    int u8_wrap;             // OUT_U8 kernels: `out` is the interleaved 8-bit RGB image [uH][uW][3]; FFTUP_FLAG_U8_WRAP
};
// min(|x|, 1) in one instruction.  (Written with fminf(fabsf(x), 1.0f) the compiler first canonicalises -- v_max x, x -- an
// x that comes out of the inline-asm butterflies, since it cannot know that it is not a signalling NaN.)
__device__ __forceinline__ float absmin1(float x)
{
    float r;
    asm("v_min_f32_e64 %0, |%1|, 1.0" : "=v"(r) : "v"(x));
    return r;
}
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

// ---- -p 2 sharpen on PAIRS of pixels in packed binary16.  The reference evaluates this shader in float16_t
// (VkResample.cpp:823-826): every addition and multiplication below is one v_pk_*_f16 instruction, i.e. rounded to
// binary16 per operation like the reference's (contraction is off: C + scale*s4 rounds twice, as there).  Same
// algebra as sharpen_eval_pair: values carried doubled (2 - smx = 2 (1 - mx) exactly), n = min(mn, 1 - mx),
// d = max(1 - mn, mx).  The inner quotient is the native binary16 reciprocal plus one residual step, the root the
// native binary16 square root (1 ulp; the Vulkan spec allows the reference's own fp16 division 2.5 ulp), the final
// quotient again reciprocal plus residual step.  k_sharpen_t keeps the exactly rounded sequence, bit for bit against the oracle.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v h2_bits(unsigned u) { return __builtin_bit_cast(h2v, u); }
__device__ __forceinline__ unsigned bits_h2(h2v h) { return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ h2v h2_splat(float f) { const _Float16 x = (_Float16)f; h2v r = {x, x}; return r; }
// gfx950 has three-input packed binary16 minimum / maximum (IEEE-754-2019 minimum/maximum; the operands here are
// finite and non-negative, where they agree with min/max): half the instructions of the two-input forms
__device__ __forceinline__ h2v pk_min3(h2v a, h2v b, h2v c)
{
    h2v r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ h2v pk_max3(h2v a, h2v b, h2v c)
{
    h2v r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v_rcp_f16 / v_sqrt_f16 of both halves of a register, the second one through SDWA straight into the upper half (left to the
// compiler: two one-half results and a v_pack_b32_f16, 24 extra instructions per thread and row pair).  The s_nop: a transcendental
// result needs a wait state before the next vector instruction reads it -- here the preserved lower half)
__device__ __forceinline__ h2v pk_rcp_h(h2v b)
{
    h2v r;
    asm("v_rcp_f16_e32 %0, %1\n\ts_nop 0\n\tv_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(r) : "v"(b));
    return r;
}
__device__ __forceinline__ h2v pk_sqrt_h(h2v b)
{
    h2v r;
    asm("v_sqrt_f16_e32 %0, %1\n\ts_nop 0\n\tv_sqrt_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(r) : "v"(b));
    return r;
}
// a / b in packed binary16: native reciprocal plus one residual step -- RN(a / b) but for rare ties
__device__ __forceinline__ h2v pk_div(h2v a, h2v b)
{
#pragma clang fp contract(off)
    const h2v rc = pk_rcp_h(b);
    const h2v q = a * rc;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-q, b, a), rc, q);
}
__device__ __forceinline__ h2v sharpen_eval_pair_half(h2v N, h2v S, h2v Wv, h2v E, h2v C, h2v mn0, h2v mn1, h2v mx0, h2v mx1, h2v ncoef)
{
#pragma clang fp contract(off)
    const h2v one = {(_Float16)1.0f, (_Float16)1.0f}, two = {(_Float16)2.0f, (_Float16)2.0f}, four = {(_Float16)4.0f, (_Float16)4.0f};
    const h2v smn = mn0 + mn1, smx = mx0 + mx1;
    const h2v u = two - smx;
    const h2v n2 = __builtin_elementwise_min(smn, u);
    const h2v d2 = two - n2;                           // = max(2 - smn, smx) exactly (see sharpen_eval_pair); in [1, 2]
    const h2v q = pk_div(n2, d2);
    const h2v r = pk_sqrt_h(q);
    const h2v scale = ncoef * r;
    const h2v s4 = ((N + Wv) + E) + S;
    const h2v prod = scale * s4;
    const h2v num = C + prod;
    const h2v den = __builtin_elementwise_fma(scale, four, one);      // 4 * scale is exact: one rounding either way, one instruction
    return pk_div(num, den);
}
// ---- the same on the FOUR pixels of a quad at once (two register pairs per value): every operation becomes two independent
// v_pk_*_f16 instructions back to back.  On gfx950 a packed binary16 result needs one wait state before the next vector
// instruction may read it, and the filter is ONE dependency chain from the minima to the quotient: evaluated pair by pair the
// compiler pads it with an s_nop behind nearly every instruction (33 per row of quads, 132 of the 880 instructions of a step);
// with the two pairs of a quad interleaved the other pair's instruction IS the wait state.  (Measured, round 6: 187 -> 71 s_nop in
// the kernel and the same time -- the padding costs the wave issue cycles, the vector ALU none, and it is the ALU the kernel is
// short of; profiles/r06_d_quad.txt.  Tabulating the weight scale(d) over its ONE binary16 argument d = min(smn, 2 - smx) in LDS --
// 15361 entries, 30 KB, the shader's per-operation roundings exactly; 10 packed + 8 transcendental instructions fewer per row
// of quads -- makes the kernel alone 6 % faster (57.5 -> 54.0 us) and the overlapped frame 1.4 % slower: 94 instead of 64 KB
// of LDS per strip leave the other frames' row and column workgroups less room on the unit.  Not kept: profiles/r06_e_lut.txt,
// the patch beside it.)
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h4v h4_cat(h2v a, h2v b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
__device__ __forceinline__ h2v h4_lo(h4v a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ h2v h4_hi(h4v a) { return __builtin_shufflevector(a, a, 2, 3); }
// v_rcp_f16 / v_sqrt_f16 of four halves: the lower halves of both registers first, then the upper ones through SDWA (each reads
// the lower half its predecessor-but-one wrote: the instruction between them is the wait state a transcendental result needs)
__device__ __forceinline__ h4v pk4_rcp_h(h4v b)
{
    h2v r0, r1;
    const h2v b0 = h4_lo(b), b1 = h4_hi(b);
    asm("v_rcp_f16_e32 %0, %2\n\tv_rcp_f16_e32 %1, %3\n\t"
        "v_rcp_f16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_rcp_f16_sdwa %1, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(r0), "=&v"(r1) : "v"(b0), "v"(b1));
    return h4_cat(r0, r1);
}
__device__ __forceinline__ h4v pk4_sqrt_h(h4v b)
{
    h2v r0, r1;
    const h2v b0 = h4_lo(b), b1 = h4_hi(b);
    asm("v_sqrt_f16_e32 %0, %2\n\tv_sqrt_f16_e32 %1, %3\n\t"
        "v_sqrt_f16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_sqrt_f16_sdwa %1, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(r0), "=&v"(r1) : "v"(b0), "v"(b1));
    return h4_cat(r0, r1);
}
__device__ __forceinline__ h4v pk4_div(h4v a, h4v b)
{
#pragma clang fp contract(off)
    const h4v rc = pk4_rcp_h(b);
    const h4v q = a * rc;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-q, b, a), rc, q);
}
__device__ __forceinline__ h4v sharpen_eval_quad_half(h4v N, h4v S, h4v Wv, h4v E, h4v C, h4v mn0, h4v mn1, h4v mx0, h4v mx1, h4v ncoef)
{
#pragma clang fp contract(off)
    const _Float16 f1 = (_Float16)1.0f, f2 = (_Float16)2.0f, f4 = (_Float16)4.0f;
    const h4v one = {f1, f1, f1, f1}, two = {f2, f2, f2, f2}, four = {f4, f4, f4, f4};
    const h4v smn = mn0 + mn1, smx = mx0 + mx1;
    const h4v u = two - smx;
    const h4v n2 = __builtin_elementwise_min(smn, u);
    const h4v d2 = two - n2;
    const h4v q = pk4_div(n2, d2);
    const h4v r = pk4_sqrt_h(q);
    const h4v scale = ncoef * r;
    const h4v s4 = ((N + Wv) + E) + S;
    const h4v prod = scale * s4;
    const h4v num = C + prod;
    const h4v den = __builtin_elementwise_fma(scale, four, one);
    return pk4_div(num, den);
}
// one window (three rows) of four pixels: P[r] = the five column pairs (-1,0) (0,1) (1,2) (2,3) (3,4) of row r
struct H2Row { h2v sa, h01, sb, h23, sc; };
__device__ __forceinline__ void sharpen_quad_half(const H2Row& r0, const H2Row& r1, const H2Row& r2, h2v ncoef, h2v& o01, h2v& o23)
{
#define V3MIN(f) pk_min3(r0.f, r1.f, r2.f)
#define V3MAX(f) pk_max3(r0.f, r1.f, r2.f)
    const h2v na = V3MIN(sa), nb = V3MIN(h01), nc = V3MIN(sb), nd = V3MIN(h23), ne = V3MIN(sc);
    const h2v xa = V3MAX(sa), xb = V3MAX(h01), xc = V3MAX(sb), xd = V3MAX(h23), xe = V3MAX(sc);
#undef V3MIN
#undef V3MAX
    // pixels 0,1: columns (-1,0) (0,1) (1,2); cross = N, C, S (the vertical triple of the centre pair) and W, E
    const h2v mn1a = pk_min3(na, nb, nc), mx1a = pk_max3(xa, xb, xc);
    const h2v mn0a = pk_min3(nb, r1.sa, r1.sb), mx0a = pk_max3(xb, r1.sa, r1.sb);
    // pixels 2,3: columns (1,2) (2,3) (3,4)
    const h2v mn1b = pk_min3(nc, nd, ne), mx1b = pk_max3(xc, xd, xe);
    const h2v mn0b = pk_min3(nd, r1.sb, r1.sc), mx0b = pk_max3(xd, r1.sb, r1.sc);
    // both pairs through the filter's arithmetic together (sharpen_eval_quad_half: same operations, interleaved)
    const h4v o = sharpen_eval_quad_half(h4_cat(r0.h01, r0.h23), h4_cat(r2.h01, r2.h23), h4_cat(r1.sa, r1.sb), h4_cat(r1.sb, r1.sc), h4_cat(r1.h01, r1.h23),
                                         h4_cat(mn0a, mn0b), h4_cat(mn1a, mn1b), h4_cat(mx0a, mx0b), h4_cat(mx1a, mx1b), h4_cat(ncoef, ncoef));
    o01 = h4_lo(o);
    o23 = h4_hi(o);
}

// =================================================================================== transform plans of the fused kernel
// What k_c2r_sharpen_g needs from a transform of length UW run by T threads: the first stage is a radix-R0 butterfly per
// thread j < NB0 = UW/R0 on registers (inputs Z[j + NB0*m]); on return thread lt < SOUT owns X[lt + SOUT*q], q < EOUT.
template <int UW_> struct FusedPlanPow2 {                   // radix-8 Stockham, 8 points per thread (reg_fft)
    static constexpr int UW = UW_, T = UW / 8, R0 = 8, NB0 = T, EOUT = 8, SOUT = T, VN = 8;

    static constexpr size_t XB = sizeof(float2) * lswz_size(UW);
    // NBUF = 3: an LDS buffer z used by the transform only.  The exchanges alternate z, c, z (c = the buffer that
    // receives this pair's L rows), one barrier each; the first one writes z, which nobody has read since the last
    // gather of the previous step, so the step needs no barrier at its end either: 4 barriers per step instead of 9.
    static constexpr int NBUF = 3;
    static constexpr bool RING_REGS = true;
    static constexpr int WPE = T * 2 / 256 > 0 ? T * 2 / 256 : 1;      // two strips can share a compute unit
    static_assert((num_stages(UW, 8) - 1) % 2 == 1, "the first exchange must go through z");
    struct Tw { TwSet<UW, 8> t; };
    // MIRROR_SHARE: the first-stage butterfly jj of a thread needs Z[jj + NB0 m] and, for the conjugate half of the
    // Hermitian spectrum, Z[KH - jj - NB0 m] -- which are the elements Z[jj' + NB0 (NI-1-m)] of butterfly jj' = NB0 - jj.
    // So butterflies jj and NB0 - jj sit in lanes l and l ^ 32 of one wave (lanes 0-31: jj = 32 w + l, lanes 32-63:
    // jj = NB0 - 32 w - (l - 32)), every thread loads its own elements only and takes the mirror ones from its partner
    // through the LDS crossbar (ds_bpermute): half the prefetch requests.  (jj = 0 has no partner and jj = NB0/2 is its
    // own: lanes 0 and 32 of wave 0, which load their mirror elements themselves.)
    static constexpr bool MIRROR_SHARE = (T % 64 == 0);
    static __device__ __forceinline__ int first_index(int lt)      // first-stage butterfly of thread lt
    {
        if constexpr (!MIRROR_SHARE) return lt;
        const int w = lt >> 6, l = lt & 63;
        return l < 32 ? 32 * w + l : (lt == 32 ? NB0 / 2 : NB0 - 32 * w - (l - 32));
    }
    static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j) { w.t.load(tw, j); }
    // jj = first_index(j): the butterfly of the first stage; from the first exchange on thread j owns X[j + T i]
    static __device__ __forceinline__ void fft(float2 (&v)[VN], float2* __restrict__ buf, float2* __restrict__ zbuf, int j, const Tw& w, int jj)
    {
        reg_fft_pp<UW, 8, -1>(v, buf, zbuf, j, w.t, jj);
    }
};

// ---- three-stage mixed-radix transform, ONE butterfly per thread and stage (radices up to 16: 3840 = 16 * 16 * 15).
// Stage s has NBs = N/Rs butterflies; butterfly j reads in[j + NBs*m], m < Rs, and writes out[(j - k)*Rs + k + m*Ns],
// k = j % Ns (Stockham autosort), in place in one LDS buffer.  LDS map: one padding element per 16 (lpad).  With radices
// 16, 16, R2 and NB1, NB2 multiples of 16 every access of a thread is ONE per-thread base plus a compile-time offset
// (lpad(16 j + m) = 17 j + m, lpad(j + NB m) = lpad(j) + (NB + NB/16) m, lpad(16 (j-k+m) + k) = 17 (j-k) + k + 17 m),
// i.e. an immediate of the ds instruction: no address arithmetic in the stages.  The scatters are conflict-free, the
// gathers pay one extra LDS cycle per 32 lanes for the padding (tools/lds_conflicts.py).
template <int R, bool PK = true> __device__ __forceinline__ void twiddle_all(float2* v, float2 w1)      // v[m] *= w1^m, m < R <= 16
{
    // PK: through cmul_tw, the power chain as well -- its results are born as aligned register pairs, which is what the
    // packed multiplies want (from plain cmul they would have to be moved together first).  Same roundings either way.
    auto mul = [](float2 a, float2 b) { if constexpr (PK) return cmul_tw(a, b); else return cmul(a, b); };
    const float2 w2 = mul(w1, w1), w3 = mul(w2, w1), w4 = mul(w2, w2), w5 = mul(w4, w1), w6 = mul(w3, w3), w7 = mul(w4, w3);
    const float2 w8 = mul(w4, w4);
    const float2 ws[16] = {make_float2(1.f, 0.f), w1, w2, w3, w4, w5, w6, w7, w8, mul(w8, w1), mul(w5, w5), mul(w8, w3),
                           mul(w6, w6), mul(w8, w5), mul(w7, w7), mul(w8, w7)};
#pragma unroll
    for (int m = 1; m < R; m++) v[m] = mul(v[m], ws[m]);
}
template <int N, int DIR, int R0, int R1, int R2> struct MrFft {
    static constexpr int NB0 = N / R0, NB1 = N / R1, NB2 = N / R2;
    static constexpr int NS1 = R0, NS2 = R0 * R1;
    static constexpr int VN = (R0 > R1 ? (R0 > R2 ? R0 : R2) : (R1 > R2 ? R1 : R2));
    struct Tw { float2 w1, w2; };                           // base twiddles of stages 1 and 2 (table sign exp(+i..))
    static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j)
    {
        w.w1 = tw[((j < NB1 ? j : 0) % NS1) * (N / (NS1 * R1))];
        w.w2 = tw[(j < NB2 ? j : 0) % NS2 * (N / (NS2 * R2))];
    }
    static_assert(R0 == 16 && R1 == 16 && NB1 % 16 == 0 && NB2 % 16 == 0, "offsets below assume 16-aligned strides");
    // zbuf != buf: the first exchange goes through zbuf (nobody has read it since the previous transform's first gather),
    // so no barrier is needed between its gather and the second exchange's scatter into buf.
    static __device__ __forceinline__ void fft(float2 (&v)[VN], float2* __restrict__ buf, float2* __restrict__ zbuf, int j, const Tw& w)
    {
        float2* const g = buf + lpad(j);                     // gather base of stage 2
        if (j < NB0) {
            bfly_reg<R0, DIR>(v);
            float2* const d = zbuf + 17 * j;
#pragma unroll
            for (int m = 0; m < R0; m++) d[m] = v[m];
        }
        __syncthreads();
        if (j < NB1) {
            float2* const g0 = zbuf + lpad(j);
#pragma unroll
            for (int m = 0; m < R1; m++) v[m] = g0[(NB1 + NB1 / 16) * m];
        }
        if (zbuf == buf) __syncthreads();                    // (uniform; in place: the buffer may be overwritten from here on)
        if (j < NB1) {
            twiddle_all<R1>(v, twid<DIR>(w.w1));
            bfly_reg<R1, DIR>(v);
            const int k = j % NS1;
            float2* const d = buf + 17 * (j - k) + k;
#pragma unroll
            for (int m = 0; m < R1; m++) d[17 * m] = v[m];
        }
        __syncthreads();
        if (j < NB2) {
#pragma unroll
            for (int m = 0; m < R2; m++) v[m] = g[(NB2 + NB2 / 16) * m];
        }
        __syncthreads();                                     // the buffer may be overwritten from here on
        if (j < NB2) {
            twiddle_all<R2>(v, twid<DIR>(w.w2));             // k = j (NS2 * R2 = N)
            bfly_reg<R2, DIR>(v);
        }
    }
};
// ---- the same with TK interleaved sequences and NO index map: when the first radix is odd the stage-0 scatter of 16
// consecutive lanes (R0 elements apart) already covers all 16 eight-byte slots, the later scatters and all gathers hit
// consecutive elements -- every access is a per-thread base plus an immediate.  Element (i, col) lives at i*TK + col.
struct MrTw { float2 w1, w2; };                             // base twiddles of stages 1 and 2 (table sign exp(+i..))
template <int N, int DIR, int TK, int R0, int R1, int R2, bool FINAL_TO_LDS> struct MrFftT {
    // (an odd first radix spreads the stage-0 scatter over the banks without a map; even ones work, with conflicts)
    static_assert(R0 * R1 * R2 == N, "radices multiply to N");
    static constexpr int NB0 = N / R0, NB1 = N / R1, NB2 = N / R2;
    static constexpr int NS1 = R0, NS2 = R0 * R1;
    static constexpr int VN = (R0 > R1 ? (R0 > R2 ? R0 : R2) : (R1 > R2 ? R1 : R2));
    static constexpr int TC = (NB0 > NB1 ? (NB0 > NB2 ? NB0 : NB2) : (NB1 > NB2 ? NB1 : NB2));     // threads per sequence
    using Tw = MrTw;
    static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j)
    {
        w.w1 = tw[((j < NB1 ? j : 0) % NS1) * (N / (NS1 * R1))];
        w.w2 = tw[((j < NB2 ? j : 0) % NS2) * (N / (NS2 * R2))];
    }
    // on entry v[m] = x[j + NB0*m] (j < NB0); on return X[j + NB2*m] in v (j < NB2) or, FINAL_TO_LDS, in buf (synced)
    static __device__ __forceinline__ void run(float2 (&v)[VN], float2* __restrict__ buf, int j, int col, const Tw& w)
    {
        float2* const g = buf + j * TK + col;
        if (j < NB0) {
            bfly_reg<R0, DIR>(v);
            float2* const d = buf + j * R0 * TK + col;
#pragma unroll
            for (int m = 0; m < R0; m++) d[m * TK] = v[m];
        }
        __syncthreads();
        if (j < NB1) {
#pragma unroll
            for (int m = 0; m < R1; m++) v[m] = g[m * NB1 * TK];
        }
        __syncthreads();
        if (j < NB1) {
            twiddle_all<R1>(v, twid<DIR>(w.w1));
            bfly_reg<R1, DIR>(v);
            const int k = j % NS1;
            float2* const d = buf + ((j - k) * R1 + k) * TK + col;
#pragma unroll
            for (int m = 0; m < R1; m++) d[m * NS1 * TK] = v[m];
        }
        __syncthreads();
        if (j < NB2) {
#pragma unroll
            for (int m = 0; m < R2; m++) v[m] = g[m * NB2 * TK];
        }
        __syncthreads();
        if (j < NB2) {
            twiddle_all<R2>(v, twid<DIR>(w.w2));             // k = j (NS2 * R2 = N)
            bfly_reg<R2, DIR>(v);
            if constexpr (FINAL_TO_LDS) {
#pragma unroll
                for (int m = 0; m < R2; m++) g[m * NB2 * TK] = v[m];
            }
        }
        if constexpr (FINAL_TO_LDS) __syncthreads();
    }
};

// ---- any number of stages, at most 8 points per butterfly except the last: T threads run ceil(NB/T) butterflies per
// stage, index map lswz.  What the run-time specialised plans (jit.hpp) instantiate for lengths without a three-stage plan.
template <int N, int DIR, int T, int TK, int... RS> struct MrFftNT {
    static constexpr int NST = sizeof...(RS);
    static constexpr int rs(int s) { constexpr int r[] = {RS...}; return r[s]; }
    static constexpr int ns(int s) { int n = 1; for (int i = 0; i < s; i++) n *= rs(i); return n; }
    static constexpr int bpt(int s) { return (N / rs(s) + T - 1) / T; }
    static constexpr int vn() { int m = 0; for (int s = 0; s < NST; s++) m = bpt(s) * rs(s) > m ? bpt(s) * rs(s) : m; return m; }
    static constexpr int mb() { int m = 0; for (int s = 0; s < NST; s++) m = bpt(s) > m ? bpt(s) : m; return m; }
    static constexpr int VN = vn(), MB = mb();
    static_assert(ns(NST) == N && bpt(0) == 1 && bpt(NST - 1) == 1, "radices multiply to N; one butterfly per thread at both ends");
    struct Tw { float2 w[NST - 1][MB]; };                    // base twiddle of stage s >= 1, butterfly b (table sign exp(+i..))
    template <int S = 1> static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j)
    {
        if constexpr (S < NST) {
            constexpr int R = rs(S), Ns = ns(S), NB = N / R;
#pragma unroll
            for (int b = 0; b < bpt(S); b++) {
                const int jb = j + T * b;
                w.w[S - 1][b] = tw[((jb < NB ? jb : 0) % Ns) * (N / (Ns * R))];
            }
            load_tw<S + 1>(w, tw, j);
        }
    }
    // on entry v[m] = x[j + NB0*m], j < NB0; on return v[m] = X[j + NBlast*m], j < NBlast.
    // The exchange behind stage s goes through z when an even number of exchanges follows it, else through c (see
    // reg_fft_pp: one barrier per exchange, the last one through z).
    static constexpr bool pow2(int n) { return (n & (n - 1)) == 0; }
    // INPLACE: one buffer (c), two barriers per exchange.
    // (TK interleaved sequences: element i of sequence col lives at lswz(i*TK + col); j = butterfly index within the sequence)
    template <bool INPLACE, int S = 0>
    static __device__ __forceinline__ void run(float2 (&v)[VN], float2* __restrict__ c, float2* __restrict__ z, int j, const Tw& w, int col = 0)
    {
        if constexpr (S < NST) {
            constexpr int R = rs(S), Ns = ns(S), NB = N / R, BPT = bpt(S), NE = NST - 1;
            float2* __restrict__ const bin = (!INPLACE && (NE - S) % 2 == 0) ? z : c;           // exchange S-1
            float2* __restrict__ const bout = (!INPLACE && (NE - 1 - S) % 2 == 0) ? z : c;      // exchange S
            asm volatile("" : "+v"(j));                      // addresses of this stage are formed here, not hoisted and kept
            if constexpr (S > 0) {
#pragma unroll
                for (int b = 0; b < BPT; b++) {
                    const int jb = j + T * b;
                    if (jb < NB) {
#pragma unroll
                        for (int m = 0; m < R; m++) v[b * R + m] = bin[lidx<TK>(jb + NB * m, col)];
                    }
                }
                if constexpr (INPLACE) __syncthreads();      // the buffer may be overwritten from here on
            }
#pragma unroll
            for (int b = 0; b < BPT; b++) {
                const int jb = j + T * b;
                if (jb < NB) {
                    if constexpr (S > 0) twiddle_all<R>(&v[b * R], twid<DIR>(w.w[S > 0 ? S - 1 : 0][b]));
                    bfly_reg<R, DIR>(&v[b * R]);
                    if constexpr (S + 1 < NST) {
                        const int k = jb % Ns, j0 = (jb - k) * R + k;
                        if constexpr (pow2(R) && pow2(Ns)) {
                            // bits of m*Ns are clear in j0: lswz(j0 + m*Ns) = lswz(j0) ^ lswz(m*Ns) (see lds_put)
                            const unsigned a0 = lds_addr(bout) + 8u * (unsigned)lidx<TK>(j0, col);
#pragma unroll
                            for (int m = 0; m < R; m++) lds_put(a0, lswz_c(m * Ns * TK), v[b * R + m]);
                        } else {
#pragma unroll
                            for (int m = 0; m < R; m++) bout[lidx<TK>(j0 + m * Ns, col)] = v[b * R + m];
                        }
                    }
                }
            }
            if constexpr (S + 1 < NST) {
                __syncthreads();
                run<INPLACE, S + 1>(v, c, z, j, w, col);
            }
        }
    }
};
template <int N, int DIR, int T, int... RS> using MrFftN = MrFftNT<N, DIR, T, 1, RS...>;

// Rows of any length UW = R0 * ... on T threads, any number of stages (MrFftN): what the run-time specialised plans
// (jit.hpp) instantiate when no three-stage 16 * 16 * R2 plan exists.  R0 must be a multiple of 2U (U = upscale factor: the non-zero 1/2U of the spectrum fills
// whole first-stage inputs), T >= UW/R0 and T >= UW/Rlast.
template <int UW_, int T_, int NBUF_, int WPE_, bool RR_, int... RS> struct FusedPlanN {
    using F = MrFftN<UW_, -1, T_, RS...>;
    static constexpr int UW = UW_, T = T_, R0 = F::rs(0), NB0 = UW / R0, EOUT = F::rs(F::NST - 1), SOUT = UW / EOUT, VN = F::VN;
    static constexpr size_t XB = sizeof(float2) * lswz_size(UW);
    // NBUF = 3 (exchanges alternate z, c, z with one barrier each, as in FusedPlanPow2) makes the 3840 kernel 5 % faster on
    // its own and the frame 2 % slower: with 61 KB of LDS two of these workgroups share a compute unit whenever consecutive
    // frames' launches overlap, with 92 KB they cannot (measured).
    static constexpr int NBUF = NBUF_;
    static constexpr bool RING_REGS = RR_;                 // false: the previous pair's L rows in a second LDS buffer instead of 12 registers per pass
    static constexpr int WPE = WPE_;                       // waves per SIMD the register allocation must allow (launch bound)
    static_assert(T >= NB0 && T >= SOUT, "one butterfly per thread at both ends");        // (and R0 a multiple of 2U: k_c2r_sharpen_g)
    static_assert(NBUF == 2 || (F::NST - 1) % 2 == 1, "three buffers: the first exchange must go through z");
    static_assert(XB % 128 == 0, "lds_put needs 128-byte aligned buffers");
    using Tw = typename F::Tw;
    static constexpr bool MIRROR_SHARE = false;
    static __device__ __forceinline__ int first_index(int lt) { return lt < NB0 ? lt : NB0 - 1; }   // (threads beyond re-read)
    static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j) { F::load_tw(w, tw, j); }
    static __device__ __forceinline__ void fft(float2 (&v)[VN], float2* __restrict__ buf, float2* __restrict__ zbuf, int j, const Tw& w, int = 0)
    {
        // The base twiddles are loop-invariant, so the compiler would hoist all power products of the twiddled stages
        // out of the strip loop and keep them (54 VGPRs for 8 * 8 * 4 * 15: spills).  Re-defining the bases here makes
        // the powers per-step work.
        Tw t = w;
#pragma unroll
        for (int st = 0; st < F::NST - 1; st++)
#pragma unroll
            for (int b = 0; b < F::MB; b++) asm volatile("" : "+v"(t.w[st][b].x), "+v"(t.w[st][b].y));
        F::template run<NBUF == 2>(v, buf, zbuf, j, t);
    }
};
// Rows of UW = 16 * 16 * R2 = 256 * R2 points on 256 threads -- one wave per SIMD -- R2 points per thread in the last stage
// (3840: R2 = 15, the default for 1920x1080; 2560: R2 = 10 for 1280x720).  120 VGPRs + the ring rows.
template <int UW_, int R2_> struct FusedPlanMr16 {
    using F = MrFft<UW_, -1, 16, 16, R2_>;
    static constexpr int UW = UW_, T = 256, R0 = 16, NB0 = F::NB0, EOUT = R2_, SOUT = F::NB2, VN = F::VN;
    static_assert(F::NB2 == T && F::NB0 <= T && F::NB1 <= T, "one butterfly per thread and stage, all threads in the last one");
    static constexpr size_t XB = (sizeof(float2) * lpad_size(UW) + 15) & ~(size_t)15;
    using Tw = typename F::Tw;
    static __device__ __forceinline__ int first_index(int lt) { return lt < NB0 ? lt : NB0 - 1; }   // (threads beyond re-read)
    static __device__ __forceinline__ void load_tw(Tw& w, const float2* __restrict__ tw, int j) { F::load_tw(w, tw, j); }
    // NBUF = 3 (first exchange through z, second through the L-row buffer: 4 barriers per step instead of 6) changes
    // nothing for the kernel alone (77 us at 3840) and costs 98 instead of 65 KB of LDS: frame 75 -> 87 us.  Kept at 2.
    static constexpr int NBUF = 2;
    static constexpr bool RING_REGS = true;
    static constexpr int WPE = 2;
    static constexpr bool MIRROR_SHARE = false;
    static __device__ __forceinline__ void fft(float2 (&v)[VN], float2* __restrict__ buf, float2* __restrict__ zbuf, int j, const Tw& w, int = 0)
    {
        F::fft(v, buf, NBUF == 3 ? zbuf : buf, j, w);
    }
};
using FusedPlan3840x16 = FusedPlanMr16<3840, 15>;

// ---------------------------------------------------------------------------------------------------
// Fused C2R + sharpen: a workgroup of T = UW/8 threads owns a strip and alternates, all
// threads together, between transforming row pair s and sharpening the two output rows that pair completes.  One such
// workgroup runs per compute unit by default (<= 96 VGPRs and 64 KB of LDS for the power-of-two plans): the rest of the
// unit is left to the row and column kernels of the frames on the other streams and to the next frame's strip.  No role
// split, no barrier counting: every thread reaches the same __syncthreads().
//   LDS: the L rows of the pair in flight (buffer c, which is also one of the transform's exchange buffers); the rows of
//   pair s-1 -- the "ring" -- live in registers (RR, FusedGLds) or, without RR, in a second L-row buffer (X[s&1] /
//   X[(s-1)&1] alternating); plans with NBUF = 3 exchange through a buffer z as well and need no barrier at the end
//   of a step (FusedPlanPow2).  The spectrum rows of pair s+1 are prefetched into REGISTERS while pair s is
//   processed (every thread loads the 8 values its first butterfly needs, mirrored ones included), so no LDS staging
//   and -- loads being older than the output stores of the same step -- no wait on a store, ever (vmcnt is in order).
template <class PL> struct FusedGLds {
    static constexpr size_t XB = PL::XB;
    // RR: the ring rows (the L rows of the previous pair) stay in the registers of the threads that read them as taps in
    // the previous step -- no second L-row buffer: LDS = the L rows of the current pair + the transform's z buffer.
    // (12 registers per sharpen pass for fp32, 10 for binary16; plans opt in with RING_REGS, at most four passes.)
    static constexpr bool RR = PL::RING_REGS && (PL::UW + 4 * PL::T - 1) / (4 * PL::T) <= 4;
    static constexpr size_t NX = RR ? 1 : 2;                                // L-row buffers
    static constexpr size_t ZOFF = NX * XB;                                 // the transform's z buffer (three-buffer plans)
    static constexpr size_t RED = NX * XB;                                  // corner partial sums: in z when there is one (free at strip start)
    static constexpr size_t TOTAL = (NX + (PL::NBUF == 3 ? 1 : 0)) * XB + (PL::NBUF == 3 ? 0 : 32 * sizeof(float));
};

// The one pixel per row pair that has to wait for the next pair, (y, UW-1): its SE tap is L(y+2, 0).  Taps: row y-1
// (n0, n1 | ne), row y (m0, m1 | me), row y+1 (s0, s1 | se) at x = UW-2, UW-1 | the wrapped right neighbour.
// `of` = index of the pixel in the dense planes; OUT_U8: (row * UW + x) * 3 + c in the interleaved 8-bit image
template <bool HALF, bool OUT_U8 = false>
__device__ __forceinline__ void deferred_pixel(const FusedParams& p, long of, float n0, float n1, float ne, float m0, float m1, float me,
                                               float s0, float s1, float se)
{
    if constexpr (HALF) {
        // the same packed evaluation as the other pixels of the row, both lanes carrying this pixel
        auto sp = [](float v) { const _Float16 x = (_Float16)v; h2v r = {x, x}; return r; };   // (exact: binary16 values)
        const h2v a0 = sp(n0), a1 = sp(n1), a2 = sp(ne), b0 = sp(m0), b1 = sp(m1), b2 = sp(me), c0 = sp(s0), c1 = sp(s1), c2 = sp(se);
        const h2v mn0 = pk_min3(a0, b0, c0), mn1 = pk_min3(a1, b1, c1), mn2 = pk_min3(a2, b2, c2);
        const h2v mx0 = pk_max3(a0, b0, c0), mx1 = pk_max3(a1, b1, c1), mx2 = pk_max3(a2, b2, c2);
        const h2v o = sharpen_eval_pair_half(a1, c1, b0, b2, b1, pk_min3(mn1, b0, b2), pk_min3(mn0, mn1, mn2),
                                             pk_max3(mx1, b0, b2), pk_max3(mx0, mx1, mx2), h2_splat(-p.coef));
        if constexpr (OUT_U8) ((uint8_t*)p.out)[of] = cvt_f_u8((float)o.x, p.u8_wrap);
        else ((_Float16*)p.out)[of] = o.x;
    } else {
        const float tt[3][6] = {{n0, n0, n1, ne, ne, ne}, {m0, m0, m1, me, me, me}, {s0, s0, s1, se, se, se}};
        float o[4];
        sharpen_quad<false>(tt, p.coef, o);
        if constexpr (OUT_U8) ((uint8_t*)p.out)[of] = cvt_f_u8(o[1], p.u8_wrap);
        else ((float*)p.out)[of] = o[1];
    }
}

// (second argument: waves per SIMD the register allocation must allow -- 4 for 512 threads = 128 VGPRs, so that two strips
// can share a compute unit; tighter caps were tried: 80 VGPRs spill in fp32 and buy nothing in binary16)
// U = the (integer) upscale factor: the spectrum rows hold kx = 0..UW/2U, output row y is row y/U of spectrum buffer y%U
// (buffer 0 = the forward spectrum S1 itself, buffers 1..U-1 = the column kernel's residue transforms, all at U times the
// reference's normalisation), p.odd_delta elements apart.  U = 2 for all ahead-of-time plans.
// Half-integer factors (-u 1.5, 2.5: k_col_pad writes all rows of the zero-padded inverse into ONE buffer at the
// reference's normalisation): U = 1 and D = 2u, the spectrum rows hold kx = 0..UW/D.
// OUT_U8 (FFTUP_FLAG_FUSE_U8_STORE): the kernel stores the interleaved 8-bit RGB image itself -- the conversion
// of VkResample.cpp:1708-1748 on the sharpened values in registers, bit for bit what k_pack_u8 makes of the stored planes
// (cvt_f_u8) -- four byte stores per quad, one base register + immediate offsets; the float / half planes are never
// written (100 MB less HBM traffic per 4096x2048 frame than planes + k_pack_u8).  A workgroup still owns ONE plane (a strip
// that owns its rows in all three has 4 instead of 12 pairs behind each halo pair: measured slower, round 3), so every
// 64-byte line of the image takes bytes from three workgroups: the strip map below puts those three on one XCD.
// What the variant costs against the planes (68 vs 56 us at 4096x2048 -p 2, profiles/r04_d_*): 14 instead of 13 steps per
// strip, and four store instructions per thread and row instead of one -- lines touched per instruction do not matter
// (a wave-transposed form with 3-4 instead of 12 lines per instruction, ds_bpermute, ran 1 % slower: removed).
// Quarter-integer factors (-u 1.25, 1.75, 2.25, 3.75; round 5): the factor is D / (2 DD) with DD = 2 -- 5/4: D = 5 -- the spectrum
// rows hold kx = 0..UW DD / D, and the first radix R0 must make R0 DD / D whole (5, 10, 15 for 5/4; 7 for 7/4; 9 for 9/4).
template <class PL, bool HALF, int TK, int U = 2, int D = 2 * U, bool OUT_U8 = false, int DD = 1>
__global__ void __launch_bounds__(PL::T, PL::WPE) k_c2r_sharpen_g(FusedParams p)
{
    constexpr int UW = PL::UW, T = PL::T, R0 = PL::R0, NB0 = PL::NB0, NI = R0 * DD / D, KH = UW * DD / D;
    static_assert((R0 * DD) % D == 0, "the first radix must be a multiple of 2u");
    static_assert(UW % 4 == 0, "the sharpen passes work on quads of pixels");
    constexpr int EOUT = PL::EOUT, SOUT = PL::SOUT, VN = PL::VN;
    constexpr int NPASS = (UW + 4 * T - 1) / (4 * T);           // sharpen passes of 4 pixels per thread
    static_assert(KH == NB0 * NI, "the non-zero half spectrum must fill whole first-stage inputs");
    constexpr float inv = (1.0f / (float)U) / (float)UW;         // 1/U: the spectrum rows carry U times the reference's scale (k_col_t)
    using L = FusedGLds<PL>;
    using LT = typename std::conditional<HALF, _Float16, float>::type;      // L rows in LDS: binary16 for -p 2
    extern __shared__ __attribute__((aligned(128))) char smem[];
    float* red = (float*)(smem + L::RED);      // [0..15] corner partial sums, [16] corner DC term (strip start only)
    int lt = threadIdx.x;                       // (made opaque at the entry of the transform phase: addresses derived from it are
                                                // formed per step instead of being hoisted and kept in registers)
    const int uH = p.uH;
    const int pairs_per_plane = uH / 2;
    const long plane = (long)UW * uH;
    typename PL::Tw tws;
    PL::load_tw(tws, p.tw, lt);                 // once: inside the loop a load would queue behind the output stores

    int f0, f1;
    if constexpr (OUT_U8) {
        // A 64-byte line of the interleaved image takes bytes from all three planes' workgroups.  Workgroups go round-robin
        // over the 8 XCDs, each with its own L2: here strips never cross planes and the three strips of the same rows are
        // workgroups b, b + 8, b + 16 -- same XCD, dispatched together, in step -- so that their partial writes of a line meet
        // in ONE L2 and the line goes to HBM once (one plane per strip in linear order: every line written three times,
        // WRITE_SIZE 75.7 MB for the 25.2 MB image; fused_grid() on the host knows the same map).
        const int b = blockIdx.x, r = b >> 3;
        const int st = (r / 3) * 8 + (b & 7);
        const int j0u = st * p.pairs_per_strip;
        if (j0u >= pairs_per_plane) return;
        f0 = (r % 3) * pairs_per_plane + j0u;
        f1 = f0 + min(p.pairs_per_strip, pairs_per_plane - j0u);
    } else {
        f0 = blockIdx.x * p.pairs_per_strip;
        f1 = min(f0 + p.pairs_per_strip, 3 * pairs_per_plane);
    }
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
        // spectrum row `row` of the 2H-row buffer: even rows are rows of S1, odd rows live odd_delta elements further on.
        // An element's address = [plane base + row part] (wave-uniform: a scalar register pair) + [column part] (per thread,
        // the same for every row: computed once per segment and kept) -- the prefetch of a step costs no vector
        // instructions for addresses.
        auto koff = [&](int k) -> unsigned {
            // (k >= 0; 24-bit multiply: full rate, v_mul_lo_u32 is quarter rate; both factors are far below 2^24)
            return (__umul24((unsigned)k / TK, tile_stride32) + ((unsigned)k % TK)) * (unsigned)sizeof(float2);
        };
        typedef const __attribute__((address_space(1))) char* gptr_t;            // (global address space: global_load, not flat_load)
        auto rowbase = [&](int row) -> gptr_t {
            const unsigned off = (((unsigned)row / U) * TK + ((unsigned)row % U) * p.odd_delta) * (unsigned)sizeof(float2);
            gptr_t r = (gptr_t)base + __builtin_amdgcn_readfirstlane(off);
            asm("" : "+s"(r));                      // (a scalar pair as it stands: not re-associated into per-thread 64-bit sums)
            return r;
        };
        auto gload = [](gptr_t r, unsigned off) -> float2 {
            asm("" : "+v"(off));                    // (the 32-bit offset is re-defined here: base + zero-extended offset is then selected as global_load v, s[..])
            const lds_f2raw t = *(const __attribute__((address_space(1))) lds_f2raw*)(r + off);
            return make_float2(t.x, t.y);
        };
        auto S2at = [&](int k, int row) -> float2 { return gload(rowbase(row), koff(k)); };
        // Im of the DC element of row `row` -- one dword: a register of a wider load that is never read would be re-used by
        // the compiler while the load is still in flight (and waited for)
        auto dc_im = [&](int row) -> float { return *(const __attribute__((address_space(1))) float*)(rowbase(row) + 4); };
        const bool need_corner = !top && (y1 + 1 < uH);
        const int rs = y1 + 1;

        // inputs of the first-stage butterfly of thread lt for pair i: A (first row) and B (second row) at k = lt + NB0*m,
        // m < NI, and at the mirror partners KH - lt - NB0*m (vkFFT.h:2096-2106); thread 0 also needs Im of the DC column
        // of the two reference partners (leak).  Threads beyond the first stage (lt >= NB0) re-read valid elements.
        // (MIRROR_SHARE plans: am, bm are loaded by the two lanes without a partner only, FusedPlanPow2)
        constexpr bool MS = PL::MIRROR_SHARE;
        struct In { float2 a[NI], am[NI], b[NI], bm[NI]; float lka, lkb; };
        unsigned ko[NI], kom[NI];                   // column parts of the 2 * NI elements this thread prefetches per row
        const int jj = PL::first_index(lt);
        {
#pragma unroll
            for (int m = 0; m < NI; m++) { ko[m] = koff(jj + NB0 * m); kom[m] = koff(KH - jj - NB0 * m); }
        }
        const bool ms_self = MS && lt < 64 && (lt & 31) == 0;        // butterflies 0 and NB0/2
        auto load_pair = [&](int i) -> In {
            In in;
            const int a = a0 + 2 * i;
            const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);       // rows past the plane: duplicate of the last row
            const gptr_t ra = rowbase(ya), rb = rowbase(yb);
            if constexpr (MS) {
#pragma unroll
                for (int m = 0; m < NI; m++) {
                    in.a[m] = gload(ra, ko[m]); in.b[m] = gload(rb, ko[m]);
                    // (am, bm: whatever the registers hold -- the two lanes below load them, all others receive them in share();
                    // an initialisation would be eight vector instructions per step for nothing)
                    asm volatile("" : "=v"(in.am[m].x), "=v"(in.am[m].y), "=v"(in.bm[m].x), "=v"(in.bm[m].y));
                }
                if (ms_self) {                      // (two lanes of wave 0; loads only inside the branch: nothing waits)
#pragma unroll
                    for (int m = 0; m < NI; m++) { in.am[m] = gload(ra, kom[m]); in.bm[m] = gload(rb, kom[m]); }
                }
            } else {
#pragma unroll
                for (int m = 0; m < NI; m++) {
                    in.a[m] = gload(ra, ko[m]); in.am[m] = gload(ra, kom[m]);
                    in.b[m] = gload(rb, ko[m]); in.bm[m] = gload(rb, kom[m]);
                }
            }
            // (loaded by every lane, raw, so that no lane-dependent branch and no arithmetic -- hence no wait -- follows the loads)
            in.lka = dc_im(ya ^ 1);
            in.lkb = dc_im(yb ^ 1);
            return in;
        };
        // Loads and stores retire through ONE in-order counter (vmcnt).  settle() is called where the prefetch is a whole
        // transform old and the previous step's stores even older, and re-defining the values keeps the compiler from copying
        // the just-issued loads' registers (and waiting for them) right behind the loads.  (Round 3 tried the wait at the end
        // of the step instead -- a constant number of stores per wave and step, rows that are not output rows stored through a
        // buffer resource with an out-of-range offset, which the hardware drops, then vmcnt(that number): correct, and not a
        // microsecond faster; the prefetch is not what the kernel waits for.  DESIGN.md section 4.)
        auto settle = [](In& in) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0), nothing else
#pragma unroll
            for (int m = 0; m < NI; m++)
                asm volatile("" : "+v"(in.a[m].x), "+v"(in.a[m].y), "+v"(in.am[m].x), "+v"(in.am[m].y), "+v"(in.b[m].x), "+v"(in.b[m].y),
                                  "+v"(in.bm[m].x), "+v"(in.bm[m].y));
            asm volatile("" : "+v"(in.lka), "+v"(in.lkb));
        };

        // MIRROR_SHARE: the mirror elements of butterfly jj are the own elements of butterfly NB0 - jj, in lane l ^ 32.  Fetched
        // through the LDS crossbar right behind settle(), they arrive while the L rows are written and the rows sharpened.
        auto share = [&](In& in) __attribute__((always_inline)) {
            if constexpr (MS) {
                const int pl = (((lt & 63) ^ 32) << 2);
                auto part = [&](float2 z) -> float2 {
                    return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pl, __builtin_bit_cast(int, z.x))),
                                       __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pl, __builtin_bit_cast(int, z.y))));
                };
                if (!ms_self) {                      // (the permutes run under the execution mask: no selects)
#pragma unroll
                    for (int m = 0; m < NI; m++) { in.am[m] = part(in.a[NI - 1 - m]); in.bm[m] = part(in.b[NI - 1 - m]); }
                }
            }
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
        In in = load_pair(0);
        settle(in);
        share(in);
        __syncthreads();            // red[] published; the previous segment's last reads of X[] are over
        // the corner sample L(y1+1, 0) (SE tap of the strip's last pixel), kept in a register by every thread
        float corner = 0.f;
        if (need_corner) {
            float sum = 0.f;
            for (int w2 = 0; w2 < (T + 63) / 64; w2++) sum += red[w2];
            corner = (red[16] + 2.0f * sum) * inv;
        }
        if constexpr (PL::NBUF == 3) __syncthreads();       // red[] lives in z: all reads before the first exchange writes it
        float pn0 = 0.f, pn1 = 0.f;                         // thread T-1: taps (a-3 | a-1 clamped, UW-2 / UW-1) of the deferred pixel
        constexpr bool RR = L::RR;
        using SavedRow = typename std::conditional<HALF, H2Row, TapRow>::type;
        SavedRow sv[RR ? NPASS : 1][2];                     // RR: tap rows a, a+1 of the previous step = ring rows a-2, a-1 of this one
        float lprev0 = 0.f;                                 // RR, thread T-1: L(a-2, 0)
        if constexpr (RR) {
#pragma unroll
            for (int h = 0; h < NPASS; h++) sv[h][0] = sv[h][1] = SavedRow{};
        }

        for (int s = 0; s < npairs; s++) {
            const int a = a0 + 2 * s;
            float2* buf = (float2*)(smem + (RR ? 0 : (s & 1)) * L::XB);
            LT* cur = (LT*)buf;                                                     // rows a, a+1 after the transform
            const LT* ring = (const LT*)(smem + ((s + 1) & 1) * L::XB);             // rows a-2, a-1 (not RR)
            // ================= transform of pair s
            asm volatile("" : "+v"(lt));
            float2 v[VN];
#pragma unroll
            for (int m = 0; m < VN; m++) v[m] = make_float2(0.f, 0.f);
#pragma unroll
            for (int m = 0; m < NI; m++) {
                v[m] = cadd_i(in.a[m], in.b[m]);                       // A + i B          (a.x - b.y, a.y + b.x)
                v[R0 - NI + m] = cadd_conj_i(in.am[m], in.bm[m]);      // conj(A) + i conj(B)   (am.x + bm.y, -am.y + bm.x)
            }
            if (lt == 0) {
                v[NI] = make_float2(in.am[0].x - in.bm[0].y, in.am[0].y + in.bm[0].x);       // k = KH = W/2
                // DC terms incl. the pair leak: row y gets -Im D[y+1] (y even) / +Im D[y-1] (y odd)
                const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
                v[0] = make_float2(in.a[0].x + ((ya & 1) ? in.lka : -in.lka), in.b[0].x + ((yb & 1) ? in.lkb : -in.lkb));
            }
            in = load_pair(min(s + 1, npairs - 1));                                 // lands during this step (last step: a harmless re-read)
            PL::fft(v, buf, (float2*)(smem + L::ZOFF), lt, tws, jj);
            settle(in);
            share(in);
            if constexpr (HALF) {
                // C2R output stored as binary16 (vkFFT.h:7289-7290), then |u^2 g| clamped, each step rounded like the shader's
                const h2v up2 = h2_splat(p.upsq), one2 = h2_splat(1.0f);
                if (SOUT == T || lt < SOUT) {
#pragma unroll
                    for (int i = 0; i < EOUT; i++) {
                        const f2v sv = mk2(v[i].x, v[i].y) * mk2(inv, inv);
                        const h2v g = {(_Float16)sv.x, (_Float16)sv.y};
                        const h2v Lv = __builtin_elementwise_min(h2_bits(bits_h2(up2 * g) & 0x7fff7fffu), one2);
                        cur[lt + SOUT * i] = Lv.x;
                        cur[UW + lt + SOUT * i] = Lv.y;
                    }
                }
            } else {
                // one packed multiply: (v / UW) * u^2 == v * (u^2 / UW), bit for bit when UW is a power of two
                const float ks = inv * p.upsq;
                if (SOUT == T || lt < SOUT) {
#pragma unroll
                    for (int i = 0; i < EOUT; i++) {
                        const f2v sv = mk2(v[i].x, v[i].y) * mk2(ks, ks);
                        cur[lt + SOUT * i] = absmin1(sv.x);
                        cur[UW + lt + SOUT * i] = absmin1(sv.y);
                    }
                }
            }
            __syncthreads();                                                        // L rows a, a+1 visible
            // ================= sharpen rows a-1 and a
            // L(a, 0) and L(a + 1, 0): the wrapped right neighbours of the row ends (quirk B5) -- read by EVERY lane (two broadcast
            // reads issued with the first taps) and put in place with selects.  (They used to be read by the one lane that
            // needs them, inside lane-dependent branches, each read a round trip through an LDS pipe busy with the other waves'
            // taps: that wave finished its sharpen 1700 cycles after the others -- in-kernel s_memtime marks -- and the whole
            // workgroup waited for it at the next barrier, every step.)
            const float la0 = (float)cur[0], la1 = (float)cur[UW];
            constexpr int LT_LAST = (UW / 4 - 1) % T, H_LAST = (UW / 4 - 1) / T;
            // The pixel deferred by the previous step, (a-2, UW-1): by the thread that owns the last four pixels of a row, from
            // registers only (it needs the ring rows as the previous step left them: called inside pass H_LAST, behind the
            // issue of that pass's tap reads and before it replaces the saved rows).
            auto finish_deferred = [&]() __attribute__((always_inline)) {
                if constexpr (RR) {
                    if (lt == LT_LAST) {
                        const SavedRow& R2 = sv[H_LAST][0];         // row a-2, pixels UW-4 .. UW-1 and L(a-1, 0)
                        const SavedRow& R1 = sv[H_LAST][1];         // row a-1
                        float r2a, r2b, r1a, r1b, r10;
                        if constexpr (HALF) { r2a = (float)R2.h23.x; r2b = (float)R2.h23.y; r1a = (float)R1.h23.x; r1b = (float)R1.h23.y; r10 = (float)R2.sc.y; }
                        else { r2a = R2.q.z; r2b = R2.q.w; r1a = R1.q.z; r1b = R1.q.w; r10 = R2.r; }
                        if (s > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1)
                            deferred_pixel<HALF, OUT_U8>(p, OUT_U8 ? ((long)(a - 2) * UW + (UW - 1)) * 3 + c : c * plane + (long)(a - 2) * UW + (UW - 1),
                                                         pn0, pn1, (a - 2 == 0) ? r10 : lprev0, r2a, r2b, r10, r1a, r1b, la0);
                        if (a == 0) { pn0 = (float)cur[UW - 2]; pn1 = (float)cur[UW - 1]; }       // (top strip, first step only)
                        else { pn0 = r1a; pn1 = r1b; }
                        lprev0 = la0;
                    }
                }
            };
            // the SE tap of pixel (a, UW-1) is L(a+2, 0): past the plane it clamps to row uH-1; in the last step it is the
            // corner sample; otherwise the pixel is finished next step (placeholder now)
            float lse = la1;
            {
                const int r2 = min(a + 2, uH - 1) - a;
                if (r2 == 0) lse = la0;
                else if (r2 > 1 && s == npairs - 1) lse = to_L<HALF>(corner, p.upsq);
            }
            auto rowp = [&](int r) -> const LT* { return r < 0 ? ring + (r + 2) * UW : cur + r * UW; };      // (r < 0: not RR)
            const bool out0 = (a - 1) >= y0 && (a - 1) < y1;                        // row y = a-1
            const bool out1 = a >= y0 && a < y1;                                    // row y = a
            if constexpr (HALF) {
                if (out0 || out1 || RR) {
                    const h2v ncoef = h2_splat(-p.coef);
                    auto pass = [&](auto hc) __attribute__((always_inline)) {
                        const int h = hc;                                  // (a constant when called with an integral_constant)
                        const int x0 = 4 * (lt + T * h);
                        if (UW % (4 * T) != 0 && x0 >= UW) return;        // (wave-uniform: UW/4 is a multiple of 64)
                        // row -1 clamps to row 0: for a == 0 the "a-1" slot aliases row a
                        const LT* rows[4] = {RR ? nullptr : rowp(-2), RR ? nullptr : ((a == 0) ? rowp(0) : rowp(-1)), rowp(0), rowp(1)};
                        // own four pixels (8 bytes) and the quads of both neighbours of every row that comes from LDS: all reads
                        // of the pass in flight together (one wait, not one per row); the shifted pairs by v_alignbit
                        uint2 q[4], ql[4], qr[4];
#pragma unroll
                        for (int r = 3; r >= (RR ? 2 : 0); r--) {
                            const LT* rp = rows[r] + x0;
                            q[r] = *(const uint2*)rp; ql[r] = *(const uint2*)(rp - 4); qr[r] = *(const uint2*)(rp + 4);
                        }
                        if constexpr (RR) { if (h == H_LAST) finish_deferred(); }
                        H2Row R[4];
#pragma unroll
                        for (int r = 3; r >= 0; r--) {
                            if constexpr (RR) {
                                if (r < 2) {
                                    R[r] = sv[h][r];
                                    if (r == 1 && a == 0) {     // (a real, wave-uniform branch: top strip, first step only)
                                        asm volatile("");
                                        R[1].sa = R[2].sa; R[1].h01 = R[2].h01; R[1].sb = R[2].sb; R[1].h23 = R[2].h23; R[1].sc = R[2].sc;
                                    }
                                    continue;
                                }
                            }
                            if (r == 0 && !out0) {
                                R[0].sa = R[0].h01 = R[0].sb = R[0].h23 = R[0].sc = h2_splat(0.f);
                                continue;
                            }
                            asm volatile("" : "+v"(q[r].x), "+v"(q[r].y), "+v"(ql[r].x), "+v"(ql[r].y), "+v"(qr[r].x), "+v"(qr[r].y));   // keep them 8-byte reads
                            R[r].h01 = h2_bits(q[r].x);
                            R[r].h23 = h2_bits(q[r].y);
                            R[r].sa = h2_bits(__builtin_amdgcn_alignbit(q[r].x, ql[r].y, 16));        // (x0-1, x0)
                            R[r].sb = h2_bits(__builtin_amdgcn_alignbit(q[r].y, q[r].x, 16));         // (x0+1, x0+2)
                            R[r].sc = h2_bits(__builtin_amdgcn_alignbit(qr[r].x, q[r].y, 16));        // (x0+3, x0+4)
                            if (x0 == 0) R[r].sa = h2_bits((bits_h2(R[r].h01) & 0xffffu) * 0x10001u);      // id_x_m clamp (VkResample.cpp:889)
                        }
                        if (x0 + 4 == UW) {
                            auto set_hi = [](h2v& d, float v) { d.y = (_Float16)v; };      // (exact: binary16 values)
                            if (a != 0) set_hi(R[1].sc, la0);       // row a-1 wraps into row a
                            set_hi(R[3].sc, lse);
                        }
                        if constexpr (RR) { sv[h][0] = R[2]; sv[h][1] = R[3]; }
#pragma unroll
                        for (int w = 0; w < 2; w++) {
                            if (w == 0 ? !out0 : !out1) continue;
                            h2v o01, o23;
                            sharpen_quad_half(R[w], R[w + 1], R[w + 2], ncoef, o01, o23);
                            if constexpr (OUT_U8) {
                                uint8_t b8[4];
                                cvt4_h_u8(o01, o23, p.u8_wrap, b8);
                                uint8_t* d8 = (uint8_t*)p.out + ((long)(a - 1 + w) * UW * 3 + c) + (unsigned)x0 * 3u;    // (uniform base + x0 * 3)
                                d8[0] = b8[0]; d8[3] = b8[1]; d8[6] = b8[2]; d8[9] = b8[3];
                                continue;
                            }
                            const long row_of = c * plane + (long)(a - 1 + w) * UW;      // wave-uniform
                            f2v val = {__builtin_bit_cast(float, o01), __builtin_bit_cast(float, o23)};
                            __builtin_nontemporal_store(val, (f2v*)((char*)((__half*)p.out + row_of) + (unsigned)x0 * 2u));
                        }
                    };
                    if constexpr (RR) {                                    // saved rows are registers: the pass index must be static
                        pass(std::integral_constant<int, 0>{});
                        if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
                        if constexpr (NPASS > 2) pass(std::integral_constant<int, 2>{});
                        if constexpr (NPASS > 3) pass(std::integral_constant<int, 3>{});
                    } else {
#pragma unroll 1
                        for (int h = 0; h < NPASS; h++) pass(h);
                    }
                }
            } else {
            if (out0 || out1 || RR) {
                auto pass = [&](auto hc) __attribute__((always_inline)) {
                    const int h = hc;                                      // (a constant when called with an integral_constant)
                    const int x0 = 4 * (lt + T * h);
                    if (UW % (4 * T) != 0 && x0 >= UW) return;            // (wave-uniform: UW/4 is a multiple of 64)
                    // row -1 clamps to row 0: for a == 0 the "a-1" slot aliases row a
                    const float* rows[4] = {RR ? nullptr : rowp(-2), RR ? nullptr : ((a == 0) ? rowp(0) : rowp(-1)), rowp(0), rowp(1)};
                    // own four pixels and the quads of both neighbours of every row that comes from LDS: conflict-free 16-byte
                    // reads, no cross-lane moves, no wave-edge branches -- all reads of the pass in flight together (one wait,
                    // not one per row).  The two rows of a pair are contiguous in LDS, so x = UW of the first one IS x = 0 of the
                    // second (quirk B5); x0 == 0 reads 16 bytes in front of the row (at worst out of range: LDS returns 0) and is
                    // replaced below.  (Whole tuples are pinned: left alone the compiler narrows the neighbour loads to the
                    // one dword that is used, and single-dword reads 16 bytes apart are a 4-way bank conflict.)
                    f4t q[4], ql[4], qr[4];
#pragma unroll
                    for (int r = 3; r >= (RR ? 2 : 0); r--) {
                        const float* rp = rows[r] + x0;
                        q[r] = *(const f4t*)rp; ql[r] = *(const f4t*)(rp - 4); qr[r] = *(const f4t*)(rp + 4);
                    }
                    if constexpr (RR) { if (h == H_LAST) finish_deferred(); }
                    TapRow t[4];
#pragma unroll
                    for (int r = 3; r >= 0; r--) {
                        if constexpr (RR) {
                            if (r < 2) {
                                t[r] = sv[h][r];
                                if (r == 1 && a == 0) {         // (a real, wave-uniform branch: top strip, first step only)
                                    asm volatile("");
                                    t[1].q = t[2].q; t[1].l = t[2].l; t[1].r = t[2].r;
                                }
                                continue;
                            }
                        }
                        if (r == 0 && !out0) {
                            t[0].q = (f4t)(0.f);
                            t[0].l = t[0].r = 0.f;
                            continue;
                        }
                        asm volatile("" : "+v"(q[r]), "+v"(ql[r]), "+v"(qr[r]));
                        t[r].q = q[r];
                        t[r].l = ql[r].w;
                        t[r].r = qr[r].x;
                        if (x0 == 0) t[r].l = t[r].q.x;            // id_x_m clamp (VkResample.cpp:889)
                    }
                    if (x0 + 4 == UW) {
                        if (a != 0) t[1].r = la0;                  // row a-1 wraps into row a
                        t[3].r = lse;
                    }
                    if constexpr (RR) { sv[h][0] = t[2]; sv[h][1] = t[3]; }
#pragma unroll
                    for (int w = 0; w < 2; w++) {
                        if (w == 0 ? !out0 : !out1) continue;
                        float vmn[6], vmx[6];
                        sharpen_vminmax(t, w, vmn, vmx);
                        const f4t o = sharpen_quad_packed(t, w, vmn, vmx, p.coef);
                        if constexpr (OUT_U8) {
                            uint8_t b8[4];
                            cvt4_f_u8(o.x, o.y, o.z, o.w, p.u8_wrap, b8);
                            uint8_t* d8 = (uint8_t*)p.out + ((long)(a - 1 + w) * UW * 3 + c) + (unsigned)x0 * 3u;        // (uniform base + x0 * 3)
                            d8[0] = b8[0]; d8[3] = b8[1]; d8[6] = b8[2]; d8[9] = b8[3];
                            continue;
                        }
                        const long row_of = c * plane + (long)(a - 1 + w) * UW;      // wave-uniform
                        __builtin_nontemporal_store(o, (f4t*)((char*)((float*)p.out + row_of) + (unsigned)x0 * 4u));
                    }
                };
                if constexpr (RR) {
                    pass(std::integral_constant<int, 0>{});
                    if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
                    if constexpr (NPASS > 2) pass(std::integral_constant<int, 2>{});
                    if constexpr (NPASS > 3) pass(std::integral_constant<int, 3>{});
                } else {
#pragma unroll 1
                    for (int h = 0; h < NPASS; h++) pass(h);
                }
            }
            }
            if constexpr (!RR) {
                if (lt == T - 1) {
                    // finish the pixel deferred by the previous pair: (a-2, UW-1); L(a,0) is known now
                    const LT* r2 = rowp(-2);
                    const LT* r1 = rowp(-1);
                    const float r10 = (float)r1[0], r00 = (float)rowp(0)[0];
                    if (s > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1)
                        deferred_pixel<HALF, OUT_U8>(p, OUT_U8 ? ((long)(a - 2) * UW + (UW - 1)) * 3 + c : c * plane + (long)(a - 2) * UW + (UW - 1), pn0, pn1, (a - 2 == 0) ? r10 : (float)r2[0],
                                             (float)r2[UW - 2], (float)r2[UW - 1], r10, (float)r1[UW - 2], (float)r1[UW - 1], r00);
                    const LT* rn = (a == 0) ? rowp(0) : rowp(-1);
                    pn0 = (float)rn[UW - 2];
                    pn1 = (float)rn[UW - 1];
                }
            }
            // NBUF = 2 (in-place exchanges): the next transform's first scatter goes into the buffer the sharpen just read.
            // NBUF = 3: that buffer is first written behind the next step's first exchange barrier (FusedPlanPow2).
            if constexpr (PL::NBUF == 2) __syncthreads();
        }
    }
}

}  // namespace fftup
