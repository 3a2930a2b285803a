// fft_engine.hpp -- LDS-staged Stockham FFT building blocks for gfx950 (wave64).
//
// Radix-2/3/4/5/7/8 butterflies and one Stockham autosort stage that works on TK interleaved
// sequences held in LDS as float2 [n][TK].  Sign convention follows the reference: DIR=+1 is
// exp(+2 pi i nk/N) (VkFFT "forward", vkFFT.h:4545/751), DIR=-1 the inverse kernel; the 1/N
// of the inverse (vkFFT.h:2921-2923) is applied by the caller when it stores the result.
// Twiddles come from a per-length table tw[k] = exp(+2 pi i k/N) computed on the host in double
// and rounded once to fp32 (the reference evaluates fp32 cos/sin in-shader, vkFFT.h:2417-2421).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fftup {

struct StagePlan {           // radix sequence of one 1-D transform (product = N)
    int32_t n;
    int32_t nstages;
    uint8_t radix[16];
};

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by DIR*i
template <int DIR> __device__ __forceinline__ float2 mul_i(float2 a)
{
    return DIR > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <int DIR> __device__ __forceinline__ float2 twid(float2 w)   // table holds exp(+i..)
{
    return DIR > 0 ? w : make_float2(w.x, -w.y);
}

// LDS index padding: one float2 of padding per 16 (keeps stride-R*TK scatter writes of the
// first Stockham stages off a single pair of banks; ds_write_b64 banks = (addr/4) % 32).
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int lpad_size(int n) { return n + (n >> 4) + 1; }

// ---------------------------------------------------------------- butterflies (in place on v[])
template <int DIR> __device__ __forceinline__ void bfly2(float2* v)
{
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <int DIR> __device__ __forceinline__ void bfly4(float2* v)
{
    float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    float2 t2 = cadd(v[1], v[3]), t3 = mul_i<DIR>(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = cadd(t1, t3);
    v[3] = csub(t1, t3);
}
template <int DIR> __device__ __forceinline__ void bfly8(float2* v)
{
    const float h = 0.70710678118654752440f;
    float2 e[4] = {v[0], v[2], v[4], v[6]};
    float2 o[4] = {v[1], v[3], v[5], v[7]};
    bfly4<DIR>(e);
    bfly4<DIR>(o);
    // w^q, w = exp(DIR*2 pi i/8)
    float2 o1 = cscale(cadd(o[1], mul_i<DIR>(o[1])), h);            // (1 + DIR*i)/sqrt2
    float2 o2 = mul_i<DIR>(o[2]);
    float2 o3 = cscale(csub(mul_i<DIR>(o[3]), o[3]), h);            // (-1 + DIR*i)/sqrt2
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
    v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
}
template <int DIR> __device__ __forceinline__ void bfly3(float2* v)
{
    const float s3 = 0.86602540378443864676f;
    float2 t1 = cadd(v[1], v[2]);
    float2 t2 = make_float2(fmaf(-0.5f, t1.x, v[0].x), fmaf(-0.5f, t1.y, v[0].y));
    float2 t3 = cscale(mul_i<DIR>(csub(v[1], v[2])), s3);
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, t3);
    v[2] = csub(t2, t3);
}
template <int DIR> __device__ __forceinline__ void bfly5(float2* v)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    float2 a = v[0];
    float2 p1 = make_float2(a.x + c1 * t1.x + c2 * t2.x, a.y + c1 * t1.y + c2 * t2.y);
    float2 p2 = make_float2(a.x + c2 * t1.x + c1 * t2.x, a.y + c2 * t1.y + c1 * t2.y);
    float2 q1 = mul_i<DIR>(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    float2 q2 = mul_i<DIR>(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    v[0] = cadd(a, cadd(t1, t2));
    v[1] = cadd(p1, q1);
    v[4] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[3] = csub(p2, q2);
}
template <int DIR> __device__ __forceinline__ void bfly7(float2* v)
{
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    float2 t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
    float2 u1 = csub(v[1], v[6]), u2 = csub(v[2], v[5]), u3 = csub(v[3], v[4]);
    float2 a = v[0];
    float2 p1 = make_float2(a.x + c1 * t1.x + c2 * t2.x + c3 * t3.x, a.y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
    float2 p2 = make_float2(a.x + c2 * t1.x + c3 * t2.x + c1 * t3.x, a.y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
    float2 p3 = make_float2(a.x + c3 * t1.x + c1 * t2.x + c2 * t3.x, a.y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
    float2 q1 = mul_i<DIR>(make_float2(s1 * u1.x + s2 * u2.x + s3 * u3.x, s1 * u1.y + s2 * u2.y + s3 * u3.y));
    float2 q2 = mul_i<DIR>(make_float2(s2 * u1.x - s3 * u2.x - s1 * u3.x, s2 * u1.y - s3 * u2.y - s1 * u3.y));
    float2 q3 = mul_i<DIR>(make_float2(s3 * u1.x - s1 * u2.x + s2 * u3.x, s3 * u1.y - s1 * u2.y + s2 * u3.y));
    v[0] = cadd(a, cadd(t1, cadd(t2, t3)));
    v[1] = cadd(p1, q1); v[6] = csub(p1, q1);
    v[2] = cadd(p2, q2); v[5] = csub(p2, q2);
    v[3] = cadd(p3, q3); v[4] = csub(p3, q3);
}
template <int R, int DIR> __device__ __forceinline__ void bfly(float2* v)
{
    if constexpr (R == 2) bfly2<DIR>(v);
    else if constexpr (R == 3) bfly3<DIR>(v);
    else if constexpr (R == 4) bfly4<DIR>(v);
    else if constexpr (R == 5) bfly5<DIR>(v);
    else if constexpr (R == 7) bfly7<DIR>(v);
    else if constexpr (R == 8) bfly8<DIR>(v);
}

// twiddle the R inputs of one butterfly: v[m] *= exp(DIR * 2 pi i * m * tidx / N), tidx = k*tstep.
// One table fetch for m=1; powers 2 and 4 are fetched too (cheap, L1/L2 resident), the rest are
// products -- two roundings instead of one, ~1e-7 relative.
template <int R, int DIR>
__device__ __forceinline__ void apply_twiddles(float2* v, const float2* __restrict__ tw, int tidx)
{
    if constexpr (R == 2) {
        v[1] = cmul(v[1], twid<DIR>(tw[tidx]));
    } else if constexpr (R == 3) {
        float2 w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2);
    } else if constexpr (R == 4) {
        float2 w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]);
        float2 w3 = cmul(w1, w2);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
    } else if constexpr (R == 5) {
        float2 w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        float2 w3 = cmul(w1, w2);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
    } else if constexpr (R == 7) {
        float2 w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        float2 w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
        v[4] = cmul(v[4], w4); v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6);
    } else if constexpr (R == 8) {
        float2 w1 = twid<DIR>(tw[tidx]), w2 = twid<DIR>(tw[2 * tidx]), w4 = twid<DIR>(tw[4 * tidx]);
        float2 w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
    }
}

// ---------------------------------------------------------------- one generic Stockham stage
// in/out: LDS, float2 [n][TK] with lpad() applied to the flattened element index.
// Butterfly j (0 <= j < N/R) of sequence col reads in[j + m*N/R], writes
// out[(j - k)*R + k + m*Ns], k = j % Ns (Stockham autosort, decimation in time).
template <int R, int DIR, int TK>
__device__ __forceinline__ void stage_lds(const float2* __restrict__ in, float2* __restrict__ out,
                                          int N, int Ns, const float2* __restrict__ tw, int tid, int T)
{
    const int nb = N / R;
    const int tstep = nb / Ns;                 // N / (Ns*R)
    const bool ns_pow2 = (Ns & (Ns - 1)) == 0;
    for (int g = tid; g < nb * TK; g += T) {
        const int col = g % TK;                // TK is a compile-time power of two
        const int j = g / TK;
        const int k = ns_pow2 ? (j & (Ns - 1)) : (j % Ns);
        float2 v[R];
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
template <int DIR, int TK>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const StagePlan& P,
                                           const float2* __restrict__ tw, int tid, int T)
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
        float2* t = a; a = b; b = t;
    }
    return a;
}

}  // namespace fftup
