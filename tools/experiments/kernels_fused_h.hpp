// NOT PART OF THE BUILD -- kept for the next round.  Last state: 128-VGPR variant, two
// workgroups per CU, no register prefetch of the spectrum rows.  Parity-green when hooked into launch_fused_t.
// kernels_fused_h.hpp -- fused C2R + sharpen kernel that fills HALF a compute unit.
//
// k_c2r_sharpen_t fills a CU with one workgroup (16 waves x 128 VGPRs, 137 KB LDS), so nothing of the other stream
// can run beside it.  This variant is sized for two workgroups per CU -- or one next to a column-kernel workgroup:
//   * 2*T threads (T = UW/16): a permanent TRANSFORM group (T threads x 16 points, radix-16 stages, two exchanges,
//     synchronised among its own waves by GroupBarrier) and a permanent SHARPEN group (T threads);
//   * the transform group parks the two new L rows in a hand-off buffer; the sharpen group keeps the two older
//     rows of its sixteen pixel columns (with their horizontal neighbours) in registers;
//   * the spectrum rows of a pair are staged through the exchange buffer itself before the first stage.
// LDS: exchange buffer (34.8 KB) + hand-off rows (32.8 KB) = 67.6 KB for UW = 4096.  One workgroup barrier per step.
// Semantics (pairing, DC leak, quirk B5, corner sample, clamps) are those of k_c2r_sharpen_t; see kernels_pow2.hpp.
#pragma once
#include "../../vkresample_amd/csrc/kernels_pow2.hpp"

namespace fftup {

// ---- pieces that lived in kernels_pow2.hpp while this experiment was built (barrier policy of the register FFT)
struct GroupBarrier {
    unsigned* cnt;            // LDS word, zeroed before first use
    unsigned target;          // arrivals expected so far
    unsigned waves;           // waves in the group
    __device__ __forceinline__ void operator()()
    {
        target += waves;
        asm volatile("" ::: "memory");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (bounded: a protocol error must show up as a wrong image in the tests, never as a hung GPU)
        for (unsigned spin = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && spin < (1u << 20); spin++)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    }
};

template <int N, int E, int DIR, int TK, bool FINAL_TO_LDS, int S = 0, int RMAX = 8, typename Bar>
__device__ __forceinline__ void reg_fft_b(float2 (&v)[E], float2* __restrict__ buf, int p, int col,
                                          const TwSet<N, E, RMAX>& tws, Bar& bar)
{
    constexpr int Ns = stage_ns(N, S, RMAX);
    constexpr int R = stage_radix(N, Ns, RMAX);
    static_assert(E % R == 0, "radix must divide the per-thread point count");
    reg_butterflies<N, E, R, Ns, DIR>(v, tws.w[S > 0 ? S - 1 : 0]);
    constexpr bool last = (Ns * R == N);
    if constexpr (!last || FINAL_TO_LDS) {
        reg_scatter<N, E, R, Ns, TK>(v, buf, p, col);
        bar();
    }
    if constexpr (!last) {
        reg_gather<N, E, TK>(v, buf, p, col);
        bar();
        reg_fft_b<N, E, DIR, TK, FINAL_TO_LDS, S + 1, RMAX>(v, buf, p, col, tws, bar);
    }
}

template <int UW> struct FusedHLds {
    static constexpr size_t XB = fused_buf_bytes(UW);
    static constexpr size_t LB = 2 * (size_t)UW * sizeof(float);
    static constexpr size_t RED = XB + LB;
    static constexpr size_t TOTAL = RED + 64 * sizeof(float);
};

// one segment = the part of a strip that lies in one colour plane
struct FusedHSeg {
    int c, y0, y1, a0, npairs, rs;
    bool need_corner;
};

template <int UW, int TK> struct FusedHAddr {
    const float2* base;
    unsigned tile_stride32;
    __device__ __forceinline__ float2 at(int k, int row) const
    {
        // byte offset kept in 32 bits so that the load takes the "SGPR base + VGPR offset" form
        const unsigned off = ((unsigned)(k / TK) * tile_stride32 + (unsigned)row * TK + (unsigned)(k % TK)) * (unsigned)sizeof(float2);
        return *(const float2*)((const char*)base + off);
    }
};

// ---- transform group: one call per segment.  Separate, non-inlined functions give each group its own register
// allocation (inlined, the kernel needs 172 VGPRs; the groups need 124 and 130).
template <int UW, bool HALF, int TK>
__device__ __noinline__ void fused_h_transform(const FusedParams& p, char* smem, const FusedHSeg& g, int lt, GroupBarrier& gbar, unsigned& consumed)
{
    constexpr int E = 16, T = UW / E, KH = UW / 4, NW = T / 64;
    constexpr float inv = 1.0f / (float)UW;
    using L = FusedHLds<UW>;
    float2* X = (float2*)smem;
    float* Lb = (float*)(smem + L::XB);
    unsigned* cnt = (unsigned*)((float*)(smem + L::RED) + 32);
    const int uH = p.uH, a0 = g.a0, npairs = g.npairs;
    const FusedHAddr<UW, TK> S2{p.S2 + (long)g.c * p.NT * ((long)uH * TK), (unsigned)uH * TK};
    auto S2at = [&](int k, int row) -> float2 { return S2.at(k, row); };
    struct Stage { float2 a[4], b[4], x0, x1; };
    auto stage_issue = [&](int i) -> Stage {
        Stage st;
        const int a = a0 + 2 * i;
        const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);       // rows past the plane: duplicate of the last row
#pragma unroll
        for (int q = 0; q < 4; q++) { st.a[q] = S2at(lt + T * q, ya); st.b[q] = S2at(lt + T * q, yb); }
        const int kx = (lt == 0) ? KH : 0;                            // lane 0: k = W/2; lane 1: DC of the reference partners
        st.x0 = S2at(kx, (lt == 1) ? (ya ^ 1) : ya);
        st.x1 = S2at(kx, (lt == 1) ? (yb ^ 1) : yb);
        return st;
    };
    auto stage_commit = [&](int i, const Stage& st) {
        float2* SA = X;
        float2* SBp = SA + (KH + 1);
#pragma unroll
        for (int q = 0; q < 4; q++) { SA[lt + T * q] = st.a[q]; SBp[lt + T * q] = st.b[q]; }
        if (lt == 0) { SA[KH] = st.x0; SBp[KH] = st.x1; }
        if (lt == 1) {
            const int a = a0 + 2 * i;
            const int ya = min(a, uH - 1), yb = min(a + 1, uH - 1);
            SBp[KH + 1] = make_float2((ya & 1) ? st.x0.y : -st.x0.y, (yb & 1) ? st.x1.y : -st.x1.y);   // leak terms
        }
    };
    __syncthreads();                                  // (the sharpen group publishes its corner sums)
    for (int s = 0; s <= npairs; s++) {
        // ================= transform group: pair s
        consumed += (unsigned)NW;                     // the sharpen group signals once per step
        if (s < npairs) {
            // (no register prefetch across the transform: with two workgroups on a CU the other one covers this
            // wait, and the transform keeps its registers for the radix-16 butterflies)
            {
                const Stage st = stage_issue(s);
                stage_commit(s, st);
            }
            TwSet<UW, E, 16> tws;
            tws.load(p.tw, lt);
            gbar();
            float2 v[E];
            {
                const float2* SA = X;
                const float2* SBp = SA + (KH + 1);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float2 A = SA[lt + T * q], B = SBp[lt + T * q];
                    v[q] = make_float2(A.x - B.y, A.y + B.x);
                }
#pragma unroll
                for (int q = 4; q < 12; q++) v[q] = make_float2(0.f, 0.f);
#pragma unroll
                for (int q = 12; q < 16; q++) {
                    const int kk = (16 - q) * T - lt;
                    const float2 A = SA[kk], B = SBp[kk];
                    v[q] = make_float2(A.x + B.y, -A.y + B.x);
                }
                if (lt == 0) {
                    const float2 A = SA[KH], B = SBp[KH];
                    v[4] = make_float2(A.x - B.y, A.y + B.x);                          // k = W/2
                    const float2 lk = SBp[KH + 1];
                    v[0] = make_float2(SA[0].x + lk.x, SBp[0].x + lk.y);                 // DC terms incl. the pair leak
                }
            }
            gbar();                                   // staging consumed: the exchange buffer is free
            reg_fft_b<UW, E, -1, 1, false, 0, 16>(v, X, lt, 0, tws, gbar);
            // the previous pair's rows must have been picked up before they are overwritten
            for (unsigned spin = 0; __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < consumed && spin < (1u << 20); spin++)
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < E; i++) {
                Lb[lt + T * i] = to_L<HALF>(v[i].x * inv, p.upsq);
                Lb[UW + lt + T * i] = to_L<HALF>(v[i].y * inv, p.upsq);
            }
        }
        __syncthreads();
    }
}

// ---- sharpen group: one call per segment
template <int UW, bool HALF, int TK>
__device__ __noinline__ void fused_h_sharpen(const FusedParams& p, char* smem, const FusedHSeg& g, int lt)
{
    constexpr int E = 16, T = UW / E, NW = T / 64, SPAN = UW / NW, NG = SPAN / 256;
    static_assert(NG * 256 == SPAN && NG >= 1, "span must be a multiple of 64 lanes x 4 pixels");
    constexpr float inv = 1.0f / (float)UW;
    using L = FusedHLds<UW>;
    float* Lb = (float*)(smem + L::XB);
    float* red = (float*)(smem + L::RED);
    unsigned* cnt = (unsigned*)(red + 32);
    const int uH = p.uH, a0 = g.a0, npairs = g.npairs, y0 = g.y0, y1 = g.y1, c = g.c;
    const long plane = (long)UW * uH;
    if (g.need_corner) {
        const FusedHAddr<UW, TK> S2{p.S2 + (long)g.c * p.NT * ((long)uH * TK), (unsigned)uH * TK};
        auto S2at = [&](int k, int row) -> float2 { return S2.at(k, row); };
        const int rs = g.rs;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) part += S2at(lt + 1 + T * q, rs).x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
        if ((lt & 63) == 0) red[lt >> 6] = part;
        if (lt == T - 1) {
            float2 d = S2at(0, rs), dp = S2at(0, rs ^ 1);
            red[16] = (rs & 1) ? d.x + dp.y : d.x - dp.y;
        }
    }
    __syncthreads();
    // sharpen group state: rows a-2 (ring[0]) and a-1 (ring[1]) of NG column groups, 6 values each (x0-1 .. x0+4)
    float ring[2][NG][6];
    float pn0 = 0.f, pn1 = 0.f, sv0 = 0.f;        // last thread only: row a-1 at UW-2, UW-1 and L(a, 0) of the previous step
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int h = 0; h < NG; h++)
#pragma unroll
            for (int k = 0; k < 6; k++) ring[r][h][k] = 0.f;
    for (int s = 0; s <= npairs; s++) {
        // ================= sharpen group: rows of pair i = s-1 arrive in Lb; outputs rows a-1 and a
        const int i = s - 1;
        const int a = a0 + 2 * i;
        const float* rowA = Lb;                       // row a
        const float* rowB = Lb + UW;                  // row a+1
        const bool out0 = i >= 0 && (a - 1) >= y0 && (a - 1) < y1;
        const bool out1 = i >= 0 && a >= y0 && a < y1;
        if (i >= 0) {
            const int wv = lt >> 6, lane = lt & 63;
#pragma unroll
            for (int h = 0; h < NG; h++) {
                const int x0 = SPAN * wv + 256 * h + 4 * lane;
                const bool last_chunk = (x0 + 4 == UW);
                float n2[6], n3[6];
                {
                    const float4 q2 = *(const float4*)(rowA + x0), q3 = *(const float4*)(rowB + x0);
                    n2[1] = q2.x; n2[2] = q2.y; n2[3] = q2.z; n2[4] = q2.w;
                    n3[1] = q3.x; n3[2] = q3.y; n3[3] = q3.z; n3[4] = q3.w;
                    n2[0] = (x0 > 0) ? rowA[x0 - 1] : q2.x;                       // id_x_m clamp (VkResample.cpp:889)
                    n3[0] = (x0 > 0) ? rowB[x0 - 1] : q3.x;
                    // x = UW wraps to x = 0 of the next row (quirk B5); row a+2 is not known yet
                    n2[5] = last_chunk ? rowB[0] : rowA[x0 + 4];
                    n3[5] = last_chunk ? rowB[0] : rowB[x0 + 4];
                }
                if (last_chunk) ring[1][h][5] = rowA[0];          // row a-1 wraps to L(a, 0), known now
                if (out0 || out1) {
                    float t[4][6];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        t[0][k] = ring[0][h][k];
                        t[1][k] = (a == 0) ? n2[k] : ring[1][h][k];   // row -1 clamps to row 0
                        t[2][k] = n2[k];
                        t[3][k] = n3[k];
                    }
                    if (last_chunk) {
                        // SE tap of pixel (a, UW-1) is L(a+2, 0): past the plane it clamps to row uH-1; in the
                        // last step it is the corner sample; otherwise the pixel is finished next step
                        const int r2 = min(a + 2, uH - 1) - a;
                        if (r2 <= 1) t[3][5] = (r2 == 0 ? rowA : rowB)[0];
                        else if (i == npairs - 1) {
                            float sum = 0.f;
                            for (int w2 = 0; w2 < NW; w2++) sum += red[w2];
                            t[3][5] = to_L<HALF>((red[16] + 2.0f * sum) * inv, p.upsq);
                        }
                    }
#pragma unroll
                    for (int w = 0; w < 2; w++) {
                        if (w == 0 ? !out0 : !out1) continue;
                        float hmn[3][4], hmx[3][4];          // rows w, w+1, w+2
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                hmn[r][k] = fminf(fminf(t[w + r][k], t[w + r][k + 1]), t[w + r][k + 2]);
                                hmx[r][k] = fmaxf(fmaxf(t[w + r][k], t[w + r][k + 1]), t[w + r][k + 2]);
                            }
                        float o[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float N = t[w][k + 1], S = t[w + 2][k + 1], Wv = t[w + 1][k], C = t[w + 1][k + 1], Ee = t[w + 1][k + 2];
                            const float mn0 = fminf(fminf(N, S), hmn[1][k]);
                            const float mx0 = fmaxf(fmaxf(N, S), hmx[1][k]);
                            const float mn1 = fminf(fminf(hmn[0][k], hmn[2][k]), mn0);
                            const float mx1 = fmaxf(fmaxf(hmx[0][k], hmx[2][k]), mx0);
                            if constexpr (HALF) o[k] = sharpen_eval_half_fast(N, S, Wv, Ee, C, mn0, mn1, mx0, mx1, p.coef);
                            else o[k] = sharpen_eval_fast(((N + Wv) + Ee) + S, C, mn0, mn1, mx0, mx1, p.coef);
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
                if (h == NG - 1 && lt == T - 1) {
                    // finish the pixel deferred by the previous pair: (a-2, UW-1); L(a,0) is known now.
                    // ring[0] = row a-2, ring[1] = row a-1 (still the old rows here); [3],[4] = x UW-2, UW-1
                    if (i > 0 && (a - 2) >= y0 && (a - 2) < y1 && a <= uH - 1) {
                        const float r2_0 = sv0;                         // L(a-2, 0)
                        const float r1_0 = ring[0][h][5];               // L(a-1, 0): row a-2's wrap neighbour
                        const float r0_0 = rowA[0];                     // L(a, 0)
                        const float ne = (a - 2 == 0) ? r1_0 : r2_0;
                        const float tq[3][6] = {{pn0, pn0, pn1, ne, ne, ne},
                                                {ring[0][h][3], ring[0][h][3], ring[0][h][4], r1_0, r1_0, r1_0},
                                                {ring[1][h][3], ring[1][h][3], ring[1][h][4], r0_0, r0_0, r0_0}};
                        float oq[4];
                        sharpen_quad<HALF>(tq, p.coef, oq);
                        const long of = c * plane + (long)(a - 2) * UW + (UW - 1);
                        if constexpr (HALF) ((__half*)p.out)[of] = __float2half_rn(oq[1]);
                        else ((float*)p.out)[of] = oq[1];
                    }
                    // row a-1 (row a for a == 0) at UW-2, UW-1 and L(a, 0) for the next step
                    pn0 = (a == 0) ? n2[3] : ring[1][h][3];
                    pn1 = (a == 0) ? n2[4] : ring[1][h][4];
                    sv0 = rowA[0];
                }
#pragma unroll
                for (int k = 0; k < 6; k++) { ring[0][h][k] = n2[k]; ring[1][h][k] = n3[k]; }
            }
        }
        // hand-off rows consumed (all LDS reads of this wave were issued above; LDS serves them in order)
        asm volatile("" ::: "memory");
        if ((lt & 63) == 0) __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }
}

template <int UW, bool HALF, int TK>
__global__ void __launch_bounds__(UW / 8) __attribute__((amdgpu_waves_per_eu(4))) k_c2r_sharpen_h(FusedParams p)
{
    constexpr int T = UW / 16;
    using L = FusedHLds<UW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* cnt = (unsigned*)((float*)(smem + L::RED) + 32);   // [0] transform-group barrier, [1] hand-off rows consumed
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / T);
    const int lt = tid - grp * T;
    const int uH = p.uH;
    const int pairs_per_plane = uH / 2;
    if (tid < 2) cnt[tid] = 0u;
    __syncthreads();
    GroupBarrier gbar{cnt, 0u, (unsigned)(T / 64)};
    unsigned consumed = 0u;                       // sharpen-wave arrivals the transform group has accounted for

    int f0 = blockIdx.x * p.pairs_per_strip;
    const int f1 = min(f0 + p.pairs_per_strip, 3 * pairs_per_plane);
    while (f0 < f1) {
        FusedHSeg g;
        g.c = f0 / pairs_per_plane;
        const int j0 = f0 - g.c * pairs_per_plane;
        const int j1 = min(j0 + (f1 - f0), pairs_per_plane);
        f0 += j1 - j0;
        g.y0 = 2 * j0; g.y1 = 2 * j1;
        const bool top = (g.y0 == 0);
        g.a0 = top ? 0 : g.y0 - 1;
        g.npairs = (j1 - j0) + 1;
        g.need_corner = !top && (g.y1 + 1 < uH);
        g.rs = g.y1 + 1;
        if (grp == 0) fused_h_transform<UW, HALF, TK>(p, smem, g, lt, gbar, consumed);
        else fused_h_sharpen<UW, HALF, TK>(p, smem, g, lt);
    }
}

}  // namespace fftup
