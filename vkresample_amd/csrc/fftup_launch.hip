// fftup_launch.hip -- the frame's kernel launches: the ONLY translation unit that instantiates the frame kernels (kernels_*.hpp).
// One frame = row R2C -> column FFT / zero-pad / iFFT -> row C2R + sharpen (fused) on the stream of lane P->cur; the
// reference's 20 dispatches per frame (performVulkanUpscale, VkResample.cpp:1249-1279).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"
#include "kernels_generic.hpp"
#include "kernels_pow2.hpp"
#include "kernels_mixed.hpp"
#include "kernels_dswap.hpp"

using namespace fftup;

// ---- facts about the kernels the planner needs
int kernels_generic_max_threads(bool dbl) { return dbl ? GenericMaxThreads<double2>::value : GenericMaxThreads<float2>::value; }
int kernels_aot_mixed_plan(uint32_t W, uint32_t H)
{
    if (W == MixedCfg1080::W && H == MixedCfg1080::H) return 1;
    if (W == MixedCfg720::W && H == MixedCfg720::H) return 2;
    return 0;
}
size_t kernels_tuned_col_lds(uint32_t H) { return sizeof(float2) * (size_t)lswz_size((int)H * TUNED_TK); }   // both transforms of the column kernel have length H

// four-step rows (k_row4_a / k_row4_b): launch both passes; ATTR: only allow their dynamic LDS sizes (plan creation)
template <typename C, int DIR, int MODE, int TKA> static hipError_t four_pass_a(const fftup_plan::Four& f, const Row4Params<C>& q, int rows, hipStream_t st, bool attr)
{
    if (attr) return hipFuncSetAttribute((const void*)(k_row4_a<DIR, TKA, MODE, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.ldsA);
    hipLaunchKernelGGL((k_row4_a<DIR, TKA, MODE, C>), dim3(rows, f.n2 / TKA, 3), dim3(f.thrA), f.ldsA, st, q);
    return hipSuccess;
}
template <typename C, int DIR, int OUT, int TKB> static hipError_t four_pass_b(const fftup_plan::Four& f, const Row4Params<C>& q, int rows, hipStream_t st, bool attr)
{
    if (attr) return hipFuncSetAttribute((const void*)(k_row4_b<DIR, TKB, OUT, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.ldsB);
    hipLaunchKernelGGL((k_row4_b<DIR, TKB, OUT, C>), dim3(rows, f.n1 / TKB, 3), dim3(f.thrB), f.ldsB, st, q);
    return hipSuccess;
}
// (the two passes pick their tile widths independently: kernels are instantiated per pass and width, not per pair)
template <typename C, int DIR, int MODE, int OUT, bool ATTR>
static hipError_t four_run(const fftup_plan::Four& f, const Row4Params<C>& q, int rows, hipStream_t st)
{
    hipError_t e;
    switch (f.tka) {
    case 16: e = four_pass_a<C, DIR, MODE, 16>(f, q, rows, st, ATTR); break;
    case 8: e = four_pass_a<C, DIR, MODE, 8>(f, q, rows, st, ATTR); break;
    case 4: e = four_pass_a<C, DIR, MODE, 4>(f, q, rows, st, ATTR); break;
    case 2: e = four_pass_a<C, DIR, MODE, 2>(f, q, rows, st, ATTR); break;
    default: e = four_pass_a<C, DIR, MODE, 1>(f, q, rows, st, ATTR); break;
    }
    if (e != hipSuccess) return e;
    switch (f.tkb) {
    case 16: return four_pass_b<C, DIR, OUT, 16>(f, q, rows, st, ATTR);
    case 8: return four_pass_b<C, DIR, OUT, 8>(f, q, rows, st, ATTR);
    case 4: return four_pass_b<C, DIR, OUT, 4>(f, q, rows, st, ATTR);
    case 2: return four_pass_b<C, DIR, OUT, 2>(f, q, rows, st, ATTR);
    default: return four_pass_b<C, DIR, OUT, 1>(f, q, rows, st, ATTR);
    }
}
// forward rows of a plan: input mode from the slot's kind and the precision; inverse rows: output type from the precision
template <typename C, bool ATTR> static hipError_t four_forward(fftup_plan* P, const Row4Params<C>& q, int kind, hipStream_t st)
{
    if constexpr (sizeof(scalar_t<C>) == 8) return four_run<C, +1, IN_F64, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
    else {
        if (kind == 2) return P->half ? four_run<C, +1, IN_U8_F16, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st) : four_run<C, +1, IN_U8_F32, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
        return P->half ? four_run<C, +1, IN_F16, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st) : four_run<C, +1, IN_F32, OUT4_TILES, ATTR>(P->fourF, q, (int)P->H, st);
    }
}
template <typename C, bool ATTR> static hipError_t four_inverse(fftup_plan* P, const Row4Params<C>& q, hipStream_t st)
{
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->half) return four_run<C, -1, IN4_TILES, OUT4_HALF, ATTR>(P->fourI, q, (int)P->uH, st);
    }
    return four_run<C, -1, IN4_TILES, OUT4_DENSE, ATTR>(P->fourI, q, (int)P->uH, st);
}
// columns longer than the LDS: forward in place in S1 (tiles of one column = dense columns), inverse S1 -> S2 with shift and guard
template <typename C, bool ATTR> static hipError_t four_columns(fftup_plan* P, hipStream_t st)
{
    using S = scalar_t<C>;
    Row4Params<C> q{};
    const fftup_plan::Four &f = P->colF, &g = P->colI;
    q.spec = (const C*)P->lanes[P->cur].S1; q.T = (C*)P->lanes[P->cur].T4; q.R = P->lanes[P->cur].S1;
    q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twH; q.plan1 = f.p1; q.plan2 = f.p2;
    q.N = (int)P->H; q.N1 = f.n1; q.N2 = f.n2; q.rows = P->ncols; q.W = (int)P->H; q.TK = 1; q.NT = P->ncols; q.inv_norm = (S)1;
    hipError_t e = four_run<C, +1, IN4_DENSE, OUT4_DENSE, ATTR>(f, q, P->ncols, st);
    if (e != hipSuccess) return e;
    q.R = P->lanes[P->cur].S2; q.tw1 = (const C*)g.tw1; q.tw2 = (const C*)g.tw2; q.twN = (const C*)P->twUH; q.plan1 = g.p1; q.plan2 = g.p2;
    q.N = (int)P->uH; q.N1 = g.n1; q.N2 = g.n2; q.zlx = P->zly; q.zrx = P->zry; q.inv_norm = (S)(1.0 / (double)P->uH);
    return four_run<C, -1, IN4_DENSE_SHIFT, OUT4_DENSE, ATTR>(g, q, P->ncols, st);
}

// allow > 64 KB dynamic LDS -- for the kernels THIS plan launches, nothing else
int kernels_set_attributes(fftup_plan* P)
{
    const bool cplx = P->cplx;
    const uint32_t H = P->H, uW = P->uW;
#define PLAN_TRY(expr) HIP_TRY(expr)
#define SET_LDS(kern, bytes) PLAN_TRY(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
        const bool generic = !P->tuned && !P->mixed;
        // (same predicate as launch_frame: a plan-time plan without a row factorization runs the size-generic row kernel)
        const bool generic_rows = generic || (P->mixed == 3 && P->jit->choice.row_kind == 2);
        if (generic_rows && !cplx && !P->dbl) {
            if (P->half) { SET_LDS(k_row_r2c<IN_F16>, P->ldsRowF); SET_LDS(k_row_r2c<IN_U8_F16>, P->ldsRowF); }
            else { SET_LDS(k_row_r2c<IN_F32>, P->ldsRowF); SET_LDS(k_row_r2c<IN_U8_F32>, P->ldsRowF); }
        }
        if (generic && !cplx && !P->dbl) {
            if (P->half) SET_LDS(k_row_c2r<true>, P->ldsRowI); else SET_LDS(k_row_c2r<false>, P->ldsRowI);
        }
        if (P->colF.on) {                                    // columns in four steps (k_row4_a / k_row4_b on dense columns)
            if (P->dbl) PLAN_TRY((four_columns<double2, true>(P, nullptr))); else PLAN_TRY((four_columns<float2, true>(P, nullptr)));
        }
        if (generic && P->poly) {
            if (P->dbl) switch (P->TK) {
            case 8: SET_LDS((k_col_poly<8, double2>), P->ldsCol); break;
            case 4: SET_LDS((k_col_poly<4, double2>), P->ldsCol); break;
            case 2: SET_LDS((k_col_poly<2, double2>), P->ldsCol); break;
            default: SET_LDS((k_col_poly<1, double2>), P->ldsCol); break;
            }
            else switch (P->TK) {
            case 8: SET_LDS(k_col_poly<8>, P->ldsCol); break;
            case 4: SET_LDS(k_col_poly<4>, P->ldsCol); break;
            case 2: SET_LDS(k_col_poly<2>, P->ldsCol); break;
            default: SET_LDS(k_col_poly<1>, P->ldsCol); break;
            }
        }
        else if (generic && !P->dbl && !P->colF.on) {
            switch (P->TK) {
            case 8: SET_LDS(k_col<8>, P->ldsCol); break;
            case 4: SET_LDS(k_col<4>, P->ldsCol); break;
            case 2: SET_LDS(k_col<2>, P->ldsCol); break;
            default: SET_LDS(k_col<1>, P->ldsCol); break;
            }
        }
        if (cplx) {
            // (one instantiation per input type / output type / one-or-two-buffer form: only this plan's)
            const bool f1 = !P->fourF.on, i1 = !P->fourI.on;         // rows in one launch (else: four steps, below)
            if (P->dbl) { if (f1) SET_LDS((k_row_c2c_fwd<IN_F64, double2>), P->ldsRowF); if (i1) SET_LDS((k_row_c2c_inv<double2>), P->ldsRowI); }
            else if (P->half) {
                if (!f1) {}
                else if (P->inplaceF) { SET_LDS((k_row_c2c_fwd<IN_F16, float2, true>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F16, float2, true>), P->ldsRowF); }
                else { SET_LDS((k_row_c2c_fwd<IN_F16, float2>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F16, float2>), P->ldsRowF); }
                if (!i1) {}
                else if (P->inplaceI) SET_LDS((k_row_c2c_inv<float2, true, true>), P->ldsRowI);
                else SET_LDS((k_row_c2c_inv<float2, true, false>), P->ldsRowI);
            } else {
                if (!f1) {}
                else if (P->inplaceF) { SET_LDS((k_row_c2c_fwd<IN_F32, float2, true>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F32, float2, true>), P->ldsRowF); }
                else { SET_LDS((k_row_c2c_fwd<IN_F32, float2>), P->ldsRowF); SET_LDS((k_row_c2c_fwd<IN_U8_F32, float2>), P->ldsRowF); }
                if (!i1) {}
                else if (P->inplaceI) SET_LDS((k_row_c2c_inv<float2, false, true>), P->ldsRowI);
                else SET_LDS((k_row_c2c_inv<float2, false, false>), P->ldsRowI);
            }
            if (P->fourF.on) {
                if (P->dbl) PLAN_TRY((four_forward<double2, true>(P, Row4Params<double2>{}, 1, nullptr)));
                else { PLAN_TRY((four_forward<float2, true>(P, Row4Params<float2>{}, 1, nullptr))); PLAN_TRY((four_forward<float2, true>(P, Row4Params<float2>{}, 2, nullptr))); }
            }
            if (P->fourI.on) {
                if (P->dbl) PLAN_TRY((four_inverse<double2, true>(P, Row4Params<double2>{}, nullptr)));
                else PLAN_TRY((four_inverse<float2, true>(P, Row4Params<float2>{}, nullptr)));
            }
        }
        if (P->dbl) {
            if (!cplx) {
                if (P->inplaceF) SET_LDS((k_row_r2c<IN_F64, double2, true>), P->ldsRowF); else SET_LDS((k_row_r2c<IN_F64, double2>), P->ldsRowF);
                if (P->inplaceI) SET_LDS((k_row_c2r<false, double2, true>), P->ldsRowI); else SET_LDS((k_row_c2r<false, double2>), P->ldsRowI);
            }
            if (P->poly) {}
            else if (P->inplaceC) switch (P->TK) {
            case 8: SET_LDS((k_col<8, double2, true>), P->ldsCol); break;
            case 4: SET_LDS((k_col<4, double2, true>), P->ldsCol); break;
            case 2: SET_LDS((k_col<2, double2, true>), P->ldsCol); break;
            default: SET_LDS((k_col<1, double2, true>), P->ldsCol); break;
            }
            else if (!P->colF.on) switch (P->TK) {
            case 8: SET_LDS((k_col<8, double2>), P->ldsCol); break;
            case 4: SET_LDS((k_col<4, double2>), P->ldsCol); break;
            case 2: SET_LDS((k_col<2, double2>), P->ldsCol); break;
            default: SET_LDS((k_col<1, double2>), P->ldsCol); break;
            }
        }
#define SET_FUSED(PL, TKK) do { if (P->u8out) { if (P->half) SET_LDS((k_c2r_sharpen_g<PL, true, TKK, 2, 4, true>), FusedGLds<PL>::TOTAL); \
                                                else SET_LDS((k_c2r_sharpen_g<PL, false, TKK, 2, 4, true>), FusedGLds<PL>::TOTAL); } \
                                else if (P->half) SET_LDS((k_c2r_sharpen_g<PL, true, TKK>), FusedGLds<PL>::TOTAL); \
                                else SET_LDS((k_c2r_sharpen_g<PL, false, TKK>), FusedGLds<PL>::TOTAL); } while (0)
#define SET_MIXED(CFG) do { SET_LDS(k_col_m<CFG>, P->ldsCol); \
        if (P->half) SET_LDS((k_row_c2r_ct<CFG::CT, true>), P->ldsRowI); else SET_LDS((k_row_c2r_ct<CFG::CT, false>), P->ldsRowI); \
        SET_FUSED(CFG::FUSED, 4); } while (0)
        if (P->mixed == 1) { SET_MIXED(MixedCfg1080); }
        if (P->mixed == 2) { SET_MIXED(MixedCfg720); }
#undef SET_MIXED
        if (P->tuned) {
            switch (uW) {
            case 1024: SET_FUSED(FusedPlanPow2<1024>, TUNED_TK); break;
            case 2048: SET_FUSED(FusedPlanPow2<2048>, TUNED_TK); break;
            default: SET_FUSED(FusedPlanPow2<4096>, TUNED_TK); break;
            }
            switch (H) {                                      // (digit-swap column kernels: 4 KB of LDS per wave)
            case 256: SET_LDS((k_col_v<TUNED_TK, 256>), 8192); break;
            case 512: SET_LDS((k_col_v<TUNED_TK, 512>), 16384); break;
            default: SET_LDS((k_col_v<TUNED_TK, 1024>), 32768); break;
            }
        }
#undef SET_FUSED
#undef SET_LDS
#undef PLAN_TRY
    return FFTUP_OK;
}

// ------------------------------------------------------------------------------------------------
// the two host loops of the reference as kernels (VR:1636-1685, VR:1708-1748)
void launch_unpack(fftup_plan* P, uint32_t slot, hipStream_t st)
{
    dim3 grid((P->W + 255) / 256, P->H);
    if (P->dbl)
        hipLaunchKernelGGL(k_unpack_u8_f64, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, (double*)P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
    else if (P->half)
        hipLaunchKernelGGL(k_unpack_u8<true>, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
    else
        hipLaunchKernelGGL(k_unpack_u8<false>, grid, dim3(256), 0, st, P->in_u8[slot], (long)3 * P->W, P->in_planar[slot],
                           (int)P->W, (int)P->H, (long)P->in_plane_stride);
}

void launch_pack(fftup_plan* P, uint32_t slot, uint8_t* dst, hipStream_t st)
{
    dim3 grid(P->dbl ? (P->uW + 255) / 256 : (P->uW + 1023) / 1024, P->uH);      // (float / half: four pixels per thread)
    const int wrap = (P->cfg.flags & FFTUP_FLAG_U8_WRAP) ? 1 : 0;
    if (P->dbl) hipLaunchKernelGGL(k_pack_u8_f64, grid, dim3(256), 0, st, (const double*)P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
    else if (P->half) hipLaunchKernelGGL(k_pack_u8<true>, grid, dim3(256), 0, st, P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
    else hipLaunchKernelGGL(k_pack_u8<false>, grid, dim3(256), 0, st, P->out[slot], dst, (int)P->uW, (int)P->uH, wrap);
}

// (test builds, FFTUP_EXPERIMENT planes=1: the row and column passes of the power-of-two plans on the first colour plane only --
// what one plane's pass costs alone on the whole GPU, the start-up a frame pipelined by planes cannot hide; results invalid)
static unsigned pass_planes()
{
    const char* e = fftup_jit::experiment("planes");
    return e ? (unsigned)std::max(1, std::min(3, atoi(e))) : 3u;
}
template <int W> static void launch_r2c_t(fftup_plan* P, const RowR2CTParams& p, int mode)
{
    dim3 grid(P->H / 2, pass_planes()), block(W / 8);
    switch (mode) {
    case IN_F32: hipLaunchKernelGGL((k_row_r2c_t<W, IN_F32, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    case IN_F16: hipLaunchKernelGGL((k_row_r2c_t<W, IN_F16, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    case IN_U8_F32: hipLaunchKernelGGL((k_row_r2c_t<W, IN_U8_F32, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    default: hipLaunchKernelGGL((k_row_r2c_t<W, IN_U8_F16, TUNED_TK>), grid, block, 0, P->lanes[P->cur].stream, p); break;
    }
}
template <int UW> static void launch_c2r_t(fftup_plan* P, const RowC2RTParams& p)
{
    dim3 grid(P->uH / 2, 3), block(UW / 8);
    if (P->half) hipLaunchKernelGGL((k_row_c2r_t<UW, true, TUNED_TK, true>), grid, block, 0, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_row_c2r_t<UW, false, TUNED_TK, true>), grid, block, 0, P->lanes[P->cur].stream, p);
}

// workgroups of the fused C2R+sharpen kernel.  Planes: strips in linear order over the 3 uH/2 row pairs.  Fused 8-bit store:
// strips per plane, the three planes' strips of the same rows 8 workgroups apart, rows of 8 strips (k_c2r_sharpen_g, OUT_U8)
static unsigned fused_grid(const fftup_plan* P, int pairs_per_strip)
{
    const int ppp = (int)P->uH / 2;
    if (P->u8out) return (unsigned)(((ppp + pairs_per_strip - 1) / pairs_per_strip + 7) / 8 * 24);
    return (unsigned)((3 * ppp + pairs_per_strip - 1) / pairs_per_strip);
}
template <class PL> static void launch_fused_t(fftup_plan* P, const FusedParams& p)
{
    dim3 grid(fused_grid(P, p.pairs_per_strip)), block(PL::T);
    hipStream_t st = P->lanes[P->cur].stream;
    if (P->u8out) {
        if (P->half) hipLaunchKernelGGL((k_c2r_sharpen_g<PL, true, TUNED_TK, 2, 4, true>), grid, block, FusedGLds<PL>::TOTAL, st, p);
        else hipLaunchKernelGGL((k_c2r_sharpen_g<PL, false, TUNED_TK, 2, 4, true>), grid, block, FusedGLds<PL>::TOTAL, st, p);
    }
    else if (P->half) hipLaunchKernelGGL((k_c2r_sharpen_g<PL, true, TUNED_TK>), grid, block, FusedGLds<PL>::TOTAL, st, p);
    else hipLaunchKernelGGL((k_c2r_sharpen_g<PL, false, TUNED_TK>), grid, block, FusedGLds<PL>::TOTAL, st, p);
}
static FusedParams fused_params(fftup_plan* P, uint32_t out_slot)
{
    FusedParams p{};
    p.S1 = P->lanes[P->cur].S1; p.odd_delta = (unsigned)(P->lanes[P->cur].S2 - P->lanes[P->cur].S1);
    if (P->U == 1) { p.S1 = P->lanes[P->cur].S2; p.odd_delta = 0; }      // half-integer factor: one buffer with all rows (k_col_pad)
    p.out = P->out[out_slot]; p.tw = P->twUW; p.uH = (int)P->uH; p.NT = P->NT;
    p.pairs_per_strip = P->pairs_per_strip; p.upsq = P->upsq; p.coef = P->coef;
    p.u8_wrap = (P->cfg.flags & FFTUP_FLAG_U8_WRAP) ? 1 : 0;
    return p;
}

static bool fast_sharpen_ok(const fftup_plan* P) { return !P->dbl && P->uW % 256 == 0 && P->uH % 16 == 0; }

static int launch_frame_tuned(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    const int kind = P->in_kind[in_slot];
    if (which < 0 || which == 0) {
        RowR2CTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.tw = P->twW; p.H = (int)P->H; p.NT = P->NT;
        int mode;
        if (kind == 2) { p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0; mode = P->half ? IN_U8_F16 : IN_U8_F32; }
        else { p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride; mode = P->half ? IN_F16 : IN_F32; }
        switch (P->W) {
        case 512: launch_r2c_t<512>(P, p, mode); break;
        case 1024: launch_r2c_t<1024>(P, p, mode); break;
        default: launch_r2c_t<2048>(P, p, mode); break;
        }
    }
    if (which < 0 || which == 1) {
        ColTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.twH = P->twH; p.twUH = P->twUH; p.W = (int)P->W; p.NT = P->NT;
        p.zly = P->zly; p.zry = P->zry;
        switch (P->H) {
        case 256: hipLaunchKernelGGL((k_col_v<TUNED_TK, 256>), dim3(P->NT, pass_planes()), dim3(128), 8192, P->lanes[P->cur].stream, p); break;
        case 512: hipLaunchKernelGGL((k_col_v<TUNED_TK, 512>), dim3(P->NT, pass_planes()), dim3(256), 16384, P->lanes[P->cur].stream, p); break;
        default: hipLaunchKernelGGL((k_col_v<TUNED_TK, 1024>), dim3(P->NT, pass_planes()), dim3(512), 32768, P->lanes[P->cur].stream, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fused) {
        const FusedParams p = fused_params(P, out_slot);
        switch (P->uW) {
        case 1024: launch_fused_t<FusedPlanPow2<1024>>(P, p); break;
        case 2048: launch_fused_t<FusedPlanPow2<2048>>(P, p); break;
        default: launch_fused_t<FusedPlanPow2<4096>>(P, p); break;
        }
        P->R_valid = false;
    } else if (which < 0 || which == 2 || which == 22) {   // 22: pre-sharpen tap requested for a fused plan
        RowC2RTParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = P->twUW; p.uH = (int)P->uH; p.NT = P->NT;
        switch (P->uW) {
        case 1024: launch_c2r_t<1024>(P, p); break;
        case 2048: launch_c2r_t<2048>(P, p); break;
        default: launch_c2r_t<4096>(P, p); break;
        }
        P->R_valid = true;
    }
    return FFTUP_OK;
}

static void launch_sharpen_fast(fftup_plan* P, uint32_t out_slot)
{
    SharpenTParams p{};
    p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
    dim3 grid(P->uW / 256, P->uH / 16, 3), block(64, 4);
    if (P->half) hipLaunchKernelGGL((k_sharpen_t<true, 4>), grid, block, 0, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_sharpen_t<false, 4>), grid, block, 0, P->lanes[P->cur].stream, p);
}

// -p 1: the size-generic kernels instantiated on double2 + the double sharpen
static int launch_frame_f64(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    hipStream_t st = P->lanes[P->cur].stream;
    if (which < 0 || which == 0) {
        RowR2CParamsT<double2> p{};
        p.S1 = (double2*)P->lanes[P->cur].S1; p.tw = (const double2*)P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
        if (P->inplaceF) hipLaunchKernelGGL((k_row_r2c<IN_F64, double2, true>), dim3(P->H / 2, 3), dim3(P->thrW), P->ldsRowF, st, p);
        else hipLaunchKernelGGL((k_row_r2c<IN_F64, double2>), dim3(P->H / 2, 3), dim3(P->thrW), P->ldsRowF, st, p);
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<double2, false>(P, st);
    else if (which < 0 || which == 1) {
        ColParamsT<double2> p{};
        p.S1 = (const double2*)P->lanes[P->cur].S1; p.S2 = (double2*)P->lanes[P->cur].S2;
        p.twH = (const double2*)P->twH; p.twUH = (const double2*)P->twUH; p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = 1.0 / (double)P->uH;
        dim3 grid(P->NT, 3), block(P->thrCol);
        if (P->poly) switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col_poly<8, double2>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col_poly<4, double2>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col_poly<2, double2>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col_poly<1, double2>), grid, block, P->ldsCol, st, p); break;
        }
        else if (P->inplaceC) switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col<8, double2, true>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col<4, double2, true>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col<2, double2, true>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col<1, double2, true>), grid, block, P->ldsCol, st, p); break;
        }
        else switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col<8, double2>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col<4, double2>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col<2, double2>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col<1, double2>), grid, block, P->ldsCol, st, p); break;
        }
    }
    if (which < 0 || which == 2) {
        RowC2RParamsT<double2> p{};
        p.S1 = (const double2*)P->lanes[P->cur].S1; p.S2 = (const double2*)P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = (const double2*)P->twUW; p.plan = P->planUW;
        p.W = (int)P->W; p.uW = (int)P->uW; p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = 1.0 / (double)P->uW; p.poly = P->poly;
        if (P->inplaceI) hipLaunchKernelGGL((k_row_c2r<false, double2, true>), dim3(P->uH / 2, 3), dim3(P->thrUW), P->ldsRowI, st, p);
        else hipLaunchKernelGGL((k_row_c2r<false, double2>), dim3(P->uH / 2, 3), dim3(P->thrUW), P->ldsRowI, st, p);
    }
    if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        const dim3 sgrid((P->uW + 511) / 512, (P->uH + SHARPEN_F64_RPT - 1) / SHARPEN_F64_RPT, 3);
        const char* ex = fftup_jit::experiment("f64_exact_sharpen");          // (test builds: IEEE divisions and root)
        if (ex && atoi(ex)) {
            if (P->uW % 2 == 0) hipLaunchKernelGGL((k_sharpen_f64<true, true>), sgrid, dim3(64, 4), 0, st, p);
            else hipLaunchKernelGGL((k_sharpen_f64<false, true>), sgrid, dim3(64, 4), 0, st, p);
        }
        else if (P->uW % 2 == 0) hipLaunchKernelGGL(k_sharpen_f64<true>, sgrid, dim3(64, 4), 0, st, p);
        else hipLaunchKernelGGL(k_sharpen_f64<false>, sgrid, dim3(64, 4), 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

// non-R2C path: four launches of size-generic kernels on complex data
template <typename C, int MODE> static void launch_c2c_fwd(fftup_plan* P, const RowR2CParamsT<C>& p, hipStream_t st)
{
    const dim3 grid(P->H, 3);
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->inplaceF) { hipLaunchKernelGGL((k_row_c2c_fwd<MODE, C, true>), grid, dim3(1024), P->ldsRowF, st, p); return; }
    }
    hipLaunchKernelGGL((k_row_c2c_fwd<MODE, C, false>), grid, dim3(P->thrW), P->ldsRowF, st, p);
}
template <typename C, bool HALF_OUT> static void launch_c2c_inv(fftup_plan* P, const RowC2RParamsT<C>& p, hipStream_t st)
{
    const dim3 grid(P->uH, 3);
    if constexpr (sizeof(scalar_t<C>) == 4) {
        if (P->inplaceI) { hipLaunchKernelGGL((k_row_c2c_inv<C, HALF_OUT, true>), grid, dim3(1024), P->ldsRowI, st, p); return; }
    }
    hipLaunchKernelGGL((k_row_c2c_inv<C, HALF_OUT, false>), grid, dim3(P->thrUW), P->ldsRowI, st, p);
}
template <typename C> static int launch_frame_cplx(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    hipStream_t st = P->lanes[P->cur].stream;
    const int kind = P->in_kind[in_slot];
    using S = scalar_t<C>;
    if ((which < 0 || which == 0) && P->fourF.on) {             // rows beyond one LDS buffer: four steps through HBM
        Row4Params<C> q{};
        const fftup_plan::Four& f = P->fourF;
        q.T = (C*)P->lanes[P->cur].T4; q.S1 = (C*)P->lanes[P->cur].S1; q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twW;
        q.plan1 = f.p1; q.plan2 = f.p2; q.N = (int)P->W; q.N1 = f.n1; q.N2 = f.n2; q.rows = (int)P->H; q.W = (int)P->W; q.TK = P->TK; q.NT = P->NT;
        if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; }
        else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; }
        (void)four_forward<C, false>(P, q, kind, st);
    } else if (which < 0 || which == 0) {
        RowR2CParamsT<C> p{};
        p.S1 = (C*)P->lanes[P->cur].S1; p.tw = (const C*)P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        if (kind == 2) {
            p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0;
            if constexpr (sizeof(S) == 4) {
                if (P->half) launch_c2c_fwd<C, IN_U8_F16>(P, p, st);
                else launch_c2c_fwd<C, IN_U8_F32>(P, p, st);
            }
        } else {
            p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
            if constexpr (sizeof(S) == 8) launch_c2c_fwd<C, IN_F64>(P, p, st);
            else if (P->half) launch_c2c_fwd<C, IN_F16>(P, p, st);
            else launch_c2c_fwd<C, IN_F32>(P, p, st);
        }
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<C, false>(P, st);
    else if (which < 0 || which == 1) {
        ColParamsT<C> p{};
        p.S1 = (const C*)P->lanes[P->cur].S1; p.S2 = (C*)P->lanes[P->cur].S2; p.twH = (const C*)P->twH; p.twUH = (const C*)P->twUH;
        p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = (S)(1.0 / (double)P->uH);
        dim3 grid(P->NT, 3), block(P->thrCol);
        switch (P->TK) {
        case 8: hipLaunchKernelGGL((k_col<8, C>), grid, block, P->ldsCol, st, p); break;
        case 4: hipLaunchKernelGGL((k_col<4, C>), grid, block, P->ldsCol, st, p); break;
        case 2: hipLaunchKernelGGL((k_col<2, C>), grid, block, P->ldsCol, st, p); break;
        default: hipLaunchKernelGGL((k_col<1, C>), grid, block, P->ldsCol, st, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fourI.on) {
        Row4Params<C> q{};
        const fftup_plan::Four& f = P->fourI;
        q.spec = (const C*)P->lanes[P->cur].S2; q.T = (C*)P->lanes[P->cur].T4; q.R = P->lanes[P->cur].R;
        q.tw1 = (const C*)f.tw1; q.tw2 = (const C*)f.tw2; q.twN = (const C*)P->twUW; q.plan1 = f.p1; q.plan2 = f.p2;
        q.N = (int)P->uW; q.N1 = f.n1; q.N2 = f.n2; q.rows = (int)P->uH; q.W = (int)P->W; q.TK = P->TK; q.NT = P->NT; q.zlx = P->zlx; q.zrx = P->zrx;
        q.inv_norm = (S)(1.0 / (double)P->uW);
        (void)four_inverse<C, false>(P, q, st);
        P->R_valid = true;
    } else if (which < 0 || which == 2) {
        RowC2RParamsT<C> p{};
        p.S2 = (const C*)P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = (const C*)P->twUW; p.plan = P->planUW;
        p.W = (int)P->W; p.uW = (int)P->uW; p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = (S)(1.0 / (double)P->uW);
        bool done = false;
        if constexpr (sizeof(S) == 4) {
            if (P->half) { launch_c2c_inv<C, true>(P, p, st); done = true; }
        }
        if (!done) launch_c2c_inv<C, false>(P, p, st);
        P->R_valid = true;
    }
    if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        bool done = false;
        if constexpr (sizeof(S) == 4) {
            if (P->half) { hipLaunchKernelGGL((k_sharpen_c<C, true>), dim3((P->uW + 255) / 256, P->uH, 3), dim3(256), 0, st, p); done = true; }
        }
        if (!done) hipLaunchKernelGGL((k_sharpen_c<C>), dim3((P->uW + 255) / 256, P->uH, 3), dim3(256), 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

// row R2C / stand-alone C2R launches of the register-resident mixed-radix plans (kernels_mixed.hpp)
template <class CFG> static void launch_row_mixed(fftup_plan* P, uint32_t in_slot, int kind)
{
    RowR2CTParams q{};
    q.S1 = P->lanes[P->cur].S1; q.tw = P->twW; q.H = (int)P->H; q.NT = P->NT;
    int mode;
    if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; mode = P->half ? IN_U8_F16 : IN_U8_F32; }
    else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; mode = P->half ? IN_F16 : IN_F32; }
    hipStream_t st = P->lanes[P->cur].stream;
    const dim3 grid(P->H / 2, 3), block(CFG::ROW_T);
    switch (mode) {
    case IN_F32: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_F32>), grid, block, 0, st, q); break;
    case IN_F16: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_F16>), grid, block, 0, st, q); break;
    case IN_U8_F32: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_U8_F32>), grid, block, 0, st, q); break;
    default: hipLaunchKernelGGL((k_row_r2c_m<CFG, IN_U8_F16>), grid, block, 0, st, q); break;
    }
}
template <class CT> static void launch_c2r_ct(fftup_plan* P, dim3 grid, const RowC2RParams& p)
{
    if (P->half) hipLaunchKernelGGL((k_row_c2r_ct<CT, true>), grid, dim3(CT::T), P->ldsRowI, P->lanes[P->cur].stream, p);
    else hipLaunchKernelGGL((k_row_c2r_ct<CT, false>), grid, dim3(CT::T), P->ldsRowI, P->lanes[P->cur].stream, p);
}

// (a frame's later launches must not mask the failure of an earlier one)
static void keep_first(hipError_t& first, hipError_t e) { if (first == hipSuccess) first = e; }

int launch_frame(fftup_plan* P, uint32_t in_slot, uint32_t out_slot, int which)
{
    const int kind = P->in_kind[in_slot];
    if (kind == 0 && which != 22) return fail(FFTUP_E_NO_INPUT, "no input uploaded for this slot");     // (22, the pre-sharpen tap, reads spectra only)
    if (P->cplx) return P->dbl ? launch_frame_cplx<double2>(P, in_slot, out_slot, which) : launch_frame_cplx<float2>(P, in_slot, out_slot, which);
    if (P->dbl) return launch_frame_f64(P, in_slot, out_slot, which);
    if (P->tuned) {
        launch_frame_tuned(P, in_slot, out_slot, which);
        if ((which < 0 || which == 3) && !P->fused) launch_sharpen_fast(P, out_slot);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        return FFTUP_OK;
    }
    hipError_t jerr = hipSuccess;       // launches from a run-time specialised code object report their errors directly
    if (which < 0 || which == 0) {
        RowR2CParams p{};
        p.S1 = P->lanes[P->cur].S1; p.tw = P->twW; p.plan = P->planW; p.W = (int)P->W; p.H = (int)P->H;
        p.TK = P->TK; p.NT = P->NT;
        dim3 grid(P->H / 2, 3), block(P->thrW);
        if (P->mixed == 3 && P->jit->choice.row_kind != 2) {
            RowR2CTParams q{};
            q.S1 = P->lanes[P->cur].S1; q.tw = P->twW; q.H = (int)P->H; q.NT = P->NT;
            if (kind == 2) { q.in = P->in_u8[in_slot]; q.in_row_stride = 3l * P->W; q.in_plane_stride = 0; }
            else { q.in = P->in_planar[in_slot]; q.in_row_stride = P->W; q.in_plane_stride = (long)P->in_plane_stride; }
            keep_first(jerr, fftup_jit::launch(P->jit->fn[kind == 2 ? fftup_jit::K_ROW_U8 : fftup_jit::K_ROW_PLANAR], grid, dim3(P->jit->choice.row_block), 0,
                                     P->lanes[P->cur].stream, q));
        } else if (P->mixed == 1 || P->mixed == 2) {
            if (P->mixed == 1) launch_row_mixed<MixedCfg1080>(P, in_slot, kind); else launch_row_mixed<MixedCfg720>(P, in_slot, kind);
        } else if (kind == 2) {
            p.in = P->in_u8[in_slot]; p.in_row_stride = 3l * P->W; p.in_plane_stride = 0;
            if (P->half) hipLaunchKernelGGL(k_row_r2c<IN_U8_F16>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
            else hipLaunchKernelGGL(k_row_r2c<IN_U8_F32>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
        } else {
            p.in = P->in_planar[in_slot]; p.in_row_stride = P->W; p.in_plane_stride = (long)P->in_plane_stride;
            if (P->half) hipLaunchKernelGGL(k_row_r2c<IN_F16>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
            else hipLaunchKernelGGL(k_row_r2c<IN_F32>, grid, block, P->ldsRowF, P->lanes[P->cur].stream, p);
        }
    }
    if ((which < 0 || which == 1) && P->colF.on) (void)four_columns<float2, false>(P, P->lanes[P->cur].stream);
    else if (which < 0 || which == 1) {
        ColParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.twH = P->twH; p.twUH = P->twUH; p.planH = P->planH; p.planUH = P->planUH;
        p.W = (int)P->W; p.H = (int)P->H; p.uH = (int)P->uH; p.NT = P->NT; p.ncols = P->ncols; p.zly = P->zly; p.zry = P->zry;
        p.inv_norm = 1.0f / (float)P->uH;
        dim3 grid(P->NT, 3), block(P->thrCol);
        if (P->mixed) {
            ColTParams q{};
            q.S1 = P->lanes[P->cur].S1; q.S2 = P->lanes[P->cur].S2; q.twH = P->twH; q.twUH = P->twUH; q.W = (int)P->W; q.NT = P->NT;
            q.zly = P->zly; q.zry = P->zry;
            if (P->mixed == 3) {
                const auto& ch = P->jit->choice;
                const dim3 jgrid(P->NT * (ch.col_kind >= 3 ? 4 / ch.col_cols : 1), 3);        // (long columns: two per workgroup)
                keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_COL], jgrid, dim3(ch.col_block), P->ldsCol, P->lanes[P->cur].stream, q));
            }
            else if (P->mixed == 1) hipLaunchKernelGGL(k_col_m<MixedCfg1080>, grid, dim3(4 * MixedCfg1080::COL_TPC), P->ldsCol, P->lanes[P->cur].stream, q);
            else hipLaunchKernelGGL(k_col_m<MixedCfg720>, grid, dim3(4 * MixedCfg720::COL_TPC), P->ldsCol, P->lanes[P->cur].stream, q);
        } else if (P->poly) switch (P->TK) {
        case 8: hipLaunchKernelGGL(k_col_poly<8>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 4: hipLaunchKernelGGL(k_col_poly<4>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 2: hipLaunchKernelGGL(k_col_poly<2>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        default: hipLaunchKernelGGL(k_col_poly<1>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        } else switch (P->TK) {
        case 8: hipLaunchKernelGGL(k_col<8>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 4: hipLaunchKernelGGL(k_col<4>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        case 2: hipLaunchKernelGGL(k_col<2>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        default: hipLaunchKernelGGL(k_col<1>, grid, block, P->ldsCol, P->lanes[P->cur].stream, p); break;
        }
    }
    if ((which < 0 || which == 2) && P->fused) {
        if (P->mixed == 3) {
            const FusedParams fp = fused_params(P, out_slot);
            keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_FUSED], dim3(fused_grid(P, fp.pairs_per_strip)),
                                     dim3(P->jit->choice.fused_t), P->jit->choice.fused_lds, P->lanes[P->cur].stream, fp));
        } else if (P->mixed == 2) launch_fused_t<MixedCfg720::FUSED>(P, fused_params(P, out_slot));
        else launch_fused_t<MixedCfg1080::FUSED>(P, fused_params(P, out_slot));       // (only the mixed plans are fused on this path)
        P->R_valid = false;
    } else if (which < 0 || which == 2 || which == 22) {                 // 22: pre-sharpen tap requested for a fused plan
        RowC2RParams p{};
        p.S1 = P->lanes[P->cur].S1; p.S2 = P->lanes[P->cur].S2; p.R = P->lanes[P->cur].R; p.tw = P->twUW; p.plan = P->planUW; p.W = (int)P->W; p.uW = (int)P->uW;
        p.uH = (int)P->uH; p.TK = P->TK; p.NT = P->NT; p.zlx = P->zlx; p.zrx = P->zrx;
        p.inv_norm = 1.0f / (float)P->uW; p.poly = P->poly && !P->mixed;
        dim3 grid(P->uH / 2, 3), block(P->thrUW);
        if (P->mixed) {
            if (P->mixed == 3) {
                if (P->U == 1) p.S1 = p.S2;                              // half-integer factor: all rows in S2
                keep_first(jerr, fftup_jit::launch(P->jit->fn[fftup_jit::K_C2R_CT], grid, dim3(P->jit->choice.ct_t), P->ldsRowI, P->lanes[P->cur].stream, p));
            }
            else if (P->mixed == 1) launch_c2r_ct<MixedCfg1080::CT>(P, grid, p);
            else launch_c2r_ct<MixedCfg720::CT>(P, grid, p);
        } else if (P->half) hipLaunchKernelGGL(k_row_c2r<true>, grid, block, P->ldsRowI, P->lanes[P->cur].stream, p);
        else hipLaunchKernelGGL(k_row_c2r<false>, grid, block, P->ldsRowI, P->lanes[P->cur].stream, p);
        P->R_valid = true;
    }
    if (P->fused) {
        // sharpen is part of launch 2
    } else if ((which < 0 || which == 3) && fast_sharpen_ok(P)) {
        launch_sharpen_fast(P, out_slot);
    } else if (which < 0 || which == 3) {
        SharpenParams p{};
        p.R = P->lanes[P->cur].R; p.out = P->out[out_slot]; p.uW = (int)P->uW; p.uH = (int)P->uH; p.upsq = P->upsq; p.coef = P->coef;
        dim3 grid((P->uW + 1023) / 1024, P->uH, 3), block(256);      // (four pixels per thread; a width of 2 gave an empty grid until round 5)
        if (P->half) hipLaunchKernelGGL(k_sharpen<true>, grid, block, 0, P->lanes[P->cur].stream, p);
        else hipLaunchKernelGGL(k_sharpen<false>, grid, block, 0, P->lanes[P->cur].stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = jerr;
    if (e != hipSuccess) return fail(FFTUP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return FFTUP_OK;
}

// 64-bit wrapping sum of 32-bit words (fftup_output_checksum): per-thread partial sums, wave reduction, one atomic per wave
__global__ void __launch_bounds__(256) k_checksum(const uint32_t* __restrict__ w, size_t n, unsigned long long* sum)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += w[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

void launch_checksum(fftup_plan* P, uint32_t slot, hipStream_t st)
{
    const size_t nwords = (size_t)3 * P->uW * P->uH * (P->u8out ? 1 : P->esz) / 4;       // (uW, uH even: whole words for binary16 and bytes too)
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, st, (const uint32_t*)P->out[slot], nwords, (unsigned long long*)P->d_sum);
}

#ifdef FFTUP_PLANE_STAMPS
// measurement build: the workgroups' begin / end stamps of the last frame's row and column pass (tools/plane_stamps.py)
extern "C" __attribute__((visibility("default"))) int fftup_debug_plane_stamps(unsigned long long* out)
{
    return out && hipMemcpyFromSymbol(out, HIP_SYMBOL(fftup::g_plane_stamps), sizeof(unsigned long long) * 3 * 2048 * 3) != hipSuccess;
}
#endif
