#!/usr/bin/env python3
"""Timing-only variants of the shipping kernels (results INVALID): source patches applied to a temporary copy of csrc/, the tree
is not touched.   python tools/vexp.py name...   ->  tools/scratch/lib_<name>.so      (tools/gpu_ab.sh takes them as variants)"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KO_FN = """template <int N, int E, int DIR, int TK, bool FINAL_TO_LDS, int S = 0, int RMAX = 8, bool WAVE = false>
__device__ __forceinline__ void reg_fft_ko(float2 (&v)[E], float2* __restrict__ buf, int p, int col,
                                        const TwSet<N, E, RMAX>& tws)
{
    constexpr int Ns = stage_ns(N, S, RMAX);
    constexpr int R = stage_radix(N, Ns, RMAX);
    reg_butterflies<N, E, R, Ns, DIR>(v, tws.w[S > 0 ? S - 1 : 0]);
    constexpr bool last = (Ns * R == N);
    if constexpr ((!last && S == 0) || (last && FINAL_TO_LDS)) {
        reg_scatter<N, E, R, Ns, TK>(v, buf, p, col);
        lds_sync<WAVE>();
    }
    if constexpr (!last) {
        if constexpr (S == 0) { reg_gather<N, E, TK>(v, buf, p, col); lds_sync<WAVE>(); }
        reg_fft_ko<N, E, DIR, TK, FINAL_TO_LDS, S + 1, RMAX, WAVE>(v, buf, p, col, tws);
    }
}

// =================================================================================== row R2C
struct RowR2CTParams {"""
PATCHES = {
    "base": [],
    "colv_nomid": [("kernels_dswap.hpp", "    __syncthreads();                                         // everybody has read the forward exchange\n", "")],
    "colv_nobar": [("kernels_dswap.hpp", "    __syncthreads();                                         // everybody has read the forward exchange\n", ""),
                   ("kernels_dswap.hpp", "    for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)(aw + 4096u * k) = r; }\n    __syncthreads();\n    const unsigned ar = zbase + 8u * (w * 512u + l);",
                    "    for (int k = 0; k < 8; k++) { lds_f2raw r = {v[k].x, v[k].y}; *(lds_f2*)(size_t)(aw + 4096u * k) = r; }\n    lds_sync<true>();\n    const unsigned ar = zbase + 8u * (w * 512u + l);")],
    "rowwave": [("kernels_pow2.hpp", "    reg_fft<W, E, +1, 1, true>(v, buf, tid, 0, tws);", "    reg_fft<W, E, +1, 1, true, 0, 8, true>(v, buf, tid, 0, tws);")],
    # column kernel with ONE barrier-synchronised exchange per transform instead of three (results invalid): what a digit-swap column kernel could gain
    "colnoex": [("kernels_pow2.hpp", """// =================================================================================== row R2C
struct RowR2CTParams {""", KO_FN),
                ("kernels_pow2.hpp", "    reg_fft<H, 8, +1, TK, true>(v, buf, pp, col, tws);            // F[k] natural order in LDS", "    reg_fft_ko<H, 8, +1, TK, true>(v, buf, pp, col, tws);"),
                ("kernels_pow2.hpp", "    reg_fft<H, 8, -1, TK, false>(v, buf, pp, col, tws);\n    float2* dst = p.S2", "    reg_fft_ko<H, 8, -1, TK, false>(v, buf, pp, col, tws);\n    float2* dst = p.S2")],
    "colwave": [("kernels_pow2.hpp", "    reg_fft<H, 8, +1, TK, true>(v, buf, pp, col, tws);            // F[k] natural order in LDS", "    reg_fft<H, 8, +1, TK, true, 0, 8, true>(v, buf, pp, col, tws);"),
                ("kernels_pow2.hpp", "    reg_fft<H, 8, -1, TK, false>(v, buf, pp, col, tws);\n    float2* dst = p.S2", "    reg_fft<H, 8, -1, TK, false, 0, 8, true>(v, buf, pp, col, tws);\n    float2* dst = p.S2")],
    # k_c2r_sharpen_g without the loads of the mirror partners (their values replaced by the thread's own elements)
    "nomirror": [("kernels_pow2.hpp", """                in.a[m] = gload(ra, ko[m]); in.am[m] = gload(ra, kom[m]);
                in.b[m] = gload(rb, ko[m]); in.bm[m] = gload(rb, kom[m]);""", """                in.a[m] = gload(ra, ko[m]); in.am[m] = in.a[m];
                in.b[m] = gload(rb, ko[m]); in.bm[m] = in.b[m];""")],
}
def build(name):
    tmp = "/tmp/vexp_" + name
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/vkresample_amd")
    shutil.copytree(ROOT + "/vkresample_amd/csrc", tmp + "/vkresample_amd/csrc")
    shutil.copytree(ROOT + "/include", tmp + "/include")
    for pt in PATCHES[name]:
        f, a, b = pt if len(pt) == 3 else ("kernels_dswap.hpp",) + tuple(pt)
        p = tmp + "/vkresample_amd/csrc/" + f
        s = open(p).read()
        assert s.count(a) == 1, (name, a, s.count(a))
        open(p, "w").write(s.replace(a, b))
    if name == "tk8":        # spectrum tiles of 8 columns instead of 4 (64-byte row pieces)
        q = tmp + "/vkresample_amd/csrc/plan.hpp"
        t = open(q).read()
        assert t.count("static constexpr int TUNED_TK = 4;") == 1
        open(q, "w").write(t.replace("static constexpr int TUNED_TK = 4;", "static constexpr int TUNED_TK = 8;"))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_variant(ROOT + "/tools/scratch/lib_%s.so" % name, csrc=tmp + "/vkresample_amd/csrc")
    print("built", name)
if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(build, sys.argv[1:]))
