#!/usr/bin/env python3
"""Where does a step of the fused C2R+sharpen kernel spend its cycles?  Builds an INSTRUMENTED copy of the library (the tree
itself is not touched): thread <tid> of workgroup 97 of k_c2r_sharpen_g records s_memtime at the phase boundaries of every
step (butterflies done / scatter issued / barrier passed / gather landed per exchange, L rows written, sharpen done);
the library dumps the marks to $FFTUP_DBG_OUT when a plan is destroyed, tools/phase_marks_read.py prints the timeline.

    python tools/phase_marks.py <tid>        ->  tools/scratch/lib_dbg<tid>.so      (tid = 0, 64, .. 448: one wave each)
    on the GPU box:  cp tools/scratch/lib_dbg0.so vkresample_amd/libfftup.so
                     FFTUP_DBG_OUT=marks.bin python bench.py --steps 1 --warmup 1 --repeats 1 --frames-per-step 8 --streams 1 --no-cpu-baseline --no-others
                     python tools/phase_marks_read.py marks.bin
Results of round 3: profiles/r03_a_phase_timeline.txt."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/tmp/dbg_build_" + (sys.argv[1] if len(sys.argv) > 1 else "0")
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp + "/vkresample_amd")
shutil.copytree(ROOT + "/vkresample_amd/csrc", tmp + "/vkresample_amd/csrc")
shutil.copytree(ROOT + "/include", tmp + "/include")
p = tmp + "/vkresample_amd/csrc/kernels_pow2.hpp"
s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)
rep("namespace fftup {\n\nconstexpr int ilog2c", '''namespace fftup {
__device__ unsigned long long g_dbg[4096];
__device__ __forceinline__ void mark(int id, int& n)
{
    if (blockIdx.x == 97 && threadIdx.x == DBG_TID) {
        if (id == 0) { n = 0; g_dbg[4000 + DBG_TID / 64] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
        if (n >= 0 && n < 4096) g_dbg[n] = ((unsigned long long)id << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);
        n++;
    }
}

constexpr int ilog2c''')
# marks inside reg_fft_pp: before scatter, after barrier, after gather
rep('''        reg_scatter<N, E, R, Ns, 1>(v, b, p, 0);
        __syncthreads();
        reg_gather<N, E, 1>(v, b, p, 0);
        reg_fft_pp<N, E, DIR, S + 1>(v, c, z, p, tws);''','''        mark(10 + S, dn);
        reg_scatter<N, E, R, Ns, 1>(v, b, p, 0);
        mark(20 + S, dn);
        __syncthreads();
        mark(30 + S, dn);
        reg_gather<N, E, 1>(v, b, p, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mark(40 + S, dn);
        reg_fft_pp<N, E, DIR, S + 1>(v, c, z, p, tws, dn);''')
rep('''        for (int s = 0; s < npairs; s++) {
            const int a = a0 + 2 * s;''','''        for (int s = 0; s < npairs; s++) {
            mark(s == 0 ? 0 : 1, dn);
            const int a = a0 + 2 * s;''')
rep('''            PL::fft(v, buf, (float2*)(smem + L::ZOFF), lt, tws);''','''            mark(2, dn);
            PL::fft(v, buf, (float2*)(smem + L::ZOFF), lt, tws, dn);
            mark(3, dn);''')
rep('''            __syncthreads();                                                        // L rows a, a+1 visible''','''            mark(4, dn);
            __syncthreads();                                                        // L rows a, a+1 visible
            mark(5, dn);''')
rep('''            // NBUF = 2 (in-place exchanges): the next transform's first scatter goes into the buffer the sharpen just read.''','''            mark(6, dn);
            // NBUF = 2 (in-place exchanges): the next transform's first scatter goes into the buffer the sharpen just read.''')
rep("__device__ __forceinline__ void reg_fft_pp(float2 (&v)[E], float2* __restrict__ c, float2* __restrict__ z, int p, const TwSet<N, E, 8>& tws)",
    "__device__ __forceinline__ void reg_fft_pp(float2 (&v)[E], float2* __restrict__ c, float2* __restrict__ z, int p, const TwSet<N, E, 8>& tws, int& dn)")
s = s.replace("int j, const Tw& w)\n    {", "int j, const Tw& w, int& dn)\n    {")
rep("        F::fft(v, buf, NBUF == 3 ? zbuf : buf, j, w);", "        F::fft(v, buf, NBUF == 3 ? zbuf : buf, j, w, dn);")
rep("reg_fft_pp<UW, 8, -1>(v, buf, zbuf, j, w.t);", "reg_fft_pp<UW, 8, -1>(v, buf, zbuf, j, w.t, dn);")
rep("    int lt = threadIdx.x;                       // (made opaque", "    int dn = -100000;\n    int lt = threadIdx.x;                       // (made opaque")
open(p, "w").write(s)
# the marks live in the launch unit (the one that includes kernels_pow2.hpp): it gets an exported dump routine, plan destruction calls it
p = tmp + "/vkresample_amd/csrc/fftup_launch.hip"
s = open(p).read() + '''
void fftup_dbg_dump()
{
    if (const char* f = getenv("FFTUP_DBG_OUT")) {
        static unsigned long long h[4096];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fftup::g_dbg), sizeof h) == hipSuccess) { FILE* o = fopen(f, "wb"); if (o) { fwrite(h, 1, sizeof h, o); fclose(o); } }
    }
}
'''
open(p, "w").write(s)
p = tmp + "/vkresample_amd/csrc/fftup_plan.hip"
s = open(p).read()
rep('''    if (P->stream) (void)hipStreamSynchronize(P->stream);
    for (size_t l = 1;''', '''    if (P->stream) (void)hipStreamSynchronize(P->stream);
    { void fftup_dbg_dump(); fftup_dbg_dump(); }
    for (size_t l = 1;''')
open(p, "w").write(s)
tid = sys.argv[1] if len(sys.argv) > 1 else "0"
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build_variant(ROOT + "/tools/scratch/lib_dbg%s.so" % tid, csrc=tmp + "/vkresample_amd/csrc", extra=["-DDBG_TID=" + tid])
print("built dbg", tid)
