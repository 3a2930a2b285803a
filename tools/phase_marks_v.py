#!/usr/bin/env python3
"""In-kernel s_memtime marks for k_c2r_sharpen_v (kernels_vpair.hpp), as tools/phase_marks.py does for k_c2r_sharpen_g:
builds an INSTRUMENTED copy of the library (the tree is not touched); thread <tid> of workgroup 97 records the phase
boundaries of every step; the library dumps them to $FFTUP_DBG_OUT at plan destruction; tools/phase_marks_read.py prints.

    python tools/phase_marks_v.py <tid> [-DNAME ...]   ->  tools/scratch/lib_vdbg<tid>.so"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tid = sys.argv[1] if len(sys.argv) > 1 else "0"
extra = sys.argv[2:]
tmp = "/tmp/vdbg_build_" + tid
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp + "/vkresample_amd")
shutil.copytree(ROOT + "/vkresample_amd/csrc", tmp + "/vkresample_amd/csrc")
shutil.copytree(ROOT + "/include", tmp + "/include")
p = tmp + "/vkresample_amd/csrc/kernels_vpair.hpp"
s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)
rep("namespace fftup {\n", '''namespace fftup {
__device__ unsigned long long g_dbg[4096];
__device__ __forceinline__ void mark(int id, int& n)
{
    if (blockIdx.x == 97 && threadIdx.x == DBG_TID) {
        if (id == 0) { n = 0; g_dbg[4000 + DBG_TID / 64] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
        if (n >= 0 && n < 4000) g_dbg[n] = ((unsigned long long)id << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);
        n++;
    }
}
''')
rep("const VTwid& tw, Hook hook)\n{", "const VTwid& tw, Hook hook, int& dn)\n{")
rep("        __syncthreads();\n        const unsigned ar = zbase", "        mark(20, dn);\n        __syncthreads();\n        mark(30, dn);\n        const unsigned ar = zbase")
rep("#pragma unroll\n    for (int m = 1; m < 8; m++) v[m] = cmul_tw_s(v[m], tw.a[m - 1]);", "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); mark(40, dn);\n#pragma unroll\n    for (int m = 1; m < 8; m++) v[m] = cmul_tw_s(v[m], tw.a[m - 1]);")
rep("        lds_sync<true>();\n        const unsigned ard", "        mark(21, dn);\n        lds_sync<true>();\n        const unsigned ard")
rep("    twiddle_powers<8>(v, tw.b);", "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); mark(41, dn);\n    twiddle_powers<8>(v, tw.b);")
rep("    hook(2);\n    lane_transpose_hi3(v);\n    hook(3);\n", "    mark(12, dn);\n    hook(2);\n    lane_transpose_hi3(v);\n    hook(3);\n    mark(13, dn);\n")
rep("        for (int s = 0; s < npairs; s++) {\n            const int a = a0 + 2 * s;", "        for (int s = 0; s < npairs; s++) {\n            mark(s == 0 ? 0 : 1, dn);\n            const int a = a0 + 2 * s;")
rep("            vfft4096(v, zb, lt, tws, issue_store);", "            mark(2, dn);\n            vfft4096(v, zb, lt, tws, issue_store, dn);\n            mark(3, dn);")
rep("            __syncthreads();                                                        // L pairs of rows a, a+1 visible", "            mark(4, dn);\n            __syncthreads();\n            mark(5, dn);")
rep("            const float la0 = (float)la.x, la1 = (float)la.y;", "            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); mark(9, dn);\n            const float la0 = (float)la.x, la1 = (float)la.y;")
rep("            for (int i = 0; i < 10; i++) P0[i] = P1[i];\n", "            for (int i = 0; i < 10; i++) P0[i] = P1[i];\n            mark(6, dn);\n")
rep("    int lt = threadIdx.x;\n    const int uH = p.uH;", "    int dn = -100000;\n    int lt = threadIdx.x;\n    const int uH = p.uH;")
open(p, "w").write(s)
p = tmp + "/vkresample_amd/csrc/fftup.hip"
s = open(p).read()
rep('''void fftup_plan_destroy(fftup_plan* P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    if (P->stream) (void)hipStreamSynchronize(P->stream);''','''void fftup_plan_destroy(fftup_plan* P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    if (P->stream) (void)hipStreamSynchronize(P->stream);
    if (const char* f = getenv("FFTUP_DBG_OUT")) {
        static unsigned long long h[4096];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fftup::g_dbg), sizeof h) == hipSuccess) { FILE* o = fopen(f, "wb"); if (o) { fwrite(h, 1, sizeof h, o); fclose(o); } }
    }''')
open(p, "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=on",
                       "-Wno-unused-function", "-DDBG_TID=" + tid] + extra + ["-shared", "-o", ROOT + "/tools/scratch/lib_vdbg%s.so" % tid,
                       tmp + "/vkresample_amd/csrc/fftup.hip"], stderr=subprocess.DEVNULL)
print("built vdbg", tid)
