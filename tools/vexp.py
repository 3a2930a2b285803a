#!/usr/bin/env python3
"""Timing-only variants of k_c2r_sharpen_v (results INVALID): source patches applied to a temporary copy of csrc/, the tree
is not touched.   python tools/vexp.py name...   ->  tools/scratch/lib_<name>.so      (tools/gpu_ab.sh takes them as variants)"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = {
    "base": [],
    # both 16-byte stores of a row as if lane l owned quads l and 64 + l of its wave's 512 pixels (contiguous kilobytes per instruction)
    "contigstore": [("char* dst = (char*)((float*)p.out + row_of) + (unsigned)lt * 32u;",
                     "char* dst = (char*)((float*)p.out + row_of) + ((unsigned)lt >> 6) * 2048u + ((unsigned)lt & 63u) * 16u;"),
                    ("__builtin_nontemporal_store(hi, (f4t*)(dst + 16));", "__builtin_nontemporal_store(hi, (f4t*)(dst + 1024));")],
    "nob": [("        lds_sync<true>();\n        const unsigned ard", "        const unsigned ard_unused"), ],
    "noc": [("    lane_transpose_hi3(v);\n    twiddle_powers", "    twiddle_powers")],
    "tk8": [],
    # k_c2r_sharpen_g without the loads of the mirror partners (their values replaced by the thread's own elements)
    "nomirror": [("kernels_pow2.hpp", """                in.a[m] = gload(ra, ko[m]); in.am[m] = gload(ra, kom[m]);
                in.b[m] = gload(rb, ko[m]); in.bm[m] = gload(rb, kom[m]);""", """                in.a[m] = gload(ra, ko[m]); in.am[m] = in.a[m];
                in.b[m] = gload(rb, ko[m]); in.bm[m] = in.b[m];""")],
    "nostore_unused": [("__builtin_nontemporal_store(lo, (f4t*)dst);", "if (p.uH < 0) __builtin_nontemporal_store(lo, (f4t*)dst);"),
                ("__builtin_nontemporal_store(hi, (f4t*)(dst + 16));", "if (p.uH < 0) __builtin_nontemporal_store(hi, (f4t*)(dst + 16));")],
}
def build(name):
    tmp = "/tmp/vexp_" + name
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/vkresample_amd")
    shutil.copytree(ROOT + "/vkresample_amd/csrc", tmp + "/vkresample_amd/csrc")
    shutil.copytree(ROOT + "/include", tmp + "/include")
    for pt in PATCHES[name]:
        f, a, b = pt if len(pt) == 3 else ("kernels_vpair.hpp",) + tuple(pt)
        p = tmp + "/vkresample_amd/csrc/" + f
        s = open(p).read()
        assert s.count(a) == 1, (name, a, s.count(a))
        open(p, "w").write(s.replace(a, b))
    if name == "tk8":        # spectrum tiles of 8 columns instead of 4 (64-byte row pieces)
        q = tmp + "/vkresample_amd/csrc/fftup.hip"
        t = open(q).read()
        assert t.count("static constexpr int TUNED_TK = 4;") == 1
        open(q, "w").write(t.replace("static constexpr int TUNED_TK = 4;", "static constexpr int TUNED_TK = 8;"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=on",
                           "-Wno-unused-function", "-shared", "-o", ROOT + "/tools/scratch/lib_%s.so" % name, tmp + "/vkresample_amd/csrc/fftup.hip"],
                          stderr=subprocess.DEVNULL)
    print("built", name)
if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(build, sys.argv[1:]))
