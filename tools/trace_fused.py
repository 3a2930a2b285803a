"""Phase timeline of the fused C2R+sharpen kernel (needs tools/scratch/libfftup_trace.so built with -DFFTUP_TRACE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vkresample_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "libfftup_trace.so")
import vkresample_amd as v
from vkresample_amd import synth
up = v.Upscaler(2048, 1024, 2.0, 0)
up.upload_rgb8(synth.frame(0, 2048, 1024))
up.execute(5)
lib = _lib.load()
buf = np.zeros(96 * 64, dtype=np.uint64)
lib.fftup_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print("rc", lib.fftup_debug_trace(up._h, buf.ctypes.data, buf.size))
names = {0: "start", 1: "iter", 2: "loads-in", 3: "fft-done", 4: "Lrows+bar", 5: "sharpen-done", 6: "end-bar"}
for wg in range(0, 8):
    row = buf[wg * 96:(wg + 1) * 96]
    row = row[row != 0]
    slots = (row >> np.uint64(56)).astype(int)
    t = (row & np.uint64((1 << 56) - 1)).astype(np.int64)
    t = t - t[0]
    print("WG %d:" % (wg * 64), " ".join("%s@%d" % (names[s], x) for s, x in zip(slots, t)))
    d = np.diff(t)
    agg = {}
    for s, x in zip(slots[1:], d):
        agg.setdefault(names[s], []).append(int(x))
    print("    mean delta to reach: " + ", ".join("%s=%d" % (k, np.mean(vv)) for k, vv in agg.items()))
