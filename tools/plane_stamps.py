#!/usr/bin/env python3
"""When do the workgroups of each colour plane begin and end inside ONE ordered frame?  Needs a library built with -DFFTUP_PLANE_STAMPS
(tools/build_variant.sh stamps -DFFTUP_PLANE_STAMPS; every workgroup stores two wall-clock stamps):
    FFTUP_LIBRARY=tools/scratch/lib_stamps.so python tools/plane_stamps.py [W H]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vkresample_amd as v
from vkresample_amd import synth, _lib
lib = _lib.load()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 1024)
buf = (C.c_ulonglong * (3 * 2048 * 3))()
with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 1) as up:
    up.upload_rgb8(synth.frame(0, W, H, "U"))
    up.execute(20)
    rows = []
    for it in range(15):
        up.execute(1)
        lib.fftup_debug_plane_stamps(buf)
        a = np.array(list(buf), dtype=np.float64).reshape(3, 2048, 3)
        t0 = a[0, :1536, 0].min()
        out = np.zeros((3, 3, 4))
        for k, n in enumerate((3 * (H // 2), 3 * ((W // 2 + 4) // 4), None)):
            wg = a[k][a[k][:, 1] > 0] if n is None else a[k][:n]
            for c in range(3):
                m = wg[wg[:, 2] == c]
                if len(m):
                    out[k, c] = ((m[:, 0].min() - t0) * 0.01, (np.median(m[:, 0]) - t0) * 0.01, (np.median(m[:, 1]) - t0) * 0.01, (m[:, 1].max() - t0) * 0.01)
        rows.append(out)
    med = np.median(np.array(rows), axis=0)
    print("per plane: first begin / median begin / median end / last end of its workgroups, us from the first row workgroup's begin (median of 15 frames)")
    for k, name in enumerate(("row pass", "column pass", "fused kernel")):
        print("%-13s" % name, "   ".join("plane %d: %5.1f /%5.1f /%5.1f /%5.1f" % ((c,) + tuple(med[k, c])) for c in range(3)))
