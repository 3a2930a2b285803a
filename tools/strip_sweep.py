#!/usr/bin/env python3
"""Strip geometry of the fused C2R+sharpen kernel, measured the three ways a plan runs: frames overlapping on a ring of slots
(fftup_execute_ring), fftup_execute(1000) of a plan without a ring, iterations overlapped (extension) and in order, and the kernel alone (fftup_profile_kernels).
Needs the test build of the library (FFTUP_EXPERIMENT knobs g_per_cu / pairs_per_strip).
    python tools/strip_sweep.py W H precision flags "knob=value" "knob=value;knob=value" ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FFTUP_LIBRARY"] = os.path.join(ROOT, "vkresample_amd", "libfftup_knobs.so")
import vkresample_amd as v
from vkresample_amd import synth

W, H, p, flags = (int(x) for x in sys.argv[1:5])
for knobs in ["", *sys.argv[5:]]:
    os.environ["FFTUP_EXPERIMENT"] = knobs
    with v.Upscaler(W, H, 2.0, p, 0.2, 0, flags, 8) as up:
        for s in range(8):
            up.upload_rgb8(synth.frame(s, W, H, "U"), slot=s)
        up.execute_ring(512, 0)
        ring = sorted(up.execute_ring(2048, 0) / 2048 for _ in range(7))[3]
        iso = up.profile_kernels(30)
    with v.Upscaler(W, H, 2.0, p, 0.2, 0, flags | v.FLAG_OVERLAP_ITERATIONS, 1) as up:
        up.upload_rgb8(synth.frame(0, W, H, "U"))
        up.execute(200)
        n1000 = sorted(up.execute(1000) for _ in range(5))[2]
    with v.Upscaler(W, H, 2.0, p, 0.2, 0, flags, 1) as up:
        up.upload_rgb8(synth.frame(0, W, H, "U"))
        up.execute(200)
        seq = sorted(up.execute(1000) for _ in range(5))[2]
        iso_seq = up.profile_kernels(30)
    print("%-40s ring %.2f us | overlapped iterations %.2f us | ordered iterations %.2f us | fused kernel alone %.2f us (sequential plan: %.2f)"
          % (knobs or "(default)", ring * 1e3, n1000 * 1e3, seq * 1e3, iso[2] * 1e3, iso_seq[2] * 1e3), flush=True)
