#!/usr/bin/env python3
"""fftup_execute(n = 1000) per frame with recorded frames (hipGraph replay, default) and with eager launches
(FFTUP_EXPERIMENT graphs=0), for frames small enough to be launch-bound and for the BASELINE sizes:  python tools/gpu_graphs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vkresample_amd as v
from vkresample_amd import synth

for (W, H, p, flags) in [(256, 128, 0, 0), (640, 480, 0, 0), (1024, 512, 0, 0), (1280, 720, 0, 0), (1920, 1080, 0, 0), (2048, 1024, 0, 0), (2048, 1024, 2, 2)]:
    line = "%4dx%-4d p%d flags %d:" % (W, H, p, flags)
    for mode in ("0", "1"):
        os.environ["FFTUP_EXPERIMENT"] = "graphs=" + mode
        with v.Upscaler(W, H, 2.0, p, 0.2, 0, flags) as up:
            up.upload_rgb8(synth.frame(1, W, H))
            up.execute(50)
            t = sorted(up.execute(1000) for _ in range(5))[2]
            with v.Upscaler(W, H, 2.0, p, 0.2, 0, flags, ring=8) as ur:
                for s in range(8):
                    ur.upload_rgb8(synth.frame(s, W, H), slot=s)
                ur.execute_ring(64, 0)
                tr = sorted(ur.execute_ring(1024, 0) / 1024 for _ in range(5))[2]
        line += "   %s: execute(1000) %7.2f us/iter, ring %7.2f us/frame" % ("graphs" if mode == "1" else "eager ", t * 1e3, tr * 1e3)
    print(line, flush=True)
