#!/usr/bin/env python3
"""Roofline fractions of the four-step launches (k_row4_a / k_row4_b: rows and columns too long for the LDS, the reference's
multi-upload plans, vkFFT.h:4773-4992): per kernel slot of the plan, algorithmic bytes (fftup_info.kernel_alg_bytes) over the
isolated duration (fftup_profile_kernels), against 8 TB/s.  A four-step slot is TWO launches that move the row four times
(read, write T, read T, write) where a one-launch transform moves it twice.   python tools/four_step_roofline.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkresample_amd as v
from vkresample_amd import synth

CASES = [  # W, H, u, precision: what runs in four steps
    (16384, 128, 2.0, 0),    # inverse rows of 32768 = 128 * 256 points (tiles of 4 / 4); forward rows of 16384 in one buffer
    (9216, 256, 2.0, 0),     # inverse rows of 18432 = 128 * 144
    (8748, 256, 2.0, 0),     # inverse rows of 17496 = 108 * 162: tiles of 2 / 4 (round 4: one sequence per workgroup)
    (17280, 128, 1.0, 0),    # -u 1: forward and inverse rows of 17280 = 120 * 144
    (256, 9800, 2.0, 0),     # columns: forward 9800 = 98 * 100 (tiles of 4 / 2), inverse 19600 = 140 * 140
    (256, 4900, 2.0, 0),     # columns: forward 4900 = 70 * 70 (tiles of 2 / 2), inverse 9800 = 98 * 100
    (4096, 128, 2.0, 1),     # -p 1: inverse rows of 8192 double2 points = 64 * 128
]
for (W, H, u, p) in CASES:
    with v.Upscaler(W, H, u, p) as up:
        up.upload_rgb8(synth.frame(1, W, H, "U"))
        up.execute(3)
        ms = up.profile_kernels(20)
        parts = []
        for name, t, b in zip(up.kernel_names, ms, up.kernel_alg_bytes):
            if name != "-" and t > 0:
                parts.append("%s %.1f us %.0f GB/s (%.3f)" % (name, t * 1e3, b / (t * 1e-3) / 1e9, b / (t * 1e-3) / 8e12))
        print("%5dx%-5d -u %g -p %d | %s\n      plan: %s" % (W, H, u, p, " | ".join(parts), up.description))
