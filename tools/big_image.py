"""One-off: an image beyond every one-launch limit -- 8192x5000 -> 16384x10000 (non-R2C path, rows of 16384 points in place, columns of
10000 points in four steps) and 10240x4096 -> 20480x8192 (rows in four steps): runs, preserves the plane means (DC), is deterministic;
time per frame.  python tools/big_image.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vkresample_amd as v
from vkresample_amd import synth
for (W, H) in ((8192, 5000), (10240, 4096)):
    rgb = synth.frame(5, W, H, "N")
    os.environ["FFTUP_STREAMS"] = "1"
    t0 = time.time()
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, v.FLAG_FUSE_U8_LOAD) as up:
        t1 = time.time()
        up.upload_rgb8(rgb)
        ms = up.execute(1)
        ms = min(up.execute(2) for _ in range(2))
        pre = up.download_presharpen()
        a = up.download_rgb8().copy()
        up.execute(1)
        b = up.download_rgb8()
        desc = up.description
        dev_gb = up.device_bytes / 1e9
    m_in = (rgb.astype(np.float64) / 255).mean(axis=(0, 1))
    m_pre = pre.astype(np.float64).mean(axis=(1, 2)) * 4
    print("%dx%d -> %dx%d: %.1f ms per frame, plan %.1f s, %.1f GB on the device; plane means in %s pre-sharpen %s (max diff %.2e); deterministic: %s" %
          (W, H, 2 * W, 2 * H, ms, t1 - t0, dev_gb, np.round(m_in, 6), np.round(m_pre, 6), np.abs(m_in - m_pre).max(), bool(np.array_equal(a, b))))
    print("   ", desc)
    del pre, a, b
