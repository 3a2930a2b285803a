"""Device-side PNG encoding: what it costs on the GPU and what it saves on the link.
 (1) rocprofv3-free timing: frames through fftup_submit_rgb8 / fftup_wait and through fftup_submit_png / fftup_wait_png with
     2 .. 8 frames in flight from one host thread (page-locked buffers): frames/s, bytes over PCIe per frame;
 (2) one frame alone: latency of submit -> wait for both.
    python tools/png_device_time.py [--width 2048 --height 1024] [--frames 256]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkresample_amd as v
from vkresample_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--depths", default="1,2,4,8", help="frames in flight to try (1: a frame's kernels run alone -- isolated durations under rocprofv3)")
    a = ap.parse_args()
    W, H = a.width, a.height
    for depth in (int(x) for x in a.depths.split(",")):
        with v.Upscaler(W, H, 2.0, a.precision, 0.2, 0, 0, max(depth, 2)) as up:
            pin = v.PinnedArray((depth, H, W, 3))
            for k in range(depth):
                pin.array[k] = synth.frame(k, W, H, "N")
            pout = v.PinnedArray((depth, up.out_height, up.out_width, 3))
            ppng = v.PinnedArray((depth, (up.png_bound() + 63) // 64 * 64))      # (rows 16-byte aligned: the GPU writes into them)
            res = {}
            for mode in ("rgb8", "png"):
                def submit(k):
                    return up.submit_png(pin.array[k % depth], ppng.array[k % depth]) if mode == "png" else up.submit_rgb8(pin.array[k % depth], pout.array[k % depth])

                def wait(t, k):
                    return up.wait_png(t, ppng.array[k % depth]) if mode == "png" else (up.wait(t) or up.out_height * up.out_width * 3)
                for rep in range(2):                      # first pass warms (slot buffers, code objects)
                    tk = [submit(k) for k in range(depth)]
                    nbytes = 0
                    t0 = time.perf_counter()
                    for k in range(a.frames):
                        nbytes += wait(tk[k % depth], k)
                        tk[k % depth] = submit(k + depth)
                    for k in range(depth):
                        wait(tk[(a.frames + k) % depth], a.frames + k)
                    dt = time.perf_counter() - t0
                res[mode] = (a.frames / dt, nbytes / a.frames)
            print("%dx%d -p %d, %d frame(s) in flight: pixels %7.0f frames/s (%.1f MB down per frame) | PNG from the device %7.0f frames/s (%.1f MB)"
                  % (W, H, a.precision, depth, res["rgb8"][0], res["rgb8"][1] / 1e6, res["png"][0], res["png"][1] / 1e6), flush=True)
            pin.close(); pout.close(); ppng.close()


if __name__ == "__main__":
    main()
