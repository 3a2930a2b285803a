#!/usr/bin/env python3
"""Host-streamed frames through ONE plan from several host threads (fftup_submit_* / fftup_wait_* are thread-safe): pixels back
(fftup_submit_rgb8) against finished PNG files back (fftup_submit_png, encoded on the device, delivered by the device into the page-
locked buffer).  One thread is bound by its own launches and waits; with two or more the pixel path sits on the PCIe link (25.2 MB per
4096x2048 frame) while the PNG path moves 14.3 MB.      python tools/png_threads.py [--threads 1,2,4] [--frames 256]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkresample_amd as v
from vkresample_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--threads", default="1,2,4")
ap.add_argument("--frames", type=int, default=256, help="frames per thread")
ap.add_argument("--width", type=int, default=2048)
ap.add_argument("--height", type=int, default=1024)
ap.add_argument("--depth", type=int, default=2, help="frames in flight per thread")
a = ap.parse_args()
W, H = a.width, a.height
for T in (int(x) for x in a.threads.split(",")):
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, max(2, T * a.depth)) as up:
        res = {}
        for mode in ("rgb8", "png"):
            bufs = []
            for t in range(T):
                pin = v.PinnedArray((a.depth, H, W, 3))
                for k in range(a.depth):
                    pin.array[k] = synth.frame(t * a.depth + k, W, H, "N")
                pout = v.PinnedArray((a.depth, (up.png_bound() + 63) // 64 * 64)) if mode == "png" else v.PinnedArray((a.depth, up.out_height, up.out_width, 3))
                bufs.append((pin, pout))
            nbytes = [0] * T

            def worker(t):
                pin, pout = bufs[t]
                sub = (lambda k: up.submit_png(pin.array[k % a.depth], pout.array[k % a.depth])) if mode == "png" else \
                      (lambda k: up.submit_rgb8(pin.array[k % a.depth], pout.array[k % a.depth]))
                wait = (lambda tk, k: up.wait_png(tk, pout.array[k % a.depth])) if mode == "png" else (lambda tk, k: up.wait(tk) or pout.array[0].size)
                tk = [sub(k) for k in range(a.depth)]
                for k in range(a.frames):
                    nbytes[t] += wait(tk[k % a.depth], k)
                    tk[k % a.depth] = sub(k + a.depth)
                for k in range(a.depth):
                    wait(tk[(a.frames + k) % a.depth], a.frames + k)
            for rep in range(2):                      # (first pass warms: slot buffers, code objects)
                nbytes = [0] * T
                th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
                t0 = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                dt = time.perf_counter() - t0
            res[mode] = (T * a.frames / dt, sum(nbytes) / (T * a.frames))
            for pin, pout in bufs:
                pin.close()
                pout.close()
        print("%dx%d, %d host thread(s) x %d frames in flight: pixels %6.0f frames/s (%.1f MB down per frame, %.1f GB/s) | PNG from the device %6.0f frames/s (%.1f MB, %.1f GB/s)"
              % (W, H, T, a.depth, res["rgb8"][0], res["rgb8"][1] / 1e6, res["rgb8"][0] * res["rgb8"][1] / 1e9, res["png"][0], res["png"][1] / 1e6,
                 res["png"][0] * res["png"][1] / 1e9), flush=True)
