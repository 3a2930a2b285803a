"""What a host thread of the batched CLI pays before its first file: plan creation, page-locked buffers, the first submit
(queue_init) -- alone and with T threads doing the same at once (the HIP runtime serialises allocations).
    python tools/plan_create_time.py [--threads 1,4,16,64]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkresample_amd as v
from vkresample_amd import synth


def one(W, H, res, i, do_submit):
    t0 = time.perf_counter()
    up = v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 2)
    t1 = time.perf_counter()
    pin = v.PinnedArray((2, H, W, 3))
    pout = v.PinnedArray((2, 2 * H, 2 * W, 3))
    t2 = time.perf_counter()
    t3 = t2
    if do_submit:
        pin.array[0] = 7
        tk = up.submit_rgb8(pin.array[0], pout.array[0])
        up.wait(tk)
        t3 = time.perf_counter()
    pin.close(); pout.close()
    t4 = time.perf_counter()
    up.close()
    t5 = time.perf_counter()
    res[i] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,4,16,64")
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    a = ap.parse_args()
    res = [None]
    one(a.width, a.height, res, 0, True)         # first plan of the process: context, code objects
    print("first plan of the process: create %.0f ms, pinned buffers %.0f ms, first frame %.0f ms, unpin %.0f ms, destroy %.0f ms" % tuple(x * 1e3 for x in res[0]))
    for T in [int(t) for t in a.threads.split(",")]:
        res = [None] * T
        th = [threading.Thread(target=one, args=(a.width, a.height, res, i, True)) for i in range(T)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        m = np.array(res).mean(axis=0) * 1e3
        print("%3d threads at once: wall %.0f ms; per thread (mean): create %.0f ms, pinned buffers %.0f ms, first frame %.0f ms, unpin %.0f ms, destroy %.0f ms"
              % (T, wall * 1e3, m[0], m[1], m[2], m[3], m[4]))


if __name__ == "__main__":
    main()
