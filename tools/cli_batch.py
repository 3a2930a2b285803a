"""End-to-end rate of the drop-in CLI's batched mode (PNG files in -> PNG files out):
`vkresample -ifolder .. -ofolder .. -numfiles N -numthreads T [-workqueue]` over T, with the wall clock around the process
(plan creation included, as a user sees it) and the CLI's own "Total time".  Per-file host work (PNG decode, PNG encode,
file I/O) is timed alone with one thread for scale.

    python tools/cli_batch.py [--files 128] [--width 2048 --height 1024] [--out gpurun_out/cli_batch.txt]
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "vkresample_amd", "vkresample")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=128)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--threads", default="1,2,4,8,16,32,64")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from PIL import Image
    from vkresample_amd import synth

    base = tempfile.mkdtemp(prefix="fftup_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    inp, outp = os.path.join(base, "in"), os.path.join(base, "out")
    os.makedirs(inp)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    try:
        t0 = time.perf_counter()
        for k in range(a.distinct):                       # "N": smooth structure + noise of sigma 4 codes (compresses like a photograph)
            Image.fromarray(synth.frame(k, a.width, a.height, "N")).save(os.path.join(inp, "%06d.png" % (k + 1)), compress_level=3)
        for k in range(a.distinct, a.files):
            shutil.copy(os.path.join(inp, "%06d.png" % (k % a.distinct + 1)), os.path.join(inp, "%06d.png" % (k + 1)))
        in_mb = os.path.getsize(os.path.join(inp, "000001.png")) / 1e6
        say("# %d files of %dx%d (%d distinct, %.1f MB each as PNG), written in %.1f s; host: %d hardware threads"
            % (a.files, a.width, a.height, a.distinct, in_mb, time.perf_counter() - t0, os.cpu_count()))
        quota = "none"
        for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            if os.path.exists(p):
                quota = open(p).read().strip() + " (" + p + ")"
                break
        say("# CPUs this process may run on: %d; cgroup CPU quota: %s" % (len(os.sched_getaffinity(0)), quota))
        say("# wall = clock around the process (plan creation, page-locked buffers, %d files); CLI = its own 'Total time' line" % a.files)
        say("%-44s %8s %8s %10s %12s" % ("vkresample -u 2 ... -numfiles %d" % a.files, "wall s", "CLI s", "files/s", "ms per file"))

        def run(label, extra, threads, files=a.files):
            if os.path.isdir(outp):
                shutil.rmtree(outp)
            os.makedirs(outp)
            cmd = [CLI, "-ifolder", inp, "-ofolder", outp, "-numfiles", str(files), "-numthreads", str(threads), "-u", "2"] + extra
            t = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            wall = time.perf_counter() - t
            n_out = len(os.listdir(outp))
            tot = [l for l in r.stdout.splitlines() if l.startswith("Total time")]
            cli_s = float(tot[0].split()[2]) if tot else float("nan")
            ok = "" if (r.returncode == 0 and n_out == files) else "  FAILED rc %d, %d outputs" % (r.returncode, n_out)
            say("%-44s %8.2f %8.2f %10.1f %12.1f%s" % (label, wall, cli_s, files / wall, wall / files * 1e3, ok))
            st = [re.findall(r"([a-z+]+) (\d+)", l.split(":", 2)[2]) for l in r.stdout.splitlines() if l.startswith("Thread") and " files in " in l]
            if st:                                       # -stagetimes: mean over the threads, ms per thread
                keys = [k for k, _ in st[0]]
                mean = {k: np.mean([float(dict(x)[k]) for x in st]) for k in keys}
                nf = np.mean([float(l.split(":")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("Thread") and " files in " in l])
                say("    per thread (mean of %d, %.1f files each), ms: " % (len(st), nf) + ", ".join("%s %.0f" % (k, mean[k]) for k in keys))
            return wall

        threads = [int(t) for t in a.threads.split(",")]
        run("warm-up (first plan of the process image)", ["-p", "0", "-workqueue"], 4, files=8)
        for T in threads:
            run("-p 0 -numthreads %d -workqueue" % T, ["-p", "0", "-workqueue", "-stagetimes"], T)
        mid = threads[len(threads) // 2 + 1] if len(threads) > 2 else threads[-1]
        run("-p 0 -numthreads %d (the reference's stripe)" % mid, ["-p", "0"], mid)
        for T in (mid, threads[-1]):
            run("-p 2 -fuseu8 -fuseu8out -numthreads %d -workqueue" % T, ["-p", "2", "-fuseu8", "-fuseu8out", "-workqueue"], T)
        for T in threads:
            run("-p 0 -gpupng -numthreads %d -workqueue" % T, ["-p", "0", "-gpupng", "-workqueue", "-stagetimes"], T)
        out_mb = os.path.getsize(os.path.join(outp, "000001.png")) / 1e6
        say("# output PNG: %.1f MB per %dx%d file" % (out_mb, 2 * a.width, 2 * a.height))

        # one thread's host work per file, for scale: decode of the input, encode of the output (the CLI's own codec through the CLI:
        # -u 1 keeps the size, a second run on the upscaled files shows the encode of 4x the pixels)
        one = run("one thread, 8 files (decode + GPU + encode, serial)", ["-p", "0"], 1, files=8)
        say("# => %.0f ms of host work per file and thread; the GPU needs 0.06 ms (kernels) + 0.5 ms (PCIe both ways)" % (one / 8 * 1e3))
    finally:
        shutil.rmtree(base, ignore_errors=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
