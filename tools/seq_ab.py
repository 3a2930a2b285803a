#!/usr/bin/env python3
"""A/B of library builds on the figures of the reference's own loop (performVulkanUpscale, VR:1260-1278):
   python tools/seq_ab.py [--configs 2,3,4] [--reps 2] base tools/scratch/lib_x.so "FFTUP_EXPERIMENT=... " ...
per variant and configuration: ordered iterations (fftup_execute(1000), FFTUP_FLAG_SEQUENTIAL_EXECUTE), overlapped frames
(fftup_execute_ring over a ring of 8), the kernels' isolated durations (HIP events).  One process per measurement."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import vkresample_amd as v
from vkresample_amd import synth
W, H, prec, flags = [int(x) for x in sys.argv[1:5]]
out = {}
with v.Upscaler(W, H, 2.0, prec, 0.2, 0, flags | v.FLAG_SEQUENTIAL_EXECUTE, 1) as up:
    up.upload_rgb8(synth.frame(0, W, H, "U"))
    up.execute(100)
    out["seq_us"] = round(sorted(up.execute(1000) for _ in range(5))[2] * 1e3, 2)
    out["seq_kernels_us"] = [round(x * 1e3, 1) for x in up.profile_kernels(50)]
with v.Upscaler(W, H, 2.0, prec, 0.2, 0, flags, 8) as up:
    for s in range(8):
        up.upload_rgb8(synth.frame(s, W, H, "U"), slot=s)
    up.execute_ring(256, 0)
    out["ring_us"] = round(sorted(up.execute_ring(2048, 0) / 2048 for _ in range(5))[2] * 1e3, 2)
    out["ring_kernels_us"] = [round(x * 1e3, 1) for x in up.profile_kernels(50)]
print(json.dumps(out))
""" % ROOT
CONFIGS = {"2": (2048, 1024, 0, 0), "3": (2048, 1024, 2, 2), "4": (1920, 1080, 0, 0)}
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2,3,4"); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
for rep in range(a.reps):
    for var in a.variants:
        env = dict(os.environ)
        if var.endswith(".so"):
            env["FFTUP_LIBRARY"] = os.path.join(ROOT, var)
        elif var != "base":
            for kv in var.split():
                k, _, val = kv.partition("=")
                env[k] = val
        for c in a.configs.split(","):
            r = subprocess.run([sys.executable, "-c", CHILD] + [str(x) for x in CONFIGS[c]], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print("[%s] config%s %s" % (var, c, line[0] if line else "FAILED: " + r.stderr[-300:]), flush=True)
