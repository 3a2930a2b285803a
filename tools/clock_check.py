#!/usr/bin/env python3
"""rocprofv3 --kernel-trace and fftup_profile_kernels (HIP events) disagreed by 15 % on the 1080p fused kernel.
Runs ONE kind of launch sequence of ONE plan, so that a rocprofv3 --kernel-trace --stats of this process holds nothing else:
  --mode profile : fftup_profile_kernels(n)  (kernels one after the other, HIP event pair around each, net of an empty pair)
  --mode ring    : fftup_execute_ring        (frames overlapping on the plan's streams)
and prints what the events say.  tools/gpu_clock_check.sh puts the trace's averages beside it."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--precision", type=int, default=0); ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--streams", type=int, default=3); ap.add_argument("--ring", type=int, default=8)
ap.add_argument("--mode", default="profile"); ap.add_argument("--n", type=int, default=200)
a = ap.parse_args()
os.environ["FFTUP_STREAMS"] = str(a.streams)
import vkresample_amd as v
from vkresample_amd import synth
with v.Upscaler(a.width, a.height, 2.0, a.precision, 0.2, 0, a.flags, a.ring) as up:
    for s in range(a.ring):
        up.upload_rgb8(synth.frame(s, a.width, a.height, "U"), slot=s)
    if a.mode == "profile":
        iso = up.profile_kernels(a.n)
        print(json.dumps({"mode": "profile_kernels", "streams": a.streams, "ring": a.ring, "plan": up.description,
                          "events_us": {k: round(x * 1e3, 2) for k, x in zip(up.kernel_names, iso)}}))
    else:
        up.execute_ring(64, 0)
        ms = up.execute_ring(a.n * 4, 0) / (a.n * 4)
        print(json.dumps({"mode": "execute_ring", "streams": a.streams, "ring": a.ring, "us_per_frame": round(ms * 1e3, 2)}))
