"""Builds the built-in wisdom table of csrc/jit.hpp: runs the plan-time tuner (FFTUP_FLAG_TUNE_PLAN) over common widths
and upscale factors on this device and prints the decisions that differ from the chooser's pick.
  python tools/gpu_wisdom.py > gpurun_out/<tag>/wisdom_builtin.txt"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cache = tempfile.mkdtemp(prefix="fftup_wisdom_")
os.environ["FFTUP_CACHE_DIR"] = cache
os.environ["FFTUP_JIT_VERBOSE"] = "1"
import vkresample_amd as v  # noqa: E402

SIZES = [(640, 480), (720, 480), (720, 576), (800, 600), (960, 540), (1000, 1000), (1024, 768), (1152, 864), (1280, 960), (1440, 900),
         (1600, 900), (1680, 1050), (1920, 1200), (2000, 2000), (2048, 1536), (2560, 1440), (3000, 2000), (3200, 1800), (3440, 1440),
         (3840, 2160), (4096, 2160), (1280, 720), (1920, 1080), (2048, 1024), (512, 512), (896, 504), (1536, 864), (1792, 1008)]
FACTORS = [2.0, 1.5, 2.5, 3.0, 4.0, 5.0]


def smooth(n):
    for q in (2, 3, 5, 7):
        while n % q == 0:
            n //= q
    return n == 1


done = set()
for (W, H) in SIZES:
    for u in FACTORS:
        uW, uH = u * W, u * H
        if uW != int(uW) or uH != int(uH) or uW > 8192 or int(uW) % 4 or int(uH) % 2 or not smooth(int(uW)) or not smooth(int(uH)):
            continue
        if (int(uW), u) in done:
            continue
        try:
            with v.Upscaler(W, H, u, int(os.environ.get("WISDOM_PRECISION", "0")), 0.2, 0, v.FLAG_TUNE_PLAN, int(os.environ.get("WISDOM_RING", "4"))) as up:
                if up.specialised_at_plan_time:
                    done.add((int(uW), u))
        except Exception as e:       # noqa: BLE001
            print("#", W, H, u, "failed:", e, file=sys.stderr)
print(open(os.path.join(cache, "wisdom.txt")).read() if os.path.exists(os.path.join(cache, "wisdom.txt")) else "")
