#!/usr/bin/env python3
"""bench.py -- headline benchmark of the upscale hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one batch of --frames-per-step synthetic 2048x1024 RGB frames pushed through the whole path (R2C rows ->
column FFT / zero-pad / iFFT -> C2R rows -> sharpen) on each GPU, inputs resident in HBM (a ring of --ring distinct
frames per GPU).  The default step is 1024 frames, i.e. the reference's `-n 1000` run (BASELINE config 2) per step.
Frames are independent, so N GPUs = N independent shards, no data-path collective ("weak" scaling: per-GPU work
fixed).  The K-step timed region (barrier + synchronize on both sides, MAX over ranks) is repeated --repeats times
and the MEDIAN region gives `value`.  Rank 0 prints ONE JSON line.

At N = 1 the same invocation also times, briefly, the other single-GPU BASELINE configurations and the reference's own
`-n 1000` figure (`others`: config3 = -p 2 with the fused uint8 load, config4 = 1920x1080, execute_n1000 =
fftup_execute(plan, 1000) on a plan without a ring of slots, i.e. performVulkanUpscale(..., 1000), VkResample.cpp:1260-1278).
`--preset config5` (BASELINE's 512-frame batch over 8 GPUs) carries the job accounting: every rank processes the frames
shard.frames_for_rank() gives it (the reference's -numthreads stripe, VkResample.cpp:1622-1629), each frame distinct and
resident, and one RCCL all-reduce of {frames, checksum of the outputs, max time} closes the job (`job` in the JSON line).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PRESETS = {
    # BASELINE.json configs[1..4]
    "config2": dict(width=2048, height=1024, precision=0, fuse_u8=False),
    "config3": dict(width=2048, height=1024, precision=2, fuse_u8=True),
    "config4": dict(width=1920, height=1080, precision=0, fuse_u8=False),
    "720p": dict(width=1280, height=720, precision=0, fuse_u8=False),          # not a BASELINE config: second mixed-radix plan
    # 512 synthetic 2048x1024 frames, -u 2 -p 2, sharded over 8 GPUs: 64 frames per rank and step
    # (64 distinct resident frames per rank: frame f * world + rank of the job goes to slot f)
    "config5": dict(width=2048, height=1024, precision=2, fuse_u8=True, frames_per_step=64, ring=64, job=True),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed region is run this many times; the median is reported")
    ap.add_argument("--frames-per-step", type=int, default=1024)
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None, help="a BASELINE.json configuration (default: config2 = headline)")
    ap.add_argument("--ring", type=int, default=8,
                    help="distinct resident input/output frame slots per GPU (8 x 131 MB > the 256 MB Infinity Cache)")
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--upscale", type=float, default=2.0)
    ap.add_argument("--precision", type=int, default=0, help="0 = fp32 (headline), 1 = fp64, 2 = fp16 memory")
    ap.add_argument("--fuse-u8", action="store_true", help="row kernel reads uint8 RGB directly")
    ap.add_argument("--fuse-u8-store", action="store_true", help="the fused C2R+sharpen kernel stores 8-bit RGB (FFTUP_FLAG_FUSE_U8_STORE): 8-bit in, 8-bit out")
    ap.add_argument("--generic", action="store_true", help="size-generic kernels (FFTUP_FLAG_GENERIC_KERNELS): no tuned, no plan-time specialised plan")
    ap.add_argument("--tune", action="store_true", help="FFTUP_FLAG_TUNE_PLAN: plan-time tuner for sizes specialised at plan time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rccl-check", action="store_true", help="skip the one-rank RCCL bring-up of N = 1 lines (`rccl_selfcheck`)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over this very configuration that measure "
                         "`roofline.traffic` live at N = 1; the committed figures of profiles/hbm_traffic.json are reported instead")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the bounded CPU-baseline sample (~11 s on 128 threads)")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other BASELINE configurations (`others`)")
    ap.add_argument("--job", action="store_true",
                    help="job accounting: rank r processes frames shard.frames_for_rank(frames_per_step * world, world, r), all distinct "
                         "(needs --ring >= --frames-per-step), output checksums reduced over the ranks (implied by --preset config5)")
    ap.add_argument("--queue", action="store_true",
                    help="with --job: dynamic sharding -- ranks claim chunks of frames from one shared counter (shard.FrameQueue, "
                         "torch.distributed's store) instead of the static stripe; every rank keeps all frames of a step resident")
    ap.add_argument("--queue-chunk", type=int, default=8, help="frames per claim of --queue")
    ap.add_argument("--profile-iters", type=int, default=50)
    ap.add_argument("--event-stride", type=int, default=32, help="kernel-timing events on every n-th frame of the first timed region")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams (lanes) consecutive frames alternate on; 1 = strictly sequential kernels")
    ap.add_argument("--host-streamed", action="store_true",
                    help="NOT the headline: frames start and end in pinned host memory (fftup_submit_rgb8 queue, "
                         "uint8 RGB over PCIe both ways); the line is marked pcie_inclusive")
    ap.add_argument("--png", action="store_true",
                    help="with --host-streamed: the frames come back as finished PNG files encoded on the device (fftup_submit_png: "
                         "row filters, Huffman-only deflate, checksums), the GPU writes each stream into the pinned buffer itself")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"),
                    help="per-launch HBM bytes of the kernels from committed rocprofv3 --pmc runs, keyed by configuration")
    a = ap.parse_args()
    if a.preset:
        for k, v in PRESETS[a.preset].items():
            setattr(a, k, v)
    if a.queue and not a.job:
        ap.error("--queue is a mode of --job")
    if a.queue:
        a.ring = a.frames_per_step * int(os.environ.get("WORLD_SIZE", "1"))       # any rank may be handed any frame of the step
    if a.job and a.ring < a.frames_per_step:
        ap.error("--job needs --ring >= --frames-per-step (every frame of a step resident and distinct)")
    return a


def config_key(args):
    """key of profiles/hbm_traffic.json: size, precision and input type"""
    return "%dx%d_p%d_%s" % (args.width, args.height, args.precision, "u8" if (args.fuse_u8 and args.precision != 1) else "planar")


def cpu_baseline(args):
    """The oracle (CPU restatement, `kind: port`) timed on this host, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    from vkresample_amd import synth
    frames = [synth.frame(k, args.width, args.height, "U") for k in range(args.cpu_frames)]
    O.upscale_rgb8(synth.frame(99, 64, 32), args.upscale, args.precision)     # load + warm the library
    t0 = time.perf_counter()
    for f in frames:
        O.upscale_rgb8(f, args.upscale, args.precision)
    dt = time.perf_counter() - t0
    return {"value": len(frames) / dt, "unit": "frames/s", "cores": O.num_threads(), "cpu_quota": O.cpu_quota(), "kind": "port",
            "sample": "%d synthetic %dx%d frames through oracle/fftup_oracle.c (the fp64 restatement of the reference's algorithm, "
                      "naive mixed-radix FFTs, OpenMP over rows/columns/planes), %.1f s" % (len(frames), args.width, args.height, dt)}


def reference_vulkan_baseline(args):
    """the reference itself on this GPU through Vulkan -- only possible when the box has a Vulkan loader
    and a prebuilt VkResample binary is supplied out of band (VKRESAMPLE_REF_BIN); never fabricated."""
    import ctypes.util
    import re
    import subprocess
    import tempfile
    ref_bin = os.environ.get("VKRESAMPLE_REF_BIN")
    if not ref_bin or not os.path.exists(ref_bin) or not ctypes.util.find_library("vulkan"):
        return {"available": False,
                "note": "reference Vulkan baseline unavailable (no libvulkan / VKRESAMPLE_REF_BIN on this box); the reference "
                        "publishes < 2 ms per 2048x1024->4096x2048 frame on a GTX 1660 Ti (README.md:12)"}
    from PIL import Image
    from vkresample_amd import synth
    with tempfile.TemporaryDirectory() as tmp:
        Image.fromarray(synth.frame(0, args.width, args.height, "U")).save(os.path.join(tmp, "in.png"))
        r = subprocess.run([ref_bin, "-i", "in.png", "-o", "out.png", "-u", str(args.upscale), "-p", str(args.precision), "-n", "1000"],
                           cwd=tmp, capture_output=True, text=True, timeout=300)
        m = re.search(r"Time: ([0-9.]+) ms", r.stdout)
        if not m:
            return {"available": False, "note": "reference binary ran but printed no time: " + r.stdout[-200:]}
        return {"available": True, "ms_per_frame": float(m.group(1)), "frames_per_s": 1e3 / float(m.group(1)), "command": "-u 2 -n 1000"}


FRAME_KERNEL_SOURCES = ("fft_engine.hpp", "kernels_dswap.hpp", "kernels_generic.hpp", "kernels_mixed.hpp", "kernels_pow2.hpp")


def kernel_sources_sha256():
    """fingerprint of the sources of the frame's kernels (the headers the library embeds for its plan-time compiler, __graft_entry__.
    KERNEL_HEADERS): committed counter / trace summaries carry the one they were measured on, so that a line can say whether its
    static figures belong to the kernels that just ran"""
    import hashlib
    h = hashlib.sha256()
    for f in FRAME_KERNEL_SOURCES:
        h.update(f.encode() + b"\0" + open(os.path.join(ROOT, "vkresample_amd", "csrc", f), "rb").read() + b"\0")
    return h.hexdigest()


def rocprof_kernel_us(key, short_name):
    """average duration (us) of a kernel from the committed `rocprofv3 --kernel-trace --stats` summary of this configuration
    (profiles/kernel_stats_index.json, written by tools/index_kernel_stats.py): (us, file, measured on these very kernel sources?)"""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "kernel_stats_index.json"))).get(key)
        if not e or short_name not in e["kernels_avg_ns"]:
            return None, None, None
        return e["kernels_avg_ns"][short_name] * 1e-3, e["file"], e.get("kernel_sources_sha256") == kernel_sources_sha256()
    except Exception:
        return None, None, None


def frame_traffic(traffic_json, key, kernel_names):
    """measured HBM bytes per launch of a configuration's kernels (committed rocprofv3 --pmc run, profiles/hbm_traffic.json):
    (per-kernel dict, frame total, source text) or (None, None, None)"""
    try:
        tj = json.load(open(traffic_json)).get(key)
        if not tj:
            return None, None, None
        per = {n: tj.get(n, {}).get("hbm_bytes_per_launch") for n in kernel_names if n != "-"}
        total = sum(per.values()) if all(x is not None for x in per.values()) else None
        fresh = tj.get("_kernel_sources_sha256") == kernel_sources_sha256()
        return per, total, "static %s [%s] (%s)%s" % (os.path.relpath(traffic_json, ROOT), key, tj.get("_source", "?"),
                                                       "" if fresh else " STALE: measured on other kernel sources")
    except Exception:
        return None, None, None


def traffic_is_current(traffic_json, key):
    try:
        return json.load(open(traffic_json))[key].get("_kernel_sources_sha256") == kernel_sources_sha256()
    except Exception:
        return None


def valu_busy(traffic_json, key, kernel_names, ms_per_frame, sclk_mhz, n_cu):
    """Share of the frame's vector-ALU issue slots in use: wave-level vector instructions of the frame's launches (SQ_INSTS_VALU of
    the committed PMC run) x 4 cycles each, over 4 SIMDs per compute unit x shader clock x frame time.  With `real_traffic_frac`
    it says what the frame is NOT bound by; the power figure says what it is.  None where a number is missing."""
    try:
        tj = json.load(open(traffic_json)).get(key) or {}
        insts = [tj.get(n, {}).get("valu_insts_per_launch") for n in kernel_names if n != "-"]
        if not insts or any(x is None for x in insts) or not sclk_mhz or not n_cu:
            return None, None
        total = float(sum(insts))
        return total, total * 4.0 / (4.0 * n_cu * sclk_mhz * 1e6 * ms_per_frame * 1e-3)
    except Exception:
        return None, None


def other_configs(v, synth, dev, traffic_json, n_cu, ring=8, live=True):
    """Short runs of the other single-GPU BASELINE configurations on the same GPU, same method as the headline (ring of
    resident frames, three streams, HIP events per kernel), and the reference's -n 1000 figure on ring-less plans.  Every entry
    carries, beside the B_alg fractions, the numbers that can still move: the fraction by MEASURED HBM bytes
    (`real_traffic_frac`), by the bytes that must move at all (`b_min_frac`: input + output), and the energy per frame."""
    out = {}
    for name in ("config3", "config4", "config3_u8_store"):
        c = PRESETS[name.replace("_u8_store", "")]
        flags = (v.FLAG_FUSE_U8_LOAD if c["fuse_u8"] else 0) | (v.FLAG_FUSE_U8_STORE if name.endswith("_u8_store") else 0)
        with v.Upscaler(c["width"], c["height"], 2.0, c["precision"], 0.2, dev, flags, ring) as up:
            for s in range(ring):
                up.upload_rgb8(synth.frame(s, c["width"], c["height"], "U"), slot=s)
            up.execute_ring(256, 0)
            # (~1.5 s of frames: the driver's power figure is a moving average; only the second half of the samples is used)
            power = PowerSampler(v.device_pci_bus_id(dev), period=0.02).start()
            t = sorted(up.execute_ring(2048, 0) / 2048 for _ in range(11))[5]         # device ms per frame, median of eleven
            power.samples = power.samples[len(power.samples) // 2:] if len(power.samples) > 8 else power.samples
            pw = power.stop()
            iso = up.profile_kernels(30)
            dom = max(range(len(iso)), key=lambda i: iso[i])
            esz = {0: 4, 1: 8, 2: 2}[c["precision"]]
            b_min = 3.0 * (c["width"] * c["height"] * (1 if c["fuse_u8"] else esz) + up.out_width * up.out_height * (1 if up.u8_store else esz))
            key = "%dx%d_p%d_%s%s" % (c["width"], c["height"], c["precision"], "u8" if c["fuse_u8"] else "planar", "_u8out" if up.u8_store else "")
            per, frame_hbm, tsrc = frame_traffic(traffic_json, key, up.kernel_names)
            if live:                                  # measured now, like the headline's (live_traffic); the committed figures otherwise
                lv, lsrc = live_traffic(["--width", str(c["width"]), "--height", str(c["height"]), "--precision", str(c["precision"]), "--ring", str(ring)]
                                        + (["--fuse-u8"] if c["fuse_u8"] else []) + (["--fuse-u8-store"] if up.u8_store else []))
                if lv and all(n in lv for n in up.kernel_names if n != "-"):
                    per = {n: lv[n]["hbm_bytes_per_launch"] for n in up.kernel_names if n != "-"}
                    frame_hbm, tsrc = sum(per.values()), lsrc
            out[name] = {"workload": "%dx%d -u 2 -p %d%s%s" % (c["width"], c["height"], c["precision"], ", fused uint8 load" if c["fuse_u8"] else "",
                                                                  ", fused 8-bit RGB store (8-bit in, 8-bit out)" if up.u8_store else ""),
                         "ms_per_frame": t, "frames_per_s": 1e3 / t,
                         "frame_frac": up.alg_bytes_per_frame / (t * 1e-3) / 8e12,
                         "kernel_ms": dict(zip(up.kernel_names, iso)),
                         "kernel_frac": up.kernel_alg_bytes[dom] / (max(iso[dom], 1e-6) * 1e-3) / 8e12,
                         "kernel_frac_real_bytes": up.kernel_min_bytes[dom] / (max(iso[dom], 1e-6) * 1e-3) / 8e12,
                         "B_min": b_min, "b_min_frac": b_min / (t * 1e-3) / 8e12,
                         "frame_hbm_bytes_measured": frame_hbm, "traffic_source": tsrc,
                         "real_traffic_frac": (frame_hbm / (t * 1e-3) / 8e12) if frame_hbm else None,
                         "kernel_hbm_bytes_measured": per,
                         "frame_valu_insts": valu_busy(traffic_json, key, up.kernel_names, t, pw.get("sclk_mhz_median"), n_cu)[0],
                         "valu_busy_frac": valu_busy(traffic_json, key, up.kernel_names, t, pw.get("sclk_mhz_median"), n_cu)[1],
                         "socket_power_w_median": pw.get("socket_power_w_median"), "sclk_mhz_median": pw.get("sclk_mhz_median"),
                         "energy_mj_per_frame": pw["socket_power_w_median"] * t if pw.get("socket_power_w_median") else None,
                         "plan": up.description}
    # the reference's own figure, performVulkanUpscale(.., 1000) (VkResample.cpp:1260-1278), on the CLI's single-image plan (no
    # ring): 1000 identical iterations IN ORDER on one stream -- the reference's one command buffer puts a pipeline barrier behind
    # every stage (VR:1217, vkFFT.h:7678), iteration i+1 cannot start before iteration i's sharpen pass has ended -- `ms_per_iter`
    # (= `sequential_ms_per_iter`, the name of earlier rounds), and the EXTENSION FFTUP_FLAG_OVERLAP_ITERATIONS beside it
    # (`overlapped_*`: the iterations alternate on the plan's streams; a throughput figure the reference has no counterpart of)
    n1000 = {}
    for name in ("config2", "config3", "config4"):
        c = PRESETS[name]
        flags = v.FLAG_FUSE_U8_LOAD if c["fuse_u8"] else 0
        e = {}
        for mode, fl in (("", 0), ("overlapped_", v.FLAG_OVERLAP_ITERATIONS)):
            with v.Upscaler(c["width"], c["height"], 2.0, c["precision"], 0.2, dev, flags | fl, 1) as up:
                up.upload_rgb8(synth.frame(0, c["width"], c["height"], "U"))
                up.execute(100)
                ms = sorted(up.execute(1000) for _ in range(3))[1]
                e[mode + "ms_per_iter"] = ms
                e[mode + "frame_frac"] = up.alg_bytes_per_frame / (ms * 1e-3) / 8e12
                if not mode:
                    e["kernel_ms"] = dict(zip(up.kernel_names, up.profile_kernels(30)))      # this plan's kernels, one at a time
        e["sequential_ms_per_iter"], e["sequential_frame_frac"] = e["ms_per_iter"], e["frame_frac"]
        n1000[name] = e
    # outside BASELINE's list, one line each (short runs, ring of 3, no counters): the reference's third precision (-p 1, SURVEY 8 f4:
    # size-generic kernels) and a size / factor the plan-time compiler serves (1080p -> 1440p, -u 4/3)
    import numpy as _np
    for name, (w, h, u, prec) in (("fp64", (2048, 1024, 2.0, 1)), ("fhd_to_qhd_u4_3", (1920, 1080, float(_np.float32(4.0 / 3.0)), 0))):
        with v.Upscaler(w, h, u, prec, 0.2, dev, 0, 3) as up:
            for s in range(3):
                up.upload_rgb8(synth.frame(s, w, h, "U"), slot=s)
            up.execute_ring(48, 0)
            t = sorted(up.execute_ring(192, 0) / 192 for _ in range(5))[2]
            out[name] = {"workload": "%dx%d -> %dx%d -p %d" % (w, h, up.out_width, up.out_height, prec), "ms_per_frame": t, "frames_per_s": 1e3 / t,
                         "frame_frac": up.alg_bytes_per_frame / (t * 1e-3) / 8e12, "kernel_ms": dict(zip(up.kernel_names, up.profile_kernels(10))),
                         "plan": up.description}
    out["execute_n1000"] = dict(n1000, note="fftup_execute(plan, 1000) on a plan without a ring = performVulkanUpscale(.., 1000), VkResample.cpp:1260-1278: "
                                            "iterations in order on one stream, like the reference's barriers; the CLI prints ms_per_iter as Time: (-n 1000); "
                                            "overlapped_* = FFTUP_FLAG_OVERLAP_ITERATIONS (the CLI's -overlap), an extension")
    return out


class PowerSampler:
    """Socket power and shader clock of one GPU sampled in a background thread while the timed regions run (sysfs hwmon files of
    the amdgpu driver; best effort: every field is None where the files are missing).  The overlapped frame of this workload
    runs INTO the board's power limit -- 1400 W, shader clock 2.05 instead of 2.4 GHz -- so the bench line says so itself."""

    def __init__(self, pci_bus_id, period=0.1):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.dir = None
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*")):
            # the card whose PCI address is the HIP device's (several boards may be present, one visible)
            addr = os.path.basename(os.path.realpath(os.path.join(c, "..", "..")))
            if addr.lower() == (pci_bus_id or "").lower() and any(os.path.exists(os.path.join(c, f)) for f in ("power1_average", "power1_input")):
                self.dir = c
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            pw = self._read("power1_average")
            if pw is None:
                pw = self._read("power1_input")
            self.samples.append((pw, self._read("freq1_input")))
            self._stop.wait(self.period)

    def start(self):
        if self.dir:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self.dir and self._thread.is_alive():
            self._thread.join(timeout=2)
        pw = [a * 1e-6 for a, _ in self.samples if a]
        fr = [b * 1e-6 for _, b in self.samples if b]
        cap = self._read("power1_cap") if self.dir else None
        med = lambda x: sorted(x)[len(x) // 2] if x else None
        return {"socket_power_w_median": med(pw), "socket_power_w_max": max(pw) if pw else None, "power_cap_w": cap * 1e-6 if cap else None,
                "sclk_mhz_median": med(fr), "sclk_mhz_min": min(fr) if fr else None, "samples": len(self.samples),
                "note": "sampled during the timed regions; at the cap the shader clock is throttled (DESIGN.md section 4)"}


# kernel name (as rocprofv3 prints it, template arguments cut) -> the plan's kernel slot, and the FETCH_SIZE correction of its access
# pattern: on gfx950 a 128-byte read request is tallied as 64 bytes, so fully coalesced streaming reads (the row and column kernels:
# whole rows / whole 64 KB tiles) report half their bytes; the two-launch C2R kernels read the blocked spectrum in 64-byte requests
# (factor 1); the fused kernel's requests are a mix -- calibrated on itself with the caches evicted (round 6): 593 538 requests for
# 54.6 MB of distinct bytes, factor 1.44 (tools/make_traffic_json.py `_method`, profiles/r06_f_fetch_calibration_evict.txt).
# WRITE_SIZE matched every kernel's known output within 1 %.
PMC_KERNELS = (("k_row_r2c", "row_r2c", 2.0), ("k_row_c2c_fwd", "row_c2c", 2.0), ("k_col", "col_fwd_pad_inv", 2.0),
               ("k_c2r_sharpen", "row_c2r_sharpen", 1.44), ("k_row_c2c_inv", "row_c2c_inv", 1.0), ("k_row_c2r", "row_c2r", 1.0), ("k_sharpen", "sharpen", 1.0))


def live_traffic(argv_config):
    """HBM bytes per launch of this configuration's kernels, measured NOW: two short runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters only, with --kernel-trace, as the microarchitecture
    guide prescribes), averaged per kernel over the run's launches, corrected as PMC_KERNELS says.  None where rocprofv3 is missing,
    fails or takes too long -- the committed figures then stand in (`traffic_source` says which it is)."""
    import csv
    import glob
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled"
    child = [sys.executable, os.path.abspath(__file__)] + argv_config + ["--steps", "1", "--warmup", "1", "--repeats", "1", "--frames-per-step", "8",
                                                                        "--profile-iters", "2", "--no-cpu-baseline", "--no-others", "--no-rccl-check", "--no-live-traffic"]
    acc = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="fftup_pmc_") as tmp:
        env = dict(os.environ, TMPDIR=tmp)
        for i, counters in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"])):
            out = os.path.join(tmp, "pass%d" % i)
            try:
                r = subprocess.run([rp, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child,
                                   cwd=tmp, env=env, capture_output=True, text=True, timeout=240)
            except Exception as e:
                return None, "rocprofv3 pass %d: %s" % (i, type(e).__name__)
            files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 pass %d failed (exit code %d)" % (i, r.returncode)
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].replace("void ", "").replace("fftup::", "")
                    a = acc.setdefault(name.split("<")[0].split("(")[0], {}).setdefault(row["Counter_Name"], [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    per = {}
    for kname, c in acc.items():
        for prefix, slot, factor in PMC_KERNELS:
            if kname.startswith(prefix) and slot not in per and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rd = c["FETCH_SIZE"][0] / c["FETCH_SIZE"][1] * 1024 * factor          # (KiB of 64-byte fabric requests)
                wr = c["WRITE_SIZE"][0] / c["WRITE_SIZE"][1] * 1024
                per[slot] = {"kernel": kname, "fetch_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "fetch_correction": factor,
                             "launches_averaged": c["FETCH_SIZE"][1]}
                break
    if not per:
        return None, "no kernel of the plan in the counter files"
    return per, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes over this configuration in this run (%.0f s)" % (time.perf_counter() - t0)


def launch_command(n, argv, port):
    """the driver's own launch line for N ranks on one node (one process per GPU, rendezvous on 127.0.0.1)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a torch.distributed.run environment: start the N ranks here -- the counterpart of the
    reference fanning its threads out inside the process (VkResample.cpp:1959-1969).  Rank 0's JSON line passes through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return subprocess.call(launch_command(n, argv, port), env=env)


RCCL_SELFCHECK = r"""
import json, socket, sys, time
t0 = time.perf_counter()
import torch, torch.distributed as dist
from datetime import timedelta
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
torch.cuda.set_device(int(sys.argv[1]))
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, timeout=timedelta(seconds=60))
x = torch.arange(8, dtype=torch.float64, device="cuda")
dist.all_reduce(x)
dist.barrier()
torch.cuda.synchronize()
ok = bool((x.cpu() == torch.arange(8, dtype=torch.float64)).all())
ver = ".".join(str(v) for v in torch.cuda.nccl.version())
dist.destroy_process_group()
print("RCCL " + json.dumps({"ok": ok, "backend": "nccl (RCCL)", "version": ver, "ranks": 1, "device": torch.cuda.get_device_name(),
                            "seconds": time.perf_counter() - t0}))
"""


def rccl_selfcheck(dev):
    """N = 1 only: bring RCCL up as a one-rank communicator on this GPU and push one all-reduce through it -- the library the
    N > 1 runs depend on is loaded and initialised in every single-GPU line, not first on the day an 8-GPU node shows up.
    In a process of its own with a time limit: a communicator that does not come up must not take the bench line with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", RCCL_SELFCHECK, str(dev)], capture_output=True, text=True, timeout=180)
        for l in r.stdout.splitlines():
            if l.startswith("RCCL "):
                return json.loads(l[5:])
        return {"ok": False, "error": "exit code %d: %s" % (r.returncode, r.stderr[-400:])}
    except Exception as e:                                   # reported, never hidden: tests/test_gpu_bench.py asserts ok
        return {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}


def main():
    args = parse_args()
    # the pool's host driver supports dmabuf IPC only: without this RCCL's multi-process bring-up fails in hipIpcGetMemHandle
    # (exported on the boxes already; kept in every environment built here, ranks started by spawn_ranks included)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line that says n_gpus = WORLD_SIZE while the caller asked for --gpus N would be a wrong scaling point: refuse
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run `python bench.py --gpus %d`, "
                         "which starts its ranks itself)\n" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (backend "nccl") over xGMI; FFTUP_BENCH_BACKEND=gloo lets several ranks share one GPU in a dry run
        backend = os.environ.get("FFTUP_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl" and torch.cuda.device_count() < world:
            sys.stderr.write("bench.py: %d ranks but %d visible GPU(s): one rank per GPU over RCCL (FFTUP_BENCH_BACKEND=gloo is the "
                             "dry run in which ranks share a device)\n" % (world, torch.cuda.device_count()))
            sys.exit(2)
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    os.environ["FFTUP_STREAMS"] = str(args.streams)
    # plan-time tuner findings of earlier runs on this machine (<cache dir>/wisdom.txt) must not steer a measurement
    wisdom = "user cache (%s)" % os.environ["FFTUP_CACHE_DIR"] if "FFTUP_CACHE_DIR" in os.environ else "built-in only (private cache dir)"
    tmp_cache = None
    if "FFTUP_CACHE_DIR" not in os.environ and not args.tune:
        tmp_cache = tempfile.TemporaryDirectory(prefix="fftup_bench_")
        os.environ["FFTUP_CACHE_DIR"] = tmp_cache.name
    import vkresample_amd as v
    from vkresample_amd import shard, synth
    if v.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    dev = local_rank % v.device_count()
    try:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count if torch.cuda.is_available() else None
    except Exception:
        n_cu = None
    flags = (v.FLAG_FUSE_U8_LOAD if args.fuse_u8 else 0) | (v.FLAG_GENERIC_KERNELS if args.generic else 0) | (v.FLAG_TUNE_PLAN if args.tune else 0) | \
            (v.FLAG_FUSE_U8_STORE if args.fuse_u8_store else 0)
    up = v.Upscaler(args.width, args.height, args.upscale, args.precision, 0.2, dev, flags, args.ring)
    queue_store = None
    if args.queue:
        # dynamic sharding: frame g of the step lives in slot g on EVERY rank; who processes it is decided by the shared counter
        my_frames = list(range(args.frames_per_step * world))
        for g in my_frames:
            up.upload_rgb8(synth.frame(g, args.width, args.height, "U"), slot=g)
        queue_store = shard.FrameQueue.default_store(dist) if dist is not None else None
    elif args.job:
        # the reference's stripe (VR:1622-1629): thread/rank t of T takes files f*T + t; slot f holds this rank's f-th frame
        my_frames = shard.frames_for_rank(args.frames_per_step * world, world, rank)
        assert len(my_frames) == args.frames_per_step
        for s, g in enumerate(my_frames):
            up.upload_rgb8(synth.frame(g, args.width, args.height, "U"), slot=s)
    else:
        # distinct frames per rank and slot: rank r owns frames r*ring .. r*ring+ring-1 of the job
        for s in range(args.ring):
            up.upload_rgb8(synth.frame(rank * args.ring + s, args.width, args.height, "U"), slot=s)

    def barrier():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    pins = None
    png_bytes = [0, 0]                         # --png: bytes and files of the step that ran last
    if args.host_streamed:
        out_shape = (args.ring, (up.png_bound() + 63) // 64 * 64) if args.png else (args.ring, up.out_height, up.out_width, 3)
        pins = (v.PinnedArray((args.ring, args.height, args.width, 3)), v.PinnedArray(out_shape))
        for s in range(args.ring):
            # (--png: frames with structure -- uniform noise does not compress, and what PNG carries is images)
            pins[0].array[s] = synth.frame(rank * args.ring + s, args.width, args.height, "N" if args.png else "U")

    def streamed_step():
        if args.png:                           # a ticket is collected before its ring slot comes round again
            tickets, nb = [None] * args.ring, 0
            for k in range(args.frames_per_step + args.ring):
                s = k % args.ring
                if tickets[s] is not None:
                    nb += up.wait_png(tickets[s], pins[1].array[s])
                    tickets[s] = None
                if k < args.frames_per_step:
                    tickets[s] = up.submit_png(pins[0].array[s], pins[1].array[s])
            png_bytes[0], png_bytes[1] = nb, args.frames_per_step
            return
        for k in range(args.frames_per_step):
            up.submit_rgb8(pins[0].array[k % args.ring], pins[1].array[k % args.ring])
        up.drain()

    class LocalStore:                          # one process: the same counter without a server
        def __init__(self):
            self.d = {}

        def add(self, k, n):
            self.d[k] = self.d.get(k, 0) + n
            return self.d[k]

    if args.queue and queue_store is None:
        queue_store = LocalStore()
    claimed = []                               # --queue: the chunks this rank processed in the step that ran last
    step_no = [0]

    def queue_step():
        """one step of the job through the shared counter: claim, run, claim ... until the step's frames are handed out"""
        q = shard.FrameQueue(queue_store, args.frames_per_step * world, args.queue_chunk, key="step%d" % step_no[0])
        step_no[0] += 1
        del claimed[:]
        ms = 0.0
        for (a, b) in q:
            ms += up.execute_ring(b - a, a)
            claimed.append((a, b))
        return ms

    slot = 0
    for _ in range(args.warmup):
        if args.queue:
            queue_step()
            continue
        if args.host_streamed:
            streamed_step()
            continue
        up.execute_ring(args.frames_per_step, slot)
        slot = (slot + args.frames_per_step) % args.ring

    # ---- timed regions: EXACTLY --steps steps each, barrier + synchronize on both sides, MAX over ranks; median of --repeats
    power = PowerSampler(v.device_pci_bus_id(dev)).start() if rank == 0 else None
    region_s, region_dev_ms, region_own_s = [], [], []
    kms = [0.0] * len(up.kernel_names)
    for rep in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(args.steps):
            if args.host_streamed:
                t1 = time.perf_counter()
                streamed_step()
                dev_ms += (time.perf_counter() - t1) * 1e3
                continue
            if args.queue:
                dev_ms += queue_step()
                continue
            if rep == 0:
                # HIP events before/after every kernel launch of every --event-stride-th frame, on the stream that runs it
                ms, km = up.execute_ring_timed(args.frames_per_step, slot, args.event_stride)   # blocks until the batch is done
                kms = [a + b / args.steps for a, b in zip(kms, km)]
            else:
                ms = up.execute_ring(args.frames_per_step, slot)
            dev_ms += ms
            slot = (slot + args.frames_per_step) % args.ring
        t_own = time.perf_counter() - t0      # this rank's own work, before the closing barrier: a sagging rank shows here
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        region_s.append(dt)
        region_own_s.append(t_own)
        region_dev_ms.append(dev_ms)
    power_stats = power.stop() if power is not None else None
    order = sorted(range(len(region_s)), key=lambda i: region_s[i])
    med = order[len(order) // 2]
    dt = region_s[med]

    frames_per_region = world * args.steps * args.frames_per_step
    fps = frames_per_region / dt
    # every rank's own rate in the median region (frames it processed / time until ITS work was done): with N ranks on one host the
    # host-streamed regime shares PCIe root complexes and DRAM bandwidth -- a rank that sags is visible here, not only in the MAX
    per_rank = None
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "frames_per_s": args.steps * args.frames_per_step / region_own_s[med], "own_s": region_own_s[med]})
    job = None
    if not args.job and dist is not None:
        # any multi-rank run says what the collective and the devices really were (the driver's scaling runs use the default
        # preset): ranks seen by an all-reduce, one device per rank, a fingerprint of every rank's distinct resident frames
        sums = [up.output_checksum(s) for s in range(args.ring)]
        me = {"rank": rank, "device": dev, "pci_bus_id": v.device_pci_bus_id(dev), "name": up.device_name,
              "frames": [rank * args.ring, rank * args.ring + args.ring - 1], "checksum": sum(sums) % (1 << 52)}
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        one = torch.ones(1, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        job = {"collective_ranks": int(one.item()), "backend": dist.get_backend(), "ranks": ranks,
               "distinct_devices": len({r["pci_bus_id"] for r in ranks}), "distinct_resident_frames": world * args.ring,
               "frames_done": frames_per_region, "checksum": sum(r["checksum"] for r in ranks) % (1 << 52),
               "sharding": "rank r owns resident frames r*ring .. r*ring+ring-1; no data-path collective"}
    if args.job:
        # every output slot holds the result of one distinct frame of the job: fingerprint them on the device, reduce
        # (--queue: the frames this rank was handed in the last step -- over the ranks, every frame of the step exactly once)
        mine = [g for (a, b) in claimed for g in range(a, b)] if args.queue else list(range(args.frames_per_step))
        sums = [up.output_checksum(s) for s in mine]
        checksum = sum(sums) % (1 << 52)
        frames_done, total, tmax = shard.reduce_summary(dist, len(mine), checksum, dt)
        me = {"rank": rank, "device": dev, "pci_bus_id": v.device_pci_bus_id(dev), "name": up.device_name,
              "first_frames": (mine if args.queue else my_frames)[:3], "frames": len(mine)}
        ranks, nranks = [me], 1
        if dist is not None:
            ranks = [None] * world
            dist.all_gather_object(ranks, me)
            one = torch.ones(1, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(one)                         # how many ranks the communicator really has
            nranks = int(one.item())
        job = {"frames_per_step_total": args.frames_per_step * world, "frames_done": frames_done, "checksum": total % (1 << 52),
               "elapsed_max_s": tmax, "collective_ranks": nranks, "backend": dist.get_backend() if dist is not None else None,
               "ranks": ranks, "distinct_devices": len({(r["pci_bus_id"]) for r in ranks}),
               "stripe": ("shared counter, chunks of %d frames (shard.FrameQueue)" % args.queue_chunk) if args.queue
                         else "frame f*world + rank (VkResample.cpp:1622-1629)"}
    line = None
    if rank == 0:
        # kms: average kernel durations inside the first timed region (consecutive frames overlap on --streams lanes, so a
        # kernel's duration there includes the time it shares the GPU with the other lanes' kernels);
        # iso: the same kernels launched strictly one after the other right after the timed regions, HIP events on the
        # launching stream, net of the cost of an empty event pair (= what rocprofv3 --kernel-trace --stats reports)
        iso = up.profile_kernels(args.profile_iters)
        if max(iso) <= 0:                      # kernels shorter than an event pair's own cost (tiny frames, few iterations)
            iso = list(kms)
        dom = max(range(len(iso)), key=lambda i: iso[i])
        achieved = up.kernel_alg_bytes[dom] / (max(iso[dom], 1e-6) * 1e-3) / 1e9
        if args.host_streamed or args.queue:
            kms = list(iso)                   # no per-kernel events in the streamed loop
        achieved_ovl = up.kernel_alg_bytes[dom] / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else None
        frame_ms = region_dev_ms[med] / (args.steps * args.frames_per_step)
        wall_frame_ms = dt / (args.steps * args.frames_per_step) * 1e3
        # measured HBM bytes (rocprofv3 --pmc, corrected as profiles/hbm_traffic.json documents) -- static: taken from the
        # committed profile of THIS configuration, not re-measured by this run
        key = config_key(args) + ("_u8out" if up.u8_store else "")
        per, frame_hbm, tsrc = frame_traffic(args.traffic_json, key, up.kernel_names)
        traffic = per.get(up.kernel_names[dom]) if per else None
        static_traffic, live = traffic, None
        if world == 1 and not args.no_live_traffic and not args.host_streamed and not args.queue:
            # ... and measured now (the static figures stay in the line for comparison)
            cfg_argv = ["--width", str(args.width), "--height", str(args.height), "--upscale", repr(args.upscale), "--precision", str(args.precision),
                        "--ring", str(args.ring), "--streams", str(args.streams)] + (["--fuse-u8"] if args.fuse_u8 else []) + \
                       (["--fuse-u8-store"] if args.fuse_u8_store else []) + (["--generic"] if args.generic else [])
            live, live_src = live_traffic(cfg_argv)
            if live and all(n in live for n in up.kernel_names if n != "-"):
                per = {n: live[n]["hbm_bytes_per_launch"] for n in up.kernel_names if n != "-"}
                frame_hbm, tsrc, traffic = sum(per.values()), live_src, per[up.kernel_names[dom]]
            else:
                tsrc = (tsrc or "none") + " [live measurement unavailable: %s]" % (live_src if not live else "kernels missing")
                live = None
        esz = {0: 4, 1: 8, 2: 2}[args.precision]
        b_in = 1 if (args.fuse_u8 and args.precision != 1) else esz
        b_min = 3.0 * (args.width * args.height * b_in + up.out_width * up.out_height * (1 if up.u8_store else esz))
        line = {
            "metric": "frames/s, %dx%d->%dx%d %s FFT upscale (R2C+zero-pad+C2R+sharpen)"
                      % (args.width, args.height, up.out_width, up.out_height, {0: "fp32", 1: "fp64", 2: "fp16-memory"}[args.precision]),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {0: "f32", 1: "f64", 2: "f16-memory/f32-math"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "%dx%d->%dx%d -u %g -p %d, %d frames/step/GPU, ring of %d resident %s frames/GPU"
                                   % (args.width, args.height, up.out_width, up.out_height, args.upscale,
                                      args.precision, args.frames_per_step, args.ring,
                                      "uint8 RGB (fused load)" if args.fuse_u8 and args.precision != 1 else "planar fp%d" % {0: 32, 1: 64, 2: 16}[args.precision])
                                   + (", 8-bit RGB out (fused store)" if up.u8_store else ""),
                       "preset": args.preset or "config2", "frames_per_step": args.frames_per_step,
                       "sharding": "independent frames, no collective",
                       "kernels": ("plan-time" if up.specialised_at_plan_time else "tuned") if up.tuned else "generic", "plan": up.description, "streams": args.streams, "device": up.device_name,
                       "wisdom": wisdom},
            "repeats": len(region_s), "region_s": region_s, "timed_region_s_median": dt,
            "ms_per_frame": wall_frame_ms, "ms_per_frame_device_events": frame_ms,
            "frame_alg_bytes": up.alg_bytes_per_frame, "B_min": b_min, "b_min_frac": b_min / (wall_frame_ms * 1e-3) / 8e12,
            "frame_roofline_frac": up.alg_bytes_per_frame / (wall_frame_ms * 1e-3) / 8e12,
            "frame_hbm_bytes_measured": frame_hbm,
            "real_traffic_frac": (frame_hbm / (wall_frame_ms * 1e-3) / 8e12) if frame_hbm else None,
            "kernel_ms": dict(zip(up.kernel_names, iso)),
            "kernel_ms_timed_region": dict(zip(up.kernel_names, kms)),
            "roofline": {"bound": "hbm", "kernel": up.kernel_names[dom], "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": tsrc,
                         # the same kernel priced by the bytes it really has to move (a fused launch never writes R)
                         "achieved_real_bytes": up.kernel_min_bytes[dom] / (max(iso[dom], 1e-6) * 1e-3) / 1e9,
                         "frac_real_bytes": up.kernel_min_bytes[dom] / (max(iso[dom], 1e-6) * 1e-3) / 8e12,
                         "achieved_overlapped": achieved_ovl,
                         "frame_achieved": up.alg_bytes_per_frame / (wall_frame_ms * 1e-3) / 1e9,      # per GPU
                         "frame_frac": up.alg_bytes_per_frame / (wall_frame_ms * 1e-3) / 8e12,
                         # do the static figures (traffic, vector instructions) belong to the kernel sources that just ran?
                         "traffic_kernel_sources_current": traffic_is_current(args.traffic_json, key),
                         "traffic_static": static_traffic, "traffic_live": live},
        }
        # the same fraction from the committed rocprofv3 --kernel-trace --stats summary of this configuration, when there is one
        rp_us, rp_file, rp_fresh = rocprof_kernel_us(key, up.kernel_names[dom])
        if rp_us:
            line["roofline"].update({"rocprof_avg_us": rp_us, "frac_rocprof": up.kernel_alg_bytes[dom] / (rp_us * 1e-6) / 8e12,
                                     "rocprof_source": rp_file, "rocprof_kernel_sources_current": rp_fresh})
        fvi, vbusy = valu_busy(args.traffic_json, key, up.kernel_names, wall_frame_ms, (power_stats or {}).get("sclk_mhz_median"), n_cu)
        line["frame_valu_insts"], line["valu_busy_frac"] = fvi, vbusy
        if power_stats and power_stats.get("socket_power_w_median"):
            # at the power limit this is the number a kernel change has to move: joules per frame
            power_stats["energy_mj_per_frame"] = power_stats["socket_power_w_median"] * wall_frame_ms       # (one GPU works on a frame)
        line["power"] = power_stats
        if args.host_streamed:
            pcie = 3.0 * (args.width * args.height + up.out_width * up.out_height)
            if args.png:
                pcie = 3.0 * args.width * args.height + png_bytes[0] / max(png_bytes[1], 1)
                line["png_bytes_per_frame"] = png_bytes[0] / max(png_bytes[1], 1)
            line["pcie_inclusive"] = True
            line["config"]["workload"] += ", HOST-STREAMED: uint8 RGB frames from pinned host memory, " + \
                ("finished PNG files (encoded on the device) back" if args.png else "uint8 RGB frames back")
            line["pcie_bytes_per_frame"] = pcie
            line["pcie_GBps"] = pcie / (wall_frame_ms * 1e-3) / 1e9
        if per_rank is not None:
            line["per_rank_frames_per_s"] = [round(r["frames_per_s"], 1) for r in sorted(per_rank, key=lambda r: r["rank"])]
        if job is not None:
            line["job"] = job
            line["rccl_ranks"] = job["collective_ranks"]
            line["frames_done"] = job["frames_done"]
            line["checksum"] = job["checksum"]
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args)
            line["reference_vulkan_baseline"] = reference_vulkan_baseline(args)
        if world == 1 and not args.no_rccl_check and torch.cuda.is_available():
            line["rccl_selfcheck"] = rccl_selfcheck(dev)
    up.close()
    if rank == 0 and world == 1 and not args.no_others and not args.host_streamed and (args.preset or "config2") == "config2" \
            and (args.width, args.height, args.precision) == (2048, 1024, 0):
        line["others"] = o = other_configs(v, synth, dev, args.traffic_json, n_cu, live=not args.no_live_traffic)
        # the figures a reader of the one line needs without opening `others`: the reference's -n 1000 number per configuration
        # and the frame fractions of configs 3 and 4
        keys = ("ms_per_iter", "frame_frac", "sequential_ms_per_iter", "sequential_frame_frac", "overlapped_ms_per_iter", "overlapped_frame_frac")
        line["execute_n1000"] = {k: {m: e[m] for m in keys} for k, e in o["execute_n1000"].items() if isinstance(e, dict)}
        # (the reference-definition figure once more under the name earlier rounds and the verdicts use, as top-level keys)
        line["sequential_ms_per_iter"] = {k: e["ms_per_iter"] for k, e in line["execute_n1000"].items()}
        line["sequential_frame_frac"] = {k: e["frame_frac"] for k, e in line["execute_n1000"].items()}
        line["config3_frame_frac"], line["config3_ms_per_frame"] = o["config3"]["frame_frac"], o["config3"]["ms_per_frame"]
        line["config4_frame_frac"], line["config4_ms_per_frame"] = o["config4"]["frame_frac"], o["config4"]["ms_per_frame"]
    if pins:
        pins[0].close()
        pins[1].close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
