"""GPU: the drop-in CLI end to end (PNG in -> PNG out), single-image and batched modes, against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import oraclelib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "vkresample_amd", "vkresample")


def _png_write(path, rgb):
    from PIL import Image
    Image.fromarray(rgb).save(path)


def _png_read(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def test_cli_single_image_1080p(tmp_path):
    """BASELINE config 1 shape: 1920x1080 (radix 3/5 path) -u 2 -p 0 -n 1, default output name (quirk B11)."""
    from vkresample_amd import synth
    rgb = synth.frame(11, 1920, 1080, "N")
    _png_write(tmp_path / "in.png", rgb)
    r = subprocess.run([CLI, "-i", "in.png", "-u", "2", "-p", "0", "-n", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "VkResample - FFT based upscaling" in r.stdout
    assert re.search(r"VkResample 2\.0x upscale: 1920x1080 to 3840x2160 Time: [0-9.]+ ms", r.stdout)
    assert "Thread 0 finished." in r.stdout and "Total time:" in r.stdout
    out = _png_read(tmp_path / "1920_3840_upscaled.png")
    _, _, ou8 = O.upscale_rgb8(rgb, 2.0, 0, 0.2)
    d = np.abs(out[:-1].astype(int) - ou8[:-1].astype(int))
    assert out.shape == (2160, 3840, 3) and d.max() <= 1 and (d != 0).mean() <= 5e-3


def test_cli_upscale_factor_as_a_ratio(tmp_path):
    """-u 4/3 (extension: the ratio divided in float): 960 x 540 -> exactly 1280 x 720, the result of -u 1.3333334"""
    from vkresample_amd import synth
    rgb = synth.frame(13, 960, 540, "N")
    _png_write(tmp_path / "in.png", rgb)
    outs = []
    for u in ("4/3", "1.3333334"):
        r = subprocess.run([CLI, "-i", "in.png", "-o", "out.png", "-u", u, "-n", "1"], capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert re.search(r"upscale: 960x540 to 1280x720 Time: [0-9.]+ ms", r.stdout), r.stdout
        outs.append(_png_read(tmp_path / "out.png"))
    assert outs[0].shape == (720, 1280, 3) and np.array_equal(outs[0], outs[1])
    _, _, ou8 = O.upscale_rgb8(rgb, float(np.float32(4.0 / 3.0)), 0, 0.2)
    d = np.abs(outs[0][:-1].astype(int) - ou8[:-1].astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 5e-3


@pytest.mark.parametrize("W,H,u,p", [(800, 600, "2", "0"), (1280, 720, "1.5", "0"), (960, 540, "4", "2")])
def test_cli_sizes_specialised_at_plan_time(tmp_path, W, H, u, p):
    """sizes and factors the CLI gets plan-time specialised kernels for (csrc/jit.hpp): 800x600 -u 2, 720p -> 1080p,
    540p -> 2160p with -p 2"""
    from vkresample_amd import synth
    rgb = synth.frame(12, W, H, "N")
    _png_write(tmp_path / "in.png", rgb)
    r = subprocess.run([CLI, "-i", "in.png", "-o", "out.png", "-u", u, "-p", p, "-n", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    uW, uH = int(float(u) * W), int(float(u) * H)
    assert re.search(r"VkResample %.1fx upscale: %dx%d to %dx%d Time: [0-9.]+ ms" % (float(u), W, H, uW, uH), r.stdout)
    out = _png_read(tmp_path / "out.png")
    _, _, ou8 = O.upscale_rgb8(rgb, float(u), int(p), 0.2)
    d = np.abs(out[:-1].astype(int) - ou8[:-1].astype(int))
    assert out.shape == (uH, uW, 3)
    if p == "0":
        assert d.max() <= 1 and (d != 0).mean() <= 5e-3
    else:
        assert d.max() <= 2 and (d > 1).mean() <= 1e-3          # fp16 storage: a one-ulp flip can move a code by 2


def test_cli_time_is_the_ordered_figure_and_overlap_is_an_extension(tmp_path):
    """-n N runs its iterations in order (the reference's barriers, VR:1217): the output file does not depend on N -- same plan, same
    strip cuts -- and `-overlap` (FFTUP_FLAG_OVERLAP_ITERATIONS) gives the same picture to rounding, faster per iteration."""
    from vkresample_amd import synth
    rgb = synth.frame(5, 2048, 1024, "N")
    _png_write(tmp_path / "in.png", rgb)
    outs, times = {}, {}
    for tag, extra in (("n1", ["-n", "1"]), ("n40", ["-n", "40"]), ("n40_overlap", ["-n", "40", "-overlap"])):
        r = subprocess.run([CLI, "-i", "in.png", "-o", tag + ".png", "-u", "2"] + extra, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        times[tag] = float(re.search(r"Time: ([0-9.]+) ms", r.stdout).group(1))
        outs[tag] = _png_read(tmp_path / (tag + ".png"))
    assert np.array_equal(outs["n1"], outs["n40"])
    d = np.abs(outs["n40"].astype(int) - outs["n40_overlap"].astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
    print("MEASURED CLI Time: -n 1 %.3f ms, -n 40 %.3f ms, -n 40 -overlap %.3f ms" % (times["n1"], times["n40"], times["n40_overlap"]))
    assert times["n40_overlap"] <= times["n40"] * 1.05


def test_cli_config1_literal_image(tmp_path):
    """BASELINE config 1 end to end: the pixels of the reference's samples/no_upscaling.png (committed as data,
    tests/golden/no_upscaling_rgb.npz; the reference decodes RGBA to 3 channels, VR:1362) -u 2 -p 0 -n 1."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "no_upscaling_rgb.npz"))
    _png_write(tmp_path / "no_upscaling.png", d["rgb"])
    r = subprocess.run([CLI, "-i", "no_upscaling.png", "-o", "up.png", "-u", "2", "-p", "0", "-n", "1"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"VkResample 2\.0x upscale: 1920x1080 to 3840x2160 Time: [0-9.]+ ms", r.stdout)
    out = _png_read(tmp_path / "up.png")
    _, _, ou8 = O.upscale_rgb8(d["rgb"], 2.0, 0, 0.2)
    assert np.array_equal(ou8[1000:1064, 1800:1864], d["u8_crop"])
    dd = np.abs(out[:-1].astype(int) - ou8[:-1].astype(int))
    assert out.shape == (2160, 3840, 3) and dd.max() <= 1 and (dd != 0).mean() <= 5e-3


def test_cli_batched_two_threads(tmp_path):
    from vkresample_amd import synth
    os.makedirs(tmp_path / "inp")
    os.makedirs(tmp_path / "outp")
    frames = [synth.frame(20 + k, 256, 128, "N") for k in range(5)]
    for k, f in enumerate(frames):
        _png_write(tmp_path / "inp" / ("%06d.png" % (k + 1)), f)
    r = subprocess.run([CLI, "-ifolder", "inp", "-ofolder", "outp", "-numfiles", "5", "-numthreads", "2", "-u", "2",
                        "-p", "2", "-s", "0.1", "-alldevices"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("finished.") == 2
    for k, f in enumerate(frames):
        out = _png_read(tmp_path / "outp" / ("%06d.png" % (k + 1)))
        _, _, ou8 = O.upscale_rgb8(f, 2.0, 2, 0.1)
        d = np.abs(out[:-1].astype(int) - ou8[:-1].astype(int))
        assert d.max() <= 2 and (d > 1).mean() <= 1e-3          # fp16 storage: a one-ulp flip can move a code by 2


@pytest.mark.parametrize("threads,extra", [(3, ["-workqueue"]), (3, ["-workqueue", "-n", "2"]), (9, ["-workqueue", "-alldevices"])])
def test_cli_batched_work_queue(tmp_path, threads, extra):
    """-workqueue: the threads share ONE file counter instead of the static stripe (VR:1622-1629); every file is processed
    exactly once whatever the thread count (more threads than files: the idle ones end at once), same pixels as the stripe."""
    from vkresample_amd import synth
    for d in ("inp", "oq", "os"):
        os.makedirs(tmp_path / d)
    frames = [synth.frame(60 + k, 128, 64, "N") for k in range(7)]
    for k, f in enumerate(frames):
        _png_write(tmp_path / "inp" / ("%06d.png" % (k + 1)), f)
    base = [CLI, "-ifolder", "inp", "-numfiles", "7", "-u", "2", "-p", "0"]
    r = subprocess.run(base + ["-ofolder", "oq", "-numthreads", str(threads)] + extra, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.count("finished.") == threads, r.stdout + r.stderr
    r = subprocess.run(base + ["-ofolder", "os", "-numthreads", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(tmp_path / "oq")) == ["%06d.png" % (k + 1) for k in range(7)]
    for k, f in enumerate(frames):
        a = _png_read(tmp_path / "oq" / ("%06d.png" % (k + 1)))
        assert np.array_equal(a, _png_read(tmp_path / "os" / ("%06d.png" % (k + 1)))), k
        _, _, ou8 = O.upscale_rgb8(f, 2.0, 0, 0.2)
        d = np.abs(a[:-1].astype(int) - ou8[:-1].astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 5e-3


def test_cli_batched_threads_share_the_plan_of_their_size(tmp_path):
    """Batched threads of one GPU share one plan per image size (the reference: one application per thread, sized by the
    thread's first file): stripe of two threads over files of two sizes -- thread 0 gets the 128x64 files 1, 3, 5, 7, thread 1 the
    64x32 files 2, 4, 6 -- and four threads over eight files of one size; every output against the oracle."""
    from vkresample_amd import synth
    for d in ("inp", "outp", "in8", "out8"):
        os.makedirs(tmp_path / d)
    frames = [synth.frame(80 + k, *((128, 64) if k % 2 == 0 else (64, 32)), "N") for k in range(7)]
    for k, f in enumerate(frames):
        _png_write(tmp_path / "inp" / ("%06d.png" % (k + 1)), f)
    r = subprocess.run([CLI, "-ifolder", "inp", "-ofolder", "outp", "-numfiles", "7", "-numthreads", "2", "-u", "2", "-p", "0"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.count("finished.") == 2, r.stdout + r.stderr
    same = [synth.frame(90 + k, 128, 64, "N") for k in range(8)]
    for k, f in enumerate(same):
        _png_write(tmp_path / "in8" / ("%06d.png" % (k + 1)), f)
    r = subprocess.run([CLI, "-ifolder", "in8", "-ofolder", "out8", "-numfiles", "8", "-numthreads", "4", "-u", "2", "-p", "0"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.count("finished.") == 4, r.stdout + r.stderr
    for folder, fr in (("outp", frames), ("out8", same)):
        for k, f in enumerate(fr):
            a = _png_read(tmp_path / folder / ("%06d.png" % (k + 1)))
            _, _, ou8 = O.upscale_rgb8(f, 2.0, 0, 0.2)
            d = np.abs(a[:-1].astype(int) - ou8[:-1].astype(int))
            assert a.shape == ou8.shape and d.max() <= 1 and (d != 0).mean() <= 5e-3, (folder, k)


def test_cli_batched_png_from_the_device(tmp_path):
    """-gpupng: the files the GPU encoded hold the pixels of the files the host encoded (three threads, one shared plan)"""
    from vkresample_amd import synth
    for d in ("inp", "og", "oh"):
        os.makedirs(tmp_path / d)
    frames = [synth.frame(70 + k, 256, 128, "N" if k % 2 else "U") for k in range(7)]
    for k, f in enumerate(frames):
        _png_write(tmp_path / "inp" / ("%06d.png" % (k + 1)), f)
    base = [CLI, "-ifolder", "inp", "-numfiles", "7", "-numthreads", "3", "-u", "2", "-p", "0", "-workqueue"]
    r = subprocess.run(base + ["-ofolder", "og", "-gpupng", "-stagetimes"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.count("finished.") == 3, r.stdout + r.stderr
    r = subprocess.run(base + ["-ofolder", "oh"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for k in range(7):
        a = _png_read(tmp_path / "og" / ("%06d.png" % (k + 1)))
        assert np.array_equal(a, _png_read(tmp_path / "oh" / ("%06d.png" % (k + 1)))), k


def test_cli_batched_queue_one_thread_and_missing_file(tmp_path):
    """one thread, 7 files: the double-buffered queue (fftup_submit_rgb8) writes every frame; -n 3 takes the
    blocking path with identical files; a missing file ends the thread like the reference (VR:1631-1634)."""
    from vkresample_amd import synth
    for d in ("inp", "o1", "o3", "o4"):
        os.makedirs(tmp_path / d)
    frames = [synth.frame(40 + k, 128, 64, "U" if k % 2 else "N") for k in range(7)]
    for k, f in enumerate(frames):
        _png_write(tmp_path / "inp" / ("%06d.png" % (k + 1)), f)
    base = [CLI, "-ifolder", "inp", "-numfiles", "7", "-numthreads", "1", "-u", "2", "-p", "0"]
    r = subprocess.run(base + ["-ofolder", "o1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and r.stdout.count("finished.") == 1, r.stdout + r.stderr
    r = subprocess.run(base + ["-ofolder", "o3", "-n", "3"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for k, f in enumerate(frames):
        a = _png_read(tmp_path / "o1" / ("%06d.png" % (k + 1)))
        b = _png_read(tmp_path / "o3" / ("%06d.png" % (k + 1)))
        assert np.array_equal(a, b), k
        _, _, ou8 = O.upscale_rgb8(f, 2.0, 0, 0.2)
        d = np.abs(a[:-1].astype(int) - ou8[:-1].astype(int))
        assert d.max() <= 1 and (d != 0).mean() <= 5e-3
    os.remove(tmp_path / "inp" / "000005.png")
    r = subprocess.run(base + ["-ofolder", "o4"], capture_output=True, text=True, cwd=tmp_path)
    assert "Image not found" in r.stdout and r.returncode != 0
    assert sorted(os.listdir(tmp_path / "o4")) == ["%06d.png" % k for k in (1, 2, 3, 4)]


def test_cli_double_precision(tmp_path):
    """-p 1 end to end (the reference's double path, VkResample.cpp:1422)"""
    from vkresample_amd import synth
    rgb = synth.frame(12, 240, 126, "N")
    _png_write(tmp_path / "in.png", rgb)
    r = subprocess.run([CLI, "-i", "in.png", "-o", "out.png", "-u", "2", "-p", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _png_read(tmp_path / "out.png")
    _, _, ou8 = O.upscale_rgb8(rgb, 2.0, 1, 0.2)
    d = np.abs(out[:-1].astype(int) - ou8[:-1].astype(int))
    assert out.shape == (252, 480, 3) and d.max() <= 1 and (d != 0).mean() <= 1e-4


def test_cli_devices_and_errors(tmp_path):
    r = subprocess.run([CLI, "-devices"], capture_output=True, text=True)
    assert r.returncode == 0 and "Device id: 0 name:" in r.stdout
    from vkresample_amd import synth
    _png_write(tmp_path / "odd.png", synth.frame(1, 22 * 11, 64, "U"))       # 242 = 2*11*11: not 2,3,5,7-smooth
    r = subprocess.run([CLI, "-i", "odd.png", "-u", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 2 and "unsupported size" in r.stdout
