#!/usr/bin/env python3
"""Golden vectors from the reference's OWN output: tests/golden/readme_*.npz.

/root/reference/samples/{car,close_people,distant_people,skyscraper,trees}.png are the README's comparison strips
(README.md:20-26), 900x300 = [NN | FFT | Native] panels of one 300x300 window of the 2x upscaled frame.
  * The NN panel is an exact 2x2 pixel duplication (99.3 % of its blocks; the rest is the label), i.e. it holds the
    EXACT input pixels of a 151x151 window of the image the reference was run on.
  * The FFT panel is the reference's real output (VkResample -u 2, default -s 0.2) on the same window.
  * The image the reference was run on is not in the repository; the NN panels match samples/no_upscaling*.png
    resampled by 2048/1920 (bicubic; RMS 1-3 grey levels, nothing better than RMS 16 at scale 1): the author fed a
    2048x1152 copy of the 1920x1080 screenshot.
So the input is known exactly inside the window and to ~2 grey levels outside it.  The upscale of a pixel is dominated by
its neighbourhood (sinc tails, alternating signs), hence: input := bicubic 2048x1152 resample with the window overwritten by
the NN panel's pixels -> CPU oracle -> compare with the FFT panel in the window's interior.  Measured here with the
whole 2048x1152 frame: mean |diff| 0.12-0.17 grey levels, 99th percentile 1, max 1-3 (5 strips, 86 400 pixels each).
The committed fixtures hold a 512x512 sub-window around each panel (mean 0.16-0.25, p99 1, max <= 3), so that the CPU
and GPU suites can run them anywhere; `--full` re-runs the whole-frame check here (needs /root/reference).
"""
import os
import sys

import numpy as np
from PIL import Image
from scipy.signal import fftconvolve

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as O  # noqa: E402

D = "/root/reference/samples/"
STRIPS = {"car": "no_upscaling.png", "close_people": "no_upscaling_2.png", "distant_people": "no_upscaling_2.png",
          "skyscraper": "no_upscaling.png", "trees": "no_upscaling.png"}
INNER = (100, 280, 20, 180)          # round 2's comparison box (kept in the fixtures); the tests now compare O.readme_panel_mask():
                                     # every panel pixel outside the label corner and a 12-pixel border


def phase_of(nn):
    best = None
    for py in (0, 1):
        for px in (0, 1):
            a = nn[py:, px:]
            h, w = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
            b = a[:h, :w].reshape(h // 2, 2, w // 2, 2, 3)
            same = ((b[:, 0, :, 0] == b[:, 1, :, 0]) & (b[:, 0, :, 0] == b[:, 0, :, 1]) & (b[:, 0, :, 0] == b[:, 1, :, 1])).all(-1).mean()
            if best is None or same > best[0]:
                best = (same, py, px)
    return best


def locate(full, key):
    g, k = full.mean(-1), key.mean(-1)
    ssd = fftconvolve(g * g, np.ones_like(k), mode="valid") - 2 * fftconvolve(g, k[::-1, ::-1], mode="valid") + (k * k).sum()
    y, x = np.unravel_index(np.argmin(ssd), ssd.shape)
    return int(y), int(x), float(np.sqrt(max(ssd[y, x], 0) / k.size))


def build(name):
    src = STRIPS[name]
    full = np.asarray(Image.open(D + src).convert("RGB").resize((2048, 1152), Image.BICUBIC)).astype(np.float64)
    strip = np.asarray(Image.open(D + name + ".png")).astype(np.int64)
    nn, fft = strip[:, :300], strip[:, 300:600]
    dup, py, px = phase_of(nn)
    iy, ix = (np.arange(300) + py) // 2, (np.arange(300) + px) // 2
    ny, nx = int(iy.max()) + 1, int(ix.max()) + 1
    patch = np.zeros((ny, nx, 3))
    for Y in range(300):
        patch[iy[Y], ix] = nn[Y]
    ky, kx, rms = locate(full, patch[50:140, 5:95])
    yA, xA = ky - 50, kx - 5
    mask = np.ones((ny, nx), bool)
    mask[:35, 95:] = False                      # the "NN" label
    comp = full.copy()
    reg = comp[yA:yA + ny, xA:xA + nx]
    reg[mask] = patch[mask]
    rgb = np.clip(np.rint(comp), 0, 255).astype(np.uint8)
    return dict(rgb=rgb, fft=fft.astype(np.uint8), yA=yA, xA=xA, py=py, px=px, ny=ny, nx=nx, dup=dup, rms=rms)


def compare(u8, fft, Yo, Xo):
    r0, r1, c0, c1 = INNER
    d = np.abs(u8[Yo:Yo + 300, Xo:Xo + 300].astype(np.int64) - fft.astype(np.int64))[r0:r1, c0:c1]
    return float(d.mean()), float(np.percentile(d, 99)), int(d.max())


if __name__ == "__main__":
    full_mode = "--full" in sys.argv
    for name in STRIPS:
        b = build(name)
        rgb, yA, xA, py, px = b["rgb"], b["yA"], b["xA"], b["py"], b["px"]
        if full_mode:
            for s in (0.2, 0.1, 0.3):
                _, _, u8 = O.upscale_rgb8(rgb, 2.0, 0, s)
                print("%-15s whole frame, -s %.1f: mean %.2f p99 %.0f max %d" % ((name, s) + compare(u8, b["fft"], 2 * yA + py, 2 * xA + px)), flush=True)
        WW = WH = 512
        wy = min(max(yA + b["ny"] // 2 - WH // 2, 0), 1152 - WH)
        wx = min(max(xA + b["nx"] // 2 - WW // 2, 0), 2048 - WW)
        win = np.ascontiguousarray(rgb[wy:wy + WH, wx:wx + WW])
        Yo, Xo = 2 * (yA - wy) + py, 2 * (xA - wx) + px
        _, _, u8 = O.upscale_rgb8(win, 2.0, 0, 0.2)
        m = compare(u8, b["fft"], Yo, Xo)
        np.savez_compressed(os.path.join(HERE, "readme_%s.npz" % name), rgb=win, fft_panel=b["fft"], Yo=Yo, Xo=Xo,
                            inner=np.array(INNER), upscale=2.0, sharpen=0.2, precision=0)
        print("%-15s NN duplication %.3f, input match rms %.1f; 512x512 window at (%d,%d), panel at output (%d,%d): mean %.2f p99 %.0f max %d"
              % ((name, b["dup"], b["rms"], wy, wx, Yo, Xo) + m), flush=True)
