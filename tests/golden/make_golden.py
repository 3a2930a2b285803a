#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs + expected outputs of the CPU oracle (oracle/fftup_oracle.c), which at
generation time was cross-checked against oracle/ref_layout_emulation.py (index-faithful replay of the
reference's buffers) and numpy's rfft2/irfft2 closed form.  The reference itself ships no golden data and
cannot run here (no Vulkan): these vectors pin the ORACLE, not the Vulkan binary (the vectors that come from the
reference's own output are made by make_readme_crops.py).
The 64x64 crop comes from the reference's samples/no_upscaling.png decoded by the reference's own
stb_image (oracle/_ref/libref_host.so), forced to 3 channels like VkResample.cpp:1362."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as O  # noqa: E402
from oracle import ref_layout_emulation as E  # noqa: E402
from vkresample_amd import synth  # noqa: E402


def save(name, rgb, u, precision, sharpen=0.2):
    pre, out, u8 = O.upscale_rgb8(rgb, u, precision, sharpen)
    if precision in (0, 1):
        lut = O.load_lut(precision)
        R, _ = E.emulate(np.stack([lut[rgb[..., c]] for c in range(3)]), u)
        assert np.abs(R - pre).max() < 1e-14
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rgb=rgb, upscale=u, precision=precision, sharpen=sharpen,
                        pre=pre, out=out, u8=u8)
    print(name, rgb.shape, "->", u8.shape)


def sample_full():
    """stb_image decode (forced to 3 channels, VkResample.cpp:1362) of the reference's samples/no_upscaling.png"""
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_host.so"))
    ref.ref_png_load_rgb.restype = C.POINTER(C.c_ubyte)
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    p = ref.ref_png_load_rgb(b"/root/reference/samples/no_upscaling.png", C.byref(w), C.byref(h), C.byref(ch))
    img = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    ref.ref_free(p)
    return img


def sample_crop():
    return sample_full()[500:564, 900:964].copy()


def save_config1():
    """BASELINE config 1: the decoded pixels of samples/no_upscaling.png (data, not source) + digests of the oracle's
    -u 2 -p 0 output so that a GPU-box run can tell a broken oracle build from a broken HIP path."""
    rgb = sample_full()
    assert rgb.shape == (1080, 1920, 3)
    pre, out, u8 = O.upscale_rgb8(rgb, 2.0, 0, 0.2)
    np.savez_compressed(os.path.join(HERE, "no_upscaling_rgb.npz"), rgb=rgb, upscale=2.0, precision=0, sharpen=0.2,
                        out_plane_means=out.mean(axis=(1, 2)), out_crop=out[:, 1000:1064, 1800:1864],
                        u8_crop=u8[1000:1064, 1800:1864], u8_sum=np.int64(u8.astype(np.int64).sum()))
    print("no_upscaling_rgb", rgb.shape, "->", u8.shape)


if __name__ == "__main__":
    save("g16x8_u2_p0", synth.frame(100, 16, 8, "U"), 2.0, 0)
    save("g20x12_u2_p0", synth.frame(101, 20, 12, "U"), 2.0, 0)
    save("g20x12_u2_p1", synth.frame(101, 20, 12, "U"), 2.0, 1)
    save("g64x32_u2_p0", synth.frame(102, 64, 32, "N"), 2.0, 0)
    save("g64x32_u2_p2", synth.frame(102, 64, 32, "N"), 2.0, 2)
    save("gsample64_u2_p0", sample_crop(), 2.0, 0)
    save_config1()
