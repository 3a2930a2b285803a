"""CPU suite: the oracle against numpy.fft, the index-faithful layout emulation, analytic known-answer
tests and the committed golden vectors.  No GPU."""
import os
import sys

import numpy as np
import pytest

import oraclelib as O
from oracle import ref_layout_emulation as E

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("n", [2, 4, 8, 12, 30, 64, 210, 1080, 2048, 2160])
def test_fft1d_vs_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.abs(O.fft1d(x, -1) - np.fft.fft(x)).max() <= 1e-11
    assert np.abs(O.fft1d(x, +1) - np.fft.ifft(x) * n).max() <= 1e-11   # reference "forward" = exp(+i..)


@pytest.mark.parametrize("W,H,u", [(16, 8, 2.0), (20, 12, 2.0), (24, 30, 2.0), (16, 8, 1.5), (16, 8, 1.0),
                                   (12, 10, 3.0), (32, 16, 2.5)])
def test_oracle_equals_layout_emulation_and_closed_form(W, H, u):
    """C restatement == replay of the reference's buffers/strides/guards == SURVEY App. A.3 closed form."""
    rng = np.random.default_rng(W * 1000 + H)
    planes = rng.random((3, H, W))
    pre, _, poison = O.upscale_planes(planes, u)
    R, temp = E.emulate(planes, u)
    assert poison == 0 and not np.isnan(R).any()       # nothing stale is ever read
    assert np.abs(pre - R).max() <= 1e-14
    assert np.abs(R - E.closed_form(planes, u)).max() <= 1e-14


def test_out_dims_and_checks():
    assert O.out_dims(2048, 1024, 2.0) == (4096, 2048)
    assert O.out_dims(1920, 1080, 2.0) == (3840, 2160)
    assert O.out_dims(16, 8, 1.5) == (24, 12)
    assert O.check(2048, 1024) == 0
    assert O.check(2 * 11 * 64, 64) == 2          # KAT8: not 2,3,5,7-smooth (vkFFT.h:4719-4726)
    assert O.check(63, 64) == 1
    assert O.check(64, 64, precision=1) == 0 and O.check(64, 64, precision=3) == 3


def test_load_conversion_table():
    """VkResample.cpp:1644 / 1676 for all 256 codes, against independent numpy float32/float16 arithmetic."""
    v = np.arange(256)
    f32 = (v.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    assert np.array_equal(O.load_lut(0), f32)
    f16 = (v.astype(np.float16).astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float16)
    assert np.array_equal(O.load_lut(2), f16.astype(np.float64))
    assert np.array_equal(O.load_lut(1), v / 255.0)              # -p 1: (double)v / 255.0 (VkResample.cpp:1657)
    # fused device conversion = fp32 division (checked on the GPU side too)
    assert np.array_equal(O.load_lut(0), (v.astype(np.float32) / np.float32(255)).astype(np.float64))


REFSO = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "oracle", "_ref", "libref_host.so")


@pytest.mark.skipif(not os.path.exists(REFSO), reason="oracle/_ref is built only where /root/reference exists")
def test_oracle_load_matches_reference_half_hpp():
    """The oracle's load/store conversions against the REFERENCE's own host code compiled where it lies
    (oracle/ref_host_shim.cpp -> oracle/_ref/libref_host.so): VkResample.cpp:1644 (float), :1676 (half.hpp
    arithmetic, -p 2), :1715 (unsigned char cast of 255.0*x) and half.hpp's float->half rounding."""
    import ctypes as C
    ref = C.CDLL(REFSO)
    ref.ref_pack_half.restype = C.c_double
    ref.ref_pack_half.argtypes = [C.c_ubyte]
    ref.ref_pack_float.restype = C.c_double
    ref.ref_pack_float.argtypes = [C.c_ubyte]
    ref.ref_unpack_float.restype = C.c_ubyte
    ref.ref_unpack_float.argtypes = [C.c_float]
    ref.ref_half_round.restype = C.c_double
    ref.ref_half_round.argtypes = [C.c_double]
    lut0, lut2 = O.load_lut(0), O.load_lut(2)
    assert np.array_equal(lut0, np.array([ref.ref_pack_float(v) for v in range(256)]))
    assert np.array_equal(lut2, np.array([ref.ref_pack_half(v) for v in range(256)]))
    # every value the -p 2 path can hold round-trips through half.hpp unchanged, and the oracle's
    # round-to-binary16 equals half.hpp's on a dense sweep incl. ties, subnormals and the overflow edge
    L = O.lib()
    L.orc_round_half.restype = C.c_double
    L.orc_round_half.argtypes = [C.c_double]
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-2, 2, 20000), rng.uniform(-1e-4, 1e-4, 5000),
                         (np.arange(0, 4096) + 0.5) * 2.0 ** -11, [0.0, 6.0e-8, 2.0 ** -25, 65504.0, 65519.0]])
    xs = xs.astype(np.float32).astype(np.float64)              # half.hpp converts from float
    for x in xs:
        assert L.orc_round_half(float(x)) == ref.ref_half_round(float(x)), x
    # store: in range the reference's cast truncates exactly like the oracle (both modes agree there)
    for x in np.concatenate([np.arange(256) / 255.0, rng.uniform(0, 1, 4000)]).astype(np.float32):
        want = ref.ref_unpack_float(float(x))
        assert L.orc_store_u8(float(x), 0) == want and L.orc_store_u8(float(x), 1) == want, x
    # out of range the reference is UB; the wrap mode reproduces what x86-64 gcc does with it
    for x in (-1.6 / 255, -0.3, 256.3 / 255, 1.5):
        assert L.orc_store_u8(float(np.float32(x)), 1) == ref.ref_unpack_float(float(np.float32(x)))


def test_store_u8():
    L = O.lib()
    assert L.orc_store_u8(0.5, 0) == 127 and L.orc_store_u8(1.0, 0) == 255 and L.orc_store_u8(0.999, 0) == 254
    assert L.orc_store_u8(-0.1, 0) == 0 and L.orc_store_u8(1.2, 0) == 255          # saturating (ours)
    assert L.orc_store_u8(-1.6 / 255, 1) == 255 and L.orc_store_u8(256.3 / 255, 1) == 0   # x86 wrap of VR:1715


# ------------------------------------------------------------------ known-answer tests (pre-sharpen, t = u^2 g)
def _pre(planes, u=2.0):
    pre, _, _ = O.upscale_planes(np.asarray(planes, dtype=np.float64), u)
    return pre * u * u


def test_kat1_constant():
    planes = np.stack([np.full((8, 16), c) for c in (0.25, 0.5, 1.0)])
    pre, out, _ = O.upscale_planes(planes, 2.0)
    for c, v in enumerate((0.25, 0.5, 1.0)):
        assert np.abs(pre[c] * 4 - v).max() <= 1e-14
        assert np.abs(out[c] - v).max() <= 1e-14           # sharpen of a constant is the constant


def test_kat2_kat3_kat5_cosines():
    W, H, u = 32, 16, 2.0
    x = np.arange(W)[None, :]
    y = np.arange(H)[:, None]
    X = np.arange(int(u * W))[None, :]
    Y = np.arange(int(u * H))[:, None]
    k0 = 5
    planes = np.stack([0.5 + 0.25 * np.cos(2 * np.pi * k0 * x / W) + 0 * y,
                       0.5 + 0.1 * (-1.0) ** x + 0 * y,
                       0.5 + 0.1 * (-1.0) ** y * np.cos(2 * np.pi * 3 * x / W)])
    t = _pre(planes, u)
    assert np.abs(t[0] - (0.5 + 0.25 * np.cos(2 * np.pi * k0 * X / (u * W)) + 0 * Y)).max() <= 1e-14   # KAT2
    assert np.abs(t[1] - (0.5 + 0.2 * np.cos(np.pi * X / u) + 0 * Y)).max() <= 1e-14                   # KAT3 (quirk B1)
    assert np.abs(t[2] - (0.5 + 0.1 * np.cos(2 * np.pi * 3 * X / (u * W) - np.pi * Y / u))).max() <= 1e-14   # KAT5


def test_kat4_nyquist_row_cancelled_by_leak():
    """0.5 + a(-1)^y, u = 2: the (kx=0, ky=H/2) component is cancelled exactly by the row-pair DC leak (B2+B3)."""
    W, H = 16, 8
    y = np.arange(H)[:, None]
    p = 0.5 + 0.1 * (-1.0) ** y + np.zeros((H, W))
    t = _pre(np.stack([p, p, p]))
    assert np.abs(t - 0.5).max() <= 1e-14


def test_kat6_impulse():
    W, H, u = 16, 8, 2
    p = np.zeros((H, W))
    p[0, 0] = 1.0
    t = _pre(np.stack([p, p, p]))[0]
    S = np.ones((H, W // 2 + 1), dtype=complex)
    S[H // 2, 0] = 0.0                                        # u = 2: leak == zeroing S[H/2,0]
    G = np.zeros((u * H, u * W // 2 + 1), dtype=complex)
    G[:H // 2, :W // 2 + 1] = S[:H // 2]
    G[u * H - H // 2:, :W // 2 + 1] = S[H // 2:]
    assert np.abs(t - u * u * np.fft.irfft2(G, s=(u * H, u * W))).max() <= 1e-14


def _sharpen_ref(L, coef):
    """direct numpy transcription of App. A.4 for interior pixels of one plane (already clamped |t|)"""
    out = np.zeros_like(L)
    Hh, Ww = L.shape
    for yy in range(1, Hh - 1):
        for xx in range(1, Ww - 1):
            n = L[yy - 1:yy + 2, xx - 1:xx + 2]
            cross = [n[0, 1], n[1, 0], n[1, 1], n[1, 2], n[2, 1]]
            mn0, mx0 = min(cross), max(cross)
            mn1, mx1 = min(mn0, n[0, 0], n[0, 2], n[2, 0], n[2, 2]), max(mx0, n[0, 0], n[0, 2], n[2, 0], n[2, 2])
            mn, mx = 0.5 * (mn0 + mn1), 0.5 * (mx0 + mx1)
            a = mn / (1 - mn) if mn < 1 else np.inf
            b = (1 - mx) / mx if mx > 0 else np.inf
            sc = -coef * np.sqrt(min(a, b))
            out[yy, xx] = (n[1, 1] + sc * (n[0, 1] + n[1, 0] + n[1, 2] + n[2, 1])) / (1 + 4 * sc)
    return out


def test_kat7_sharpen_step_edge_and_quirks():
    uW, uH = 24, 10
    R = np.zeros((3, uH, uW))
    R[0, :, uW // 2:] = 0.2                       # step edge, t = 0.8
    R[1] = np.linspace(-0.05, 0.3, uW)[None, :]    # negative -> abs (quirk B4), > 0.25 -> clamp
    R[2] = np.random.default_rng(5).random((uH, uW)) * 0.25
    out = O.sharpen(R, 2.0, 0, 0.2)
    coef = float(np.float32(0.2))
    for c in range(3):
        L = np.clip(np.abs(4.0 * R[c]), 0, 1)
        ref = _sharpen_ref(L, coef)
        assert np.abs(out[c, 1:-1, 1:-1] - ref[1:-1, 1:-1]).max() <= 1e-14
    # quirk B5: right neighbour of x = uW-1 is x = 0 of the next row; left/top clamp
    L = np.clip(np.abs(4.0 * R[2]), 0, 1)
    y = 4
    ext = np.column_stack([L[:, uW - 2], L[:, uW - 1], np.roll(L[:, 0], -1)])   # (y, uW) -> (y+1, 0)
    n = ext[y - 1:y + 2]
    cross = [n[0, 1], n[1, 0], n[1, 1], n[1, 2], n[2, 1]]
    mn0, mx0 = min(cross), max(cross)
    mn1, mx1 = min(mn0, n[0, 0], n[0, 2], n[2, 0], n[2, 2]), max(mx0, n[0, 0], n[0, 2], n[2, 0], n[2, 2])
    mn, mx = 0.5 * (mn0 + mn1), 0.5 * (mx0 + mx1)
    sc = -coef * np.sqrt(min(mn / (1 - mn), (1 - mx) / mx))
    expect = (n[1, 1] + sc * (n[0, 1] + n[1, 0] + n[1, 2] + n[2, 1])) / (1 + 4 * sc)
    assert abs(out[2, y, uW - 1] - expect) <= 1e-14


def test_sharpen_constants_via_percent_f():
    """the shader sees its constants as '%f' text (VkResample.cpp:893-920): s = 0.123456789 acts as 0.123457"""
    R = np.random.default_rng(1).random((3, 8, 8)) * 0.25
    a = O.sharpen(R, 2.0, 0, 0.123456789)
    b = O.sharpen(R, 2.0, 0, 0.123457)
    assert np.array_equal(a, b)


def test_double_mode_differs_from_single_only_in_the_load():
    """-p 1: same restatement, un-rounded load; its shader literals are float constants promoted to double
    (VkResample.cpp:893-920 prints them without a suffix), e.g. s = 0.2 acts as float(0.2) in both modes."""
    rgb = np.random.default_rng(4).integers(0, 256, (12, 20, 3), dtype=np.uint8)
    pre1, out1, _ = O.upscale_rgb8(rgb, 2.0, 1)
    planes = (rgb.astype(np.float64) / 255.0).transpose(2, 0, 1)
    pre0, out0, _ = O.upscale_planes(planes, 2.0, 0)
    assert np.array_equal(pre1, pre0) and np.array_equal(out1, out0)
    pre32, _, _ = O.upscale_rgb8(rgb, 2.0, 0)
    assert 0 < np.abs(pre32 - pre1).max() < 1e-6
    R = np.random.default_rng(1).random((3, 8, 8)) * 0.25
    assert np.array_equal(O.sharpen(R, 2.0, 1, 0.2), O.sharpen(R, 2.0, 0, 0.2))
    x = 0.05                                   # constant neighbourhood: closed form with the FLOAT constant
    got = O.sharpen(np.full((3, 4, 4), x), 2.0, 1, 0.2)[0, 1, 1]
    l = 4.0 * x
    sc = -float(np.float32(0.2)) * np.sqrt(min(l / (1 - l), (1 - l) / l))
    assert got == (l + sc * (l + l + l + l)) / (1.0 + sc * 4.0)


def test_fp16_mode_values_are_halves():
    from vkresample_amd import synth
    rgb = synth.frame(3, 32, 16, "N")
    pre, out, u8 = O.upscale_rgb8(rgb, 2.0, 2)
    assert np.array_equal(pre, pre.astype(np.float16).astype(np.float64))
    assert np.array_equal(out, out.astype(np.float16).astype(np.float64))
    pre0, out0, _ = O.upscale_rgb8(rgb, 2.0, 0)
    assert np.abs(pre - pre0).max() <= 2e-4 and np.abs(out - out0).max() <= 2e-2


# ------------------------------------------------------------------ the non-R2C path
@pytest.mark.parametrize("W,H,u", [(16, 8, 2.0), (20, 12, 2.0), (24, 30, 2.0), (16, 8, 3.0), (12, 8, 2.5), (16, 8, 1.5)])
def test_complex_path_oracle_equals_layout_emulation_and_closed_form(W, H, u):
    """VR:1424 false: full complex transform, four-quadrant shift (VR:527-546, replayed with the shader's own index
    arithmetic), read guards VR:1497-1502; imaginary input parts defined as 0."""
    rng = np.random.default_rng(W * 31 + H)
    planes = rng.random((3, H, W))
    z, out, poison = O.upscale_planes_complex(planes, u)
    assert poison == 0
    assert np.abs(z - E.emulate_complex(planes, u)).max() <= 1e-14
    assert np.abs(z - E.closed_form_complex(planes, u)).max() <= 1e-14
    # sharpen on the modulus: for u = 2 the imaginary part is the Nyquist-row/column remainder, small but not zero
    if u == 2.0:
        assert 1e-4 < np.abs(z.imag).max() < 0.2
    usq = float(np.float32(np.float32(u) * np.float32(u)))
    L = np.clip(np.abs(round(usq, 6) * z), 0.0, 1.0)
    ref = np.stack([_sharpen_ref(L[c], float(np.float32(0.2))) for c in range(3)])
    assert np.abs(out[:, 1:-1, 1:-1] - ref[:, 1:-1, 1:-1]).max() <= 1e-12      # (_sharpen_ref: interior pixels)


def test_complex_path_half_memory():
    """-p 2 beyond the R2C limit (VR:1420-1424): the complex pre-sharpen image is stored as binary16 pairs (the inverse's last
    write, VF:7282-7292), the shader works on f16vec2 in float16_t.  KAT1: a constant image stays constant; the pre-sharpen
    values are binary16 numbers; against the fp32-memory run of the same (binary16-rounded) input the image differs by the
    storage rounding only."""
    rng = np.random.default_rng(5)
    W, H = 4608, 8
    assert O.uses_complex_path(W, H, 2.0, 2)
    const = np.full((H, W, 3), 77, np.uint8)
    pre, out, u8 = O.upscale_rgb8(const, 2.0, 2, 0.2)
    cval = float(np.float16(np.float64(np.float32(np.float16(77))) / 255.0))
    assert np.abs(pre * 4 - cval).max() <= 2.0 ** -11 and np.ptp(out[:, :-1]) == 0.0 and np.all(u8[:-1] == u8[0, 0])
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pre, out, _ = O.upscale_rgb8(rgb, 2.0, 2, 0.2)
    assert np.array_equal(pre, pre.astype(np.float16).astype(np.float64)) and np.array_equal(out, out.astype(np.float16).astype(np.float64))
    planes = np.ascontiguousarray(rgb.transpose(2, 0, 1)).astype(np.float16)
    planes = (planes.astype(np.float32).astype(np.float64) / 255.0).astype(np.float16).astype(np.float64)     # the -p 2 load conversion
    z, _, _ = O.upscale_planes_complex(planes, 2.0, 0)
    ulp = np.maximum(np.abs(z.real), 2.0 ** -14) * 2.0 ** -11
    assert (np.abs(pre - z.real) <= ulp * 1.0001).all()


def test_complex_path_selection_and_kat():
    assert not O.uses_complex_path(4096, 64) and O.uses_complex_path(4608, 64)
    assert not O.uses_complex_path(2048, 64, precision=1) and O.uses_complex_path(2304, 64, precision=1)
    # a constant image stays constant; a cosine below Nyquist is resampled exactly (real result, no imaginary part)
    W, H = 32, 16
    x = np.arange(W)[None, :] + 0 * np.arange(H)[:, None]
    planes = np.stack([np.full((H, W), 0.3), 0.5 + 0.25 * np.cos(2 * np.pi * 3 * x / W), 0.5 + 0.1 * np.cos(2 * np.pi * 5 * x / W)])
    z, _, _ = O.upscale_planes_complex(planes, 2.0)
    X = np.arange(2 * W)[None, :] + 0 * np.arange(2 * H)[:, None]
    assert np.abs(4 * z[0] - 0.3).max() <= 1e-14
    assert np.abs(4 * z[1] - (0.5 + 0.25 * np.cos(2 * np.pi * 3 * X / (2 * W)))).max() <= 1e-14
    # through the size rule: 4608 x 8 takes the complex path inside orc_upscale_planes; pre = real part
    rng = np.random.default_rng(3)
    big = rng.random((3, 8, 4608))
    pre, out, _ = O.upscale_planes(big, 2.0)
    z2, out2, _ = O.upscale_planes_complex(big, 2.0)
    assert np.array_equal(pre, z2.real) and np.array_equal(out, out2)


# ------------------------------------------------------------------ golden vectors (tests/golden/make_golden.py)
@pytest.mark.parametrize("name", ["g16x8_u2_p0", "g20x12_u2_p0", "g20x12_u2_p1", "g64x32_u2_p0", "g64x32_u2_p2", "gsample64_u2_p0"])
def test_golden_vectors(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    pre, out, u8 = O.upscale_rgb8(d["rgb"], float(d["upscale"]), int(d["precision"]), float(d["sharpen"]))
    assert np.abs(pre - d["pre"]).max() <= 1e-13
    assert np.abs(out - d["out"]).max() <= 1e-12
    assert np.array_equal(u8, d["u8"])


def test_config1_fixture_digests():
    """BASELINE config 1 (samples/no_upscaling.png, stb_image-decoded, -u 2 -p 0): the oracle on the committed
    pixels reproduces the digests stored with them (bounded: one 1080p frame, ~2 s)."""
    d = np.load(os.path.join(GOLDEN, "no_upscaling_rgb.npz"))
    assert d["rgb"].shape == (1080, 1920, 3)
    _, out, u8 = O.upscale_rgb8(d["rgb"], 2.0, 0, 0.2)
    assert np.abs(out.mean(axis=(1, 2)) - d["out_plane_means"]).max() <= 1e-12
    assert np.abs(out[:, 1000:1064, 1800:1864] - d["out_crop"]).max() <= 1e-12
    assert np.array_equal(u8[1000:1064, 1800:1864], d["u8_crop"]) and int(u8.astype(np.int64).sum()) == int(d["u8_sum"])


# ------------------------------------------------------------------ the reference's own output (README strips)
README_STRIPS = ["car", "close_people", "distant_people", "skyscraper", "trees"]


@pytest.mark.parametrize("name", README_STRIPS)
def test_oracle_reproduces_reference_output_crops(name):
    """tests/golden/make_readme_crops.py: the FFT panels of the README's comparison strips are crops of images the
    reference itself produced; their NN panels give the exact input pixels of the same window.  The oracle run on
    that input reproduces the reference's pixels to the 8-bit grid (the residue comes from the input outside the
    window, known only to ~2 grey levels), and only with the reference's default sharpen strength.  Compared: every
    panel pixel outside the label corner and a 12-pixel border (O.readme_panel_mask: 68 476 pixels x 3 channels per strip;
    measured mean 0.19-0.30 grey levels, p99 <= 2, max <= 4 -- profiles/r03_b_readme_crops_histograms.txt)."""
    d = np.load(os.path.join(GOLDEN, "readme_%s.npz" % name))
    _, _, u8 = O.upscale_rgb8(d["rgb"], float(d["upscale"]), int(d["precision"]), float(d["sharpen"]))
    st = O.readme_panel_stats(u8, d)
    print("README %s oracle: %s" % (name, st))
    assert st["mean"] <= 0.35 and st["p99"] <= 2 and st["p99.9"] <= 3 and st["max"] <= 5, st
    _, _, weak = O.upscale_rgb8(d["rgb"], 2.0, 0, 0.1)          # discriminating power: -s 0.1 is visibly off
    sw = O.readme_panel_stats(weak, d)
    assert sw["mean"] > 2 * st["mean"] and sw["max"] >= 20 and sw["p99"] >= 5, sw


def _textbook_variant(rgb, sharpen, nyquist_half, dc_leak):
    """the upscale with ONE of the reference's quirks taken out (numpy closed form of oracle/ref_layout_emulation.closed_form, then
    the oracle's own sharpen and 8-bit store): nyquist_half -- the Nyquist row and column of the input spectrum at half weight, as a
    textbook zero-padding would split them (quirk B1 keeps them whole); dc_leak = False -- without the row-pair leak of the DC
    column's imaginary part (quirk B3)"""
    planes = np.stack([np.float32(rgb[:, :, c]) / np.float32(255.0) for c in range(3)]).astype(np.float64)
    C_, H, W = planes.shape
    uW, uH = 2 * W, 2 * H
    out = np.empty((3, uH, uW))
    y = np.arange(uH)
    for c in range(3):
        S = np.fft.rfft2(planes[c])
        if nyquist_half:
            S[:, W // 2] *= 0.5
            S[H // 2, :] *= 0.5
        G = np.zeros((uH, uW // 2 + 1), dtype=np.complex128)
        G[:H // 2, :W // 2 + 1] = S[:H // 2]
        G[uH - H // 2:, :W // 2 + 1] = S[H // 2:]
        if nyquist_half:
            G[H // 2, :W // 2 + 1] = S[H // 2]                     # the split Nyquist row on both sides of the padding
        g = np.fft.irfft2(G, s=(uH, uW))
        if dc_leak:
            q = S[H // 2, 0].real * np.sin(np.pi * y / 2.0) / (uH * uW)
            g[0::2, :] -= q[1::2, None]
            g[1::2, :] += q[0::2, None]
        out[c] = g
    sh = O.sharpen(out, 2.0, 0, sharpen)
    return np.clip(np.floor(255.0 * np.clip(sh, 0.0, None)), 0, 255).astype(np.uint8).transpose(1, 2, 0)


def _panel_diff_stats(u8, panel, Yo, Xo, mask):
    diff = np.abs(u8[Yo:Yo + 300, Xo:Xo + 300].astype(np.int64) - panel.astype(np.int64))[mask]
    return float(diff.mean()), float(np.percentile(diff, 99)), int(diff.max()), np.bincount(np.minimum(diff.ravel(), 7), minlength=8).tolist()


def test_readme_pin_border_and_sharpen_strengths():
    """VERDICT r5 #8, on the committed 512x512 fixtures: (1) the pixels the parity test masks out -- the 12-pixel border of a panel (the
    input beyond the window is known to ~2 grey levels only), split into its inner 8 and its outer 4 pixels, and the label corner --
    against the oracle with looser bounds, histograms printed; (2) the sharpen strength: -s 0.1 and -s 0.3 are told apart from the
    reference's default 0.2 by a wide margin.  (The reference's quirks B1 -- Nyquist row and column at full weight -- and B3 -- the
    row-pair leak of Im DC -- cannot be judged on a 512x512 window: its Nyquist bins are not the frame's.  On the whole frame they
    can, weakly: next test.)  profiles/r06_i_readme_pin.txt holds the printed table."""
    m_in = O.readme_panel_mask()
    ring_in = np.ones((300, 300), bool)
    ring_in[12:288, 12:288] = False
    ring_in[:94, 166:] = False                                         # (the label corner and its 12-pixel surroundings)
    ring_out = ring_in.copy()
    ring_out[4:296, 4:296] = False
    ring_in &= ~ring_out
    corner = np.zeros((300, 300), bool)
    corner[:82, 178:] = True
    for name in README_STRIPS:
        d = np.load(os.path.join(GOLDEN, "readme_%s.npz" % name))
        Yo, Xo = int(d["Yo"]), int(d["Xo"])
        res = {}
        for s_ in (0.2, 0.1, 0.3):
            u8 = O.upscale_rgb8(d["rgb"], 2.0, 0, s_)[2]
            res[s_] = _panel_diff_stats(u8, d["fft_panel"], Yo, Xo, m_in)
            print("README pin %-15s -s %.1f interior (68 476 px)     mean %.3f p99 %3.0f max %3d hist %s" % ((name, s_) + res[s_]))
            if s_ == 0.2:
                for rn, mask in (("border, pixels 4..12", ring_in), ("border, pixels 0..4 ", ring_out), ("label corner        ", corner)):
                    st = _panel_diff_stats(u8, d["fft_panel"], Yo, Xo, mask)
                    print("README pin %-15s -s 0.2 %s        mean %.3f p99 %3.0f max %3d hist %s" % ((name, rn) + st))
                    if mask is ring_in:
                        assert st[0] <= 0.6 and st[1] <= 3 and st[2] <= 10, (name, st)      # as good as the interior, a little noisier
                    if mask is ring_out:
                        assert st[0] <= 6.0, (name, st)                                      # the outermost pixels: the unknown surroundings show
        assert res[0.1][0] > 2 * res[0.2][0] and res[0.3][0] > 10 * res[0.2][0] and res[0.3][2] > 100


@pytest.mark.skipif(not os.path.exists("/root/reference/samples/car.png"), reason="needs the reference's samples")
def test_readme_pin_quirks_on_the_whole_frame():
    """... and what the artefact says about quirks B1 and B3 (here only: the whole 2048x1152 frame is rebuilt from the reference's
    samples).  With either quirk replaced by its textbook form the agreement with the reference's own pixels gets WORSE, in every
    strip -- by hundredths of a grey level in the mean, one to three thousand exact pixels of 205 428: the reference's output carries
    both quirks, as the oracle restates them; the margin is small because the effects are (the leak is 1.4e-4 of full scale)."""
    sys.path.insert(0, GOLDEN)
    import make_readme_crops as M
    m_in = O.readme_panel_mask()
    for name in ("car", "distant_people"):
        b = M.build(name)
        Yo, Xo = 2 * b["yA"] + b["py"], 2 * b["xA"] + b["px"]
        st = {"oracle (B1 + B3)": _panel_diff_stats(O.upscale_rgb8(b["rgb"], 2.0, 0, 0.2)[2], b["fft"], Yo, Xo, m_in),
              "Nyquist bins halved (no B1)": _panel_diff_stats(_textbook_variant(b["rgb"], 0.2, True, True), b["fft"], Yo, Xo, m_in),
              "no DC leak (no B3)": _panel_diff_stats(_textbook_variant(b["rgb"], 0.2, False, False), b["fft"], Yo, Xo, m_in)}
        for k, v in st.items():
            print("README pin %-15s whole frame, %-28s mean %.3f p99 %3.0f max %3d hist %s" % ((name, k) + v))
        assert st["oracle (B1 + B3)"][0] < st["Nyquist bins halved (no B1)"][0] and st["oracle (B1 + B3)"][0] < st["no DC leak (no B3)"][0]
        assert st["oracle (B1 + B3)"][3][0] > st["Nyquist bins halved (no B1)"][3][0] and st["oracle (B1 + B3)"][3][0] > st["no DC leak (no B3)"][3][0]


@pytest.mark.skipif(not os.path.exists("/root/reference/samples/car.png"), reason="needs the reference's samples")
def test_oracle_reproduces_reference_output_whole_frame():
    """the same with the whole 2048x1152 frame instead of the committed 512x512 windows (here only)"""
    sys.path.insert(0, GOLDEN)
    import make_readme_crops as M
    b = M.build("car")
    assert b["dup"] > 0.99 and b["rms"] < 1.5
    _, _, u8 = O.upscale_rgb8(b["rgb"], 2.0, 0, 0.2)
    mean, p99, mx = M.compare(u8, b["fft"], 2 * b["yA"] + b["py"], 2 * b["xA"] + b["px"])
    assert mean <= 0.2 and p99 <= 1 and mx <= 2, (mean, p99, mx)


@pytest.mark.parametrize("W,H", [(32, 16), (24, 20), (64, 12)])
def test_x_split_identities_of_the_u2_inverse(W, H):
    """The x direction of the u = 2 inverse in the folded form DESIGN.md section 4 prices ("Why the x direction is not split"; VERDICT r3
    next 1 asked for the derivation before any kernel): of every output row, the even and the odd columns are two real transforms
    of length W of the row's half spectrum Z[0..W/2] -- the odd ones behind the phase exp(2 pi i k / 2W), the Nyquist column (quirk
    B1: kept at full weight on both sides) folded in as 2 Re Z[W/2] resp. -2 Im Z[W/2] -- and on the EVEN rows the even columns are the
    input image itself plus (-1)^x alt_y / W (B1) plus one constant per row (the DC leak between the rows of a C2R pair, B2 + B3):
    three complex transforms of length W per four output rows instead of two of length 2W.  Checked against the oracle's pre-sharpen
    image (the reference's literal kernel sequence)."""
    rng = np.random.default_rng(W * 1000 + H)
    x = rng.random((3, H, W))
    pre, _, _ = O.upscale_planes(x, 2.0, 1)                  # fp64 arithmetic: g, before the sharpen shader multiplies by u^2
    uW = 2 * W
    k = np.arange(W // 2 + 1)
    for c in range(3):
        for r in range(2 * H - 1):                           # (the last output row is excluded from parity everywhere: quirk B5)
            Z = np.fft.rfft(pre[c, r])                       # the row's spectrum: kx = 0..uW/2, zero beyond W/2
            assert np.abs(Z[W // 2 + 1:]).max() <= 1e-12
            Xe = Z[:W // 2 + 1].copy()
            Xe[W // 2] = 2 * Z[W // 2].real
            Xo = Z[:W // 2 + 1] * np.exp(2j * np.pi * k / uW)
            Xo[W // 2] = -2 * Z[W // 2].imag
            assert np.abs(pre[c, r, 0::2] - 0.5 * np.fft.irfft(Xe, n=W)).max() <= 1e-13
            assert np.abs(pre[c, r, 1::2] - 0.5 * np.fft.irfft(Xo, n=W)).max() <= 1e-13
        alt = (x[c] * (-1.0) ** np.arange(W)).sum(axis=1)    # Re F[W/2] of every input row (its imaginary part is 0)
        res = 4 * pre[c, 0:2 * H - 1:2, 0::2] - x[c, :H] - np.outer(alt[:H], (-1.0) ** np.arange(W)) / W
        assert np.abs(res - res[:, :1]).max() <= 1e-12      # one constant per even row ...
        s_nyq = (x[c] * ((-1.0) ** np.arange(H))[:, None]).sum()          # S[H/2, 0]
        assert np.abs(np.abs(res[:, 0]) - abs(s_nyq) / (W * H)).max() <= 1e-12 and np.abs(res[0::2, 0] + res[1::2, 0]).max() <= 1e-12   # ... +- S[H/2,0] / WH, alternating
