"""CPU suite: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side C/C++ that sees untrusted or
irregular input: the CLI's PNG decoder fed malformed files (the reference's stb_image rejects
them; `-i` / `-ifolder` files are untrusted), and the CPU oracle over a sweep of odd configurations."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _png(w, h, depth, ctype, raw=None, plte=None, comp=0, filt=0, inter=0, ihdr_len=13):
    ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, comp, filt, inter)[:ihdr_len]
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr)
    if plte is not None:
        out += _chunk(b"PLTE", plte)
    if raw is None:
        nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype, 1)
        rb = (w * nch * max(depth, 1) + 7) // 8
        raw = (b"\x00" + b"\x55" * rb) * h
    return out + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b"")


@pytest.fixture(scope="module")
def png_driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    exe = str(d / "png_driver")
    cli = os.path.join(ROOT, "vkresample_amd", "csrc", "cli")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + ["-I", cli, os.path.join(ROOT, "tests", "sanitize", "png_driver.cpp"),
                                                         os.path.join(cli, "png_codec.cpp"), "-o", exe, "-lz"])
    return exe


def test_png_decoder_rejects_malformed_headers_under_asan_ubsan(png_driver, tmp_path):
    bad = {
        "depth0": _png(4, 4, 0, 0), "depth32": _png(4, 4, 32, 2), "depth3": _png(8, 2, 3, 0), "depth5": _png(8, 2, 5, 0),
        "depth7": _png(8, 2, 7, 3, plte=b"\x01\x02\x03"), "rgb_depth4": _png(4, 4, 4, 2), "rgba_depth2": _png(4, 4, 2, 6),
        "pal_depth16": _png(4, 4, 16, 3, plte=b"\x00" * 6), "ctype5": _png(4, 4, 8, 5), "ctype7": _png(4, 4, 8, 7),
        "plte1": _png(4, 4, 8, 3, plte=b"\x01"), "plte2": _png(4, 4, 8, 3, plte=b"\x01\x02"), "plte4": _png(4, 4, 8, 3, plte=b"\x01\x02\x03\x04"),
        "plte_missing": _png(4, 4, 8, 3), "huge": _png(0x7FFFFFFF, 0x7FFFFFFF, 8, 2, raw=b"\x00" * 16),
        "huge2": _png(1 << 20, 1 << 20, 16, 6, raw=b"\x00" * 16), "zero_w": _png(0, 4, 8, 2, raw=b"\x00"),
        "comp1": _png(4, 4, 8, 2, comp=1), "filt1": _png(4, 4, 8, 2, filt=1), "inter2": _png(4, 4, 8, 2, inter=2),
        "short_ihdr": _png(4, 4, 8, 2, ihdr_len=9), "bad_filter_byte": _png(4, 2, 8, 0, raw=b"\x07" + b"\x00" * 4 + b"\x00" * 5),
        "short_idat": _png(16, 16, 8, 2, raw=b"\x00" * 20), "long_idat": _png(2, 2, 8, 0, raw=b"\x00" * 64),
        "no_idat": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)) + _chunk(b"IEND", b""),
        "plte_first": b"\x89PNG\r\n\x1a\n" + _chunk(b"PLTE", b"\x00" * 3), "truncated": _png(4, 4, 8, 2)[:40], "not_png": b"GIF89a" + b"\x00" * 32,
        "empty": b"",
    }
    good = {
        "grey1": _png(9, 3, 1, 0), "grey2": _png(9, 3, 2, 0), "grey4": _png(9, 3, 4, 0), "grey16": _png(5, 3, 16, 0),
        "pal2": _png(7, 3, 2, 3, plte=b"\x10\x20\x30" * 2), "pal_index_beyond_palette": _png(7, 3, 8, 3, plte=b"\x10\x20\x30"),
        "ga8": _png(5, 3, 8, 4), "rgb16": _png(5, 3, 16, 2), "rgba8_adam7_1x1": _png(1, 1, 8, 6, inter=1, raw=b"\x00\x01\x02\x03\x04"),
    }
    names = []
    for k, v in {**bad, **good}.items():
        p = tmp_path / (k + ".png")
        p.write_bytes(v)
        names.append(str(p))
    r = subprocess.run([png_driver] + names + [str(tmp_path / "missing.png")], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    lines = {os.path.basename(l.split(":")[0])[:-4]: l for l in r.stdout.splitlines()}
    for k in bad:
        assert "rejected" in lines[k], lines[k]
    for k in good:
        assert " ok" in lines[k], lines[k]
    assert "rejected" in lines["missing"]


def test_png_roundtrip_under_asan_ubsan(png_driver, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for name, (w, h) in {"a": (1, 1), "b": (2, 5), "c": (67, 33)}.items():
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / (name + ".png"))
    Image.fromarray(rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)).save(tmp_path / "d.png")
    # the encoder's two paths: smooth content over several 256 KB blocks (Huffman-only deflate), flat content (zlib run lengths)
    yy, xx = np.mgrid[0:300, 0:500]
    smooth = np.stack([128 + 100 * np.sin(xx / 37.0 + yy / 91.0), 128 + 90 * np.cos(xx / 53.0), 40 + 0.3 * yy + 0.1 * xx], axis=2)
    Image.fromarray(np.clip(smooth + rng.normal(0, 2, smooth.shape), 0, 255).astype(np.uint8)).save(tmp_path / "e.png")
    flat = np.full((90, 120, 3), 200, np.uint8)
    flat[20:40, 30:60] = (10, 250, 0)
    Image.fromarray(flat).save(tmp_path / "f.png")
    Image.fromarray((rng.integers(0, 2, (64, 80, 3)) * 255).astype(np.uint8)).save(tmp_path / "g.png")
    files = [str(tmp_path / (n + ".png")) for n in "abcdefg"]
    r = subprocess.run([png_driver] + files, capture_output=True, text=True, env=ENV)
    assert r.returncode == 0 and r.stdout.count(" ok") == 7, r.stdout + r.stderr[-3000:]
    for f in files:
        a = np.asarray(Image.open(f).convert("RGB"))
        b = np.asarray(Image.open(f + ".out.png"))
        assert np.array_equal(a, b)


def test_png_encoder_huffman_deflate_under_asan_ubsan(tmp_path):
    """The encoder's own deflate (Huffman-only dynamic blocks): code lengths complete and within 15 / 7 bits for Fibonacci
    frequencies (trees deeper than the limit), single symbols and 3000 random sets; streams of flat, uniform, geometric and
    Fibonacci-distributed bytes and of lengths around the 256 KB block size inflate (zlib) to the input."""
    exe = str(tmp_path / "huffman_driver")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "sanitize", "huffman_driver.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_png_decoder_fast_inflate_against_zlib_under_asan_ubsan(tmp_path):
    """The PNG reader's own zlib-stream decoder (csrc/cli/inflate_fast.hpp), differentially against zlib: 200 streams (eight kinds
    of data x five levels x five strategies incl. stored, fixed and Huffman-only blocks) decode to the input; wrong expected lengths
    are refused; of ~4300 truncated / bit-flipped streams and 2000 garbage inputs every one is either refused (zlib then decides)
    or decodes to exactly what zlib makes of it."""
    exe = str(tmp_path / "inflate_driver")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "sanitize", "inflate_driver.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_driver")
    subprocess.check_call(["gcc", "-std=c11", "-fopenmp", "-fno-fast-math", "-ffp-contract=off",] + SAN +
                          [os.path.join(ROOT, "tests", "sanitize", "oracle_driver.c"), os.path.join(ROOT, "oracle", "fftup_oracle.c"),
                           "-o", exe, "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(ENV, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "sweep ok" in r.stdout, r.stdout + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
