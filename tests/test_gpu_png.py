"""GPU: the device-side PNG encoder (fftup_submit_png / fftup_wait_png, csrc/kernels_png.hpp).  A PNG stream has no reference
bytes to be equal to; what must hold: the file is a valid PNG (PIL decodes it: chunk CRCs, zlib's Huffman-table and Adler-32
checks) and its pixels are byte-identical to what fftup_submit_rgb8 returns for the same frame."""
import io
import struct
import threading
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(W, H, n):
    from vkresample_amd import synth
    out = []
    for k in range(n):
        kind = k % 5
        if kind == 0:
            f = synth.frame(500 + k, W, H, "N")
        elif kind == 1:
            f = synth.frame(500 + k, W, H, "U")                       # noise: near-uniform residuals
        elif kind == 2:
            f = np.full((H, W, 3), (17, 200, 90), np.uint8)          # flat: one symbol carries everything (codes of 1 bit)
        elif kind == 3:
            yy, xx = np.mgrid[0:H, 0:W]
            f = np.stack([xx * 255 // max(W - 1, 1), yy * 255 // max(H - 1, 1), (xx + yy) % 256], axis=2).astype(np.uint8)
        else:
            f = synth.frame(500 + k, W, H, "N")
            f[H // 3: 2 * H // 3] = 0                                  # black band: blocks with very different statistics
        out.append(np.ascontiguousarray(f))
    return out


def _decode(png_bytes):
    from PIL import Image
    Image.open(io.BytesIO(png_bytes)).verify()
    return np.asarray(Image.open(io.BytesIO(png_bytes)).convert("RGB"))


def _chunks(png):
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, out = 8, []
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        typ, data = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + data) & 0xFFFFFFFF == crc, typ
        out.append((typ, data))
        pos += 12 + n
    return out


@pytest.mark.parametrize("W,H,u,precision,flags,ring", [(256, 128, 2.0, 0, 0, 3), (240, 126, 2.0, 0, 0, 2), (60, 36, 2.0, 0, 0, 1),
                                                        (512, 256, 2.0, 2, 6, 4), (96, 64, 3.0, 0, 0, 2), (128, 64, 1.5, 0, 0, 2),
                                                        (2048, 1024, 2.0, 0, 0, 2), (1920, 1080, 2.0, 2, 2, 2)])
def test_png_from_the_device_decodes_to_the_frame(W, H, u, precision, flags, ring):
    import vkresample_amd as v
    n = 2 * ring + 1 if W < 1000 else 3
    frames = _frames(W, H, max(n, 5))[:n] if W < 1000 else _frames(W, H, 5)[:3]
    with v.Upscaler(W, H, u, precision, 0.2, 0, flags, ring) as up:
        uW, uH = up.out_width, up.out_height
        want = []
        out = np.empty((uH, uW, 3), np.uint8)
        for f in frames:
            up.wait(up.submit_rgb8(f, out))
            want.append(out.copy())
        buf = v.PinnedArray((up.png_bound(),))
        sizes = []
        for k, f in enumerate(frames):
            t = up.submit_png(f, buf.array if k % 2 else None)          # odd frames: the GPU delivers the stream into the buffer itself
            nbytes = up.wait_png(t, buf.array)
            png = bytes(buf.array[:nbytes])
            ch = _chunks(png)
            assert [c[0] for c in ch] == [b"IHDR", b"IDAT", b"IEND"]
            assert struct.unpack(">IIBBBBB", ch[0][1]) == (uW, uH, 8, 2, 0, 0, 0)
            raw = zlib.decompress(ch[1][1])                            # Huffman tables, block structure, Adler-32
            assert len(raw) == uH * (3 * uW + 1) and set(raw[::3 * uW + 1]) <= {0, 1, 2, 3, 4}
            img = _decode(png)
            assert img.shape == want[k].shape and np.array_equal(img, want[k]), k
            sizes.append(nbytes)
        print("MEASURED png %dx%d: %s bytes per frame, %d raw" % (uW, uH, sizes, uH * uW * 3))
        assert sizes[2] < 0.2 * uH * uW * 3 if len(sizes) > 2 and W < 1000 else True      # the flat frame: one bit per byte + headers
        # tickets of the two kinds share the queue; a buffer that is too small is the caller's error: the stream stays on the device
        # and the ticket collectable -- with a proper buffer it delivers its frame
        t = up.submit_png(frames[0])
        with pytest.raises(v.FftupError):
            up.wait_png(t, buf.array[:64])
        assert np.array_equal(_decode(bytes(buf.array[:up.wait_png(t, buf.array)])), want[0])
        t2 = up.submit_png(frames[1 % len(frames)])
        nbytes = up.wait_png(t2, buf.array)
        assert np.array_equal(_decode(bytes(buf.array[:nbytes])), want[1 % len(frames)])
        with pytest.raises(v.FftupError):
            up.wait_png(t2, buf.array)                                  # collected already
        with pytest.raises(v.FftupError):
            up.submit_png(frames[0], np.empty(up.png_bound(), np.uint8))   # a destination the GPU cannot write to (not page-locked)
        other = v.PinnedArray((up.png_bound(),))
        t5 = up.submit_png(frames[0], buf.array)
        with pytest.raises(v.FftupError):
            up.wait_png(t5, other.array)                                # the file is in the buffer named at submission: named again, it is collected
        assert np.array_equal(_decode(bytes(buf.array[:up.wait_png(t5, buf.array)])), want[0])
        other.close()
        t3 = up.submit_png(frames[0])
        with pytest.raises(v.FftupError):
            up.wait(t3)                                                 # the pixel path's wait does not collect a PNG
        assert np.array_equal(_decode(bytes(buf.array[:up.wait_png(t3, buf.array)])), want[0])
        t4 = up.submit_rgb8(frames[0], out)
        with pytest.raises(v.FftupError):
            up.wait_png(t4, buf.array)                                  # ... and the other way round
        up.wait(t4)
        assert np.array_equal(out, want[0])
        buf.close()


def test_png_over_many_sizes():
    """row lengths and heights of every residue (rows of 49 .. 24 001 bytes, one to dozens of deflate blocks, threads with and
    without symbols, a last block of one row): one natural and one flat frame per size"""
    import vkresample_amd as v
    from vkresample_amd import synth
    sizes = [(16, 8, 2.0), (18, 10, 3.0), (36, 16, 2.5), (50, 18, 2.0), (98, 54, 2.0), (128, 100, 1.5), (250, 250, 2.0), (486, 98, 2.0),
             (1000, 490, 2.0), (2000, 36, 4.0), (80, 1250, 2.0), (4000, 16, 2.0), (162, 162, 3.0), (100, 8, 1.0),
             (4608, 16, 2.0)]                                   # 9216-pixel rows: the non-R2C path's image
    for (W, H, u) in sizes:
        with v.Upscaler(W, H, u, 0, 0.2, 0, 0, 2) as up:
            uW, uH = up.out_width, up.out_height
            buf = np.empty(up.png_bound(), np.uint8)
            out = np.empty((uH, uW, 3), np.uint8)
            for f in (synth.frame(900 + W + H, W, H, "N"), np.full((H, W, 3), 131, np.uint8)):
                up.wait(up.submit_rgb8(f, out))
                n = up.wait_png(up.submit_png(f), buf)
                png = bytes(buf[:n])
                _chunks(png)
                assert np.array_equal(_decode(png), out), (W, H, u)


def test_png_is_refused_for_double_precision_plans():
    import vkresample_amd as v
    from vkresample_amd import synth
    with v.Upscaler(64, 32, 2.0, 1, 0.2, 0, 0, 1) as up:
        with pytest.raises(v.FftupError) as e:
            up.submit_png(synth.frame(1, 64, 32, "N"))
        assert e.value.code == 3                                    # FFTUP_E_UNSUPPORTED_PRECISION: the CLI encodes -p 1 frames on the host


def test_png_tickets_from_several_threads():
    """four threads, one plan with two slots: every thread's PNGs decode to its frames (a slot with an uncollected stream makes
    the next submission of that slot wait)"""
    import vkresample_amd as v
    W, H, T, per = 256, 128, 4, 5
    frames = _frames(W, H, T * per)
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 2) as up:
        want, out = [], np.empty((2 * H, 2 * W, 3), np.uint8)
        for f in frames:
            up.wait(up.submit_rgb8(f, out))
            want.append(out.copy())
        errors = []

        def worker(t):
            try:
                buf = v.PinnedArray((up.png_bound(),))
                for g in range(t, T * per, T):
                    n = up.wait_png(up.submit_png(frames[g], buf.array if g % 3 else None), buf.array)
                    if not np.array_equal(_decode(bytes(buf.array[:n])), want[g]):
                        errors.append((t, g, "pixels differ"))
                buf.close()
            except Exception as e:                                     # noqa: BLE001
                errors.append((t, repr(e)))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors


def test_a_thread_waiting_for_its_own_png_ticket_is_an_error_not_a_hang(monkeypatch):
    """a submission that comes round to a ring slot whose PNG ticket has not been collected waits for the collector --
    unless the collector is the submitting thread itself, which would wait forever: FFTUP_E_WOULD_BLOCK (9), after a bounded wait in
    which no other thread collected anything on the plan (FFTUP_SELF_WAIT_MS, two seconds by default).  ring = 1 (the
    default): the second fftup_submit_png of one thread; ring = 2: the third; fftup_submit_rgb8 behind an open ticket likewise.
    After fftup_wait_png the slot is free again and nothing was lost."""
    import vkresample_amd as v
    from vkresample_amd import synth
    monkeypatch.setenv("FFTUP_SELF_WAIT_MS", "50")
    W, H = 128, 64
    f = [synth.frame(40 + k, W, H, "N") for k in range(3)]
    for ring in (1, 2):
        with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, ring) as up:
            buf = v.PinnedArray((up.png_bound(),))
            out = v.PinnedArray((2 * H, 2 * W, 3))
            want = []
            for x in f:
                up.wait(up.submit_rgb8(x, out.array))
                want.append(out.array.copy())
            tickets = [up.submit_png(f[k]) for k in range(ring)]
            for call in (lambda: up.submit_png(f[2]), lambda: up.submit_rgb8(f[2], out.array)):
                with pytest.raises(v.FftupError) as e:
                    call()
                assert e.value.code == 9 and "fftup_wait_png" in str(e.value)
            for k, t in enumerate(tickets):                              # nothing was lost: every open ticket still delivers its frame
                assert np.array_equal(_decode(bytes(buf.array[:up.wait_png(t, buf.array)])), want[k])
            assert np.array_equal(_decode(bytes(buf.array[:up.wait_png(up.submit_png(f[2]), buf.array)])), want[2])
            buf.close()
            out.close()


def test_producer_and_consumer_threads_with_more_frames_than_ring_slots():
    """ONE thread submits (fftup_submit_png), ANOTHER collects (fftup_wait_png), more frames than ring slots.  Every held
    slot then belongs to the producer -- which round 5 took for "would wait for itself" and answered with FFTUP_E_WOULD_BLOCK; the
    submission has to wait for the consumer instead.  ring = 2, 14 frames, the consumer deliberately late: all files decode."""
    import queue
    import time
    import vkresample_amd as v
    W, H, N = 128, 64, 14
    frames = _frames(W, H, 7)
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 2) as up:
        want, out = [], np.empty((2 * H, 2 * W, 3), np.uint8)
        for f in frames:
            up.wait(up.submit_rgb8(f, out))
            want.append(out.copy())
        tickets, errors, got = queue.Queue(), [], []

        def producer():
            try:
                for i in range(N):
                    tickets.put((up.submit_png(frames[i % 7]), i % 7))
            except Exception as e:                                     # noqa: BLE001
                errors.append(("producer", repr(e)))
            tickets.put(None)

        def consumer():
            try:
                buf = np.empty(up.png_bound(), np.uint8)
                time.sleep(0.3)                                        # the producer has filled the ring long before the first collection
                while True:
                    item = tickets.get(timeout=60)
                    if item is None:
                        return
                    n = up.wait_png(item[0], buf)
                    got.append(np.array_equal(_decode(bytes(buf[:n])), want[item[1]]))
            except Exception as e:                                     # noqa: BLE001
                errors.append(("consumer", repr(e)))

        th = [threading.Thread(target=producer, daemon=True), threading.Thread(target=consumer, daemon=True)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=120)
        assert not any(x.is_alive() for x in th), "producer and consumer wait for each other"
        assert not errors and len(got) == N and all(got), (errors, got)


def test_a_callers_mistake_in_wait_png_keeps_the_ticket():
    """a buffer too small (or not the one named at submission) is the CALLER's error -- the encoded stream sits intact on
    the device, so the ticket stays collectable; only device errors and an overflowed stream void it."""
    import vkresample_amd as v
    from vkresample_amd import synth
    W, H = 128, 64
    f = synth.frame(3, W, H, "N")
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 1) as up:
        out = np.empty((2 * H, 2 * W, 3), np.uint8)
        up.wait(up.submit_rgb8(f, out))
        t = up.submit_png(f)
        with pytest.raises(v.FftupError) as e:
            up.wait_png(t, np.empty(64, np.uint8))
        assert e.value.code == 1 and "too small" in str(e.value)
        buf = np.empty(up.png_bound(), np.uint8)
        n = up.wait_png(t, buf)                                       # the same ticket, a proper buffer: the file
        assert np.array_equal(_decode(bytes(buf[:n])), out)


def test_device_png_size_against_the_reference_writer():
    """VERDICT r4 #6 for the device-side encoder: the same 8-bit frames written by the reference's stbi_write_png
    (oracle/_ref/libref_host.so = the reference's stb_image_write compiled in place; built where /root/reference exists, it travels
    with the snapshot).  Natural frames: the device's Huffman-only stream with per-block dynamic codes is smaller than
    stb's LZ + fixed codes."""
    import ctypes as C
    import os
    import tempfile
    import vkresample_amd as v
    from vkresample_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    refso = os.path.join(root, "oracle", "_ref", "libref_host.so")
    if not os.path.exists(refso):
        pytest.skip("oracle/_ref/libref_host.so not built (no reference tree where build() ran)")
    ref = C.CDLL(refso)
    frames = {"README car strip": np.load(os.path.join(root, "tests", "golden", "readme_car.npz"))["rgb"],
              "no_upscaling.png crop": np.ascontiguousarray(np.load(os.path.join(root, "tests", "golden", "no_upscaling_rgb.npz"))["rgb"][284:796, 448:1472]),
              "synthetic N": synth.frame(1, 1024, 512, "N")}
    with tempfile.TemporaryDirectory() as tmp:
        for name, rgb in frames.items():
            H, W, _ = rgb.shape
            with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 1) as up:
                buf = np.empty(up.png_bound(), np.uint8)
                n = up.wait_png(up.submit_png(rgb), buf)
                img = _decode(bytes(buf[:n]))
            img = np.ascontiguousarray(img)
            path = os.path.join(tmp, "stb.png")
            assert ref.ref_png_write_rgb(path.encode(), img.shape[1], img.shape[0], img.ctypes.data_as(C.POINTER(C.c_ubyte)))
            b = os.path.getsize(path)
            print("MEASURED device png size %s (%dx%d): fftup_submit_png %d bytes, stbi_write_png %d bytes, ratio %.3f" % (name, img.shape[1], img.shape[0], n, b, n / b))
            assert n <= 1.0 * b, (name, n, b)


def test_a_stream_that_would_not_fit_is_refused_on_the_device(monkeypatch):
    """the stream buffer's size rests on an argument about the length-limited Huffman codes; the device checks the real
    size against the capacity before a bit is packed.  With the capacity knob of the test build set below what a noise frame needs,
    fftup_wait_png reports FFTUP_E_OVERFLOW (10) with the size that was needed, nothing is written (the words behind the
    pretended capacity stay zero: the buffer is zeroed per frame), the slot is free again, and a frame that fits still encodes."""
    import vkresample_amd as v
    from vkresample_amd import _lib, synth
    monkeypatch.setenv("FFTUP_LIBRARY", _lib.KNOBS_LIB_PATH)
    monkeypatch.setenv("FFTUP_EXPERIMENT", "png_capacity=60000")
    W, H = 128, 64
    noise, flat = synth.frame(7, W, H, "U"), np.full((H, W, 3), 90, np.uint8)
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 1) as up:
        buf = np.empty(up.png_bound(), np.uint8)
        out = np.empty((2 * H, 2 * W, 3), np.uint8)
        t = up.submit_png(noise)                                  # ~ 190 KB of residuals: does not fit 60 000 bytes
        with pytest.raises(v.FftupError) as e:
            up.wait_png(t, buf)
        assert e.value.code == 10 and "exceeds" in str(e.value)
        n = up.wait_png(up.submit_png(flat), buf)                 # a flat frame fits; the slot was released by the failed wait
        up.wait(up.submit_rgb8(flat, out))
        assert n < 60000 and np.array_equal(_decode(bytes(buf[:n])), out)


def test_threads_that_keep_png_tickets_open_do_not_wait_in_a_circle():
    """Round 4's queue took its ring slots strictly in turn: two threads that each keep a PNG ticket open while submitting the
    next could end up waiting for each other's slots (found by tools/png_threads.py: the run never ended).  Submissions now take
    the next slot WITHOUT an uncollected stream: with ring = threads x depth and depth tickets open per thread a slot is always
    free.  Four threads, two tickets open each, 40 frames each -- every file decodes to its frame, and the run ends."""
    import vkresample_amd as v
    W, H, T, depth, per = 128, 64, 4, 2, 40
    frames = _frames(W, H, 10)
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, T * depth) as up:
        want, out = [], np.empty((2 * H, 2 * W, 3), np.uint8)
        for f in frames:
            up.wait(up.submit_rgb8(f, out))
            want.append(out.copy())
        errors, done = [], []

        def worker(t):
            try:
                bufs = [v.PinnedArray((up.png_bound(),)) for _ in range(depth)]
                tk = [(up.submit_png(frames[(t + k) % 10], bufs[k].array), (t + k) % 10) for k in range(depth)]
                for i in range(per):
                    k = i % depth
                    n = up.wait_png(tk[k][0], bufs[k].array)
                    if not np.array_equal(_decode(bytes(bufs[k].array[:n])), want[tk[k][1]]):
                        errors.append((t, i, "pixels differ"))
                    g = (t + i + depth) % 10
                    tk[k] = (up.submit_png(frames[g], bufs[k].array), g)
                for k in range(depth):
                    up.wait_png(tk[(per + k) % depth][0], bufs[(per + k) % depth].array)
                for b in bufs:
                    b.close()
                done.append(t)
            except Exception as e:                                     # noqa: BLE001
                errors.append((t, repr(e)))

        th = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=120)
        assert not any(x.is_alive() for x in th), "threads still waiting for each other's ring slots"
        assert not errors and sorted(done) == list(range(T)), errors
