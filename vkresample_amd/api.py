"""Host-side mirror of the reference's per-image pipeline surface, on top of the C ABI.

Reference (VkResample.cpp): launchResample() VR:1280-1780 builds the plan and runs, per file,
pack+upload (VR:1636-1688) -> performVulkanUpscale(numIter) (VR:1249-1279, VR:1692) -> download+unpack
(VR:1697-1748).  `Upscaler` keeps those three steps and their argument meaning (`upscale`, `precision`,
`sharpen`, `num_iter` = -u/-p/-s/-n).  All compute happens in libfftup.so (HIP); there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib


class FftupError(RuntimeError):
    def __init__(self, code, where):
        lib = _lib.load()
        self.code = code
        detail = lib.fftup_last_error().decode(errors="replace")
        super().__init__("%s failed: %s (%d)%s" % (where, lib.fftup_strerror(code).decode(), code,
                                                   ": " + detail if detail else ""))


def _check(code, where):
    if code != 0:
        raise FftupError(code, where)


def device_count():
    return _lib.load().fftup_device_count()


def device_name(device=0):
    buf = C.create_string_buffer(256)
    _check(_lib.load().fftup_device_name(device, buf, 256), "fftup_device_name")
    return buf.value.decode()


def device_pci_bus_id(device=0):
    buf = C.create_string_buffer(64)
    _check(_lib.load().fftup_device_pci_bus_id(device, buf, 64), "fftup_device_pci_bus_id")
    return buf.value.decode()


class Upscaler:
    """One plan = one (width, height, upscale, precision, sharpen, device) configuration."""

    def __init__(self, width, height, upscale=2.0, precision=0, sharpen=0.2, device=0, flags=0, ring=1):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.Config(width, height, 3, upscale, precision, sharpen, device, flags, ring)
        _check(self._lib.fftup_plan_create(C.byref(self._h), C.byref(cfg)), "fftup_plan_create")
        info = _lib.Info()
        _check(self._lib.fftup_plan_info(self._h, C.byref(info)), "fftup_plan_info")
        if info.abi_version != _lib.ABI_VERSION:
            self._lib.fftup_plan_destroy(self._h)
            self._h = None
            raise RuntimeError("libfftup.so speaks ABI version %d, this binding %d: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
                               % (info.abi_version, _lib.ABI_VERSION))
        self.width, self.height = width, height
        self.out_width, self.out_height = info.out_width, info.out_height
        self.precision = precision
        self.ring = max(1, ring)
        self.alg_bytes_per_frame = info.alg_bytes_per_frame
        self.kernel_alg_bytes = list(info.kernel_alg_bytes)
        self.kernel_min_bytes = list(info.kernel_min_bytes)
        self.kernel_names = [bytes(n).split(b"\0")[0].decode() for n in info.kernel_names]
        self.device_name = info.device_name.decode()
        self.device_bytes = info.device_bytes
        self.tuned = bool(info.tuned)
        self.u8_store = bool(info.u8_store)             # output slots hold 8-bit RGB (FLAG_FUSE_U8_STORE in effect)
        _buf = C.create_string_buffer(512)
        _check(self._lib.fftup_plan_describe(self._h, _buf, 512), "fftup_plan_describe")
        self.description = _buf.value.decode()
        self.specialised_at_plan_time = info.tuned == 2      # kernels instantiated for this size through hipRTC (csrc/jit.hpp)
        self._dtype = {0: np.float32, 1: np.float64, 2: np.float16}[precision]

    def close(self):
        if self._h:
            self._lib.fftup_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- pack + transferDataFromCPU (VR:1636-1688)
    def upload_rgb8(self, rgb, slot=0):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.shape == (self.height, self.width, 3), rgb.shape
        _check(self._lib.fftup_upload_rgb8_slot(self._h, slot, rgb.ctypes.data, rgb.strides[0]), "fftup_upload_rgb8")

    def upload_planar(self, planes, slot=0):
        planes = np.ascontiguousarray(planes, dtype=self._dtype)
        assert planes.shape == (3, self.height, self.width), planes.shape
        _check(self._lib.fftup_upload_planar(self._h, slot, planes.ctypes.data, self.width, self.width * self.height),
               "fftup_upload_planar")

    # ---- performVulkanUpscale (VR:1249-1279): returns ms per iteration
    def execute(self, num_iter=1):
        ms = C.c_double()
        _check(self._lib.fftup_execute(self._h, num_iter, C.byref(ms)), "fftup_execute")
        return ms.value

    def execute_ring(self, n_frames, first_slot=0):
        ms = C.c_double()
        _check(self._lib.fftup_execute_ring(self._h, n_frames, first_slot, C.byref(ms)), "fftup_execute_ring")
        return ms.value

    def execute_ring_timed(self, n_frames, first_slot=0, stride=1):
        """-> (total ms, [ms per kernel]); events around every launch of every stride-th frame, on the launching stream"""
        ms = C.c_double()
        km = (C.c_double * _lib.FFTUP_NUM_KERNELS)()
        _check(self._lib.fftup_execute_ring_timed(self._h, n_frames, first_slot, stride, C.byref(ms), km),
               "fftup_execute_ring_timed")
        return ms.value, list(km)

    def profile_kernels(self, num_iter=10):
        ms = (C.c_double * _lib.FFTUP_NUM_KERNELS)()
        _check(self._lib.fftup_profile_kernels(self._h, num_iter, ms), "fftup_profile_kernels")
        return list(ms)

    # ---- transferDataToCPU + unpack (VR:1697-1748)
    def download_planar(self, slot=0):
        out = np.empty((3, self.out_height, self.out_width), dtype=self._dtype)
        _check(self._lib.fftup_download_planar(self._h, slot, out.ctypes.data), "fftup_download_planar")
        return out

    def download_presharpen(self):
        out = np.empty((3, self.out_height, self.out_width), dtype=self._dtype)
        _check(self._lib.fftup_download_presharpen(self._h, out.ctypes.data), "fftup_download_presharpen")
        return out

    def download_input_planar(self, slot=0):
        out = np.empty((3, self.height, self.width), dtype=self._dtype)
        _check(self._lib.fftup_download_input_planar(self._h, slot, out.ctypes.data), "fftup_download_input_planar")
        return out

    def submit_rgb8(self, rgb_in, rgb_out):
        """Enqueue one host frame end to end (H2D, convert, kernels, convert, D2H) and return its ticket; up to
        `ring` frames are in flight.  rgb_in [H][W][3] / rgb_out [uH][uW][3] uint8, C-contiguous, and they must
        stay alive and untouched until wait(ticket) returns (pinned memory: PinnedArray).  submit_rgb8 / wait / drain of
        one Upscaler may be called from several threads at once (tickets are global to the plan)."""
        assert rgb_in.dtype == np.uint8 and rgb_in.shape == (self.height, self.width, 3) and rgb_in.flags.c_contiguous
        assert rgb_out.dtype == np.uint8 and rgb_out.shape == (self.out_height, self.out_width, 3) and rgb_out.flags.c_contiguous
        t = C.c_uint64()
        _check(self._lib.fftup_submit_rgb8(self._h, rgb_in.ctypes.data, 3 * self.width, rgb_out.ctypes.data,
                                           3 * self.out_width, C.byref(t)), "fftup_submit_rgb8")
        return t.value

    def png_bound(self):
        """bytes a PNG buffer of wait_png needs"""
        return int(self._lib.fftup_png_bound(self._h))

    def submit_png(self, rgb_in, png_out=None):
        """Like submit_rgb8, but the frame is PNG-encoded on the device; collect it with wait_png(ticket, buffer).  With png_out
        (a PinnedArray's uint8 array of png_bound() bytes) the GPU writes the file's stream into it itself; pass the same array to
        wait_png."""
        assert rgb_in.dtype == np.uint8 and rgb_in.shape == (self.height, self.width, 3) and rgb_in.flags.c_contiguous
        t = C.c_uint64()
        dst, cap = (png_out.ctypes.data, png_out.size) if png_out is not None else (None, 0)
        _check(self._lib.fftup_submit_png(self._h, rgb_in.ctypes.data, 3 * self.width, dst, cap, C.byref(t)), "fftup_submit_png")
        return t.value

    def wait_png(self, ticket, buf):
        """buf: uint8 array of at least png_bound() bytes (PinnedArray for a fast copy); returns the PNG file's bytes in it"""
        assert buf.dtype == np.uint8 and buf.flags.c_contiguous
        n = C.c_size_t()
        _check(self._lib.fftup_wait_png(self._h, ticket, buf.ctypes.data, buf.size, C.byref(n)), "fftup_wait_png")
        return int(n.value)

    def wait(self, ticket):
        _check(self._lib.fftup_wait(self._h, ticket), "fftup_wait")

    def drain(self):
        _check(self._lib.fftup_drain(self._h), "fftup_drain")

    def output_checksum(self, slot=0):
        """64-bit wrapping sum of the 32-bit words of output slot `slot`, computed on the device (job accounting)."""
        v = C.c_uint64()
        _check(self._lib.fftup_output_checksum(self._h, slot, C.byref(v)), "fftup_output_checksum")
        return v.value

    def download_rgb8(self, slot=0):
        out = np.empty((self.out_height, self.out_width, 3), dtype=np.uint8)
        _check(self._lib.fftup_download_rgb8(self._h, slot, out.ctypes.data, out.strides[0]), "fftup_download_rgb8")
        return out


class PinnedArray:
    """uint8 numpy view of page-locked host memory from fftup_host_alloc (needed for truly asynchronous copies
    in Upscaler.submit_rgb8); free with .close() after the frames using it have been waited for."""

    def __init__(self, shape):
        self._lib = _lib.load()
        n = int(np.prod(shape))
        self._p = self._lib.fftup_host_alloc(n)
        if not self._p:
            raise FftupError(5, "fftup_host_alloc")
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(n,)).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            self._lib.fftup_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upscale_image(rgb, upscale=2.0, precision=0, sharpen=0.2, num_iter=1, device=0, flags=0):
    """Single-image path of launchResample(): returns (rgb_out uint8 [uH][uW][3], ms_per_iter)."""
    rgb = np.asarray(rgb)
    with Upscaler(rgb.shape[1], rgb.shape[0], upscale, precision, sharpen, device, flags) as up:
        up.upload_rgb8(rgb)
        ms = up.execute(num_iter)
        return up.download_rgb8(), ms
