"""ctypes loader of the C-ABI library (include/fftup.h).  No fallback: if libfftup.so is missing or
fails to load, importing the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfftup.so")

FFTUP_NUM_KERNELS = 4

FLAG_U8_WRAP = 1
FLAG_FUSE_U8_LOAD = 2
FLAG_GENERIC_KERNELS = 4
FLAG_UNFUSED_SHARPEN = 8
FLAG_TUNE_PLAN = 16
FLAG_FUSE_U8_STORE = 32
FLAG_SEQUENTIAL_EXECUTE = 64      # (accepted and ignored: ordered iterations are the default)
FLAG_OVERLAP_ITERATIONS = 128

# every symbol include/fftup.h declares
EXPORTS = [
    "fftup_device_count", "fftup_device_name", "fftup_plan_create", "fftup_plan_destroy", "fftup_plan_info",
    "fftup_upload_rgb8", "fftup_upload_rgb8_slot", "fftup_upload_planar", "fftup_execute", "fftup_execute_ring", "fftup_execute_ring_timed",
    "fftup_profile_kernels", "fftup_download_rgb8", "fftup_download_planar", "fftup_download_presharpen",
    "fftup_download_input_planar", "fftup_host_alloc", "fftup_host_free", "fftup_submit_rgb8", "fftup_wait",
    "fftup_drain", "fftup_strerror", "fftup_last_error", "fftup_version", "fftup_jit_check", "fftup_plan_describe",
    "fftup_device_pci_bus_id", "fftup_output_checksum", "fftup_png_bound", "fftup_submit_png", "fftup_wait_png",
]
ABI_VERSION = 2


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("upscale", C.c_float),
                ("precision", C.c_uint32), ("sharpen", C.c_float), ("device", C.c_int32), ("flags", C.c_uint32),
                ("ring", C.c_uint32)]


class Info(C.Structure):
    _fields_ = [("out_width", C.c_uint32), ("out_height", C.c_uint32), ("num_kernels", C.c_uint32),
                ("tuned", C.c_uint32), ("alg_bytes_per_frame", C.c_double),
                ("kernel_alg_bytes", C.c_double * FFTUP_NUM_KERNELS), ("device_bytes", C.c_uint64),
                ("device_name", C.c_char * 256), ("kernel_names", (C.c_char * 64) * FFTUP_NUM_KERNELS),
                # appended in ABI version 2
                ("kernel_min_bytes", C.c_double * FFTUP_NUM_KERNELS), ("abi_version", C.c_uint32), ("u8_store", C.c_uint32)]


KNOBS_LIB_PATH = os.path.join(HERE, "libfftup_knobs.so")       # the same objects + the FFTUP_EXPERIMENT parser (tests, tools)
_libs = {}


def load():
    """Load libfftup.so (built by __graft_entry__.build()).  Raises if it is absent.
    FFTUP_LIBRARY=<path> names another build of the library (tests that pin factorizations or strip lengths through
    FFTUP_EXPERIMENT load libfftup_knobs.so this way: the shipping library has no such parser); read at every call, one
    handle per path."""
    path = os.environ.get("FFTUP_LIBRARY") or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    lib.fftup_device_count.restype = C.c_int
    lib.fftup_device_name.argtypes = [C.c_int, C.c_char_p, sz]
    lib.fftup_plan_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    lib.fftup_plan_destroy.argtypes = [vp]
    lib.fftup_plan_destroy.restype = None
    lib.fftup_plan_info.argtypes = [vp, C.POINTER(Info)]
    lib.fftup_upload_rgb8.argtypes = [vp, vp, sz]
    lib.fftup_upload_rgb8_slot.argtypes = [vp, u32, vp, sz]
    lib.fftup_upload_planar.argtypes = [vp, u32, vp, sz, sz]
    lib.fftup_execute.argtypes = [vp, u32, C.POINTER(C.c_double)]
    lib.fftup_execute_ring.argtypes = [vp, u32, u32, C.POINTER(C.c_double)]
    lib.fftup_execute_ring_timed.argtypes = [vp, u32, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fftup_profile_kernels.argtypes = [vp, u32, C.POINTER(C.c_double)]
    lib.fftup_download_rgb8.argtypes = [vp, u32, vp, sz]
    lib.fftup_download_planar.argtypes = [vp, u32, vp]
    lib.fftup_download_presharpen.argtypes = [vp, vp]
    lib.fftup_download_input_planar.argtypes = [vp, u32, vp]
    lib.fftup_host_alloc.argtypes = [sz]
    lib.fftup_host_alloc.restype = vp
    lib.fftup_host_free.argtypes = [vp]
    lib.fftup_host_free.restype = None
    lib.fftup_submit_rgb8.argtypes = [vp, vp, sz, vp, sz, C.POINTER(C.c_uint64)]
    lib.fftup_wait.argtypes = [vp, C.c_uint64]
    lib.fftup_drain.argtypes = [vp]
    lib.fftup_png_bound.argtypes = [vp]
    lib.fftup_png_bound.restype = sz
    lib.fftup_submit_png.argtypes = [vp, vp, sz, vp, sz, C.POINTER(C.c_uint64)]
    lib.fftup_wait_png.argtypes = [vp, C.c_uint64, vp, sz, C.POINTER(sz)]
    lib.fftup_strerror.argtypes = [C.c_int]
    lib.fftup_strerror.restype = C.c_char_p
    lib.fftup_last_error.restype = C.c_char_p
    lib.fftup_version.restype = C.c_char_p
    lib.fftup_plan_describe.argtypes = [vp, C.c_char_p, sz]
    lib.fftup_jit_check.argtypes = [u32, u32, C.c_float, u32, C.c_char_p, C.c_char_p, sz]
    lib.fftup_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, sz]
    lib.fftup_output_checksum.argtypes = [vp, u32, C.POINTER(C.c_uint64)]
    _libs[path] = lib
    return lib
