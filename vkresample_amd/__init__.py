"""vkresample_amd -- MI355X-native FFT upscaler, drop-in for VkResample's upscale path.

Product = vkresample_amd/libfftup.so (hand-written HIP for gfx950 behind the C ABI of include/fftup.h)
plus the C++ CLI (vkresample_amd/csrc/cli).  This package is the thin Python host mirror used by tests
and bench.py.  Nothing here computes on the CPU."""
from .api import FftupError, PinnedArray, Upscaler, device_count, device_name, device_pci_bus_id, upscale_image  # noqa: F401
from ._lib import FLAG_FUSE_U8_LOAD, FLAG_FUSE_U8_STORE, FLAG_GENERIC_KERNELS, FLAG_OVERLAP_ITERATIONS, FLAG_SEQUENTIAL_EXECUTE, FLAG_TUNE_PLAN, FLAG_U8_WRAP, FLAG_UNFUSED_SHARPEN  # noqa: F401
