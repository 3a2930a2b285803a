import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The plan-time compiler keeps code objects and the tuner's findings (wisdom.txt) in a cache directory, by default the
# user's ~/.cache/fftup.  Tests must neither depend on what earlier runs left there (a wisdom entry changes the
# factorization a plan picks, and with it exact-string asserts and timings) nor leave anything behind: one private
# directory per test session, unless the caller pins one.
if "FFTUP_CACHE_DIR" not in os.environ:
    _cache = tempfile.TemporaryDirectory(prefix="fftup_test_cache_")
    os.environ["FFTUP_CACHE_DIR"] = _cache.name


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` under gpurun)")


# Order of the GPU suite under `pytest -x`: the parity tests proper first (hot path against the oracle, through the C ABI), then the
# plan-time compiler, the drop-in CLI, the PNG paths, the seeded sweeps -- and the bench line's bookkeeping LAST, so that a timing
# ratio or a stale profile on some other box can never stop the run before a single parity test has reported.
_GPU_ORDER = ["test_gpu_parity", "test_gpu_jit", "test_gpu_cli", "test_gpu_png", "test_gpu_sweep", "test_gpu_bench"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(mod) + 1 if mod in _GPU_ORDER else 0     # (everything else -- the CPU suite -- keeps its place in front)
    items.sort(key=key)                                                      # stable: the order inside a file is kept
