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
