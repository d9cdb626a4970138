import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The fast kernels walk small trees (< 1024 primitives) in the reference's order by default — cheaper there — and large ones nearer child
# first with a certificate. Most fixtures and fuzz scenes are small: the suite therefore asks for the nearest-first kernels on EVERY
# regular tree, so that they keep being held against every golden vector and random scene; the default choice has its own tests
# (test_small_trees_keep_the_reference_order_by_default, and tests/fuzz_parity.py renders every scene under both settings).
os.environ.setdefault("GPUART_HIP_NEAREST_MIN_PRIMS", "0")
