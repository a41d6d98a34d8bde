"""pytest configuration: registers the `gpu` marker and makes the repo root
importable (package `radiorust_amd`, test-only package `oracle`)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import rr_oracle

    rr_oracle.build()
    return rr_oracle


PRECISION = 1e-10  # lib.rs:52-58 (assert_approx)


def assert_approx(a, b, precision=PRECISION):
    """Port of the reference's test helper (src/lib.rs:52-58): absolute OR
    log-ratio difference within 1e-10."""
    import math

    if not (abs(a - b) <= precision or (a != 0 and b != 0 and a / b > 0 and abs(math.log(a / b)) <= precision)):
        raise AssertionError(f"{a} and {b} are not approximately equal")
