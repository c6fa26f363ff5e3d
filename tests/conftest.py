"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`python -m pytest tests/ -m "not gpu"` runs on a CPU-only box (oracle vs golden
vectors, host logic, C-ABI symbol/arg-check coverage, gloo world_size-2 sharding).
`python -m pytest tests/ -m gpu` needs one MI355X and calls the HIP path through the C ABI.
"""
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
    """The CPU oracle (test infrastructure): oracle/libcnt_oracle.so via ctypes."""
    from oracle import cnt_oracle

    cnt_oracle.build()
    return cnt_oracle


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)
