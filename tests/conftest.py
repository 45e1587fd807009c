"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"` runs in the build container (no GPU): oracle pinning against the golden
fixtures, host logic, ABI/export checks, gloo multi-process tests.
`-m gpu` runs on an MI355X box: parity of the HIP path (through the C ABI) against the
oracle and the golden fixtures.  /root/reference does not exist there.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_ctx():
    """Process-wide HIP context; fails loudly (no skip) if the library or GPU is missing."""
    import elfi_amd
    return elfi_amd.default_context()
