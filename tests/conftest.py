import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the oracle (gcc) and the product libraries (hipcc cross-compiles without a GPU) if they are missing."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libpngloss_port.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    csrc = os.path.join(ROOT, "pngloss_amd", "csrc")
    if not (os.path.exists(os.path.join(csrc, "libpngloss_hip.so")) and os.path.exists(os.path.join(csrc, "libpngloss_synth.so"))):
        subprocess.run(["make", "-C", csrc, "-j4"], check=True, capture_output=True)
    yield
