import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _have_gpu():
    try:
        from gfalign_amd import scorer
        return scorer.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run on the HIP path; no GPU -> the test fails loudly."""
    if not _have_gpu():
        pytest.fail("no HIP device / libgfalign_scorer.so not loadable: "
                    "gpu-marked tests must run on an MI355X")
    return True
