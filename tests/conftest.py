import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import ka9q_sdr_amd as kq
        return kq.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail (not skip) when the HIP library or the device is missing: no silent fallback."""
    import ka9q_sdr_amd as kq
    kq.load_library()
    n = kq.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need the MI355X"
    return n
