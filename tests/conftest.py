import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_library():
    """The HIP library normally travels prebuilt in-tree; on a checkout without it (sources only) build it once
    with hipcc -- building the product is not a fallback, the tests still fail without it."""
    import ka9q_sdr_amd as kq
    if not os.path.exists(kq.library_path()):
        kq.build_library()
    return kq


def _gpu_present():
    try:
        return _ensure_library().device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail (not skip) when the HIP library or the device is missing: no silent fallback."""
    kq = _ensure_library()
    kq.load_library()
    n = kq.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need the MI355X"
    return n


# Threshold ties (decision flips are counted, not hidden in a tolerance): the parity tests append
# (test, ties, channels, what) here and the run's summary prints the list, so the tail of a GPU test log shows the count.
TIES = []


def note_ties(test, flips, channels):
    TIES.append((test, len(flips), channels, ", ".join("ch %d %s %.2g" % f for f in flips[:6])))


def pytest_terminal_summary(terminalreporter):
    if not TIES:
        return
    terminalreporter.write_sep("-", "threshold ties of the reference's own comparisons (counted, each checked to be one)")
    for test, n, channels, what in TIES:
        terminalreporter.write_line("%-64s %3d of %4d channels%s" % (test, n, channels, ("  [" + what + "]") if what else ""))
    terminalreporter.write_line("total: %d ties in %d channel runs" % (sum(t[1] for t in TIES), sum(t[2] for t in TIES)))
