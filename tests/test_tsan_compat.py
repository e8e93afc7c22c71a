"""ThreadSanitizer run of the compat surface's threading protocol: kq_compat.cpp + kq_radio.cpp compiled for
the CPU against a device-less stand-in for the HIP runtime, driven by one producer, the three demodulator threads and a
set_filter thread (tests/tsan/harness.cpp) -- filter.c:146-172,195-237,538-543 as radio.c / display.c use them.  The GPU
pool runs no sanitizers, so this is where the condvar block counter, the response hot swap and the terminate poll are
checked for races."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_compat_threading_protocol_is_race_free():
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    probe = subprocess.run("echo 'int main(){}' | g++ -x c++ -fsanitize=thread - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    subprocess.run(["make", "-C", os.path.join(HERE, "tsan"), "clean"], capture_output=True)
    r = subprocess.run(["make", "-C", os.path.join(HERE, "tsan"), "tsan"], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert "hand-offs" in out
