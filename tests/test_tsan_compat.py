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


def test_fanout_protocol_with_a_world_of_eight_is_race_free():
    """kq_fanout.cpp (unchanged) with world = 8: one thread per rank, asynchronous mock streams, a thread-based librccl
    loaded through KQ_RCCL_LIB (tests/tsan/mock_rccl.cpp) -- 2000 batches with jittered consumers under ThreadSanitizer.
    A slot overwritten before its release or read before its batch landed is a data race and fails the run; the harness
    checks that every rank reads the root's batch k at step k, that kq_fanout_stats reports RCCL's world of 8, and that a
    rank whose set-up fails (or a communicator that cannot be formed) makes kq_fanout_create return NULL on EVERY rank.
    Stands in for multicast.c:143-237's fan-out, which has never had more than one GPU to run on in this project."""
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    probe = subprocess.run("echo 'int main(){}' | g++ -x c++ -fsanitize=thread - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    r = subprocess.run(["make", "-C", os.path.join(HERE, "tsan"), "fanout", "WORLD=8", "STEPS=2000"], capture_output=True,
                       text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert "world 8: 16000 consumer passes checked" in out and "fan-out protocol: ok" in out
    # a world of two and of three (odd: the root plus an uneven split of waiters)
    for w in (2, 3):
        r = subprocess.run(["make", "-C", os.path.join(HERE, "tsan"), "fanout", "WORLD=%d" % w, "STEPS=400"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "fan-out protocol: ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_c_host_for_several_gpus_runs_with_a_world_of_eight():
    """examples/radio_fanout.c -- one thread per GPU, plain C -- compiled unchanged by gcc and run with a world of 8 under
    ThreadSanitizer against kq_fanout.cpp (unchanged), the mock streams, the mock librccl and a stand-in bank: every rank
    must have seen bit-identical front-end samples (IF power of every block equal to rank 0's)."""
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    probe = subprocess.run("echo 'int main(){}' | g++ -x c++ -fsanitize=thread - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    r = subprocess.run(["make", "-C", os.path.join(HERE, "tsan"), "example", "WORLD=8"], capture_output=True, text=True,
                       timeout=900)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert out.count("rccl ranks 8") == 8 and "\nok\n" in out
