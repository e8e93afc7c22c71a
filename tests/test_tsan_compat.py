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


@pytest.mark.parametrize("what,old,new", [
    ("the producer's wait for the slot's release",
     'if (hipError_t e = hipStreamWaitEvent(f->side, f->freed[slot], 0); e != hipSuccess) return fail("kq_fanout_post: hipStreamWaitEvent", e);',
     "/* mutated: no wait for the consumer's release */"),
    ("the consumer's wait for the batch",
     "if (hipEventQuery(f->ready[slot]) != hipSuccess) {",
     "if (false) {  /* mutated: the consumer never waits */"),
])
def test_fanout_harness_catches_a_missing_wait(what, old, new, tmp_path):
    """The world-of-eight run means something only if it fails when the protocol is broken: kq_fanout.cpp with ONE of its two
    cross-stream waits taken out must be reported by ThreadSanitizer (a slot overwritten before its release / read before
    its batch has landed is a data race on plain memory under the mock streams)."""
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    probe = subprocess.run("echo 'int main(){}' | g++ -x c++ -fsanitize=thread - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    tsan = os.path.join(HERE, "tsan")
    r = subprocess.run(["make", "-C", tsan, "_build/librccl_mock.so"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    src = open(os.path.join(HERE, "..", "ka9q_sdr_amd", "csrc", "kq_fanout.cpp")).read()
    assert src.count(old) == 1, "kq_fanout.cpp no longer holds the statement this test takes out: " + what
    mutated = tmp_path / "kq_fanout_mutated.cpp"
    mutated.write_text(src.replace(old, new).replace('#include "../../include/ka9q_hip.h"', '#include "ka9q_hip.h"'))
    exe = tmp_path / "harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(tsan, "mock_async"),
           "-I" + os.path.join(HERE, "..", "include"), "-I" + os.path.join(HERE, "..", "ka9q_sdr_amd", "csrc"),
           "-Wno-unknown-pragmas", os.path.join(tsan, "fanout_harness.cpp"), str(mutated), "-o", str(exe),
           "-L" + os.path.join(tsan, "_build"), "-lrccl_mock", "-Wl,-rpath," + os.path.join(tsan, "_build"), "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, KQ_RCCL_LIB=os.path.join(tsan, "_build", "librccl_mock.so"), TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([str(exe), "8", "400"], capture_output=True, text=True, timeout=600, env=env)
    out = run.stdout + run.stderr
    assert run.returncode != 0 and ("ThreadSanitizer: data race" in out or "failures" in out), \
        "the run did not notice " + what + ":\n" + out[-2000:]
