"""The C example (examples/radio_bank.c) built with gcc against the C ABI and run on the GPU: a known FM signal
must come back with its deviation (fm.c:146-158)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_radio_bank_c_example_runs(gpu):
    lib = os.path.join(ROOT, "ka9q_sdr_amd", "lib")
    out = os.path.join(tempfile.gettempdir(), "kq_radio_bank_example_%d" % os.getpid())
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "radio_bank.c"), "-L", lib, "-lka9q_hip", "-Wl,-rpath," + lib,
                        "-lm", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    try:
        run = subprocess.run([out], capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(out)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = run.stdout.strip().splitlines()
    assert lines[-1] == "ok"
    assert sum(1 for ln in lines if ln.startswith("block ")) == 4


def test_radio_fanout_c_example_runs_with_a_world_of_one(gpu):
    """examples/radio_fanout.c (the plain-C host for several GPUs, one thread per GPU) on the one GPU of the test box:
    world = 1, the bank's own stream from kq_bank_stream as the fan-out's consumer stream, batches posted from host
    memory; the FM channel on the first carrier must measure its 3 kHz deviation.  (World = 8 runs on the CPU against
    the mock RCCL: tests/test_tsan_compat.py.)"""
    lib = os.path.join(ROOT, "ka9q_sdr_amd", "lib")
    out = os.path.join(tempfile.gettempdir(), "kq_radio_fanout_example_%d" % os.getpid())
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "radio_fanout.c"), "-L", lib, "-lka9q_hip", "-Wl,-rpath," + lib,
                        "-lm", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    try:
        run = subprocess.run([out, "1", "5", "0", "30"], capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(out)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = run.stdout.strip().splitlines()
    assert lines[-1] == "ok" and lines[0].startswith("rank 0: channels 0..26  rccl ranks 0")
    # the per-rank table of the timed phase (what bench.py --gpus N carries as per_rank)
    head = [i for i, ln in enumerate(lines) if ln.startswith("rank  ms_per_step")]
    assert len(head) == 1, run.stdout
    row = lines[head[0] + 1].split()
    assert row[0] == "0" and float(row[1]) > 0 and float(row[2]) > 0 and float(row[1]) >= float(row[2]), row


@pytest.mark.parametrize("pcm,operator,paced", [(1, 0, 0), (0, 0, 0), (1, 1, 0), (1, 1, 1), (0, 0, 1)])
def test_radio_realtime_c_example(gpu, pcm, operator, paced):
    """examples/radio_realtime.c: 8192 channels from plain C in the receiver's loop -- process, push the next batch, queue the
    delivery (int16 PCM words + silent masks, or floats), wait for the delivery two back -- set up with one
    kq_bank_add_channels call; every delivered channel-block carries olen samples with the squelch open, and the bank keeps
    up with real time (8192 channels are a quarter of what one GPU carries).  operator = 1: a second thread changes filters,
    drops a channel and brings it back, retunes (~1000 operations a second) beside the calls in flight -- none of which may
    cost the loop its pace: the receiver's worst wait for the handle's lock stays far below a call period.  paced = 1: the
    batches arrive by the clock (one per 1.64 ms): no delivery later than the reference player's playout buffer (monitor.c:83),
    the backlog never beyond a few periods, and at most a handful more than one call period behind (the pool's hosts are
    shared: a receiver thread taken off its core for a few milliseconds is not the library's doing -- the program prints where
    the longest interval's time went)."""
    lib = os.path.join(ROOT, "ka9q_sdr_amd", "lib")
    out = os.path.join(tempfile.gettempdir(), "kq_radio_realtime_example_%d_%d_%d_%d" % (os.getpid(), pcm, operator, paced))
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "radio_realtime.c"), "-L", lib, "-lka9q_hip", "-Wl,-rpath," + lib,
                        "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-o", out], capture_output=True, text=True)   # (no HIP library on the line)
    assert r.returncode == 0, r.stderr
    try:
        run = subprocess.run([out, "8192", "2", "2.5" if paced else "1.5", str(pcm), str(operator), str(paced)],
                             capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(out)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = run.stdout.strip().splitlines()
    assert lines[-1] == "ok", run.stdout
    rt = [ln for ln in lines if "x real time" in ln]
    factor = float(rt[0].split("=")[-1].split("x")[0])
    assert len(rt) == 1 and (0.995 < factor < 1.005 if paced else factor > 2.0), rt
    if operator:
        op = [ln for ln in lines if ln.startswith("operator thread:")]
        assert len(op) == 1 and float(op[0].split("=")[1].split()[0]) > 500, op
        worst_wait = float(op[0].split("worst")[2].split("ms")[0])
        assert worst_wait < 0.8, op             # half a call period
    if paced:
        dl = [ln for ln in lines if ln.startswith("deadline:")]
        assert len(dl) == 1 and dl[0].rstrip().endswith("(monitor.c:83): 0"), dl
        late = int(dl[0].split()[1])
        backlog = int(dl[0].split("deepest backlog")[1].split()[0])
        assert late <= 30 and backlog <= 12, dl
