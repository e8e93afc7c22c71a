"""The demodulator thread entry points of radio.h:235-237 (include/ka9q_hip_radio.h), driven from C the way radio.c
drives them (examples/radio_threads.c): struct demod filled by hand, master from create_filter_input, per-sample LO mix
with step_osc, execute_filter_input per block, audio back through send_mono_output / send_stereo_output.

Two threads are put on top of the same master code:
  * the library's own demod_fm / demod_am / demod_linear, against the oracle chain on the same input;
  * the REFERENCE's demod_am -- /root/reference/am.c compiled where it lies, unmodified, against
    include/ka9q_hip_compat.h (oracle/Makefile -> oracle/_ref/libref_am_dropin.so; built in the build container,
    travels prebuilt) -- whose loop then runs on this library's create_filter_output / set_filter /
    execute_filter_output on the GPU.  A boundary test: reference code executing above the library."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import kq_oracle as ko
from common import oracle_cfg, rel_rms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_AM = os.path.join(ROOT, "oracle", "_ref", "libref_am_dropin.so")
pytestmark = pytest.mark.gpu

FS, L, M, D = 192000, 2048, 2049, 4
NB = 8


@pytest.fixture(scope="module")
def harness(gpu):
    lib = os.path.join(ROOT, "ka9q_sdr_amd", "lib")
    exe = os.path.join(tempfile.gettempdir(), "kq_radio_threads_%d" % os.getpid())
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "radio_threads.c"), "-L", lib, "-lka9q_hip", "-Wl,-rpath," + lib,
                        "-rdynamic", "-ldl", "-lpthread", "-lm", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    yield exe
    os.unlink(exe)


def _signal(kind, seed, geom=None):
    rng = np.random.default_rng(seed)
    fs_, l_ = (geom or (FS, L, M, D))[:2]
    n = NB * l_
    t = np.arange(n) / fs_
    fc = 20000.0
    if kind == "am":
        env = 0.1 * (1 + 0.5 * np.cos(2 * np.pi * 1000 * t))
        s = env * np.exp(2j * np.pi * fc * t)
    elif kind == "fm":
        s = 0.1 * np.exp(1j * (2 * np.pi * fc * t + 3.0 * np.sin(2 * np.pi * 1000 * t)))   # 3 kHz peak deviation
    else:   # two-tone upper sideband
        s = 0.05 * (np.exp(2j * np.pi * (fc + 700) * t) + np.exp(2j * np.pi * (fc + 1900) * t))
    s = s + 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return s.astype(np.complex64)


def _run(exe, mode, iq, low, high, extra=(), geom=None):
    fs_, l_, m_, d_ = geom or (FS, L, M, D)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.cf32"), os.path.join(d, "out.bin")
        iq.tofile(fin)
        cmd = [exe, mode, str(fs_), str(l_), str(m_), str(d_), str(low), str(high), str(NB), fin, fout, "--lo", "-20000"]
        r = subprocess.run(cmd + list(extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
        blob = open(fout, "rb").read()
    recs, pos, tail = [], 0, None
    while pos < len(blob):
        (n,) = struct.unpack_from("<i", blob, pos)
        pos += 4
        if n < 0:
            tail = np.frombuffer(blob, np.float32, 8, pos)
            break
        audio = np.frombuffer(blob, np.float32, n, pos)
        pos += 4 * n
        st = np.frombuffer(blob, np.float32, 8, pos)
        pos += 32
        recs.append((audio, dict(zip(("bb_power", "n0", "snr", "foffset", "pdeviation", "gain", "plfreq", "noise_gain"), st))))
    assert tail is not None and len(recs) >= NB      # the thread may hand over one more (zero) block while it winds down
    return recs[:NB], tail


def _oracle(p, iq, geom=None):
    fs_, l_, m_, d_ = geom or (FS, L, M, D)
    cfg = oracle_cfg(p, fs_, l_, m_, d_, compute_n0=1)
    return ko.run_chain(cfg, iq.reshape(NB, l_), want_filt=False)


AM = dict(demod="am", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=0.0, recovery_rate=50.0)
FM = dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0)
USB = dict(demod="linear", low=100.0, high=3000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0)


# the thread entry points at the reference's own default -L 3840 -M 4353 (main.c:160-170: a history longer than a block) and at a
# 240 kHz front end (N = 9600, decimate 5) beside the power-of-two geometry of the rest of this file
GEOMS = [None, (192000, 3840, 4353, 4), (240000, 4800, 4801, 5)]


@pytest.mark.parametrize("geom", GEOMS)
def test_library_demod_am_thread(harness, geom):
    iq = _signal("am", 1, geom)
    recs, _ = _run(harness, "am", iq, AM["low"], AM["high"], ["--recovery", "50"], geom=geom)
    auds, sts, _ = _oracle(AM, iq, geom)
    l_, d_ = (geom or (FS, L, M, D))[1], (geom or (FS, L, M, D))[3]
    assert all(len(a) == l_ // d_ for a, _ in recs)
    assert rel_rms(np.concatenate([a for a, _ in recs]), np.concatenate(auds)) < 1e-5
    for b in range(NB):
        assert abs(recs[b][1]["n0"] / sts[b]["n0"] - 1) < 2e-4               # am.c:46-49 runs before the hand-off
        assert abs(recs[b][1]["gain"] / sts[b]["agc_gain"] - 1) < 2e-5
        if b > 0:                                                            # am.c:78 runs after it: one block behind
            assert abs(recs[b][1]["bb_power"] / sts[b - 1]["bb_power"] - 1) < 2e-5


@pytest.mark.skipif(not os.path.exists(REF_AM), reason="oracle/_ref/libref_am_dropin.so not built (needs the reference tree)")
def test_reference_am_c_runs_on_the_library(harness):
    iq = _signal("am", 2)
    ref, _ = _run(harness, "am", iq, AM["low"], AM["high"], ["--recovery", "50", "--ref", REF_AM])
    mine, _ = _run(harness, "am", iq, AM["low"], AM["high"], ["--recovery", "50"])
    auds, sts, _ = _oracle(AM, iq)
    a_ref = np.concatenate([a for a, _ in ref])
    # the reference's own envelope detector / DC removal / AGC loop on this library's filter output ...
    assert rel_rms(a_ref, np.concatenate(auds)) < 1e-5            # ... equals the oracle's chain
    assert rel_rms(a_ref, np.concatenate([a for a, _ in mine])) < 1e-5   # ... and the library's own thread
    for b in range(NB):
        assert abs(ref[b][1]["n0"] / sts[b]["n0"] - 1) < 2e-4
        assert abs(ref[b][1]["gain"] / sts[b]["agc_gain"] - 1) < 2e-5
        assert abs(ref[b][1]["noise_gain"] / mine[b][1]["noise_gain"] - 1) < 1e-6   # set_filter under both threads


@pytest.mark.parametrize("geom", GEOMS)
def test_library_demod_fm_thread(harness, geom):
    iq = _signal("fm", 3, geom)
    recs, _ = _run(harness, "fm", iq, FM["low"], FM["high"], geom=geom)
    auds, sts, _ = _oracle(FM, iq, geom)
    assert rel_rms(np.concatenate([a for a, _ in recs]), np.concatenate(auds)) < 1e-5
    for b in range(1, NB):
        assert abs(recs[b][1]["pdeviation"] - sts[b]["pdeviation"]) < 1e-4 * 3000
        assert abs(recs[b][1]["foffset"] - sts[b]["foffset"]) < 0.5
        assert abs(recs[b][1]["bb_power"] / sts[b]["bb_power"] - 1) < 2e-5
    assert abs(recs[-1][1]["pdeviation"] - 3000) < 150


@pytest.mark.parametrize("stereo,geom", [(False, None), (True, None), (False, GEOMS[1]), (True, GEOMS[2])])
def test_library_demod_linear_thread(harness, stereo, geom):
    iq = _signal("usb", 4, geom)
    p = dict(USB, channels=2 if stereo else 1)
    recs, _ = _run(harness, "linear", iq, p["low"], p["high"], ["--hang", "1.1", "--recovery", "6"] + (["--stereo"] if stereo else []),
                   geom=geom)
    auds, sts, _ = _oracle(p, iq, geom)
    l_, d_ = (geom or (FS, L, M, D))[1], (geom or (FS, L, M, D))[3]
    assert all(len(a) == (2 if stereo else 1) * l_ // d_ for a, _ in recs)
    # the first block is the AGC start-up on numerically-zero samples (linear.c:271-272): compared from block 1 on
    assert rel_rms(np.concatenate([a for a, _ in recs[1:]]), np.concatenate(auds[1:])) < 1e-5
    for b in range(1, NB):
        assert abs(recs[b][1]["gain"] / sts[b]["agc_gain"] - 1) < 2e-5


def test_linear_thread_at_65536_points(harness):
    """The cfg-5 geometry through the reference's own shell: master of 65536 points (the compat surface's two-pass
    transform), demod_linear as radio.c starts it; slave, compute_n0 (linear.c:123-126) and the demodulator run on the
    master's resident spectrum (kq_bank_process_spectrum): one forward transform per block, the master's."""
    geom = (20000000, 32768, 32769, 512)
    iq = _signal("usb", 6, geom)
    recs, _ = _run(harness, "linear", iq, USB["low"], USB["high"], ["--hang", "1.1", "--recovery", "6"], geom)
    auds, sts, _ = _oracle(USB, iq, geom)
    assert all(len(a) == geom[1] // geom[3] for a, _ in recs)
    assert rel_rms(np.concatenate([a for a, _ in recs][1:]), np.concatenate(auds[1:])) < 1e-5
    for b in range(NB):
        assert abs(recs[b][1]["n0"] / sts[b]["n0"] - 1) < 5e-3, (b, recs[b][1]["n0"], sts[b]["n0"])   # (a tie moves it 1e-3)


def _timed_run(exe, mode, geom, nblocks, low, high, extra=()):
    fs_, l_, m_, d_ = geom
    rng = np.random.default_rng(11)
    n = nblocks * l_
    t = np.arange(n) / fs_
    iq = (0.1 * np.exp(1j * (2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000 * t))) +
          1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.cf32"), os.path.join(d, "out.bin")
        iq.tofile(fin)
        cmd = [exe, mode, str(fs_), str(l_), str(m_), str(d_), str(low), str(high), str(nblocks), fin, fout, "--lo", "-20000",
               "--time"] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("timing:")][0]
    import re
    f = [float(x) for x in re.findall(r"([0-9.]+) us", line)]
    factor = float(re.search(r"([0-9.]+) x real time", line).group(1))
    return dict(mix_us=f[0], master_us=f[1], thread_us=f[2], block_us=f[3], factor=factor, line=line), iq


def test_drop_in_surface_speed(harness, capsys):
    """The drop-in surface measured: one channel through the reference's own shell -- host mix with step_osc,
    execute_filter_input, the library's demod_fm thread -- at BASELINE.json's cfg 1 geometry (192 kHz, N = 16384, D = 4;
    23.4 blocks/s in real time) must run at 20 x real time or better; beside it the oracle's chain (the reference's
    structure: per-sample NCO, N-point transform, slave, compute_n0, FM demodulator) on one host core, and the same
    shell at N = 65536 (cfg 5's geometry, demod_linear).  The figures go to INTEGRATION.md section A2."""
    import time
    g1 = (192000, 8192, 8193, 4)
    r1, iq1 = _timed_run(harness, "fm", g1, 44, FM["low"], FM["high"])
    cfg = oracle_cfg(FM, *g1, compute_n0=1)
    blocks = iq1.reshape(-1, g1[1])
    t0 = time.perf_counter()
    ko.run_chain(cfg, blocks, want_filt=False)
    oracle_us = (time.perf_counter() - t0) / len(blocks) * 1e6
    g5 = (20000000, 32768, 32769, 512)
    r5, _ = _timed_run(harness, "linear", g5, 24, USB["low"], USB["high"], ["--hang", "1.1", "--recovery", "6"])
    with capsys.disabled():
        print("\n  cfg 1 shell: " + r1["line"])
        print("  cfg 1 oracle chain on one host core: %.0f us per block = %.1f x real time" % (oracle_us, r1["block_us"] / oracle_us))
        print("  cfg 5 shell: " + r5["line"])
    assert r1["factor"] >= 20.0, r1["line"]
    assert r5["factor"] >= 1.0, r5["line"]
