"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (north_star): float outputs within 1e-5 RMS relative to the oracle's signal RMS; sample counts,
block sequence and decision state (squelch counter, hang counter, blanked-sample count) bit exact.
"""
import os

import numpy as np
import pytest

import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
import kq_oracle as ko
from common import bank_cfg, oracle_cfg, rel_rms, run_oracle

pytestmark = pytest.mark.gpu

AUDIO_TOL = 1e-5      # relative RMS, north_star
FILT_TOL = 1e-5
FIRST_LINEAR_TOL = 4e-3   # audio of a linear channel's first block (AGC start-up on numerically-zero samples, linear.c:271-272);
                          # observed <= 7e-4 at powers of two, 2.5e-3 at N = 15360 (the mixed-radix transforms round differently from
                          # the oracle's recursion: the start-up samples themselves are rounding noise)
EXACT_TOL = 2e-6          # status sums against float64 arithmetic on the kernel's own samples


def _run_bank(plan, geom, iq, nblocks, fwd_mode, compute_n0=False, per_call=None, pl_tone=True):
    per_call = per_call or nblocks
    bank = kq.Bank(geom["samprate"], geom["L"], geom["M"], geom["D"], len(plan), per_call,
                   compute_n0=compute_n0, fwd_mode=fwd_mode, pl_tone=pl_tone)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    L = geom["L"]
    res = [dict(audio=[], status=[], filt=[]) for _ in plan]
    done = 0
    while done < nblocks:
        nb = min(per_call, nblocks - done)
        bank.push_iq(iq[done * L:(done + nb) * L])
        assert bank.process() == nb
        for c in range(len(plan)):
            for b in range(nb):
                res[c]["audio"].append(bank.audio(c, b))
                res[c]["status"].append(bank.status(c, b))
                res[c]["filt"].append(bank.filter_output(c, b))
        done += nb
    mode = bank.fwd_mode
    bank.close()
    return res, mode


def _fm_readings_float64(prev, cur, dsr):
    """foffset and pdeviation (fm.c:125-154) of a fully open block without blanked samples, in float64 on the filter output
    the kernel itself produced: y_n = arg(s_n conj(s_{n-1})), the block's first sample against the last one of the block
    before; foffset = dsr mean(y) / 2 pi, pdeviation = dsr max(y_max - mean, mean - y_min) / 2 pi."""
    s = np.concatenate([prev[-1:], cur]).astype(np.complex128)
    y = np.angle(s[1:] * np.conj(s[:-1]))
    avg = y.mean()
    return dsr * avg / (2 * np.pi), dsr * max(y.max() - avg, avg - y.min()) / (2 * np.pi)


def _compare(plan, got, want, skip_blocks=0, check_n0=False, geom=None):
    for c, p in enumerate(plan):
        auds, sts, filts = want[c]
        nb = len(auds)
        g = got[c]
        assert len(g["audio"]) == nb
        for b in range(nb):
            assert g["status"][b]["nout"] == sts[b]["nout"], (c, b)
            assert len(g["audio"][b]) == len(auds[b])
        filt_g = np.concatenate(g["filt"][skip_blocks:])
        filt_w = np.concatenate(filts[skip_blocks:])
        assert rel_rms(filt_g, filt_w) < FILT_TOL, ("filter", c, p["demod"], rel_rms(filt_g, filt_w))
        # Linear AGC start-up: with an all-zero history the first filter outputs are numerically zero and
        # the AGC (linear.c:271-272) divides by them, turning float rounding noise into O(1e-4) gain
        # differences until the first real attack.  Audio of that first block is held to FIRST_LINEAR_TOL (below)
        # instead of 1e-5; its filter output, its integer state and every later block are compared like any other.
        sk = max(skip_blocks, 1) if p["demod"] == "linear" else skip_blocks
        a_g = np.concatenate(g["audio"][sk:])
        a_w = np.concatenate(auds[sk:])
        assert rel_rms(a_g, a_w) < AUDIO_TOL, ("audio (linear channels: from block 1 on)", c, p["demod"], rel_rms(a_g, a_w))
        if p["demod"] == "linear" and skip_blocks == 0:
            # That first block, explicitly: its filter output to 1e-5 (the rounding noise of a window that is half zeros),
            # its sample count and hang counter exactly (below), and its audio to FIRST_LINEAR_TOL -- the gain the AGC lands
            # on after dividing by those near-zero samples, relative to the block's own RMS.
            assert rel_rms(g["filt"][0], filts[0]) < FILT_TOL, ("first linear block, filter", c, rel_rms(g["filt"][0], filts[0]))
            if np.abs(auds[0]).max() > 0:
                e0 = rel_rms(g["audio"][0], auds[0])
                assert e0 < FIRST_LINEAR_TOL, ("first linear block, audio", c, e0)
        for b in range(nb):
            sg, sw = g["status"][b], sts[b]
            assert sg["squelch_count"] == sw["squelch_count"], (c, b)
            assert sg["hangcount"] == sw["hangcount"], (c, b)
            assert sg["blanked"] == sw["blanked"], (c, b)
            assert np.isnan(sg["plfreq"]) == np.isnan(sw["plfreq"]), (c, b, sg["plfreq"], sw["plfreq"])
            if not np.isnan(sw["plfreq"]):
                assert sg["plfreq"] == sw["plfreq"], (c, b)        # same peak bin (fm.c:260-267)
            np.testing.assert_allclose(sg["bb_power"], sw["bb_power"], rtol=1e-5)
            # (the oracle sums in sequence, fm.c:93-97, the kernel as a tree: observed 5e-7 apart.)  Against float64
            # arithmetic on the samples the kernel itself produced, bb_power = sum |s|^2 / (2 olen) holds to EXACT_TOL
            exact = float(np.sum(np.abs(g["filt"][b].astype(np.complex128)) ** 2) / (2 * len(g["filt"][b])))
            np.testing.assert_allclose(sg["bb_power"], exact, rtol=EXACT_TOL, err_msg="bb_power vs float64 (%d, %d)" % (c, b))
            # (the block-level comparison above already covers this block's filter output)
            np.testing.assert_allclose(sg["if_power"], sw["if_power"], rtol=2e-4)
            if p["demod"] == "fm":
                # snr = a^2/(2(bb - a^2)) - 1 (fm.c:101-102) cancels catastrophically at high SNR: compare the
                # well-conditioned variance ratio 1/(1+snr) = 2 var/a^2, which float summation order perturbs by ~1e-6 (up to ~1e-5 at olen = 2048)
                np.testing.assert_allclose(1.0 / (1.0 + sg["snr"]), 1.0 / (1.0 + sw["snr"]), rtol=2e-4, atol=2e-5)
                np.testing.assert_allclose(sg["foffset"], sw["foffset"], rtol=1e-4, atol=1e-2)
                np.testing.assert_allclose(sg["pdeviation"], sw["pdeviation"], rtol=1e-4, atol=1e-2)
                # ... and, where the block and the one before it are fully open with no sample blanked (so that the hold
                # rule of fm.c:128-144 is not involved), against float64 arithmetic on the kernel's own filter output:
                # what is left is atan2f's rounding, 2e-5 of the deviation and 2e-3 Hz on the offset (which may be near 0)
                if geom is not None and b > 0 and all(g["status"][k]["squelch_count"] == 0 and g["status"][k]["blanked"] == 0
                                                      for k in (b - 1, b)):
                    fo, pd = _fm_readings_float64(g["filt"][b - 1], g["filt"][b], geom["samprate"] / geom["D"])
                    np.testing.assert_allclose(sg["foffset"], fo, rtol=2e-5, atol=2e-3, err_msg="foffset vs float64 (%d, %d)" % (c, b))
                    np.testing.assert_allclose(sg["pdeviation"], pd, rtol=2e-5, atol=2e-3, err_msg="pdeviation vs float64 (%d, %d)" % (c, b))
            else:
                if b > 0 or p["demod"] == "am":
                    np.testing.assert_allclose(sg["agc_gain"], sw["agc_gain"], rtol=2e-5)
            if check_n0:
                np.testing.assert_allclose(sg["n0"], sw["n0"], rtol=2e-4)


def _compare_counting_ties(plan, got, want, nblocks, check_n0=True):
    """_compare channel by channel; a channel that fails it may still be a threshold tie of the reference's own algorithm
    -- a comparison within one float rounding of its threshold, where two correct float transforms decide differently:
      "n0":  a bin at compute_n0's 2 x mean cut (radio.c:414-420) moves n0 by a bin's worth (<= 5e-3), nothing else;
      "agc": `gain * amplitude > headroom` (linear.c:271, am.c:66) with a new envelope peak within 1e-6 of the one that
             set the gain: the hang counter restarts or not, the two gain tracks stay within 1e-5 of each other.
    Such a channel must still agree in sample counts, squelch state, filter output (1e-5) and audio (1e-4).  Returns
    [(channel, kind, worst deviation)] for the caller to hold against its budget; anything else fails here."""
    flips = []
    for c, p in enumerate(plan):
        try:
            _compare([p], [got[c]], [want[c]], check_n0=check_n0)
            continue
        except AssertionError as e:
            first = str(e)[:300]
        auds, sts, filts = want[c]
        g = got[c]
        assert all(g["status"][b]["nout"] == sts[b]["nout"] and g["status"][b]["squelch_count"] == sts[b]["squelch_count"]
                   and g["status"][b]["blanked"] == sts[b]["blanked"] for b in range(nblocks)), ("not a tie", c, p, first)
        assert rel_rms(np.concatenate(g["filt"]), np.concatenate(filts)) < FILT_TOL, ("not a tie", c, p, first)
        sk = 1 if p["demod"] == "linear" else 0
        assert rel_rms(np.concatenate(g["audio"][sk:]), np.concatenate(auds[sk:])) < 1e-4, ("not a tie", c, p, first)
        try:
            _compare([p], [got[c]], [want[c]], check_n0=False)
            worst = max(abs(g["status"][b]["n0"] / sts[b]["n0"] - 1) for b in range(nblocks))
            assert check_n0 and worst < 5e-3, ("not a threshold tie", c, p, worst, first)
            flips.append((c, "n0", worst))
        except AssertionError:
            assert p["demod"] != "fm", ("not a tie", c, p, first)
            worst = max(abs(g["status"][b]["agc_gain"] / sts[b]["agc_gain"] - 1) for b in range(sk, nblocks))
            assert worst < 1e-5, ("not an AGC tie", c, p, worst, first)
            if check_n0:
                np.testing.assert_allclose([g["status"][b]["n0"] for b in range(nblocks)], [sts[b]["n0"] for b in range(nblocks)],
                                           rtol=5e-3)
            flips.append((c, "agc", worst))
    return flips


def _n0_ties_are_ties(plan, geom, iq, nblocks, flips, rate):
    """Every channel _compare_counting_ties called an n0 tie, run once more with its master spectra captured: the
    kernel's n0 must be what float64 arithmetic gives for compute_n0 on those spectra (common.n0_float64) -- the
    kernel decided the bin at the cut the way exact arithmetic does (or within 2e-5 of it); the float32 oracle, summing its
    first pass in sequence, is the one that differs.  rate: the smoothing constant (fm.c:82 0.01, am.c:47 / linear.c:124 0.001)."""
    from common import n0_float64
    for c, kind, _ in flips:
        if kind != "n0":
            continue
        p = plan[c]
        bank = kq.Bank(geom["samprate"], geom["L"], geom["M"], geom["D"], 1, nblocks, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
        bank.add_channel(bank_cfg(p))
        try:
            bank.spectrum(0, 0)          # arms the capture
        except Exception:
            pass
        bank.push_iq(iq[:nblocks * geom["L"]])
        assert bank.process() == nblocks
        n0 = None
        for b in range(nblocks):
            fresh = n0_float64(bank.spectrum(0, b), geom["samprate"], p["low"], p["high"])
            n0 = fresh if n0 is None else n0 + rate * (fresh - n0)
            np.testing.assert_allclose(bank.status(0, b)["n0"], n0, rtol=2e-5, err_msg="channel %d block %d" % (c, b))
        bank.close()


def _small_geom(D):
    return dict(samprate=192000, L=512, M=513, D=D)


@pytest.mark.parametrize("D", [4, 16])
def test_small_mixed_full(gpu, D):
    """N=1024 chain, FM / FM-flat / AM / USB / LSB / ISB stereo / IQ stereo with shift, full forward path + n0."""
    geom = _small_geom(D)
    fs = geom["samprate"]
    plan = []
    for e, kind in ((28, "fm"), (29, "fm"), (30, "am"), (31, "ssb"), (35, "ssb")):
        p = wl._mode_params(kind if kind != "ssb" else "ssb", e)
        p.update(second_lo=-(wl.emitter_freq(e, fs) + 3.7), emitter=e)
        plan.append(p)
    plan[1]["flat"] = 1
    isb = dict(demod="linear", low=-5000.0, high=5000.0, hangtime=1.1, recovery_rate=6.0, isb=1, channels=2,
               second_lo=-(wl.emitter_freq(31, fs) + 1.0))
    iqs = dict(demod="linear", low=-5000.0, high=5000.0, hangtime=1.1, recovery_rate=6.0, channels=2, shift=700.0,
               second_lo=-(wl.emitter_freq(30, fs) - 2.0))
    plan += [isb, iqs]
    nblocks = 12
    # boost emitters so post-filter SNR leaves wide decision margins
    iq = wl.make_iq(fs, nblocks * geom["L"], seed=11, emitters=range(24, 40))
    want = run_oracle(plan, geom, iq, nblocks, compute_n0=1)
    got, mode = _run_bank(plan, geom, iq, nblocks, kq.KQ_FWD_FULL, compute_n0=True, per_call=5)
    assert mode == kq.KQ_FWD_FULL
    _compare(plan, got, want, check_n0=True)


def test_cfg1_geometry_fm(gpu):
    """BASELINE configs[0] geometry: 192 kHz, N=16384, D=4, FM; 6 blocks."""
    g = wl.GEOMETRY["cfg1"]
    plan = wl.channel_plan("cfg1", 1)
    nblocks = 6
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=5)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, compute_n0=True, per_call=3)
    _compare(plan, got, want, check_n0=True)


@pytest.mark.parametrize("name,nchan", [("cfg2", 6), ("cfg3", 8)])
def test_config_geometry_full(gpu, name, nchan):
    g = wl.GEOMETRY[name]
    plan = wl.channel_plan(name, nchan)
    nblocks = 4
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=7)
    # compute_n0 on: at these sample rates k*samprate overflows the reference's int (radio.c:407), so this also
    # pins the wrapped passband mask of the full-spectrum kernel
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, compute_n0=True, per_call=4)
    _compare(plan, got, want, check_n0=True, geom=g)


def test_full_size_pruned_and_full_spectrum_agree_on_every_channel(gpu):
    """BASELINE configs[2] at its full size (1024 mixed channels): the pruned kernel and the full-spectrum kernel
    are independent implementations of the same filter; every channel must agree between them to the parity
    tolerance, with identical sample counts and squelch decisions."""
    g = wl.GEOMETRY["cfg3"]
    plan = wl.channel_plan("cfg3", g["channels"])
    nblocks = 6
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=21)
    res = {}
    for mode in (kq.KQ_FWD_PRUNED, kq.KQ_FWD_FULL):
        bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], len(plan), nblocks, fwd_mode=mode)
        for p in plan:
            bank.add_channel(bank_cfg(p))
        bank.push_iq(iq)
        assert bank.process() == nblocks
        assert bank.fwd_mode == mode
        res[mode] = [([bank.audio(c, b) for b in range(nblocks)], [bank.status(c, b) for b in range(nblocks)])
                     for c in range(len(plan))]
        bank.close()
    worst = 0.0
    for c, p in enumerate(plan):
        (a0, s0), (a1, s1) = res[kq.KQ_FWD_PRUNED][c], res[kq.KQ_FWD_FULL][c]
        for b in range(nblocks):
            assert s0[b]["nout"] == s1[b]["nout"] and s0[b]["squelch_count"] == s1[b]["squelch_count"], (c, b)
            assert s0[b]["blanked"] == s1[b]["blanked"] and s0[b]["hangcount"] == s1[b]["hangcount"], (c, b)
        sk = 1 if p["demod"] == "linear" else 0          # AGC start-up on numerically-zero samples, see _compare
        e = rel_rms(np.concatenate(a0[sk:]), np.concatenate(a1[sk:]))
        worst = max(worst, e)
        assert e < AUDIO_TOL, (c, p["demod"], e)
    assert worst > 0          # two different code paths, not one result read twice


GEOMETRIES = [
    # N,     L,     M,    D,   samprate, forward mode, compute_n0
    (512,   256,   257,   2,   192000, "full", True),        # N/D = 256 on the LDS kernel
    (1024,  768,   257,   4,   192000, "full", True),        # L != M - 1
    (2048,  1024,  1025,  8,   384000, "full", False),
    (4096,  3072,  1025,  16,  2000000, "full", True),
    (4096,  2048,  2049,  64,  2000000, "pruned", False),    # pruned, R = 64
    (8192,  4096,  4097,  128, 2000000, "pruned", False),    # pruned, R = 128
    (8192,  4096,  4097,  64,  2000000, "auto", False),      # N/D = 128 below 65536: full path
    (16384, 12288, 4097,  64,  2000000, "full", True),       # full-spectrum kernel, L != M - 1, N/D = 256
    (16384, 8192,  8193,  32,  1000000, "full", True),       # N/D = 512
    (16384, 12800, 3585,  64,  2000000, "full", True),       # L not a multiple of 1024: no row-paired copy, 8-byte window loads
    (16384, 8192,  8193,  2,   192000, "full", False),       # N/D = 8192: LDS filter kernel, largest FM working set
    (32768, 16384, 16385, 128, 4000000, "full", False),      # beyond one LDS block: split kernel, N/D = 256
    (65536, 49152, 16385, 32,  8000000, "full", False),      # split kernel at its limits: N = 65536, N/D = 2048, L != M - 1
    (65536, 32768, 32769, 512, 20000000, "auto", False),     # cfg 5 geometry with FM / AM on it too
    (65536, 32768, 32769, 512, 20000000, "auto", True),      # the same with compute_n0: four sibling workgroups per channel-block
    (65536, 49152, 16385, 32,  8000000, "full", True),       # N = 65536 full-spectrum kernel, N/D = 2048, L != M - 1
    (65536, 33280, 32257, 256, 4000000, "full", True),       # L not a multiple of 1024: 8-byte window loads
    (65536, 32768, 32769, 8,   2000000, "full", True),       # N/D = 8192: the slave reads bins of every sub-transform row
    # sizes with factors 3 and 5 (round 6; FFTW plans them all, filter.c:78,132): front ends that are not 48 kHz x 2^k
    (9600,  4800,  4801,  5,   240000, "auto", True),        # 240 kHz: decimate = samprate / 48000 = 5 (radio_status.c:266); N/D = 1920
    (15360, 7680,  7681,  8,   384000, "full", True),        # N = 2^10 3 5, N/D = 1920, PL slave of 60 points
    (12000, 6000,  6001,  10,  480000, "auto", True),        # N = 2^5 3 5^3, N/D = 1200
    (7680,  3840,  3841,  4,   192000, "auto", False),       # the reference's default L with M = L + 1: N/D = 1920
    (3840,  2880,  961,   2,   96000,  "auto", True),        # L != M - 1, decimate 2
    (9600,  4800,  4801,  150, 7200000, "auto", True),       # N/D = 64 behind a master with factors 3 and 5: the N/D = 64 demodulators
    (6000,  3000,  3001,  5,   240000, "auto", True),        # N/D = 1200, olen = 600
    (3000,  1500,  1501,  2,   96000,  "auto", True),        # N = 2^3 3 5^3 (one radix-2 pass), N/D = 1500
    (24000, 12000, 12001, 25,  1200000, "full", False),      # beyond one LDS block with factors 3 and 5: split kernel, 2 x 12000
    (38400, 19200, 19201, 40,  1920000, "full", False),      # split kernel, 3 x 12800, N/D = 960
    # ... and 7 (a rate of 7 x 48 kHz; 2 x 7^4 = four radix-7 passes)
    (13440, 6720,  6721,  7,   336000, "auto", True),        # N = 2^7 3 5 7, N/D = 1920
    (7168,  3584,  3585,  4,   192000, "auto", True),        # N = 2^10 7, N/D = 1792 = 2^8 7, 896 samples per block
    (4802,  2401,  2402,  7,   336000, "auto", True),        # N = 2 7^4, N/D = 686 = 2 7^3, 343 samples per block
    # decimate 1 (samprate / 48000 at a 48 kHz front end, radio_status.c:266): the slave's transform is the master's size
    (2048,  1024,  1025,  1,   48000,  "auto", True),
    (1920,  960,   961,   1,   48000,  "auto", True),        # 20 ms blocks at 48 kHz: N = 2^7 3 5
    (8192,  3840,  4353,  1,   48000,  "full", False),       # the largest: two 64 KiB buffers side by side
    # an impulse response of two and three blocks (M >= 2 L): window_filter's in-place rotation (filter.c:389-390) then forms
    # its first taps from slots it has already written -- the response design follows the reference in that (both the complex
    # design and the REAL one of the FM audio filter, fm.c:64: 16 samples per block against 49 taps)
    (4096,  1024,  3073,  64,  2000000, "auto", False),
    (4096,  1024,  3073,  16,  2000000, "full", True),
]


def _sweep_case(N, L, M, D, fs):
    ds = fs / D
    bw = min(8000.0, 0.2 * ds)
    f0 = 0.11 * fs
    plan = [dict(demod="fm", low=-bw, high=bw, second_lo=-f0),
            dict(demod="am", low=-0.6 * bw, high=0.6 * bw, second_lo=-f0 - 0.37 * ds, hangtime=0.0, recovery_rate=50.0),
            dict(demod="linear", low=0.02 * bw, high=0.4 * bw, second_lo=-f0 + 0.21 * ds, hangtime=1.1, recovery_rate=6.0),
            dict(demod="linear", low=-0.5 * bw, high=0.5 * bw, second_lo=-f0 - 1.3, isb=1, channels=2, hangtime=1.1,
                 recovery_rate=6.0, shift=0.01 * ds)]
    nblocks = 5
    rng = np.random.default_rng(N + D)
    t = np.arange(nblocks * L) / fs
    dev = min(3000.0, 0.25 * bw)
    sig = 0.3 * np.exp(1j * (2 * np.pi * f0 * t + (dev / 500.0) * np.sin(2 * np.pi * 500.0 * t)))
    sig += 0.1 * np.exp(2j * np.pi * (f0 + 0.37 * ds) * t) * (1 + 0.5 * np.sin(2 * np.pi * 300.0 * t))
    sig += 0.05 * np.exp(2j * np.pi * (f0 - 0.21 * ds + 0.2 * bw) * t)       # a tone inside the USB channel's passband
    # 30 dB in-channel SNR on the FM carrier: with less noise the FM variance estimate (fm.c:101) is rounding noise
    # and the squelch decision with it
    sigma = 0.3 * 10 ** (-30 / 20) / np.sqrt(4 * bw / fs)
    iq = (sig + sigma * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    return plan, iq, nblocks


@pytest.mark.parametrize("N,L,M,D,fs,mode,n0", GEOMETRIES)
def test_geometry_sweep(gpu, N, L, M, D, fs, mode, n0):
    """Every kernel family over the geometries it claims: unequal L / M splits, small and large N/D, both sample-rate
    regimes of compute_n0's wrapped mask, FM / AM / USB / ISB-stereo on each."""
    assert L + M - 1 == N
    g = dict(samprate=fs, L=L, M=M, D=D)
    plan, iq, nblocks = _sweep_case(N, L, M, D, fs)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=int(n0))
    fwd = {"full": kq.KQ_FWD_FULL, "pruned": kq.KQ_FWD_PRUNED, "auto": kq.KQ_FWD_AUTO}[mode]
    got, used = _run_bank(plan, g, iq, nblocks, fwd, compute_n0=n0, per_call=3)
    if mode != "auto":
        assert used == fwd
    _compare(plan, got, want, check_n0=n0, geom=g)


# KQ_FUZZ_EXTRA=n in the environment: n more seeds for each of the drawn families below (a longer hunt by hand:
#   KQ_FUZZ_EXTRA=200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "drawn_at_random")
_EXTRA = [100000 + k for k in range(int(os.environ.get("KQ_FUZZ_EXTRA", "0")))]


def _smooth_sizes(lo, hi):
    out = []
    for n in range(lo, hi + 1, 2):
        m = n
        for p in (2, 3, 5, 7):
            while m % p == 0:
                m //= p
        if m == 1:
            out.append(n)
    return out


@pytest.mark.parametrize("seed", list(range(48)) + _EXTRA)
def test_geometries_drawn_at_random(gpu, seed):
    """The sweep above is a list somebody chose.  Here the geometry is drawn: N/decimate any even 2^a 3^b 5^c 7^d in 64..2048,
    decimate from 1 to 64 (the front end at decimate x 48 kHz, radio_status.c:266), the impulse response between a quarter and
    0.55 of N/decimate (so M - 1 < L, = L and > L all occur), 2 to 4 blocks per call, compute_n0 on; FM / AM / USB / ISB
    against the oracle as in the sweep."""
    rng = np.random.default_rng(9000 + seed)
    while True:
        nd = int(rng.choice(_smooth_sizes(64, 2048)))
        D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 25, 32, 64]))
        N = nd * D
        if N <= (8192 if D == 1 else 16384) and N >= 256:
            break
    k = int(rng.integers(nd // 4, int(nd * 0.55) + 1))
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    g = dict(samprate=fs, L=L, M=M, D=D)
    plan, iq, nblocks = _sweep_case(N, L, M, D, fs)
    if D == 1:                                  # (the sweep's AM channel would straddle the band edge at decimate 1)
        plan[1]["second_lo"] = -0.11 * fs + 0.2 * fs
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=int(rng.integers(2, 5)))
    try:
        _compare(plan, got, want, check_n0=True, geom=g)
    except AssertionError as e:
        raise AssertionError("geometry N = %d (L = %d, M = %d), decimate %d, %d Hz: %s" % (N, L, M, D, fs, e)) from e


@pytest.mark.parametrize("nd,D,k", [(4, 4, 2), (4, 8, 1), (6, 4, 3), (8, 2, 4), (8, 16, 3), (10, 5, 4), (12, 4, 6), (14, 8, 6), (16, 1, 8),
                                    (16, 4, 9), (20, 3, 8), (24, 2, 11), (32, 1, 16), (32, 64, 16), (36, 7, 17), (48, 5, 20)])
def test_the_smallest_geometries(gpu, nd, D, k):
    """The small end of what kq_bank_create takes (N >= 16, N/decimate >= 4): slaves of 4 ... 48 points, two to 28 samples per
    block, workgroups with more threads than points -- FM / AM / USB / ISB against the oracle as in the sweep."""
    N, M, L, fs = nd * D, k * D + 1, (nd - k) * D, 48000 * D
    if N < 16:
        pytest.skip("N < 16")
    g = dict(samprate=fs, L=L, M=M, D=D)
    plan, iq, nblocks = _sweep_case(N, L, M, D, fs)
    nblocks = 40
    plan, iq, _ = _sweep_case(N, L, M, D, fs)
    rng = np.random.default_rng(nd * 100 + D)
    t = np.arange(nblocks * L) / fs
    ds, f0 = fs / D, 0.11 * fs
    bw = min(8000.0, 0.2 * ds)
    sig = 0.3 * np.exp(1j * (2 * np.pi * f0 * t + 4.0 * np.sin(2 * np.pi * 500.0 * t)))
    sig += 0.1 * np.exp(2j * np.pi * (f0 + 0.37 * ds) * t) * (1 + 0.5 * np.sin(2 * np.pi * 300.0 * t))
    sig += 0.05 * np.exp(2j * np.pi * (f0 - 0.21 * ds + 0.2 * bw) * t)
    sigma = 0.3 * 10 ** (-30 / 20) / np.sqrt(4 * bw / fs)          # 30 dB in the FM channel, as in the sweep
    iq = (sig + sigma * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=7)
    try:
        _compare(plan, got, want, check_n0=True, geom=g)
    except AssertionError as e:
        raise AssertionError("geometry N = %d (L = %d, M = %d), decimate %d: %s" % (N, L, M, D, e)) from e


def test_filter_and_shift_changed_while_running(gpu):
    """What the UI does between blocks (display.c:161-177, radio.c:304-311): new filter edges / Kaiser beta and a new
    post-detection shift; the response is swapped at the next block (filter.c:538-543), the shift oscillator keeps
    its phase (osc.c:24-27)."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L = g["samprate"], g["L"]
    nb = 9
    plan = wl.channel_plan("cfg1", 1) + [dict(demod="linear", low=100.0, high=3000.0, second_lo=-20000.0, hangtime=1.1,
                                               recovery_rate=6.0, shift=150.0),
                                          dict(demod="am", low=-5000.0, high=5000.0, second_lo=-20000.0, recovery_rate=50.0)]
    iq = wl.make_iq(fs, nb * L, seed=51)
    t = np.arange(nb * L) / fs
    iq = (iq + 0.2 * np.exp(2j * np.pi * 21000.0 * t) * (1 + 0.4 * np.sin(2 * np.pi * 400.0 * t))).astype(np.complex64)
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), 3, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
    chans = []
    for p in plan:
        bank.add_channel(bank_cfg(p))
        chans.append(ko.Channel(oracle_cfg(p, fs, L, g["M"], g["D"], compute_n0=1)))
    changes = {3: [("filter", 0, (-6000.0, 6000.0, 5.0)), ("filter", 1, (250.0, 2400.0, 3.0)), ("shift", 1, 300.0)],
               6: [("filter", 2, (-3000.0, 4000.0, 1.0)), ("shift", 1, 0.0), ("filter", 0, (-9000.0, 7000.0, 3.0))]}
    for first in range(0, nb, 3):
        for kind, c, arg in changes.get(first, []):
            if kind == "filter":
                bank.set_filter(c, *arg)
                chans[c].set_filter(*arg)
            else:
                bank.set_shift(c, arg)
                chans[c].set_shift(arg)
        bank.push_iq(iq[first * L:(first + 3) * L])
        assert bank.process() == 3
        for b in range(3):
            for c, ch in enumerate(chans):
                wa, ws, _, _ = ch.block(iq[(first + b) * L:(first + b + 1) * L])
                ga, gs = bank.audio(c, b), bank.status(c, b)
                assert gs["nout"] == ws["nout"]
                if not (plan[c]["demod"] == "linear" and first + b == 0):       # AGC start-up, see _compare
                    assert rel_rms(ga, wa) < AUDIO_TOL, (first + b, c, rel_rms(ga, wa))
                np.testing.assert_allclose(gs["n0"], ws["n0"], rtol=2e-4)
                np.testing.assert_allclose(gs["bb_power"], ws["bb_power"], rtol=2e-5)
    for c, ch in enumerate(chans):
        np.testing.assert_allclose(bank.response(c), ch.response(), rtol=0, atol=2e-9)
    bank.close()


def test_set_mode_restarts_the_demodulator(gpu):
    """set_mode on running channels (radio.c:322-374): FM -> AM, AM -> USB with a shift, USB -> FM, and a channel
    entering a carrier-tracking mode ahead of an existing PLL channel (whose loop state has to move slots).  The new
    demodulator starts from its prologue state; oscillators, n0, foffset and pdeviation carry over."""
    g = wl.GEOMETRY["cfg1"]
    fs, L = g["samprate"], g["L"]
    nb, per = 24, 4
    t = np.arange(nb * L) / fs
    rng = np.random.default_rng(61)
    sig = 0.1 * (1 + 0.5 * np.cos(2 * np.pi * 1000.0 * t)) * np.exp(2j * np.pi * (20000.0 + 23.0) * t)        # AM / CAM
    sig += 0.2 * np.exp(1j * (2 * np.pi * 50000.0 * t + 3.0 * np.sin(2 * np.pi * 700.0 * t)))                  # FM
    iq = (sig + 2e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    fm = dict(demod="fm", low=-8000.0, high=8000.0)
    am = dict(demod="am", low=-5000.0, high=5000.0, recovery_rate=50.0)
    usb = dict(demod="linear", low=100.0, high=3000.0, hangtime=1.1, recovery_rate=6.0, shift=120.0)
    cam = dict(demod="linear", low=-5000.0, high=5000.0, recovery_rate=50.0, pll=1)
    plan = [dict(fm, second_lo=-50000.0), dict(am, second_lo=-20000.0), dict(usb, second_lo=-50000.0),
            dict(cam, second_lo=-20000.0)]
    # at block 8 the first three rotate modes; at block 16 channel 1 becomes a second PLL channel in front of channel 3
    switch = {8: {0: am, 1: usb, 2: dict(fm, low=9000.0, high=-9000.0)}, 16: {1: dict(cam, square=0), 0: fm}}
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), per, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
    chans = []
    for p in plan:
        bank.add_channel(bank_cfg(p))
        chans.append(ko.Channel(oracle_cfg(p, fs, L, g["M"], g["D"], compute_n0=1)))
    cur = [dict(p) for p in plan]
    since = [0] * len(plan)                                  # blocks since the channel's demodulator (re)started
    for first in range(0, nb, per):
        for c, m in switch.get(first, {}).items():
            cur[c] = dict(m, second_lo=cur[c]["second_lo"])
            bank.set_mode(c, bank_cfg(cur[c]))
            chans[c].set_mode(oracle_cfg(cur[c], fs, L, g["M"], g["D"], compute_n0=1))
            since[c] = 0
        bank.push_iq(iq[first * L:(first + per) * L])
        assert bank.process() == per
        for b in range(per):
            for c, ch in enumerate(chans):
                wa, ws, _, _ = ch.block(iq[(first + b) * L:(first + b + 1) * L])
                ga, gs = bank.audio(c, b), bank.status(c, b)
                assert gs["nout"] == ws["nout"], (first + b, c)
                assert (gs["squelch_count"], gs["hangcount"], gs["pll_lock"], gs["lock_count"]) == \
                       (ws["squelch_count"], ws["hangcount"], ws["pll_lock"], ws["lock_count"]), (first + b, c)
                np.testing.assert_allclose(gs["n0"], ws["n0"], rtol=2e-4)
                linear = cur[c]["demod"] == "linear"
                tol = 2e-5 if cur[c].get("pll") else AUDIO_TOL
                if not (linear and since[c] == 0) and not (cur[c].get("pll") and since[c] < 6):
                    assert rel_rms(ga, wa) < tol, (first + b, c, cur[c]["demod"], rel_rms(ga, wa))
                since[c] += 1
    bank.close()


def test_channels_come_and_go(gpu):
    """kq_bank_remove_channel / kq_bank_add_channel on a running bank.  The others keep their numbers and their
    carried state -- including a carrier loop whose slot moves when a PLL channel ahead of it leaves or arrives; a new
    channel takes the lowest hole and starts from its prologue state on the master's live history."""
    g = wl.GEOMETRY["cfg1"]
    fs, L, M = g["samprate"], g["L"], g["M"]
    nb, per = 28, 4
    t = np.arange(nb * L) / fs
    rng = np.random.default_rng(67)
    sig = 0.1 * (1 + 0.5 * np.cos(2 * np.pi * 1000.0 * t)) * np.exp(2j * np.pi * (20000.0 + 23.0) * t)        # AM / CAM
    sig += 0.1 * (1 + 0.4 * np.cos(2 * np.pi * 600.0 * t)) * np.exp(2j * np.pi * (-30000.0 - 11.0) * t)       # AM / CAM
    sig += 0.2 * np.exp(1j * (2 * np.pi * 50000.0 * t + 3.0 * np.sin(2 * np.pi * 700.0 * t)))                  # FM
    iq = (sig + 2e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    fm = dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-50000.0)
    am = dict(demod="am", low=-5000.0, high=5000.0, recovery_rate=50.0, second_lo=-20000.0)
    usb = dict(demod="linear", low=100.0, high=3000.0, hangtime=1.1, recovery_rate=6.0, shift=120.0, second_lo=-50000.0)
    cam_a = dict(demod="linear", low=-5000.0, high=5000.0, recovery_rate=50.0, pll=1, second_lo=-20000.0)
    cam_b = dict(cam_a, second_lo=30000.0)
    bank = kq.Bank(fs, L, M, g["D"], 6, per, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
    live = {}                                               # channel number -> [plan, oracle, blocks since start]

    def add(p, first, expect):
        c = bank.add_channel(bank_cfg(p))
        assert c == expect
        o = ko.Channel(oracle_cfg(p, fs, L, M, g["D"], compute_n0=1))
        if first:
            o.prime_history(iq[first * L - (M - 1):first * L])
        live[c] = [p, o, 0]

    def remove(c):
        bank.remove_channel(c)
        live.pop(c)[1].close()
        assert not bank.channel_active(c)

    for k, p in enumerate([fm, cam_a, usb, cam_b, dict(fm, flat=1)]):
        add(p, 0, k)
    events = {
        8: lambda: (remove(1), remove(4)),                   # a PLL channel ahead of cam_b leaves; the last slot goes
        12: lambda: (add(am, 12, 1), add(cam_a, 12, 4)),     # hole first, then the end (a carrier loop behind cam_b)
        20: lambda: (remove(1), add(cam_a, 20, 1), remove(0)),   # a carrier loop arrives ahead of both others
        24: lambda: add(dict(fm, second_lo=-50010.0), 24, 0),
    }
    for first in range(0, nb, per):
        if first in events:
            events[first]()
            if first == 8:
                assert bank.num_channels == 4
                with pytest.raises(kq.KqError):
                    bank.set_shift(1, 100.0)
                with pytest.raises(kq.KqError):
                    bank.remove_channel(1)
        bank.push_iq(iq[first * L:(first + per) * L])
        assert bank.process() == per
        for b in range(per):
            for c, (p, o, since) in sorted(live.items()):
                wa, ws, _, _ = o.block(iq[(first + b) * L:(first + b + 1) * L])
                ga, gs = bank.audio(c, b), bank.status(c, b)
                assert gs["nout"] == ws["nout"], (first + b, c)
                assert (gs["squelch_count"], gs["hangcount"], gs["pll_lock"], gs["lock_count"]) == \
                       (ws["squelch_count"], ws["hangcount"], ws["pll_lock"], ws["lock_count"]), (first + b, c)
                np.testing.assert_allclose(gs["n0"], ws["n0"], rtol=2e-4)
                tol = 2e-5 if p.get("pll") else AUDIO_TOL
                if not (p["demod"] == "linear" and since == 0) and not (p.get("pll") and since < 6):
                    assert rel_rms(ga, wa) < tol, (first + b, c, p["demod"], rel_rms(ga, wa))
                live[c][2] += 1
    bank.close()


@pytest.mark.parametrize("name", ["cfg4", "cfg5", "cfg2"])
def test_holes_are_skipped_on_the_pruned_paths_and_in_the_pcm_stage(gpu, name):
    """kq_bank_remove_channel on the pruned forward kernels (N/D = 64 resident, 128 streamed, 256 resident) with the PCM
    stage on: the launches go over the list of active channels, the survivors keep their slots, state and output."""
    import kq_oracle as ko
    from common import oracle_cfg
    g = wl.GEOMETRY[name]
    fs, L, M = g["samprate"], g["L"], g["M"]
    plan = wl.channel_plan(name, 11)
    for p in plan:                       # unswept: cfg 2's N/D = 256 pruned kernel takes no sweep
        p["second_lo"] -= p["doppler"]
        p["doppler"] = p["doppler_rate"] = 0.0
    per, ncalls = 2, 3
    iq = wl.make_iq(fs, per * ncalls * L, seed=37)
    chans = {c: ko.Channel(oracle_cfg(p, fs, L, M, g["D"])) for c, p in enumerate(plan)}
    bank = kq.Bank(fs, L, M, g["D"], len(plan), per, fwd_mode=kq.KQ_FWD_PRUNED)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    bank.enable_pcm(True)
    for call in range(ncalls):
        if call == 1:
            for c in (0, 4, 9):
                bank.remove_channel(c)
                chans.pop(c).close()
        if call == 2:
            bank.remove_channel(10)
            chans.pop(10).close()
        bank.push_iq(iq[call * per * L:(call + 1) * per * L])
        assert bank.process() == per
        for b in range(per):
            x = iq[(call * per + b) * L:(call * per + b + 1) * L]
            for c, ch in chans.items():
                wa, ws, wf, _ = ch.block(x, want_filt=True)
                assert rel_rms(bank.filter_output(c, b), wf) < FILT_TOL, (call, b, c)
                ga = bank.audio(c, b)
                if not (plan[c]["demod"] == "linear" and call == 0 and b == 0):
                    assert rel_rms(ga, wa) < AUDIO_TOL, (call, b, c)
                words, silent = bank.pcm(c, b)
                want_words, want_silent, _ = ko.pcm_block(ga)
                assert np.array_equal(words, want_words) and silent == want_silent, (call, b, c)
    bank.close()
    for ch in chans.values():
        ch.close()


def _random_plan(rng, fs, n):
    """Channels tuned on emitters of the synthetic band with randomised mode, filter, AGC and tuning details"""
    plan = []
    for _ in range(n):
        e = int(rng.integers(0, 64))
        kind = wl.emitter_kind(e)
        off = float(rng.uniform(-40.0, 40.0))
        p = dict(second_lo=-(wl.emitter_freq(e, fs) + off), kaiser_beta=float(rng.uniform(0.0, 8.0)),
                 headroom=float(10 ** (-rng.uniform(6.0, 25.0) / 20)))
        if kind == "fm":
            w = float(rng.uniform(5000.0, 12000.0))
            p.update(demod="fm", low=-w, high=float(rng.uniform(5000.0, 12000.0)), flat=int(rng.random() < 0.3))
        elif kind == "am":
            if rng.random() < 0.5:
                p.update(demod="am", low=-float(rng.uniform(2500.0, 6000.0)), high=float(rng.uniform(2500.0, 6000.0)),
                         hangtime=float(rng.choice([0.0, 0.02])), recovery_rate=float(rng.uniform(5.0, 60.0)))
            else:    # an AM signal through the linear demodulator: mono, stereo I/Q or ISB
                p.update(demod="linear", low=-float(rng.uniform(2500.0, 6000.0)), high=float(rng.uniform(2500.0, 6000.0)),
                         hangtime=float(rng.choice([0.0, 0.02, 1.1])), recovery_rate=float(rng.uniform(3.0, 60.0)),
                         channels=int(rng.choice([1, 2])), isb=int(rng.random() < 0.4),
                         shift=float(rng.choice([0.0, rng.uniform(-500.0, 500.0)])))
        else:
            lo = float(rng.uniform(50.0, 400.0))
            hi = float(rng.uniform(2200.0, 3500.0))
            sgn = 1.0 if (e // 4) % 2 == 0 else -1.0
            p.update(demod="linear", low=min(sgn * lo, sgn * hi), high=max(sgn * lo, sgn * hi),
                     hangtime=float(rng.choice([0.0, 1.1])), recovery_rate=float(rng.uniform(3.0, 20.0)),
                     channels=int(rng.choice([1, 2])), shift=float(rng.choice([0.0, rng.uniform(-300.0, 300.0)])))
        plan.append(p)
    return plan


@pytest.mark.parametrize("seed,mode", [(1, "pruned"), (2, "full"), (3, "pruned"), (4, "full")])
def test_random_channel_plans(gpu, seed, mode):
    """24 channels with randomised modes, filter edges, Kaiser beta, headroom, hang / recovery, stereo / ISB, shift
    and tuning offsets at the cfg 3 geometry, on both forward paths."""
    g = wl.GEOMETRY["cfg3"]
    rng = np.random.default_rng(1000 + seed)
    plan = _random_plan(rng, g["samprate"], 24)
    nblocks = 6
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=100 + seed)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=int(mode == "full"))
    fwd = kq.KQ_FWD_PRUNED if mode == "pruned" else kq.KQ_FWD_FULL
    got, used = _run_bank(plan, g, iq, nblocks, fwd, compute_n0=(mode == "full"), per_call=4)
    assert used == fwd
    _compare(plan, got, want, check_n0=(mode == "full"))


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_random_channel_plans_generic_demodulators(gpu, seed):
    """The same randomised plans at the cfg 2 geometry (N/D = 256, 128 samples per block), where the block-parallel
    FM demodulator, its audio / PL kernels and the LDS AM / linear demodulators run instead of the wave-per-channel
    ones; ten blocks in calls of four."""
    g = wl.GEOMETRY["cfg2"]
    rng = np.random.default_rng(2000 + seed)
    plan = _random_plan(rng, g["samprate"], 18)
    nblocks = 10
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=200 + seed)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, used = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=4)
    assert used == kq.KQ_FWD_FULL
    _compare(plan, got, want, check_n0=True)


@pytest.mark.parametrize("N,L,M,D,fs", [(6000, 3000, 3001, 5, 240000),       # 600 samples per block: nine groups of 64 and one of 24
                                          (2048, 1024, 1025, 8, 384000),      # 128: two whole groups
                                          (3840, 2880, 961, 2, 96000)])       # 1440: 22 groups and one of 32
def test_every_form_of_the_agc_recurrence_on_a_keyed_signal(gpu, N, L, M, D, fs):
    """The wave-per-channel AM / SSB demodulator for any block length (k_demod_agc_any) walks the AGC recurrence of
    am.c:64-74 / linear.c:269-279 in four forms -- gain held while the hang counter outlasts a group, coasting where it
    runs out inside one, the counter-free form of a channel without a hang time, and the general one -- and picks per
    group of 64 samples.  A keyed carrier (on 9 ms, off 14 ms, two levels) in noise with hang times of 0, 2.5 ms, 30 ms
    and 1.1 s makes every form and every hand-over between them occur; the hang counter after each block is compared
    exactly, gains and audio at the usual bar."""
    g = dict(samprate=fs, L=L, M=M, D=D)
    ds = fs / D
    f0 = 0.13 * fs
    nblocks = 12
    t = np.arange(nblocks * L) / fs
    rng = np.random.default_rng(N)
    key = ((t % 0.023) < 0.009) * np.where((t % 0.046) < 0.023, 1.0, 0.35)
    sig = 0.2 * key * np.exp(2j * np.pi * (f0 + 700.0) * t)
    iq = (sig + 2e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = []
    for hang in (0.0, 0.0025, 0.03, 1.1):
        plan.append(dict(demod="linear", low=100.0, high=3000.0, second_lo=-f0, hangtime=hang, recovery_rate=20.0))
        plan.append(dict(demod="am", low=-4000.0, high=4000.0, second_lo=-f0 - 700.0, hangtime=hang, recovery_rate=50.0))
    plan.append(dict(demod="linear", low=-3000.0, high=3000.0, second_lo=-f0, hangtime=0.0025, recovery_rate=6.0, channels=2,
                     shift=0.004 * ds))
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, used = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=5)
    _compare(plan, got, want, check_n0=True, geom=g)
    hangs = {c: [st["hangcount"] for st in got[c]["status"]] for c in range(len(plan))}
    assert any(h != 0 for h in hangs[2]) and any(h == 0 for h in hangs[2]), hangs[2]     # 2.5 ms: runs out and is set again
    assert all(h == 0 for h in hangs[0]) and all(h > 0 for h in hangs[6][1:]), (hangs[0], hangs[6])


@pytest.mark.parametrize("seed,per_call", [(5, 4), (6, 5), (7, 11), (8, 1)])
def test_cfg2_geometry_without_the_pl_measurement(gpu, seed, per_call):
    """cfg 2 as SURVEY 8d measures it (pltask off): the de-emphasis overlap-save then runs in k_fm_audio256 -- one wave per
    pair of blocks, both real windows through one 256-point lane-exchange transform pair -- instead of the LDS kernel.
    Eleven blocks (an odd count: the last wave has one block only) in calls of 4, 5, 11 and 1, so that the history hand-over
    between calls is met at even and odd block counts; FM with and without de-emphasis among randomised plans."""
    g = wl.GEOMETRY["cfg2"]
    rng = np.random.default_rng(3000 + seed)
    plan = _random_plan(rng, g["samprate"], 14)
    fs = g["samprate"]
    plan += [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-(wl.emitter_freq(0, fs) + 3.0)),
             dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-(wl.emitter_freq(1, fs) - 7.0), flat=1),
             dict(demod="fm", low=-6000.0, high=9000.0, second_lo=-(wl.emitter_freq(4, fs) + 11.0), kaiser_beta=5.0)]
    nblocks = 11
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=300 + seed)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=per_call, pl_tone=False)
    _compare(plan, got, want, check_n0=True, geom=g)
    for c, p in enumerate(plan):
        if p["demod"] == "fm":
            assert all(np.isnan(st["plfreq"]) for st in got[c]["status"])


@pytest.mark.parametrize("name,pl_tone", [("cfg2", False), ("cfg2", True), ("cfg3", True), ("cfg1", True)])
def test_more_than_64_blocks_in_one_call(gpu, name, pl_tone):
    """A call of 150 blocks (max_blocks is the host's choice; a file replayed faster than real time uses long calls): the FM
    kernels walk a call in chunks of 64 blocks (k_demod_fm256's LDS, k_demod_fm's phases), the wave-per-channel ones block
    by block, the squelch / hang / n0 chains run through all of them.  FM, AM and SSB against the oracle, then 23 more
    blocks in a second call."""
    g = wl.GEOMETRY[name]
    fs = g["samprate"]
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-(wl.emitter_freq(0, fs) + 3.0)),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-(wl.emitter_freq(1, fs) - 7.0), flat=1),
            dict(demod="am", low=-5000.0, high=5000.0, second_lo=-wl.emitter_freq(2, fs), recovery_rate=50.0),
            dict(demod="linear", low=100.0, high=3000.0, second_lo=-wl.emitter_freq(3, fs), hangtime=1.1, recovery_rate=6.0),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=0.9 * 21.0 / 64 * fs)]    # half way between two emitters: squelch shut
    nblocks, last = (150 + 23, 23) if name != "cfg1" else (70 + 9, 9)
    iq = wl.make_iq(fs, nblocks * g["L"], seed=77)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=nblocks - last, pl_tone=pl_tone)
    if name == "cfg1":          # (at 192 kHz the emitters stand 2.7 kHz apart: no empty channel there)
        _compare(plan, got, want, check_n0=True, geom=g)
        return
    _compare(plan[:4], got[:4], want[:4], check_n0=True, geom=g)
    # the empty channel: 60 dB below the band's power, its output is compared at 5e-5 of its own RMS (float rounding of a
    # 16384-point transform of the whole band); what matters here is its squelch counter, which runs through the chunk borders
    _, sts, filts = want[4]
    assert rel_rms(np.concatenate(got[4]["filt"]), np.concatenate(filts)) < 5e-5
    assert [st["squelch_count"] for st in got[4]["status"]] == [st["squelch_count"] for st in sts]
    assert [st["blanked"] for st in got[4]["status"]] == [st["blanked"] for st in sts]
    if name == "cfg2":      # (32 samples per block at cfg 3: the SNR estimate of noise alone reopens the squelch now and then)
        assert any(st["squelch_count"] > 64 for st in sts)
    for a, w in zip(got[4]["audio"], want[4][0]):
        assert len(a) == len(w) and (not np.any(w) or rel_rms(a, w) < 1e-3)


def test_long_run_phase_continuity(gpu):
    """2.5 million input samples (300 blocks over five process calls, more than 150 renormalisation periods of the
    reference's NCO recurrence): the closed-form oscillators of the bank must not drift away from the oracle's
    sample-by-sample recurrence, on the pruned path and, with a swept Doppler, on the full path."""
    g = wl.GEOMETRY["cfg4"]
    nblocks = 300
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=41)
    plan = wl.channel_plan("cfg4", 3, first=500)
    want = run_oracle(plan, g, iq, nblocks)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_PRUNED, per_call=64)
    assert mode == kq.KQ_FWD_PRUNED
    _compare(plan, got, want)
    # the last blocks on their own: an accumulated phase error would show there first
    for c in range(len(plan)):
        assert rel_rms(np.concatenate(got[c]["audio"][-8:]), np.concatenate(want[c][0][-8:])) < AUDIO_TOL
    swept = [dict(plan[0], doppler=4000.0, doppler_rate=-60.0, second_lo=plan[0]["second_lo"] - 4000.0)]
    want = run_oracle(swept, g, iq[:100 * g["L"]], 100)
    got, _ = _run_bank(swept, g, iq[:100 * g["L"]], 100, kq.KQ_FWD_FULL, per_call=32)
    _compare(swept, got, want)


def test_int16_ingest_and_zero_fill(gpu):
    """radio.c:110-122 int16 scaling + gain_factor, and the lost-packet zero fill of radio.c:81-100."""
    import kq_oracle as ko
    from common import oracle_cfg
    geom = _small_geom(4)
    fs, L = geom["samprate"], geom["L"]
    p = wl._mode_params("am", 30)
    p.update(second_lo=-(wl.emitter_freq(30, fs) + 0.5))
    x = wl.make_iq(fs, 8 * L, seed=3, emitters=[30])
    xi = np.stack([np.round(x.real * 20000), np.round(x.imag * 20000)], axis=1).astype(np.int16)
    ch = ko.Channel(oracle_cfg(p, fs, L, geom["M"], geom["D"], gain_factor=0.5))
    bank = kq.Bank(fs, L, geom["M"], geom["D"], 1, 8, gain_factor=0.5, fwd_mode=kq.KQ_FWD_FULL)
    bank.add_channel(bank_cfg(p))
    want_a, want_s = [], []
    # 3 blocks, then a gap of L+100 samples, then the remaining data
    for b in range(3):
        a, s = ch.block_i16(xi[b * L:(b + 1) * L])
        want_a.append(a)
        want_s.append(s)
    for a, s in ch.zero_fill(L):          # exactly one block completes inside the zero fill
        want_a.append(a)
        want_s.append(s)
    for b in range(3, 6):
        a, s = ch.block_i16(xi[b * L:(b + 1) * L])
        want_a.append(a)
        want_s.append(s)
    bank.push_iq(xi[:3 * L])
    bank.push_zeros(L)
    bank.push_iq(xi[3 * L:6 * L])
    assert bank.blocks_ready() == 7
    assert bank.process() == 7
    for b in range(7):
        st = bank.status(0, b)
        assert st["nout"] == want_s[b]["nout"]
        assert st["hangcount"] == want_s[b]["hangcount"]
        np.testing.assert_allclose(st["if_power"], want_s[b]["if_power"], rtol=2e-4)
        assert rel_rms(bank.audio(0, b), want_a[b]) < AUDIO_TOL or np.abs(want_a[b]).max() == 0
    bank.close()


@pytest.mark.parametrize("seed", list(range(24)) + _EXTRA)
def test_lost_packets_at_geometries_drawn_at_random(gpu, seed):
    """The zero fill of radio.c:81-100 (lost packets: the filter input gets zeros, the oscillators step on, IF power is not
    updated for blocks that complete inside the fill) at drawn geometries and drawn gap lengths, from a fraction of a block to
    two and a half, between pieces of int16 data of drawn lengths -- with an impulse response longer than a block the zeros
    stay in the history for several blocks.  An AM and an FM channel: sample counts, IF power, audio."""
    import kq_oracle as ko
    from common import oracle_cfg
    rng = np.random.default_rng(9700 + seed)
    while True:
        nd = int(rng.choice(_smooth_sizes(64, 1024)))
        D = int(rng.choice([1, 2, 4, 5, 8, 16, 32]))
        N = nd * D
        if N <= (8192 if D == 1 else 16384) and N >= 512:
            break
    k = int(rng.integers(nd // 4, int(nd * 0.6) + 1))
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    plan = [dict(demod="am", low=-4500.0, high=4500.0, recovery_rate=50.0, second_lo=-(wl.emitter_freq(22, fs) + 0.5)),
            dict(demod="fm", low=-7000.0, high=7000.0, second_lo=-(wl.emitter_freq(24, fs) - 2.0))]
    total = 14 * L
    x = wl.make_iq(fs, total, seed=3 + seed, emitters=range(20, 28))
    xi = np.stack([np.round(x.real * 20000), np.round(x.imag * 20000)], axis=1).astype("<i2")
    chans = [ko.Channel(oracle_cfg(p, fs, L, M, D, gain_factor=0.5)) for p in plan]
    bank = kq.Bank(fs, L, M, D, len(plan), 24, gain_factor=0.5, fwd_mode=kq.KQ_FWD_AUTO)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    want = [[] for _ in plan]
    got = [[] for _ in plan]
    pos = 0
    steps = []

    def drain():
        nb = bank.blocks_ready()
        if nb:
            assert bank.process() == nb
            for c in range(len(plan)):
                got[c] += [(bank.audio(c, b), bank.status(c, b)) for b in range(nb)]

    while pos < total - 2 * L:
        n = int(rng.integers(L // 5, 2 * L))
        n = min(n, total - pos)
        bank.push_iq(xi[pos:pos + n])
        for c, ch in enumerate(chans):
            want[c] += ch.push_raw(xi[pos:pos + n].tobytes(), n, 1)
        pos += n
        drain()
        z = int(rng.integers(1, int(2.5 * L)))
        bank.push_zeros(z)
        for c, ch in enumerate(chans):
            want[c] += ch.zero_fill(z)
        drain()
        steps.append((n, z))
    where = "N = %d (L = %d, M = %d), decimate %d, (data, zeros) %s" % (N, L, M, D, steps)
    for c in range(len(plan)):
        assert len(got[c]) == len(want[c]), where
        for b in range(len(want[c])):
            wa, ws = want[c][b]
            ga, st = got[c][b]
            assert st["nout"] == ws["nout"], where
            np.testing.assert_allclose(st["if_power"], ws["if_power"], rtol=2e-4, atol=1e-12, err_msg=where)
            if b >= 2 and np.abs(wa).max() > 0 and np.all(np.isfinite(wa)):
                if plan[c]["demod"] == "am":
                    # (a block that is mostly fill has an output 1e-3 of the others': its own RMS is no yardstick for 1e-5 --
                    #  block by block a loose bound that a wrong sample count or a misplaced zero would break by orders,
                    #  the 1e-5 bar on the whole run below)
                    assert rel_rms(ga, wa) < 1e-3, "%s: channel %d block %d: %.2e" % (where, c, b, rel_rms(ga, wa))
                else:
                    assert st["squelch_count"] == ws["squelch_count"], where
        if plan[c]["demod"] == "am":
            fin = [b for b in range(2, len(want[c])) if np.all(np.isfinite(want[c][b][0]))]
            e = rel_rms(np.concatenate([got[c][b][0] for b in fin]), np.concatenate([want[c][b][0] for b in fin]))
            assert e < AUDIO_TOL, "%s: channel %d: %.2e" % (where, c, e)
    bank.close()


@pytest.mark.parametrize("name,nchan,nblocks", [("cfg3", 40, 5), ("cfg4", 33, 4), ("cfg2", 37, 3)])
def test_config_geometry_pruned(gpu, name, nchan, nblocks):
    """Pruned forward path (only the N/D bins the slave reads) against the oracle's full N-point FFT."""
    g = wl.GEOMETRY[name]
    plan = wl.channel_plan(name, nchan)
    plan[3]["isb"] = 1 if plan[3]["demod"] == "linear" else 0
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=9)
    want = run_oracle(plan, g, iq, nblocks)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_PRUNED, per_call=3)
    assert mode == kq.KQ_FWD_PRUNED
    _compare(plan, got, want)


def test_cfg5_geometry_swept_doppler(gpu):
    """BASELINE configs[4] geometry: N=65536 (L=32768, M=32769), D=512, SSB with a swept Doppler NCO
    (doppler.freq != 0, rate != 0: osc.c:43-47 sweep).  Runs on the split full path (N > one LDS block)."""
    g = wl.GEOMETRY["cfg5"]
    plan = wl.channel_plan("cfg5", 4, first=100)
    plan[1]["channels"] = 2          # one stereo I/Q channel
    nblocks = 3
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=13)
    want = run_oracle(plan, g, iq, nblocks)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, per_call=2)
    assert mode == kq.KQ_FWD_FULL
    _compare(plan, got, want)


@pytest.mark.parametrize("variant", ["swept", "unswept", "fast_sweep"])
def test_cfg5_geometry_full_spectrum_with_n0(gpu, variant):
    """configs[4] as the reference runs it -- compute_n0 on every block (linear.c:123-126) -- on the N = 65536
    full-spectrum kernel: the swept steady-state variant, the unswept one, and a sweep beyond what its table path takes
    (per-sample closed form, 20 kHz/s at 20 MS/s)."""
    g = wl.GEOMETRY["cfg5"]
    plan = wl.channel_plan("cfg5", 9, first=300)
    plan[1]["channels"] = 2
    plan[2]["isb"] = 1
    plan[2]["channels"] = 2
    if variant == "unswept":
        for p in plan:
            p["second_lo"] -= p["doppler"]
            p["doppler"] = p["doppler_rate"] = 0.0
    elif variant == "fast_sweep":
        plan[3]["doppler_rate"] = 20000.0
        plan[4]["doppler_rate"] = -9000.0
    else:
        plan[5]["second_lo"] -= plan[5]["doppler"]    # one unswept channel in the swept launch
        plan[5]["doppler"] = plan[5]["doppler_rate"] = 0.0
    nblocks = 5
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=23)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, compute_n0=True, per_call=2)
    assert mode == kq.KQ_FWD_FULL
    # (one of these nine channels has a bin 5e-5 below the cut in its first block -- checked in float64 on the kernel's
    # own spectrum, tests/diag/diag_n0_64k.py: the kernel keeps it, the oracle's sequential float sum, 1.3e-4 low, drops it)
    flips = _compare_counting_ties(plan, got, want, nblocks)
    import conftest
    conftest.note_ties("test_cfg5_geometry_full_spectrum_with_n0[%s]" % variant, flips, len(plan))
    assert len(flips) <= 1, flips
    _n0_ties_are_ties(plan, g, iq, nblocks, flips, 0.001)


def test_cfg5_geometry_pruned_stream(gpu):
    """configs[4] on the pruned path: N/D = 128, window streamed through LDS, swept Doppler NCO per channel."""
    g = wl.GEOMETRY["cfg5"]
    plan = wl.channel_plan("cfg5", 11, first=200)
    plan[2]["isb"] = 1
    plan[2]["channels"] = 2
    plan[5]["second_lo"] -= plan[5]["doppler"]    # one unswept channel in the swept launch, still on its emitter
    plan[5]["doppler"] = plan[5]["doppler_rate"] = 0.0
    nblocks = 3
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=17)
    want = run_oracle(plan, g, iq, nblocks)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_PRUNED, per_call=2)
    assert mode == kq.KQ_FWD_PRUNED
    _compare(plan, got, want)


def test_cfg5_geometry_pruned_stream_unswept(gpu):
    g = wl.GEOMETRY["cfg5"]
    plan = wl.channel_plan("cfg5", 9)
    for p in plan:
        p["second_lo"] -= p["doppler"]
        p["doppler"] = p["doppler_rate"] = 0.0
    nblocks = 2
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=19)
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_PRUNED)
    _compare(plan, got, want)


def test_pruned_swept_doppler_cfg4_geometry(gpu):
    """Doppler-tracked FM channels (rate != 0) at the cfg 3/4 geometry on the pruned path, several calls."""
    g = wl.GEOMETRY["cfg4"]
    plan = wl.channel_plan("cfg4", 6)
    for i, p in enumerate(plan):
        d = (-1) ** i * (3000.0 + 500.0 * i)
        p.update(doppler=d, doppler_rate=(-1) ** i * (100.0 + 40.0 * i))
        p["second_lo"] += d
    plan[4]["second_lo"] -= plan[4]["doppler"]
    plan[4]["doppler"] = plan[4]["doppler_rate"] = 0.0
    nblocks = 6
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=23)
    want = run_oracle(plan, g, iq, nblocks)
    got, mode = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_PRUNED, per_call=2)
    assert mode == kq.KQ_FWD_PRUNED
    _compare(plan, got, want)


def test_pruned_refuses_huge_sweep(gpu):
    g = wl.GEOMETRY["cfg4"]
    p = wl.channel_plan("cfg4", 1)[0]
    p.update(doppler=1000.0, doppler_rate=5e6)       # 5 MHz/s: beyond the first-order treatment
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 1, fwd_mode=kq.KQ_FWD_PRUNED)
    bank.add_channel(bank_cfg(p))
    bank.push_iq(np.zeros(g["L"], np.complex64))
    with pytest.raises(kq.KqError, match="sweep"):
        bank.process()
    bank.close()


def test_pl_tone_measurement_cfg1_geometry(gpu):
    """fm.c:189-285: CTCSS tone from the decimate-32 PL slave and the 16384-point ring transform; the peak-bin
    index must agree exactly with the oracle, block by block (configs[0] geometry, 100.0 Hz tone)."""
    g = wl.GEOMETRY["cfg1"]
    fs, L = g["samprate"], g["L"]
    nblocks = 40
    t = np.arange(nblocks * L) / fs
    ph = 2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t) + 6.0 * np.sin(2 * np.pi * 100.0 * t)
    rng = np.random.default_rng(31)
    iq = (0.1 * np.exp(1j * ph) + 1e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0, flat=1)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, per_call=7)
    _compare(plan, got, want)
    tones = [s["plfreq"] for s in got[0]["status"]]
    assert np.isnan(tones[0]) and abs(tones[-1] - 100.0) < 0.2


def test_pl_tone_measurement_cfg2_geometry(gpu):
    """The same measurement at cfg 2's geometry (N/D = 256: a PL slave of 8 points, 4 samples per block), where the
    de-emphasis filter and the PL slave of a PAIR of blocks come out of one packed 256-point transform in registers
    (k_fm_audio256: the slave's bins 0..4 of each window are separated from the packed spectrum by Hermitian symmetry).
    512 PL samples = 128 blocks until the first ring transform (fm.c:251): 280 blocks in calls of 37 (an odd count, so
    pairs straddle the calls) see two of them; the tone -- and with it the peak bin, fm.c:260-267 -- must equal the
    oracle's in every block, for a de-emphasised and a flat channel."""
    g = wl.GEOMETRY["cfg2"]
    fs, L = g["samprate"], g["L"]
    nblocks = 280
    t = np.arange(nblocks * L) / fs
    ph = 2 * np.pi * 100000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t) + 6.0 * np.sin(2 * np.pi * 100.0 * t)
    rng = np.random.default_rng(33)
    # (30 dB of in-channel SNR: with less noise the envelope's variance, fm.c:101, is float rounding of either sign and the
    # squelch counter with it)
    sigma = 0.1 * 10 ** (-30 / 20) / np.sqrt(2 * 16000.0 / fs)
    iq = (0.1 * np.exp(1j * ph) + sigma / np.sqrt(2) * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-100000.0),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-100000.0, flat=1)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, per_call=37)
    _compare(plan, got, want)
    for c in range(2):
        tones = [s["plfreq"] for s in got[c]["status"]]
        assert np.isnan(tones[0]) and np.isnan(tones[126]) and abs(tones[-1] - 100.0) < 0.5, (tones[0], tones[126], tones[-1])
        assert sum(1 for a, b in zip(tones[:-1], tones[1:]) if not (a == b or (np.isnan(a) and np.isnan(b)))) >= 1


@pytest.mark.parametrize("seed", list(range(8)))
def test_pl_tone_at_geometries_drawn_at_random(gpu, seed):
    """pltask (fm.c:189-285) where 32 divides N/decimate and the block length but nothing else is chosen: PL slaves of 8 ... 64
    points (60, 42 and 28 among them), 1 ... 30 PL samples per block, 0.8 s of a 100 Hz tone under the voice -- two ring transforms;
    the tone, and with it the peak bin, equal to the oracle's in every block; everything else as in the sweep."""
    rng = np.random.default_rng(9650 + seed)
    while True:
        nd = int(rng.choice([n for n in _smooth_sizes(256, 2048) if n % 64 == 0]))      # (an even PL slave: N/decimate / 32)
        D = int(rng.choice([2, 4, 5, 8, 16, 32]))
        if nd * D <= 16384:
            break
    k = 32 * int(rng.integers(max(1, nd // 128), nd // 64 + 1))
    N, M, L, fs = nd * D, k * D + 1, (nd - k) * D, 48000 * D
    g = dict(samprate=fs, L=L, M=M, D=D)
    nblocks = int(np.ceil(0.8 * fs / L))
    t = np.arange(nblocks * L) / fs
    fc = 0.1 * fs
    ph = 2 * np.pi * fc * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t) + 6.0 * np.sin(2 * np.pi * 100.0 * t)
    sigma = 0.1 * 10 ** (-30 / 20) / np.sqrt(2 * 16000.0 / fs)
    iq = (0.1 * np.exp(1j * ph) + sigma / np.sqrt(2) * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-fc),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-fc, flat=1)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, per_call=int(rng.integers(3, 30)))
    try:
        _compare(plan, got, want)
    except AssertionError as e:
        raise AssertionError("N = %d (L = %d, M = %d), decimate %d, %d blocks: %s" % (N, L, M, D, nblocks, e)) from e
    tones = [s_["plfreq"] for s_ in got[0]["status"]]
    assert np.isnan(tones[0]) and abs(tones[-1] - 100.0) < 0.5, (N, L, M, D, tones[0], tones[-1])


@pytest.mark.parametrize("nd,D,k", [(400, 4, 160), (900, 2, 400), (270, 8, 110), (1350, 4, 600), (2000, 2, 900)])
def test_pl_tone_where_32_does_not_divide_the_sizes(gpu, nd, D, k):
    """fm.c:201-205 with N/decimate or the block length no multiple of 32: create_filter_output warns and truncates
    (filter.c:103-107,116), the PL slave then resamples by N_dec / PL_N instead of 32 and the reference reads the tone that much
    off (100 Hz as 104.2 at N/decimate = 400).  The library does what the reference does: tone and peak bin equal to the
    oracle's -- whose create_filter_output truncates the same way -- in every block."""
    N, M, L, fs = nd * D, k * D + 1, (nd - k) * D, 48000 * D
    g = dict(samprate=fs, L=L, M=M, D=D)
    pn = nd // 32
    assert nd % 32 != 0
    nblocks = int(np.ceil(0.9 * fs / L))
    t = np.arange(nblocks * L) / fs
    fc = 0.1 * fs
    rng = np.random.default_rng(nd)
    ph = 2 * np.pi * fc * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t) + 6.0 * np.sin(2 * np.pi * 100.0 * t)
    sigma = 0.1 * 10 ** (-30 / 20) / np.sqrt(2 * 16000.0 / fs)
    iq = (0.1 * np.exp(1j * ph) + sigma / np.sqrt(2) * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-fc),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-fc, flat=1)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, per_call=11)
    _compare(plan, got, want)
    tone = got[0]["status"][-1]["plfreq"]
    if pn == 62:            # (2 x 31: no transform of that size here or in the oracle -- FFTW would plan it -- and no PL slave)
        assert np.isnan(tone)
    else:                   # (off by the resampling error and by the samples dropped at every block border: a few per cent)
        assert abs(tone - 100.0) < 8.0, tone


def test_pcm_output_stage(gpu):
    """scaleclip + network byte order + per-480-word silence flags (audio.c:22-28, 45-50, 95-100),
    bit exact against the oracle applied to the same device audio; includes clipping and an all-zero (squelched) block."""
    import kq_oracle as ko
    g = dict(samprate=192000, L=2048, M=2049, D=4)        # olen = 512: two chunks (480 + 32) per mono block
    fs, L = g["samprate"], g["L"]
    nb = 8
    t = np.arange(nb * L) / fs
    sig = 0.2 * np.exp(1j * (2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
    sig[4 * L:] = 0                                         # carrier drops: squelch closes, audio becomes exact zeros
    rng = np.random.default_rng(8)
    iq = (sig + 1e-4 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0, headroom=30.0),   # loud: clips
            dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, channels=2)]
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), nb, fwd_mode=kq.KQ_FWD_FULL)
    bank.enable_pcm(True)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    bank.push_iq(iq)
    assert bank.process() == nb
    clipped = silent = 0
    for c in range(len(plan)):
        for b in range(nb):
            a = bank.audio(c, b)
            got, gmask = bank.pcm(c, b)
            want, wmask, nch = ko.pcm_block(a)
            assert np.array_equal(got, want) and gmask == wmask and nch == (len(a) + 479) // 480
            clipped += int(np.sum(np.abs(a) >= 1.0))
            silent += bin(gmask).count("1")
    assert clipped > 0 and silent > 0
    bank.close()


def test_pcm_rtp_datagrams(gpu):
    """SURVEY 8f-2, the rest of audio.c:32-132: the datagrams themselves -- header, 480-word chunks, skipped silent
    chunks with the timestamp still advancing, marker bit on resume, sequence numbers on sent packets only -- byte for
    byte against the oracle packetiser run on the same device audio (mono FM that squelches, stereo linear)."""
    import kq_oracle as ko
    g = dict(samprate=192000, L=2048, M=2049, D=4)        # olen = 512: chunks of 480 + 32 (mono), 480 + 480 + 64 (stereo)
    fs, L = g["samprate"], g["L"]
    nb = 8
    t = np.arange(nb * L) / fs
    sig = 0.2 * np.exp(1j * (2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
    sig[3 * L:5 * L] = 0                                    # the carrier drops for two blocks, then comes back
    rng = np.random.default_rng(18)
    iq = (sig + 1e-4 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0),
            dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, channels=2)]
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), 4, fwd_mode=kq.KQ_FWD_FULL)
    bank.enable_pcm(True)
    ssrc = [0xCAFE0001, 0x7FFFFFF2]
    ora = []
    for c, p in enumerate(plan):
        bank.add_channel(bank_cfg(p))
        bank.set_output_ssrc(c, ssrc[c])
        ora.append(ko.OutRtp(ssrc=ssrc[c]))
    sent = skipped = markers = 0
    for call in range(2):                                   # state carries across process calls
        bank.push_iq(iq[call * 4 * L:(call + 1) * 4 * L])
        assert bank.process() == 4
        for c in range(len(plan)):
            for b in range(4):
                a = bank.audio(c, b)
                got = bank.rtp_audio(c, b)
                want = ora[c].packetize(a, stereo=(c == 1))
                assert got == want, (call, c, b)
                nchunks = (len(a) + 479) // 480
                sent += len(got)
                skipped += nchunks - len(got)
                markers += sum(1 for d in got if d[1] & 0x80)
                for d in got:
                    assert d[0] == 0x80 and (d[1] & 0x7F) == (10 if c == 1 else 11) and len(d) <= 12 + 960
    for c in range(len(plan)):
        st = bank.output_rtp_state(c)
        assert (st["seq"], st["timestamp"], st["silent"], st["packets"], st["bytes"]) == \
               (ora[c].seq, ora[c].timestamp, ora[c].silent, ora[c].packets, ora[c].bytes)
        assert st["timestamp"] == nb * 512                  # every frame counted, sent or not
    assert sent > 0 and skipped > 0 and markers > 0
    bank.close()


@pytest.mark.parametrize("seed", list(range(20)) + _EXTRA)
def test_pcm_and_rtp_output_at_geometries_drawn_at_random(gpu, seed):
    """The output stage (audio.c:22-132) where a block is not 512 samples: drawn N/decimate (16 ... 3000 samples per block,
    odd counts among them -- 343, 675 ...), so that a block is a fraction of a 480-word packet, or several and a rest, mono and
    stereo; PCM words and silence masks bit for bit against the oracle on the same device audio, the datagrams byte for byte
    (sequence numbers, timestamps advancing over skipped packets, the marker on resume), over calls of three blocks."""
    import kq_oracle as ko
    rng = np.random.default_rng(9800 + seed)
    while True:
        nd = int(rng.choice(_smooth_sizes(32, 4096)))
        D = int(rng.choice([1, 2, 4, 5, 8, 16]))
        N = nd * D
        if N <= (8192 if D == 1 else 16384) and N >= 256:
            break
    k = int(rng.integers(nd // 4, nd // 2 + 1))
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    per, ncalls = 3, 4
    nb = per * ncalls
    t = np.arange(nb * L) / fs
    f0 = 0.21 * fs
    sig = 0.2 * np.exp(1j * (2 * np.pi * f0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
    sig[4 * L:7 * L] = 0                                    # the carrier drops for three blocks, then comes back
    iq = (sig + 1e-4 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-f0, headroom=float(rng.choice([0.1778, 30.0]))),
            dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-f0, hangtime=1.1, recovery_rate=6.0, channels=2)]
    bank = kq.Bank(fs, L, M, D, len(plan), per, fwd_mode=kq.KQ_FWD_AUTO)
    bank.enable_pcm(True)
    ssrc = [0xCAFE0001, 0x7FFFFFF2]
    ora = []
    for c, p in enumerate(plan):
        bank.add_channel(bank_cfg(p))
        bank.set_output_ssrc(c, ssrc[c])
        ora.append(ko.OutRtp(ssrc=ssrc[c]))
    where = "N = %d (L = %d, M = %d), decimate %d: %d samples per block" % (N, L, M, D, L // D)
    for call in range(ncalls):
        bank.push_iq(iq[call * per * L:(call + 1) * per * L])
        assert bank.process() == per
        for c in range(len(plan)):
            for b in range(per):
                a = bank.audio(c, b)
                got, gmask = bank.pcm(c, b)
                want, wmask, nch = ko.pcm_block(a)
                assert np.array_equal(got, want) and gmask == wmask and nch == (len(a) + 479) // 480, (where, call, c, b)
                assert bank.rtp_audio(c, b) == ora[c].packetize(a, stereo=(c == 1)), (where, call, c, b)
    for c in range(len(plan)):
        st = bank.output_rtp_state(c)
        assert (st["seq"], st["timestamp"], st["silent"], st["packets"], st["bytes"]) == \
               (ora[c].seq, ora[c].timestamp, ora[c].silent, ora[c].packets, ora[c].bytes), where
        assert st["timestamp"] == nb * (L // D), where
    bank.close()


# the generic kernels (k_filter_full steps a float oscillator through a window that has ONE unswept oscillator and evaluates per
# sample in double otherwise: both in this test; k_filter_split beyond one LDS block) at sizes of their own
RETUNE_GEOMETRIES = {"n8192": dict(samprate=192000, L=3840, M=4353, D=4),        # the reference's default -L / -M (main.c:160-170)
                     "n9600": dict(samprate=240000, L=4800, M=4801, D=5),
                     "n32768": dict(samprate=4000000, L=16384, M=16385, D=128)}


# ... and where M - 1 > L (the reference's default -L 3840 -M 4353, main.c:160-170; N = 16384 and 65536 likewise): the SECOND
# block after a retune still has M - 1 - L samples of the old oscillator in its window -- and with one block per call that
# block belongs to the next call
RETUNE_GEOMETRIES.update({"n16384_long_m": dict(samprate=192000, L=7680, M=8705, D=4),
                          "n16384_long_m_d256": dict(samprate=10000000, L=6144, M=10241, D=256),  # two blocks of old history
                          "n65536_long_m": dict(samprate=20000000, L=24576, M=40961, D=512),
                          "n4096_long_m": dict(samprate=2000000, L=1024, M=3073, D=64)})         # pruned-capable, three blocks


@pytest.mark.parametrize("name,mode,per", [("cfg4", "pruned", 2), ("cfg4", "full", 2), ("cfg5", "pruned", 2), ("cfg5", "full", 2),
                                           ("n8192", "full", 2), ("n9600", "full", 2), ("n32768", "full", 2),
                                           ("n8192", "full", 1), ("n16384_long_m", "full", 2), ("n16384_long_m", "full", 1),
                                           ("n16384_long_m_d256", "full", 1), ("n16384_long_m_d256", "pruned", 1),
                                           ("n65536_long_m", "full", 1), ("n65536_long_m", "pruned", 2),
                                           ("n4096_long_m", "pruned", 1), ("n4096_long_m", "full", 3)])
def test_retune_mid_stream_is_sample_exact(gpu, name, mode, per):
    """osc.c:22-36 + radio.c:132-139: a retune changes only the samples mixed after it; the M-1 history samples of
    the blocks that follow keep the old oscillator (phase continuous).  Second LO and Doppler retuned between calls of
    `per` blocks."""
    import kq_oracle as ko
    from common import oracle_cfg
    if name in RETUNE_GEOMETRIES:
        g = RETUNE_GEOMETRIES[name]
        ds = g["samprate"] / g["D"]
        # (each channel on an emitter of the synthetic band: the 1e-5 bar is relative to the channel's own output)
        lo = [-wl.emitter_freq(e, g["samprate"]) for e in (24, 27, 26)]           # an FM, an SSB and an AM emitter
        plan = [dict(demod="fm", low=-0.16 * ds, high=0.16 * ds, second_lo=lo[0], doppler=0.0, doppler_rate=0.0),
                dict(demod="linear", low=0.002 * ds, high=0.06 * ds, second_lo=lo[1], hangtime=1.1, recovery_rate=6.0, doppler=0.0,
                     doppler_rate=0.0),
                dict(demod="am", low=-0.1 * ds, high=0.1 * ds, second_lo=lo[2], recovery_rate=50.0, doppler=0.0, doppler_rate=0.0)]
    else:
        g = wl.GEOMETRY[name]
        plan = wl.channel_plan(name, 3)
    fs, L = g["samprate"], g["L"]
    for p in plan:
        p["second_lo"] -= p["doppler"]
        p["doppler"] = p["doppler_rate"] = 0.0
    ncalls = 3 if per == 2 else 12 // per
    nblocks = ncalls * per
    iq = wl.make_iq(fs, nblocks * L, seed=29)
    chans = [ko.Channel(oracle_cfg(p, fs, L, g["M"], g["D"])) for p in plan]
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), per, fwd_mode=kq.KQ_FWD_PRUNED if mode == "pruned" else kq.KQ_FWD_FULL)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    first, second = (1, 2) if per == 2 else (ncalls // 3, 2 * ncalls // 3)
    for call in range(ncalls):
        if call == first:  # retune channel 0's second LO by a non-bin-aligned amount, start a Doppler sweep on channel 1
            new_lo = plan[0]["second_lo"] + 777.7
            bank.set_second_lo(0, new_lo)
            chans[0].set_lo2(new_lo)
            bank.set_doppler(1, 1500.0, 80.0)
            chans[1].set_doppler(1500.0, 80.0)
        if call == second:  # and back again
            bank.set_second_lo(0, plan[0]["second_lo"])
            chans[0].set_lo2(plan[0]["second_lo"])
            bank.set_doppler(1, 0.0, 0.0)
            chans[1].set_doppler(0.0, 0.0)
        bank.push_iq(iq[per * call * L:per * (call + 1) * L])
        assert bank.process() == per
        for c, ch in enumerate(chans):
            for b in range(per):
                _, _, filt, _ = ch.block(iq[(per * call + b) * L:(per * call + b + 1) * L], want_filt=True)
                got = bank.filter_output(c, b)
                assert rel_rms(got, filt) < FILT_TOL, (call, c, b, rel_rms(got, filt))
    bank.close()


@pytest.mark.parametrize("seed", list(range(48)) + _EXTRA)
def test_retunes_at_geometries_drawn_at_random(gpu, seed):
    """The retune rule at geometries nobody chose: N/decimate, decimate and the impulse response drawn as in
    test_geometries_drawn_at_random but up to 0.7 of N/decimate (M - 1 up to 2.3 L: old samples in up to three blocks; every
    third seed up to 0.82: 4.5 L), one to three blocks per call, a second-LO retune and a Doppler sweep switched on and off again
    at drawn calls, and on every third seed a tuning knob turned on a third channel: a step before each of up to five calls in a row.  Filter output
    of every block from the third on (the first two are the leading edge of a long impulse response) against the oracle."""
    import kq_oracle as ko
    from common import oracle_cfg
    rng = np.random.default_rng(9500 + seed)
    while True:
        nd = int(rng.choice(_smooth_sizes(64, 2048)))
        D = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16, 32, 64]))
        N = nd * D
        if N <= (8192 if D == 1 else 16384) and N >= 512:
            break
    k = int(rng.integers(nd // 4, int(nd * (0.7 if seed % 3 else 0.82)) + 1))      # (every third seed: M - 1 up to 4.5 L)
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    ds = 48000.0
    lo = [-wl.emitter_freq(e, fs) for e in (24, 27, 26)]
    plan = [dict(demod="fm", low=-0.16 * ds, high=0.16 * ds, second_lo=lo[0]),
            dict(demod="linear", low=0.002 * ds, high=0.06 * ds, second_lo=lo[1], hangtime=1.1, recovery_rate=6.0),
            dict(demod="am", low=-0.1 * ds, high=0.1 * ds, second_lo=lo[2], recovery_rate=50.0)]
    per = int(rng.integers(1, 4))
    ncalls = 15 // per
    iq = wl.make_iq(fs, ncalls * per * L, seed=31 + seed, emitters=range(20, 32))
    chans = [ko.Channel(oracle_cfg(p, fs, L, M, D)) for p in plan]
    bank = kq.Bank(fs, L, M, D, len(plan), per, fwd_mode=kq.KQ_FWD_AUTO)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    first = int(rng.integers(ncalls // 3, ncalls // 2 + 1))
    second = int(rng.integers(first + 1, ncalls))
    knob = set(range(first + 1, min(second, first + 6))) if seed % 3 == 0 else set()    # a tuning knob turned: a step before every call
    for call in range(ncalls):
        if call in knob:
            hz = lo[2] + 12.5 * (call - first)
            bank.set_second_lo(2, hz)
            chans[2].set_lo2(hz)
        if call == first:
            bank.set_second_lo(0, lo[0] + 777.7)
            chans[0].set_lo2(lo[0] + 777.7)
            bank.set_doppler(1, 1500.0, 80.0)
            chans[1].set_doppler(1500.0, 80.0)
        if call == second:
            bank.set_second_lo(0, lo[0])
            chans[0].set_lo2(lo[0])
            bank.set_doppler(1, 0.0, 0.0)
            chans[1].set_doppler(0.0, 0.0)
        bank.push_iq(iq[per * call * L:per * (call + 1) * L])
        assert bank.process() == per
        for c, ch in enumerate(chans):
            for b in range(per):
                _, _, filt, _ = ch.block(iq[(per * call + b) * L:(per * call + b + 1) * L], want_filt=True)
                if per * call + b >= 2:
                    e = rel_rms(bank.filter_output(c, b), filt)
                    assert e < FILT_TOL, ("N = %d (L = %d, M = %d), decimate %d, %d blocks per call, retunes at calls %d and %d: "
                                          "call %d channel %d block %d: %.2e" % (N, L, M, D, per, first, second, call, c, b, e))
    bank.close()


@pytest.mark.parametrize("seed", list(range(48)) + [1000 + k for k in range(16)] + [2000 + k for k in range(8)] + _EXTRA)
def test_an_operator_at_geometries_drawn_at_random(gpu, seed):
    """What the UI and the Doppler thread do to a running receiver (display.c:161-177, doppler.c, radio.c:290-374) -- second LO,
    Doppler with and without a rate and off again, shift, filter edges and Kaiser beta, mode, channels leaving and joining --
    drawn at random between the
    calls of a bank whose geometry is drawn as in test_geometries_drawn_at_random (impulse responses up to 0.6 of
    N/decimate), one to three blocks per call.  Every block's filter output from the third on, the sample counts and the FM
    channels' squelch counters against the oracle given the same operations."""
    import kq_oracle as ko
    from common import oracle_cfg
    rng = np.random.default_rng(9900 + seed)
    while True:
        if 2000 <= seed < 3000:  # ... the 65536-point kernels (four sibling workgroups per channel-block; pruned-stream at N/D = 128)
            D = int(rng.choice([32, 64, 128, 256, 512]))
            nd = 65536 // D
        elif 1000 <= seed < 2000:  # ... the 16384-point register kernel at splits of L and M nobody chose (and the pruned ones)
            D = int(rng.choice([4, 8, 16, 32, 64, 128, 256]))
            nd = 16384 // D
        else:
            nd = int(rng.choice(_smooth_sizes(64, 2048)))
            D = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16, 32, 64]))
        N = nd * D
        if N <= (8192 if D == 1 else 65536 if 1000 <= seed < 3000 else 16384) and N >= 512:
            break
    k = int(rng.integers(nd // 4, int(nd * 0.6) + 1))
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    modes = dict(fm=dict(demod="fm", low=-7000.0, high=7000.0),
                 am=dict(demod="am", low=-4500.0, high=4500.0, recovery_rate=50.0),
                 usb=dict(demod="linear", low=100.0, high=3000.0, hangtime=1.1, recovery_rate=6.0),
                 isb=dict(demod="linear", low=-3000.0, high=3000.0, hangtime=1.1, recovery_rate=6.0, isb=1, channels=2))
    names = list(modes)
    C = 5
    cur = []
    for c in range(C):
        p = dict(modes[names[int(rng.integers(0, 4))]])
        # (an FM or AM emitter each: whatever mode the channel is in or is switched to, its passband holds a signal -- the
        #  1e-5 bar is relative to the channel's own output, and float rounding of the whole band is 2e-5 of an empty one)
        p["second_lo"] = -(wl.emitter_freq((20, 21, 22, 24, 25)[c], fs) + float(rng.uniform(-50, 50)))
        cur.append(p)
    per = int(rng.integers(1, 4))
    ncalls = 14 // per + 1
    iq = wl.make_iq(fs, ncalls * per * L, seed=41 + seed, emitters=range(20, 32))
    chans = [ko.Channel(oracle_cfg(p, fs, L, M, D)) for p in cur]
    bank = kq.Bank(fs, L, M, D, C, per, compute_n0=bool(seed % 2), fwd_mode=kq.KQ_FWD_AUTO)
    for p in cur:
        bank.add_channel(bank_cfg(p))
    log = []
    for call in range(ncalls):
        for _ in range(int(rng.integers(0, 3)) if call >= 1 else 0):
            c = int(rng.integers(0, C))
            op = ("lo", "doppler", "doppler_off", "shift", "filter", "mode", "leave", "join")[int(rng.integers(0, 8))]
            if op == "join":        # a channel joins on the master's live history (its oracle is primed with the M - 1 samples before)
                holes = [h for h in range(C) if chans[h] is None]
                if not holes or per * call * L < M - 1:
                    continue
                c = holes[0]
                p = dict(modes[names[int(rng.integers(0, 4))]], second_lo=cur[c]["second_lo"] + float(rng.uniform(-20, 20)))
                assert bank.add_channel(bank_cfg(p)) == c
                chans[c] = ko.Channel(oracle_cfg(p, fs, L, M, D))
                chans[c].prime_history(iq[per * call * L - (M - 1):per * call * L])
                cur[c] = p
                log.append((call, c, op))
                continue
            if chans[c] is None:
                continue
            if op == "leave":
                if sum(ch is not None for ch in chans) <= 2:
                    continue
                bank.remove_channel(c)
                chans[c].close()
                chans[c] = None
                log.append((call, c, op))
                continue
            if op == "lo":
                hz = cur[c]["second_lo"] + float(rng.uniform(-400, 400))
                bank.set_second_lo(c, hz)
                chans[c].set_lo2(hz)
                cur[c]["second_lo"] = hz
            elif op == "doppler":
                hz, rate = float(rng.uniform(-2000, 2000)), float(rng.choice([0.0, rng.uniform(-120, 120)]))
                bank.set_doppler(c, hz, rate)
                chans[c].set_doppler(hz, rate)
            elif op == "doppler_off":
                bank.set_doppler(c, 0.0, 0.0)
                chans[c].set_doppler(0.0, 0.0)
            elif op == "shift":
                if cur[c]["demod"] != "linear":
                    continue
                hz = float(rng.uniform(-300, 300))
                bank.set_shift(c, hz)
                chans[c].set_shift(hz)
            elif op == "filter":
                lo_, hi_ = cur[c]["low"] * float(rng.uniform(0.6, 1.0)), cur[c]["high"] * float(rng.uniform(0.6, 1.0))
                beta = float(rng.choice([2.0, 3.0, 5.0]))
                bank.set_filter(c, lo_, hi_, beta)
                chans[c].set_filter(lo_, hi_, beta)
            else:
                p = dict(modes[names[int(rng.integers(0, 4))]], second_lo=cur[c]["second_lo"])
                bank.set_mode(c, bank_cfg(p))
                chans[c].set_mode(oracle_cfg(p, fs, L, M, D))
                cur[c] = p
            log.append((call, c, op))
        bank.push_iq(iq[per * call * L:per * (call + 1) * L])
        assert bank.process() == per
        for c, ch in enumerate(chans):
            if ch is None:
                assert not bank.channel_active(c)
                continue
            for b in range(per):
                _, st, filt, _ = ch.block(iq[(per * call + b) * L:(per * call + b + 1) * L], want_filt=True)
                where = "N = %d (L = %d, M = %d), decimate %d, %d per call; operations %s; call %d channel %d block %d" % (
                    N, L, M, D, per, log, call, c, b)
                gs = bank.status(c, b)
                assert gs["nout"] == st["nout"], where
                if per * call + b >= 2:
                    e = rel_rms(bank.filter_output(c, b), filt)
                    assert e < FILT_TOL, "%s: %.2e" % (where, e)
                    if cur[c]["demod"] == "fm":
                        assert gs["squelch_count"] == st["squelch_count"], where
    bank.close()


@pytest.mark.parametrize("name", ["n8192", "n16384_long_m"])
def test_every_channel_of_a_large_bank_retuned_at_once(gpu, name):
    """1 300 channels at a geometry with M - 1 > L, all retuned between two calls of one block (more than the patch records of
    a call hold: the bank stages every channel afresh), half of them again before the next call (two transitions in the
    window), calls of one block: a sample of the channels against the oracle."""
    import kq_oracle as ko
    from common import oracle_cfg
    g = RETUNE_GEOMETRIES[name]
    fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
    C = 1300
    plan = [dict(demod="fm", low=-6000.0, high=6000.0, second_lo=-(wl.emitter_freq(20 + c % 10, fs) + 0.37 * (c // 10)))
            for c in range(C)]
    sample = [0, 1, 7, 511, 512, 1023, 1024, 1299]
    ncalls = 8
    iq = wl.make_iq(fs, ncalls * L, seed=5, emitters=range(20, 30))
    chans = {c: ko.Channel(oracle_cfg(plan[c], fs, L, M, D)) for c in sample}
    bank = kq.Bank(fs, L, M, D, C, 1, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
    bank.add_channels([bank_cfg(p) for p in plan])
    for call in range(ncalls):
        if call in (3, 4):
            for c in range(C):
                if call == 3 or c % 2 == 0:
                    hz = plan[c]["second_lo"] + (55.5 if call == 3 else -31.25)
                    bank.set_second_lo(c, hz)
                    plan[c]["second_lo"] = hz
                    if c in chans:
                        chans[c].set_lo2(hz)
        bank.push_iq(iq[call * L:(call + 1) * L])
        assert bank.process() == 1
        for c, ch in chans.items():
            _, _, filt, _ = ch.block(iq[call * L:(call + 1) * L], want_filt=True)
            if call >= 2:
                e = rel_rms(bank.filter_output(c, 0), filt)
                assert e < FILT_TOL, (call, c, e)
    bank.close()


def _pll_case(fs, nsamp, seed):
    """two emitters for carrier-tracking channels: a full-carrier AM signal at +20 037 Hz and a suppressed-carrier DSB one at
    -30 061 Hz"""
    t = np.arange(nsamp) / fs
    rng = np.random.default_rng(seed)
    msg = np.cos(2 * np.pi * 1000.0 * t)
    sig = 0.1 * (1 + 0.5 * msg) * np.exp(2j * np.pi * 20037.0 * t)
    sig = sig + 0.1 * np.cos(2 * np.pi * 700.0 * t) * np.exp(-2j * np.pi * 30061.0 * t + 0.7j)
    return (sig + 1e-3 * (rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp))).astype(np.complex64)


def _pll_plan(n):
    """n carrier-tracking channels: CAM (pll), DSB (pll + square) and stereo CAM in turn, each with its own offset from the
    carrier inside the +-300 Hz search window (linear.c:51-56)"""
    plan = []
    for c in range(n):
        off = (10.0 + 90.0 * ((c * 37) % 101) / 100.0) * (1 if c % 2 else -1)   # (the loop takes 59-66 blocks to lock from 10-100 Hz off)
        if c % 3 == 1:
            plan.append(dict(demod="linear", low=-5000.0, high=5000.0, second_lo=30061.0 - 61.0 + 0.5 * off, hangtime=1.1,
                             recovery_rate=6.0, pll=1, square=1))
        else:
            plan.append(dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20037.0 + off, hangtime=0.0, recovery_rate=50.0,
                             pll=1, channels=2 if c % 3 == 2 else 1))
    return plan


def test_a_thousand_carrier_tracking_channels_in_one_bank(gpu):
    """linear.c:129-246 on 1 024 channels of ONE bank (rounds 1-5 stopped at 64: slot = rank among the PLL channels, fixed
    allocations; now a slot per channel out of chunks of 64 that are allocated as the count grows, kq_bank.cpp pll_acquire).
    Every 93rd channel against the oracle: lock state, lock counter, hang counter block for block, carrier phase and offset,
    audio; all channels: the right number of samples, the carrier found."""
    g = wl.GEOMETRY["cfg1"]
    fs, L = g["samprate"], g["L"]
    C, nblocks, per_call = 1024, 64, 8       # (the search needs 16 blocks of samples, the lock counter 40-60 more: linear.c:157-200)
    iq = _pll_case(fs, nblocks * L, 43)
    plan = _pll_plan(C)
    sampled = sorted(set(range(0, C, 93)) | {C - 1, 64, 65})     # (64 / 65: the first slots of the second chunk)
    want = run_oracle([plan[c] for c in sampled], g, iq, nblocks)
    bank = kq.Bank(fs, L, g["M"], g["D"], C, per_call, fwd_mode=kq.KQ_FWD_FULL)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    got = {c: dict(audio=[], status=[]) for c in sampled}
    last = None
    for first in range(0, nblocks, per_call):
        bank.push_iq(iq[first * L:(first + per_call) * L])
        assert bank.process() == per_call
        for c in sampled:
            for b in range(per_call):
                got[c]["audio"].append(bank.audio(c, b))
                got[c]["status"].append(bank.status(c, b))
        if first + per_call == nblocks:
            last = [bank.status(c, per_call - 1) for c in range(C)]
    bank.close()
    olen = L // g["D"]
    for i, c in enumerate(sampled):
        auds, sts, _ = want[i]
        for b in range(nblocks):
            sg, sw = got[c]["status"][b], sts[b]
            assert (sg["pll_lock"], sg["lock_count"], sg["nout"], sg["hangcount"]) == \
                   (sw["pll_lock"], sw["lock_count"], sw["nout"], sw["hangcount"]), (c, b)
            np.testing.assert_allclose(sg["foffset"], sw["foffset"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(sg["cphase"], sw["cphase"], atol=2e-4)
        a_g, a_w = np.concatenate(got[c]["audio"][24:]), np.concatenate(auds[24:])
        assert rel_rms(a_g, a_w) < 2e-5, (c, rel_rms(a_g, a_w))
    # all channels: the right number of samples, a lock counter inside its rails, a loop that has found its carrier (the search
    # puts the offset within a bin); how many are locked after 64 blocks depends on each loop's transient (57 .. 80 blocks)
    locked = 0
    for c in range(C):
        assert last[c]["nout"] == (2 * olen if plan[c].get("channels", 1) == 2 else olen), (c, last[c])
        assert abs(last[c]["lock_count"]) <= 48000 and np.isfinite(last[c]["cphase"]) and abs(last[c]["foffset"]) < 12.0, (c, last[c])
        locked += last[c]["pll_lock"]
    assert locked > C // 4, locked


def test_carrier_tracking_channels_come_and_go_while_the_bank_runs(gpu):
    """Channels enter and leave the carrier-tracking set between calls in flight -- removed, added back into the holes, set_mode
    out of a PLL mode and into one -- with nothing drained: slots are handed out from a free list and started afresh by fill
    records in front of the next call's demodulators (rounds 1-5 moved the loops' state by synchronous copies).  Against a twin
    that is drained around every change: every status word and every audio sample of every channel bit for bit; 150 channels
    = three chunks of slots."""
    g = dict(samprate=192000, L=2048, M=2049, D=4)
    fs, L = g["samprate"], g["L"]
    C, nb, ncalls = 150, 4, 9
    iq = _pll_case(fs, ncalls * nb * L, 44)
    plan = _pll_plan(C)
    plain = dict(demod="linear", low=100.0, high=3000.0, second_lo=-20037.0, hangtime=1.1, recovery_rate=6.0)
    out = []
    for drained in (False, True):
        bank = kq.Bank(fs, L, g["M"], g["D"], C, nb, fwd_mode=kq.KQ_FWD_FULL)
        for p in plan[:100]:                              # the rest arrive while the bank runs
            bank.add_channel(bank_cfg(p))
        rec = []

        def change(fn, *a):
            if drained:
                bank.sync()
            return fn(*a)

        for k in range(ncalls):
            if k == 2:
                for c in range(100, 130):
                    assert change(bank.add_channel, bank_cfg(plan[c])) == c
                for c in (3, 64, 65, 99):                 # out of the set: their slots go back
                    change(bank.remove_channel, c)
            if k == 4:
                for c in (64, 3):                         # back into the holes (lowest first), fresh loops in recycled slots
                    assert change(bank.add_channel, bank_cfg(plan[c])) in (3, 64)
                for c in (10, 70):                        # a PLL mode -> plain SSB -> (two calls later) a PLL mode again
                    change(bank.set_mode, c, bank_cfg(plain))
                change(bank.set_mode, 11, bank_cfg(plan[12]))      # PLL -> PLL: a fresh loop in the same slot
            if k == 6:
                for c in (10, 70):
                    change(bank.set_mode, c, bank_cfg(plan[c]))
                for c in range(130, C):                   # the first two land in the holes 65 and 99, the rest behind
                    change(bank.add_channel, bank_cfg(plan[c]))
            bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
            assert bank.process() == nb
            live = sorted(c for c in range(bank.num_channels) if bank.channel_active(c))
            rec.append({c: ([bank.status(c, b) for b in range(nb)], [bank.audio(c, b).copy() for b in range(nb)]) for c in live})
        bank.close()
        out.append(rec)
    a, b = out
    for k in range(ncalls):
        assert a[k].keys() == b[k].keys(), k
        for c in a[k]:
            for blk in range(nb):
                sa, sb = a[k][c][0][blk], b[k][c][0][blk]
                assert all(np.array_equal(sa[f], sb[f], equal_nan=True) for f in sa), (k, c, blk, sa, sb)
                assert np.array_equal(a[k][c][1][blk], b[k][c][1][blk]), (k, c, blk)
    assert len(a[-1]) >= 148


@pytest.mark.parametrize("seed", list(range(8)))
def test_carrier_tracking_at_geometries_drawn_at_random(gpu, seed):
    """The carrier loop (linear.c:129-246) where a block is not 2048 samples: N/decimate drawn from 512 ... 4096 (7-smooth
    sizes among them), decimate 1 ... 32, four and a half seconds of signal in as many blocks as that takes, calls of a drawn length.
    CAM and DSB (squaring loop) in turn: lock state, lock counter and hang counter block for block, offset and phase, audio
    once locked."""
    rng = np.random.default_rng(9600 + seed)
    while True:
        nd = int(rng.choice(_smooth_sizes(512, 4096)))
        D = int(rng.choice([1, 2, 4, 5, 8, 16, 32]))
        N = nd * D
        if N <= (8192 if D == 1 else 16384) and N >= 2048:
            break
    k = int(rng.integers(nd // 4, nd // 2 + 1))
    M, L, fs = k * D + 1, (nd - k) * D, 48000 * D
    g = dict(samprate=fs, L=L, M=M, D=D)
    nblocks = int(np.ceil(4.5 * fs / L))
    t = np.arange(nblocks * L) / fs
    msg = np.cos(2 * np.pi * 1000.0 * t)
    fc = 0.1 * fs
    if seed % 2 == 0:
        sig = 0.1 * (1 + 0.5 * msg) * np.exp(2j * np.pi * (fc + 37.0) * t)
        p = dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-fc, hangtime=0.0, recovery_rate=50.0, pll=1)
    else:
        sig = 0.1 * msg * np.exp(2j * np.pi * (fc - 61.0) * t + 0.7j)
        p = dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-fc, hangtime=1.1, recovery_rate=6.0, pll=1, square=1)
    iq = (sig + 1e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [p, dict(p, channels=2)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_AUTO, per_call=int(rng.integers(3, 40)))
    where = "N = %d (L = %d, M = %d), decimate %d, %d blocks" % (N, L, M, D, nblocks)
    for c in range(len(plan)):
        auds, sts, _ = want[c]
        for b in range(nblocks):
            sg, sw = got[c]["status"][b], sts[b]
            assert (sg["pll_lock"], sg["lock_count"], sg["nout"], sg["hangcount"]) == \
                   (sw["pll_lock"], sw["lock_count"], sw["nout"], sw["hangcount"]), (where, c, b)
            np.testing.assert_allclose(sg["foffset"], sw["foffset"], rtol=1e-3, atol=1e-3, err_msg=where)
            np.testing.assert_allclose(sg["cphase"], sw["cphase"], atol=2e-4, err_msg=where)
        locked = [b for b in range(nblocks) if sts[b]["pll_lock"]]
        assert locked, where                     # (the lock detector's count runs in blocks: short blocks take longer)
        first = min(max(locked[0] + 2, nblocks // 3), nblocks - 1)
        a_g = np.concatenate(got[c]["audio"][first:])
        a_w = np.concatenate(auds[first:])
        assert rel_rms(a_g, a_w) < 2e-5, (where, c, rel_rms(a_g, a_w))


@pytest.mark.parametrize("mode", ["cam", "dsb"])
def test_linear_carrier_pll(gpu, mode):
    """linear.c:129-246: carrier search (65536-point transform, +-300 Hz window), coarse + fine NCO, 2nd-order loop,
    lock detector with hysteresis.  CAM = full-carrier AM tracked coherently; DSB = suppressed carrier through the
    squaring loop.  Lock state and lock counter must match block for block; audio within the float budget."""
    g = wl.GEOMETRY["cfg1"]
    fs, L = g["samprate"], g["L"]
    nblocks = 64
    t = np.arange(nblocks * L) / fs
    rng = np.random.default_rng(41)
    msg = np.cos(2 * np.pi * 1000.0 * t)
    if mode == "cam":
        sig = 0.1 * (1 + 0.5 * msg) * np.exp(2j * np.pi * (20000.0 + 37.0) * t)
        p = dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=0.0, recovery_rate=50.0, pll=1)
    else:
        sig = 0.1 * msg * np.exp(2j * np.pi * (20000.0 - 61.0) * t + 0.7j)
        p = dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, pll=1, square=1)
    iq = (sig + 1e-3 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [p, dict(p, channels=2)]
    want = run_oracle(plan, g, iq, nblocks)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, per_call=9)
    for c in range(len(plan)):
        auds, sts, _ = want[c]
        for b in range(nblocks):
            sg, sw = got[c]["status"][b], sts[b]
            assert (sg["pll_lock"], sg["lock_count"], sg["nout"], sg["hangcount"]) == \
                   (sw["pll_lock"], sw["lock_count"], sw["nout"], sw["hangcount"]), (c, b)
            np.testing.assert_allclose(sg["foffset"], sw["foffset"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(sg["cphase"], sw["cphase"], atol=2e-4)
        assert got[c]["status"][-1]["pll_lock"] == 1
        a_g = np.concatenate(got[c]["audio"][20:])
        a_w = np.concatenate(auds[20:])
        assert rel_rms(a_g, a_w) < 2e-5, (c, rel_rms(a_g, a_w))
