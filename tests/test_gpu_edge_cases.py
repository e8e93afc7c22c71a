"""Empty, ragged and limit-size inputs through the C ABI (the reference has no tests of its own; these pin the
behaviour its callers rely on: proc_samples fills input.c[] sample by sample and runs the filter whenever a block
completes, radio.c:139-146; packet.c:204-211 does the same for the AFSK master)."""
import ctypes as C

import numpy as np
import pytest

import ka9q_sdr_amd as kq
import kq_oracle as ko
from common import bank_cfg, rel_rms, run_oracle
from ka9q_sdr_amd import workload as wl

pytestmark = pytest.mark.gpu


def _geom():
    return dict(samprate=192000, L=512, M=513, D=4)


def test_empty_and_ragged_pushes(gpu):
    g = _geom()
    plan = wl.channel_plan("cfg1", 1)
    iq = wl.make_iq(g["samprate"], 5 * g["L"], seed=3)
    want = run_oracle(plan, g, iq, 5)
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 1, 8)
    bank.add_channel(bank_cfg(plan[0]))
    assert bank.process() == 0                      # nothing pushed yet
    bank.push_iq(iq[:0])                            # empty push
    assert bank.blocks_ready() == 0 and bank.process() == 0
    got = []
    # one sample short of a block, then the missing sample, then ragged pieces crossing block boundaries
    cuts = [0, g["L"] - 1, g["L"], g["L"] + 7, 3 * g["L"] - 1, 3 * g["L"] + 1, 5 * g["L"]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        bank.push_iq(iq[a:b])
        n = bank.process()
        assert n == b // g["L"] - a // g["L"]
        got += [bank.audio(0, k) for k in range(n)]
    assert len(got) == 5
    assert rel_rms(np.concatenate(got), np.concatenate(want[0][0])) < 1e-5
    bank.close()


def test_single_block_calls_and_full_batches_agree(gpu):
    g = wl.GEOMETRY["cfg4"]
    plan = wl.channel_plan("cfg4", 5)
    nb = 6
    iq = wl.make_iq(g["samprate"], nb * g["L"], seed=8)
    outs = []
    for per_call in (1, nb):
        bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], len(plan), nb)
        for p in plan:
            bank.add_channel(bank_cfg(p))
        res = [[] for _ in plan]
        for s in range(0, nb, per_call):
            bank.push_iq(iq[s * g["L"]:(s + per_call) * g["L"]])
            assert bank.process() == per_call
            for c in range(len(plan)):
                res[c] += [(bank.audio(c, k), bank.status(c, k)) for k in range(per_call)]
        outs.append(res)
        bank.close()
    for c in range(len(plan)):
        # one block per call and six per call: the same state hand-over, so the same samples to rounding (the FM
        # kernel pairs blocks within a call and the oscillators are rebased per call, so not bit for bit) and the
        # same integer state
        a1 = np.concatenate([a for a, _ in outs[0][c]])
        a6 = np.concatenate([a for a, _ in outs[1][c]])
        assert rel_rms(a1, a6) < 1e-5
        for (_, s1), (_, s6) in zip(outs[0][c], outs[1][c]):
            assert s1["squelch_count"] == s6["squelch_count"] and s1["nout"] == s6["nout"]
            assert s1["blanked"] == s6["blanked"]


def test_channel_limit_and_bad_arguments(gpu):
    g = _geom()
    bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], 2, 2)
    cfg = bank_cfg(wl.channel_plan("cfg1", 1)[0])
    assert bank.add_channel(cfg) == 0 and bank.add_channel(cfg) == 1
    with pytest.raises(kq.KqError):
        bank.add_channel(cfg)                       # bank full
    with pytest.raises(kq.KqError):
        bank.set_filter(5, -1000.0, 1000.0, 3.0)    # no such channel
    with pytest.raises(kq.KqError):
        bank.set_filter(0, float("nan"), 1000.0, 3.0)   # filter.c:504-505
    x = np.zeros(3 * g["L"], np.complex64)
    with pytest.raises(kq.KqError, match="ring overflow"):
        bank.push_iq(x)                             # the ring holds max_blocks blocks
    bank.push_iq(x[:2 * g["L"] - 5])                # (the ring takes max_blocks * L samples plus L - 1 of slack)
    assert bank.blocks_ready() == 1
    bank.push_iq(x[:5])
    assert bank.blocks_ready() == 2 and bank.process() == 2
    bank.push_iq(x[:g["L"]])
    assert bank.process() == 1
    bank.close()


def test_decimator_and_afsk_degenerate_calls(gpu):
    dec = kq.Decimator(3, 8, 1, max_out=16)
    y, s16, e = dec.process(np.zeros(0, np.complex64))
    assert len(y) == 0 and s16.shape == (0, 2)
    with pytest.raises(ValueError):
        dec.process(np.zeros(9, np.complex64))      # not a multiple of the decimation ratio
    with pytest.raises(kq.KqError):
        dec.process(np.zeros(17 << 3, np.complex64))    # beyond max_out
    y, _, _ = dec.process(np.ones(8, np.complex64))     # a single output sample
    fe = ko.FrontEndDecimator(3, 8, 1)
    wy, _, _ = fe.process(np.ones(8, np.complex64))
    assert np.array_equal(y, wy)
    dec.close()
    bank = kq.AfskBank(2, max_frames=1)
    assert bank.push(np.zeros((2, 0), np.float32)) == 0
    assert bank.push(np.zeros((2, 999), np.float32)) == 0
    assert bank.push(np.zeros((2, 1), np.float32)) == 1
    assert bank.frames(0) == [] and bank.state(1)["blocks"] == 1
    with pytest.raises(kq.KqError):
        bank.L.kq_afsk_push(bank.h, None, 0, 1, 10, 10, 0) and None
        bank._chk(-1, "kq_afsk_push")
    bank.close()


def test_handles_release_their_device_memory(gpu):
    """create / use / destroy in a loop: free device memory must come back (no leak in any of the four handle types)"""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    g = wl.GEOMETRY["cfg3"]
    plan = wl.channel_plan("cfg3", 16)
    iq = wl.make_iq(g["samprate"], 2 * g["L"], seed=1)

    def cycle():
        for mode, n0 in ((kq.KQ_FWD_PRUNED, False), (kq.KQ_FWD_FULL, True)):
            bank = kq.Bank(g["samprate"], g["L"], g["M"], g["D"], len(plan), 2, compute_n0=n0, fwd_mode=mode)
            bank.enable_pcm(True)
            for p in plan:
                bank.add_channel(bank_cfg(p))
            bank.push_iq(iq)
            assert bank.process() == 2
            bank.sync()
            bank.close()
        dec = kq.Decimator(6, 8, 1, max_out=1024)
        dec.process(np.zeros(1024 << 6, np.complex64))
        dec.close()
        af = kq.AfskBank(8)
        af.push(np.zeros((8, 2000), np.float32))
        af.close()

    cycle()                                   # first use pays for one-off tables (twiddles, code objects)
    before = free_bytes()
    for _ in range(10):
        cycle()
    after = free_bytes()
    assert before - after < 8 << 20, (before, after)


@pytest.mark.parametrize("per_call,D,pl_tone", [(150, 4, True), (64, 4, True), (7, 4, True), (150, 16, True), (7, 16, True),
                                                (150, 4, False), (64, 4, False), (7, 4, False), (1, 4, False)])
def test_fm_hand_overs_across_many_blocks(gpu, per_call, D, pl_tone):
    """The generic FM demodulator takes 64 blocks of a channel at a time and resolves what each block inherits from
    the ones before it (squelch counter, last strong sample, last audio value, offset / deviation readings) by a
    scan.  A signal that comes and goes, fades below the blanking threshold, and a stretch of exact zeros (an open
    block without a single strong sample, then a squelched one) against the oracle's sequential loop; 150 blocks in
    one call span three chunks, 7 per call leave every chunk ragged.  D = 16 (N/D = 64) runs the same signal through
    the wave-per-channel demodulator, which pairs blocks instead.  pl_tone = False at D = 4 is cfg 2's geometry as SURVEY 8d
    measures it: the whole demodulator then runs in the one fused launch k_demod_fm256 (blocks held in LDS, the audio filter
    on pairs of them; 150 blocks = three chunks with the history handed from chunk to chunk inside the kernel)."""
    g = dict(samprate=192000, L=512, M=513, D=D)            # D = 4: N/D = 256, 128 samples per block
    fs, L = g["samprate"], g["L"]
    nb = 150
    n = nb * L
    t = np.arange(n) / fs
    rng = np.random.default_rng(91)
    blk = np.arange(n) // L
    on = ((blk < 30) | ((blk >= 60) & (blk < 90)) | (blk >= 100)).astype(np.float64)
    fade = 1.0 + 0.93 * np.cos(2 * np.pi * 211.0 * t)      # dips under 0.55 of the block average
    sig = 0.2 * on * fade * np.exp(1j * (2 * np.pi * 30000.0 * t + 2.5 * np.sin(2 * np.pi * 900.0 * t)))
    iq = sig + 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    iq[(blk >= 90) & (blk < 100)] = 0                        # digital silence
    iq = iq.astype(np.complex64)
    w = 8000.0 if D == 4 else 4500.0
    plan = [dict(demod="fm", low=-w, high=w, second_lo=-30000.0 - 13.0),
            dict(demod="fm", low=-0.75 * w, high=w, second_lo=-30000.0 + 7.0, flat=1)]
    want = run_oracle(plan, g, iq, nb, compute_n0=1)
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), per_call, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=pl_tone)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    got = [[] for _ in plan]
    for s in range(0, nb, per_call):
        k = min(per_call, nb - s)
        bank.push_iq(iq[s * L:(s + k) * L])
        assert bank.process() == k
        for c in range(len(plan)):
            got[c] += [(bank.audio(c, b), bank.status(c, b)) for b in range(k)]
    bank.close()
    seen_closed = seen_blanked = 0
    for c in range(len(plan)):
        auds, sts, _ = want[c]
        for b in range(nb):
            ga, gs = got[c][b]
            assert (gs["squelch_count"], gs["blanked"], gs["nout"]) == \
                   (sts[b]["squelch_count"], sts[b]["blanked"], sts[b]["nout"]), (c, b)
            np.testing.assert_allclose(gs["foffset"], sts[b]["foffset"], rtol=1e-4, atol=2e-2)
            np.testing.assert_allclose(gs["pdeviation"], sts[b]["pdeviation"], rtol=1e-4, atol=2e-2)
            if b < 90 or b >= 100:   # n0 of an all-zero spectrum is 0/0
                np.testing.assert_allclose(gs["n0"], sts[b]["n0"], rtol=2e-4)
            seen_closed += sts[b]["squelch_count"] >= 2
            seen_blanked += sts[b]["blanked"] > 0 and sts[b]["squelch_count"] < 2
        a_g, a_w = np.concatenate([a for a, _ in got[c]]), np.concatenate(auds)
        assert rel_rms(a_g, a_w) < 1e-5, (c, rel_rms(a_g, a_w))
    assert seen_closed > 20 and seen_blanked > 20           # the case exercises what it is meant to


@pytest.mark.parametrize("L,D,pruned", [(512, 16, 0), (512, 4, 0), (8192, 256, 0), (8192, 256, 1), (32768, 512, 0)])
def test_nan_sample_propagates_like_the_reference(gpu, L, D, pruned):
    """One NaN sample in the input: every comparison in the chain that the reference makes with a NaN operand (squelch
    `snr > 2`, blanking threshold, AGC `isnan(gain)` and `gain*level > headroom`, compute_n0's `< 2*mean`, the AM
    carrier filter that never recovers) has to fall the same way.  Outputs must be NaN in the same places, equal
    where finite, and the integer state identical.  N = 1024 with D = 16 runs the wave-per-channel demodulators, D = 4 the
    generic ones; N = 16384 the register-resident full-spectrum kernel and the pruned kernel; N = 65536 the four sibling
    workgroups per channel-block, whose compute_n0 then takes the spelled-out loops with two more exchanges between the
    siblings (sum and count of the first pass)."""
    g = dict(samprate=192000 if L == 512 else 10000000, L=L, M=L + 1, D=D)
    fs = g["samprate"]
    nb = 9
    iq = wl.make_iq(fs, nb * L, seed=17, emitters=range(24, 40) if L == 512 else None).copy()
    iq[3 * L + 100] = np.nan + 0j
    plan = []
    for e, kind in ((28, "fm"), (29, "fm"), (30, "am"), (31, "ssb"), (35, "ssb")):
        p = wl._mode_params(kind, e)
        p.update(second_lo=-(wl.emitter_freq(e, fs) + 3.7))
        plan.append(p)
    plan[1]["flat"] = 1
    plan[4].update(hangtime=0.0)
    want = run_oracle(plan, g, iq, nb, compute_n0=int(not pruned))
    bank = kq.Bank(fs, L, g["M"], D, len(plan), 4, compute_n0=not pruned,
                   fwd_mode=kq.KQ_FWD_PRUNED if pruned else kq.KQ_FWD_FULL)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    got = [[] for _ in plan]
    for s in range(0, nb, 4):
        k = min(4, nb - s)
        bank.push_iq(iq[s * L:(s + k) * L])
        assert bank.process() == k
        for c in range(len(plan)):
            got[c] += [(bank.audio(c, b), bank.status(c, b)) for b in range(k)]
    bank.close()
    nan_blocks = 0
    for c, p in enumerate(plan):
        auds, sts, _ = want[c]
        for b in range(nb):
            ga, gs = got[c][b]
            wa, ws = auds[b], sts[b]
            assert (gs["squelch_count"], gs["hangcount"], gs["blanked"], gs["nout"]) == \
                   (ws["squelch_count"], ws["hangcount"], ws["blanked"], ws["nout"]), (c, b, p["demod"])
            assert np.array_equal(np.isnan(ga), np.isnan(wa)), (c, b, p["demod"], np.isnan(ga).sum(), np.isnan(wa).sum())
            assert np.isnan(gs["n0"]) == np.isnan(ws["n0"]), (c, b)
            assert np.isnan(gs["agc_gain"]) == np.isnan(ws["agc_gain"]), (c, b)
            fin = ~np.isnan(wa)
            nan_blocks += int(not fin.all())
            if fin.any() and not (p["demod"] == "linear" and b in (0, 3, 4, 5)):
                # linear: block 0 is the documented start-up exclusion; the AGC restarts from a NaN gain in the blocks the
                # NaN passes through, dividing by whatever the first finite sample is
                assert rel_rms(ga[fin], wa[fin]) < 1e-5, (c, b, p["demod"], rel_rms(ga[fin], wa[fin]))
    assert nan_blocks >= 2 * len(plan)


@pytest.mark.parametrize("L,D", [(8192, 256), (512, 8)])
def test_agc_hold_coast_and_attack_regimes(gpu, L, D):
    """AM / linear AGC with short hang times on a signal that fades in and out: groups of samples in which the gain
    is held by the hang counter, coasts (counter runs out, recovery, no attack), or attacks -- the three paths of the
    wave-per-channel AGC -- in every order, against the oracle's sample-by-sample loop.  Hang counter exact per
    block, gain and audio to the parity tolerance.  (8192, 256): 32 samples per block, two blocks per group;
    (512, 8): 64 samples per block."""
    fs = 10000000 if L == 8192 else 192000
    g = dict(samprate=fs, L=L, M=L + 1, D=D)
    nb = 48
    n = nb * L
    t = np.arange(n) / fs
    rng = np.random.default_rng(77)
    f0 = 0.11 * fs
    drate = fs / D
    # two-tone SSB-like signal and an AM carrier next to it, both under a slow fade with a period of ~12 blocks
    fade = 0.55 + 0.45 * np.cos(2 * np.pi * t / (12.3 * L / fs))
    tone = lambda f: np.exp(2j * np.pi * (f0 + f) * t)      # noqa: E731
    sig = 0.1 * fade * (tone(0.02 * drate) + 0.7 * tone(0.05 * drate))
    sig += 0.1 * fade * (1 + 0.5 * np.cos(2 * np.pi * 0.03 * drate * t)) * np.exp(2j * np.pi * (f0 + 2.5 * drate) * t)
    iq = (sig + 1e-4 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    w = 0.1 * drate
    plan = []
    for hangtime, rec in ((0.002 * 39062.5 / drate, 20.0), (0.0007 * 39062.5 / drate, 60.0), (0.0, 30.0)):
        plan.append(dict(demod="linear", low=0.005 * drate, high=w, hangtime=hangtime, recovery_rate=rec, second_lo=-f0))
        plan.append(dict(demod="am", low=-w, high=w, hangtime=hangtime, recovery_rate=rec, second_lo=-(f0 + 2.5 * drate)))
    plan.append(dict(demod="linear", low=-w, high=w, hangtime=0.001 * 39062.5 / drate, recovery_rate=15.0, channels=2,
                     second_lo=-f0, shift=0.01 * drate))
    want = run_oracle(plan, g, iq, nb)
    bank = kq.Bank(fs, L, g["M"], D, len(plan), 7, fwd_mode=kq.KQ_FWD_AUTO)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    got = [[] for _ in plan]
    for s in range(0, nb, 7):
        k = min(7, nb - s)
        bank.push_iq(iq[s * L:(s + k) * L])
        assert bank.process() == k
        for c in range(len(plan)):
            got[c] += [(bank.audio(c, b), bank.status(c, b)) for b in range(k)]
    bank.close()
    olen = L // D
    for c, p in enumerate(plan):
        auds, sts, _ = want[c]
        hangs = [sts[b]["hangcount"] for b in range(nb)]
        # the linear channels really visit "counter running", "counter at zero" and "re-armed" (the AM carrier filter is
        # still charging over these 48 blocks, so its level rises with every sample and every sample attacks)
        if p["hangtime"] > 0 and p["demod"] == "linear":
            hm = int(p["hangtime"] / (D / fs))
            assert 0 in hangs[2:] and hm in hangs[2:] and any(0 < h < hm for h in hangs[2:]), (c, hm, hangs)
        for b in range(nb):
            ga, gs = got[c][b]
            assert gs["hangcount"] == sts[b]["hangcount"], (c, b, p["demod"], gs["hangcount"], sts[b]["hangcount"])
            assert gs["nout"] == sts[b]["nout"] == olen * p.get("channels", 1)
            if b > 0:
                np.testing.assert_allclose(gs["agc_gain"], sts[b]["agc_gain"], rtol=2e-5)
        sk = 1 if p["demod"] == "linear" else 0
        a_g = np.concatenate([a for a, _ in got[c][sk:]])
        a_w = np.concatenate(auds[sk:])
        assert rel_rms(a_g, a_w) < 1e-5, (c, p["demod"], rel_rms(a_g, a_w))


def test_n0_smoothing_continues_from_a_seeded_value(gpu):
    """sig.n0 belongs to struct demod, not to the demodulator thread: a thread that set_mode starts goes on smoothing from
    what the last one left (fm.c:78-82, am.c:46-49; only NaN takes the first value as it comes).  kq_bank_set_n0 is that
    hand-over; the thread entry points use it at start-up."""
    g = _geom()
    fs, L = g["samprate"], g["L"]
    plan = wl.channel_plan("cfg1", 2)
    plan[1] = dict(plan[1], demod="am", low=-5000.0, high=5000.0, recovery_rate=50.0)
    iq = wl.make_iq(fs, 3 * L, seed=3)
    fresh = kq.Bank(fs, L, g["M"], g["D"], 2, 3, compute_n0=True)
    seeded = kq.Bank(fs, L, g["M"], g["D"], 2, 3, compute_n0=True)
    for p in plan:
        fresh.add_channel(bank_cfg(p))
        seeded.add_channel(bank_cfg(p))
    seed = 3.0e-9
    seeded.set_n0(0, seed)
    seeded.set_n0(1, seed)
    for bank in (fresh, seeded):
        bank.push_iq(iq)
        assert bank.process() == 3
    for c, rate in ((0, 0.01), (1, 0.001)):
        raw0 = fresh.status(c, 0)["n0"]                       # a fresh channel takes the first value as it comes
        want = np.float32(seed + rate * (np.float64(raw0) - seed))
        got = seeded.status(c, 0)["n0"]
        assert abs(got / want - 1) < 1e-6, (c, got, want)
        # two blocks later the seeded channel is still near its seed (the raw values are 10 x larger here), the fresh one near raw
        assert abs(seeded.status(c, 2)["n0"] / seed - 1) < 30 * rate + 0.01 and abs(fresh.status(c, 2)["n0"] / raw0 - 1) < 0.5
    fresh.close()
    seeded.close()


def test_digital_silence_at_65536_points(gpu):
    """All-zero input at N = 65536 with compute_n0: the threshold is 0, so the four sibling workgroups take the
    spelled-out loops (and their extra exchanges); n0 is 0/0 = NaN in the reference and here, then recovers."""
    g = wl.GEOMETRY["cfg5"]
    fs, L = g["samprate"], g["L"]
    plan = wl.channel_plan("cfg5", 3)
    nb = 4
    iq = wl.make_iq(fs, nb * L, seed=9).copy()
    iq[:2 * L] = 0
    want = run_oracle(plan, g, iq, nb, compute_n0=1)
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), nb, compute_n0=True)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    bank.push_iq(iq)
    assert bank.process() == nb and bank.fwd_mode == kq.KQ_FWD_FULL
    for c in range(len(plan)):
        for b in range(nb):
            gs, ws = bank.status(c, b), want[c][1][b]
            assert np.isnan(gs["n0"]) == np.isnan(ws["n0"]), (c, b, gs["n0"], ws["n0"])
            assert gs["nout"] == ws["nout"]
            if b < 2:
                assert not np.any(bank.filter_output(c, b)) and np.isnan(ws["n0"])
    bank.close()


def test_soak64k_smoke(gpu):
    """tools/soak64k.py with 300 calls: N = 65536 banks with compute_n0 created and destroyed thirty times
    without leaking device memory, one stepped 300 times with kq_bank_sync never reporting a lost sibling workgroup and
    the same block giving the same output at the end as at the start, then the streaming host I/O over 200 steps queued
    without a host wait and checked bit for bit against the blocking pulls.  A child process: its own HIP context."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak64k.py"), "300"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "soak ok", r.stdout[-2000:]


def test_soak_realtime_smoke(gpu):
    """tools/soak_realtime.py for a few seconds: 4096 FM / AM / SSB channels fed two blocks per call with PCM planes streamed
    back, while a random script retunes, sweeps, shifts, refilters, restarts, removes and re-adds channels between the calls
    (never waiting for the device); the first 200 calls bit for bit against a bank that is drained around every change."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_realtime.py"), "--channels", "4096", "--seconds", "3",
                        "--check-calls", "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "soak ok", r.stdout[-2000:]


_IIR_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
bank_cfg = wl.bank_channel_config
cfg, fwd = sys.argv[3], int(sys.argv[4])
g = wl.GEOMETRY[cfg]
fs, L = g["samprate"], g["L"]
plan = wl.channel_plan(cfg, 6)
nb = 5
iq = wl.make_iq(fs, 3 * nb * L, seed=77)
bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), nb, compute_n0=(fwd != kq.KQ_FWD_PRUNED), fwd_mode=fwd, pl_tone=False)
bank.add_channels([bank_cfg(p) for p in plan])
out = []
for k in range(3):
    if k == 1:
        bank.push_zeros(2 * L + 100)        # a lost-packet gap: blocks completed inside it leave the IF power alone (radio.c:94-98)
        bank.push_iq(iq[k * nb * L:(k + 1) * nb * L - (2 * L + 100)])
    else:
        bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
    n = bank.process()
    out.append([[bank.status(c, b)["if_power"] for b in range(n)] for c in range(len(plan))])
bank.close()
np.save(sys.argv[2], np.array(out, dtype=np.float32))
"""


@pytest.mark.parametrize("config,fwd", [("cfg4", "full"), ("cfg4", "pruned"), ("cfg5", "full")])
def test_if_power_recurrence_in_the_filter_launch_is_the_stand_alone_one(gpu, config, fwd, tmp_path):
    """The IF-power recurrence (radio.c:143-145) rides in the call's filter launch -- one wave of its first workgroup, behind
    compute_n0 (k_filter_full16k, N = 16384 and 65536) or behind its channels (k_pruned_resident).  KQ_IIR_IN_FILTER=0 launches it
    on its own as before: the status words must be the same bits either way, through three calls with a zero-filled gap in the
    second (whose blocks must not update it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fwd_mode = {"full": kq.KQ_FWD_FULL, "pruned": kq.KQ_FWD_PRUNED}[fwd]
    res = []
    for val in ("0", "1"):
        path = str(tmp_path / ("ifp_%s.npy" % val))
        env = dict(os.environ, KQ_IIR_IN_FILTER=val)
        r = subprocess.run([sys.executable, "-c", _IIR_SCRIPT, root, path, config, str(fwd_mode)], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        res.append(np.load(path))
    a, b = res
    assert a.shape == b.shape and a.size > 0
    assert np.all(np.isfinite(a)) and np.all(a > 0)
    assert a.tobytes() == b.tobytes()
    assert np.all(a[:, 0, :] == a[:, 1, :])          # one front end: every channel reports the same IF power


@pytest.mark.parametrize("L,D,C", [(8192, 256, 1), (8192, 256, 2), (8192, 256, 9), (32768, 512, 3)])
def test_full_bank_of_distinct_edges_changes_every_filter(gpu, L, D, C):
    """compute_n0's passband masks live in max_channels slot sets, one per distinct pair of edges (kq_bank.cpp
    acquire_n0slot).  A FULL bank in which every channel has its own edges has no free set: a filter change must give the
    channel's old set back before it asks for a new one (asking first wrote one set past the planes -- 2 KiB,
    8 KiB at N = 65536, beyond n0lane and 4 bytes beyond n0meta).  Every channel's filter changes twice between calls, on
    a bank of exactly C channels and on a twin with room to spare: status (n0 among the words) and audio bit for bit,
    and n0 against the oracle's compute_n0 (radio.c:383-425) after each change."""
    from common import oracle_cfg
    fs = 10_000_000 if L == 8192 else 20_000_000
    g = dict(samprate=fs, L=L, M=L + 1, D=D)
    nb, ncalls = 2, 3
    base = wl.channel_plan("cfg3" if L == 8192 else "cfg5", max(C, 4))[:C]
    plan = [dict(p) for p in base]
    for c, p in enumerate(plan):                    # all edges distinct from the start
        p["low"], p["high"] = p["low"] - 10.0 * c, p["high"] + 7.0 * c
    iq = wl.make_iq(fs, ncalls * nb * L, seed=21)

    def edges(c, k):
        p = plan[c]
        return p["low"] * (0.9 - 0.05 * k) - c, p["high"] * (0.8 + 0.07 * k) + 3 * c, 3.0

    got = []
    for room in (C, C + 55):
        bank = kq.Bank(fs, L, g["M"], D, room, nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
        for p in plan:
            bank.add_channel(bank_cfg(p))
        rec = []
        for k in range(ncalls):
            if k:
                for c in range(C):
                    bank.set_filter(c, *edges(c, k))
                for c in range(0, C, 2):            # and once more for some: same edges again keep their set
                    bank.set_filter(c, *edges(c, k))
            bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
            assert bank.process() == nb
            rec.append(([[bank.status(c, b) for b in range(nb)] for c in range(C)],
                        [[bank.audio(c, b).copy() for b in range(nb)] for c in range(C)]))
        bank.close()
        got.append(rec)
    for k in range(ncalls):
        for c in range(C):
            for b in range(nb):
                sa, sb = got[0][k][0][c][b], got[1][k][0][c][b]
                assert all(np.array_equal(sa[f], sb[f], equal_nan=True) for f in sa), (k, c, b, sa, sb)
                assert np.array_equal(got[0][k][1][c][b], got[1][k][1][c][b]), (k, c, b)
    for c in range(C):
        ch = ko.Channel(oracle_cfg(plan[c], fs, L, g["M"], D, compute_n0=1))
        for k in range(ncalls):
            if k:
                ch.set_filter(*edges(c, k))
            for b in range(nb):
                _, st, _, _ = ch.block(iq[(k * nb + b) * L:(k * nb + b + 1) * L])
                have = got[0][k][0][c][b]["n0"]
                assert abs(have - st["n0"]) <= 2e-5 * abs(st["n0"]), (c, k, b, have, st["n0"])
        ch.close()
