"""The reference's operating point -- every channel at 1.0 x the front end's rate (main.c:105, README.md:470-477) -- as one
bank of tens of thousands of channels fed in small batches: what scales with the number of channels rather than with
channels x blocks.

* 32768 channels of cfg 4's geometry, two blocks per call, three calls through the streaming host I/O; every 64th channel
  against the oracle, sample counts and squelch state of ALL channels exact.
* kq_bank_add_channels against kq_bank_add_channel, channel by channel: bit-identical.
* the steady-state path (per-call oscillator planes advanced on the device) against per-call staging on the host, with
  sweeps, shifts and a retune on the way.
* kq_bank_pull_pcm_planes_async against the PCM stage of kq_bank_enable_pcm / the oracle's scaleclip.
"""
import ctypes

import numpy as np
import pytest
import torch   # before the HIP library is loaded: both then share one HIP runtime (as tests/test_gpu_fanout.py)

import ka9q_sdr_amd as kq
from common import bank_cfg, rel_rms, run_oracle
from ka9q_sdr_amd import workload as wl
from ka9q_sdr_amd.bank import STATUS_DTYPE
from test_gpu_parity import AUDIO_TOL, FILT_TOL

pytestmark = pytest.mark.gpu


def _pinned(n, dtype):
    return torch.zeros(n, dtype=dtype).pin_memory()


def test_32768_channels_two_blocks_per_call_against_the_oracle(gpu):
    g = wl.GEOMETRY["cfg4"]
    fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
    olen = L // D
    C, B, ncalls, every = 32768, 2, 3, 64
    plan = wl.channel_plan("cfg4", C)
    iq = wl.make_iq(fs, ncalls * B * L, seed=0x6B62)
    sampled = list(range(0, C, every))
    want = run_oracle([plan[c] for c in sampled], g, iq, ncalls * B, compute_n0=1)

    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, fwd_mode=kq.KQ_FWD_AUTO, pl_tone=False)
    idx = bank.add_channels([bank_cfg(p) for p in plan])
    assert idx == list(range(C)) and bank.num_channels == C
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()
    audio = [_pinned(C * B * 2 * olen, torch.float32) for _ in range(ncalls)]
    stat = [_pinned(C * B * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(ncalls)]
    filt = {c: [] for c in sampled}
    bank.push_iq_async(iq_pin.data_ptr(), B * L)
    for k in range(ncalls):
        assert bank.process() == B
        if k + 1 < ncalls:
            bank.push_iq_async(iq_pin.data_ptr() + 8 * (k + 1) * B * L, B * L)
        bank.pull_planes_async(audio[k].data_ptr(), stat[k].data_ptr())
        bank.pull_wait(0)
        for c in sampled[::8]:          # the filter output of a few of them (a synchronous pull each)
            for b in range(B):
                filt[c].append(bank.filter_output(c, b))
    bank.host_io_wait()
    ht = bank.host_timing()
    assert ht["calls"] == ncalls
    bank.close()

    st = [np.frombuffer(s.numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B) for s in stat]
    au = [a.numpy().reshape(C, B, 2 * olen) for a in audio]
    # counts / indices on ALL channels: every block delivers olen samples; the squelch state is the one the oracle finds
    # on the sampled channels (all of them listen to an emitter 45 dB above the noise: one value per block)
    for k in range(ncalls):
        assert np.all(st[k]["nout"] == olen)
        for b in range(B):
            sq = {want[i][1][k * B + b]["squelch_count"] for i in range(len(sampled))}
            assert len(sq) == 1, sq
            assert np.all(st[k]["squelch_count"][:, b] == sq.pop()), (k, b)
            assert np.all(np.isfinite(st[k]["n0"][:, b])) and np.all(st[k]["n0"][:, b] > 0)
            assert np.all(st[k]["if_power"][:, b] == st[k]["if_power"][0, b])     # one front end: one IF power
    # every 64th channel against the oracle
    worst_a = worst_f = worst_n0 = 0.0
    ties = []
    for i, c in enumerate(sampled):
        auds, sts, filts = want[i]
        got = np.concatenate([au[k][c, b, :olen] for k in range(ncalls) for b in range(B)])
        ea = rel_rms(got, np.concatenate(auds))
        worst_a = max(worst_a, ea)
        assert ea < AUDIO_TOL, ("audio", c, ea)
        for k in range(ncalls):
            for b in range(B):
                s, w = st[k][c, b], sts[k * B + b]
                assert s["blanked"] == w["blanked"] and s["squelch_count"] == w["squelch_count"], (c, k, b)
                np.testing.assert_allclose(s["bb_power"], w["bb_power"], rtol=1e-5)
                np.testing.assert_allclose(s["if_power"], w["if_power"], rtol=1e-4)
                dn0 = abs(s["n0"] / w["n0"] - 1)
                if dn0 > 2e-4:           # a bin at compute_n0's 2 x mean cut (radio.c:414-420): counted, bounded
                    assert dn0 < 5e-3, ("n0", c, k, b, dn0)
                    ties.append((c, "n0", dn0))
                else:
                    worst_n0 = max(worst_n0, dn0)
        if filt[c]:
            ef = rel_rms(np.concatenate(filt[c]), np.concatenate(filts))
            worst_f = max(worst_f, ef)
            assert ef < FILT_TOL, ("filter", c, ef)
    assert len({t[0] for t in ties}) <= max(2, 0.01 * len(sampled)), ties
    import conftest
    conftest.note_ties("test_32768_channels_two_blocks_per_call_against_the_oracle", ties, len(sampled))
    print("worst audio %.2g filter %.2g n0 %.2g over %d channels" % (worst_a, worst_f, worst_n0, len(sampled)))


def _mixed_plan(fs, n):
    rng = np.random.default_rng(77)
    plan = []
    for c in range(n):
        e = int(rng.integers(24, 40))
        p = dict(wl._mode_params(wl.emitter_kind(e), e))
        p.update(second_lo=-(wl.emitter_freq(e, fs) + float(rng.uniform(-40, 40))), kaiser_beta=float(rng.choice([2.0, 3.0])),
                 headroom=wl.HEADROOM, channels=1, flat=0, isb=0, shift=0.0, doppler=0.0, doppler_rate=0.0)
        if p["demod"] == "fm" and c % 3 == 0:
            p["flat"] = 1
        if p["demod"] == "linear":
            if c % 2:
                p.update(isb=1, channels=2, low=-3000.0, high=3000.0)
            if c % 5 == 0:
                p["shift"] = 300.0
        plan.append(p)
    return plan


def _pull_all(bank, C, nb):
    return ([[bank.audio(c, b) for b in range(nb)] for c in range(C)], [[bank.status(c, b) for b in range(nb)] for c in range(C)])


def test_batched_add_is_the_same_as_adding_one_by_one(gpu):
    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L = g["samprate"], g["L"]
    plan = _mixed_plan(fs, 40)
    nb = 4
    iq = wl.make_iq(fs, 2 * nb * L, seed=5, emitters=range(24, 40))
    res = []
    for batched in (False, True):
        bank = kq.Bank(fs, L, g["M"], g["D"], len(plan) + 3, nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
        if batched:
            assert bank.add_channels([bank_cfg(p) for p in plan[:5]]) == list(range(5))
            assert bank.add_channels([bank_cfg(p) for p in plan[5:]]) == list(range(5, len(plan)))   # appended behind the first lot
        else:
            for p in plan:
                bank.add_channel(bank_cfg(p))
        out = []
        for k in range(2):
            bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
            assert bank.process() == nb
            out.append(_pull_all(bank, len(plan), nb))
        resp = [bank.response(c) for c in range(len(plan))]
        res.append((out, resp))
        if batched:     # all or nothing: a bad entry adds none, a batch beyond the capacity adds none
            bad = [bank_cfg(plan[0]), bank_cfg(plan[1]), bank_cfg(plan[2]), bank_cfg(plan[3])]
            bad[2].low = float("nan")
            with pytest.raises(kq.KqError):
                bank.add_channels(bad)
            with pytest.raises(kq.KqError):
                bank.add_channels([bank_cfg(plan[0])] * 4)
            assert bank.num_channels == len(plan)
            bank.remove_channel(7)          # with a hole the batch goes one by one and reuses it
            assert bank.add_channels([bank_cfg(plan[7]), bank_cfg(plan[8])]) == [7, len(plan)]
        bank.close()
    (a, ra), (b, rb) = res
    for c in range(len(plan)):
        assert np.array_equal(ra[c], rb[c]), c
        for k in range(2):
            for blk in range(nb):
                assert np.array_equal(a[k][0][c][blk], b[k][0][c][blk]), (c, k, blk)
                sa, sb = a[k][1][c][blk], b[k][1][c][blk]
                for key in sa:
                    assert sa[key] == sb[key] or (np.isnan(sa[key]) and np.isnan(sb[key])), (c, k, blk, key)


@pytest.mark.parametrize("geom", ["n1024", "cfg4", "cfg5"])
def test_steady_state_planes_advanced_on_the_device(gpu, geom):
    """A call that follows a call with no channel added or removed takes its oscillator planes from the device (advanced in
    closed form by k_block_energy_sum; the channels retuned in between are patched from a few records, k_patch_planes)
    instead of from the host.  Same audio as a bank whose every call is staged by the host (KQ_STEADY=0, read per call),
    through sweeps, shifts, retunes of the second LO, the Doppler and the shift oscillator, and a channel leaving; integer
    state identical."""
    import os
    if geom == "n1024":
        g = dict(samprate=192000, L=512, M=513, D=4)
        plan = _mixed_plan(g["samprate"], 12)
        emit = range(24, 40)
    else:
        g = wl.GEOMETRY[geom]
        plan = wl.channel_plan(geom, 24)
        emit = None
    fs, L = g["samprate"], g["L"]
    for c, p in enumerate(plan):                  # sweeps on a third of them (per-sample variant of the kernels)
        if geom != "cfg5" and c % 3 == 1:
            p.update(doppler=1500.0 + 10 * c, doppler_rate=-300.0 * (c + 1))
            p["second_lo"] += p["doppler"]
    nb, ncalls = 2, 9
    iq = wl.make_iq(fs, ncalls * nb * L, seed=9, emitters=emit)
    outs = []
    for forced in (False, True):
        bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
        bank.add_channels([bank_cfg(p) for p in plan])
        got = []
        if forced:
            os.environ["KQ_STEADY"] = "0"
        for k in range(ncalls):
            if k == 3:
                bank.set_shift(4, 250.0)                             # patches: one channel's planes, the others advance on the device
            if k == 4:
                bank.set_second_lo(2, plan[2]["second_lo"] + 750.0)  # a retune on the way: its first block keeps the old LO in the history
                bank.set_second_lo(7, plan[7]["second_lo"] - 320.0)
            if k == 5 and geom != "cfg5":
                bank.set_doppler(3, 900.0, 0.0)                      # a Doppler oscillator that was off comes on (radio.c:135)
            if k == 6:
                bank.remove_channel(5)
            if k == 7:
                bank.set_doppler(1, plan[1].get("doppler", 0.0) + 40.0, plan[1].get("doppler_rate", 0.0))
            bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
            assert bank.process() == nb
            live = [c for c in range(len(plan)) if bank.channel_active(c)]
            got.append({c: ([bank.audio(c, b) for b in range(nb)], [bank.status(c, b) for b in range(nb)]) for c in live})
        ht = bank.host_timing()
        outs.append((got, ht))
        bank.close()
        os.environ.pop("KQ_STEADY", None)
    (a, ha), (b, hb) = outs
    assert ha["calls"] == hb["calls"] == ncalls
    for k in range(ncalls):
        assert a[k].keys() == b[k].keys()
        for c in a[k]:
            for blk in range(nb):
                x, y = a[k][c][0][blk], b[k][c][0][blk]
                assert len(x) == len(y)
                if np.abs(y).max() > 0:
                    # a linear channel's first block divides by numerically-zero start-up samples (linear.c:271-272)
                    tol = 2e-3 if (k == 0 and blk == 0 and plan[c]["demod"] == "linear") else 2e-6
                    assert rel_rms(x, y) < tol, (k, c, blk, rel_rms(x, y))
                sa, sb = a[k][c][1][blk], b[k][c][1][blk]
                for key in ("nout", "squelch_count", "hangcount", "blanked"):
                    assert sa[key] == sb[key], (k, c, blk, key)


def test_pcm_planes_straight_to_the_host(gpu):
    """kq_bank_pull_pcm_planes_async: the words and silent-chunk masks of the PCM stage (audio.c:22-28, 45-50, 95-100)
    formed inside the copy kernel -- equal to kq_bank_pull_pcm's (the k_pcm stage) and to the oracle's scaleclip on the
    same device audio; a loud channel that clips, a stereo channel, a channel that squelches."""
    import kq_oracle as ko
    g = dict(samprate=192000, L=2048, M=2049, D=4)        # olen = 512: chunks of 480 + 32 (mono), 480 + 480 + 64 (stereo)
    fs, L = g["samprate"], g["L"]
    olen = L // g["D"]
    nb = 8
    t = np.arange(nb * L) / fs
    sig = 0.2 * np.exp(1j * (2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
    sig[4 * L:] = 0                                         # the carrier drops: squelch closes, audio becomes exact zeros
    rng = np.random.default_rng(8)
    iq = (sig + 1e-4 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0, headroom=30.0),   # loud: clips
            dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, channels=2),
            dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0)]
    C = len(plan)
    bank = kq.Bank(fs, L, g["M"], g["D"], C, nb, fwd_mode=kq.KQ_FWD_FULL)
    bank.enable_pcm(True)
    for p in plan:
        bank.add_channel(bank_cfg(p))
    bank.push_iq(iq)
    assert bank.process() == nb
    pcm = _pinned(C * nb * 2 * olen, torch.int16)
    mask = _pinned(C * nb, torch.int32)
    stat = _pinned(C * nb * ctypes.sizeof(kq.ChanStatus), torch.uint8)
    bank.pull_pcm_planes_async(pcm.data_ptr(), mask.data_ptr(), stat.data_ptr())
    bank.pull_wait(0)
    words = pcm.numpy().reshape(C, nb, 2 * olen)
    masks = mask.numpy().view(np.uint32).reshape(C, nb)
    st = np.frombuffer(stat.numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, nb)
    clipped = silent = 0
    for c in range(C):
        for b in range(nb):
            a = bank.audio(c, b)
            stage, smask = bank.pcm(c, b)
            want, wmask, _ = ko.pcm_block(a)
            n = len(a)
            assert st[c, b]["nout"] == n
            assert np.array_equal(words[c, b, :n], want) and np.array_equal(stage, want), (c, b)
            assert int(masks[c, b]) == wmask == smask, (c, b, masks[c, b], wmask)
            clipped += int(np.sum(np.abs(a) >= 1.0))
            silent += bin(wmask).count("1")
    assert clipped > 0 and silent > 0
    # the same delivery with the 24-byte records a receiver reads per block (kq_bank_pull_pcm_planes_compact_async): words and
    # masks unchanged, every field equal to the full record's -- FM: foffset / squelch counter, linear: agc.gain / hang counter;
    # C * nb = 24 records, so the plane ends in an odd record behind the last whole 16-byte piece
    pcm2 = _pinned(C * nb * 2 * olen, torch.int16)
    mask2 = _pinned(C * nb, torch.int32)
    comp = _pinned(C * nb * kq.bank.COMPACT_STATUS_DTYPE.itemsize + 64, torch.uint8)
    comp.fill_(0xA5)
    bank.pull_pcm_planes_async(pcm2.data_ptr(), mask2.data_ptr(), comp.data_ptr(), compact=True)
    bank.pull_wait(0)
    nbytes = C * nb * kq.bank.COMPACT_STATUS_DTYPE.itemsize
    assert kq.bank.COMPACT_STATUS_DTYPE.itemsize == 24
    assert bytes(comp.numpy()[nbytes:]) == b"\xa5" * 64              # nothing written past the plane
    cs = np.frombuffer(comp.numpy()[:nbytes].tobytes(), dtype=kq.bank.COMPACT_STATUS_DTYPE).reshape(C, nb)
    assert np.array_equal(pcm2.numpy(), pcm.numpy()) and np.array_equal(mask2.numpy(), mask.numpy())
    for c, p in enumerate(plan):
        fm = p["demod"] == "fm"
        for b in range(nb):
            full = st[c, b]
            for f in ("bb_power", "n0", "snr", "nout"):
                assert np.array_equal(cs[c, b][f], full[f], equal_nan=True), (c, b, f)
            assert np.array_equal(cs[c, b]["aux"], full["foffset" if fm else "agc_gain"], equal_nan=True), (c, b)
            assert cs[c, b]["state"] == full["squelch_count" if fm else "hangcount"], (c, b)
    with pytest.raises(kq.KqError):
        bank.pull_wait(8)               # only the last eight deliveries are remembered
    bank.close()


@pytest.mark.parametrize("config,everybody", [("cfg4", False), ("cfg4", True), ("cfg2", False)])
def test_swept_and_fixed_channels_side_by_side_against_the_oracle(gpu, config, everybody):
    """A bank of fixed-frequency channels with a few satellite passes among them (Doppler offset with a rate, radio.c:180-184):
    at N = 16384 the unswept channels run the steady-state variant of the filter kernel and the swept ones a steady-state
    variant of their own (row phasors with the sweep in them, the lane part of the cross term to first order; rates up to
    kq full16k_sweep_limit = 4.8e-11 cycles per sample^2: 4.8 kHz/s at cfg 4's 10 MS/s, 190 Hz/s at cfg 2's 2 MS/s), as two
    launches over two channel lists; a channel changes list when its Doppler rate is set or cleared while the bank runs, and one
    rate beyond the limit sends the swept list back to the per-sample variant.  everybody: all channels sweep (one launch).
    Every channel against the oracle through five calls, with such changes in between."""
    g = wl.GEOMETRY[config]
    fs, L = g["samprate"], g["L"]
    k_rate = (fs / 1e7) ** 2                 # the same rates in cycles per sample^2 at either sample rate
    plan = wl.channel_plan(config, 14)
    for c in (range(14) if everybody else (2, 5, 11)):
        plan[c].update(doppler=1800.0 + 100 * c, doppler_rate=k_rate * (-250.0 * (c + 1) if c % 2 else 330.0 * (c + 1)))
        plan[c]["second_lo"] += plan[c]["doppler"]
    nb, ncalls = 2, 5
    iq = wl.make_iq(fs, ncalls * nb * L, seed=0x6B63)
    import kq_oracle as ko
    from common import oracle_cfg
    chans = [ko.Channel(oracle_cfg(p, fs, L, g["M"], g["D"], compute_n0=1)) for p in plan]
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
    bank.add_channels([bank_cfg(p) for p in plan])
    for k in range(ncalls):
        changes = ()
        if k == 2:      # channel 7 starts to sweep (or sweeps the other way), channel 5 stops (its offset stays): both change lists
            changes = ((7, (900.0, -120.0 * k_rate)), (5, (plan[5]["doppler"], 0.0)))
        if k == 3:      # 4.6 kHz/s at 10 MS/s: just inside the limit (theta up to 2.9e-4 rad)
            changes = ((11, (2900.0, 4600.0 * k_rate)),)
        if k == 4:      # 9 kHz/s: beyond it -- the swept channels of this call take the per-sample variant
            changes = ((2, (2000.0, -9000.0 * k_rate)),)
        for c, (d, r) in changes:
            bank.set_doppler(c, d, r)
            chans[c].set_doppler(d, r)
        bank.push_iq(iq[k * nb * L:(k + 1) * nb * L])
        assert bank.process() == nb
        for b in range(nb):
            blk = iq[(k * nb + b) * L:(k * nb + b + 1) * L]
            for c, ch in enumerate(chans):
                aud, st, filt, _ = ch.block(blk, want_filt=True)
                assert rel_rms(bank.filter_output(c, b), filt) < FILT_TOL, (k, b, c, rel_rms(bank.filter_output(c, b), filt))
                assert rel_rms(bank.audio(c, b), aud) < AUDIO_TOL, (k, b, c)
                got = bank.status(c, b)
                assert got["squelch_count"] == st["squelch_count"] and got["blanked"] == st["blanked"], (k, b, c)
    bank.close()


@pytest.mark.parametrize("geom", ["n1024", "cfg4", "n9600"])
def test_control_plane_between_calls_in_flight(gpu, geom):
    """The control plane no longer waits for the device: kq_bank_set_filter / set_mode / add_channel / remove_channel /
    set_n0 / set_linear_options write their parameters on the bank's stream, behind the calls in flight (which keep the
    values they were queued with) and their demodulators (wherever those run), in front of the next call.  A host that
    never waits -- twelve calls queued back to back with such changes between them, every call's planes streamed to their
    own pinned buffers, the demodulators therefore on their second stream -- must get, bit for bit, what a host gets that
    drains the device before every change and after every call."""
    if geom == "n1024":
        g = dict(samprate=192000, L=512, M=513, D=4)
        plan = _mixed_plan(g["samprate"], 14)
        emit = range(24, 40)
    elif geom == "n9600":                             # a size with factors 3 and 5: design jobs, mask sets and the
        g = dict(samprate=240000, L=4800, M=4801, D=5)    # group-wise AGC demodulator (960 samples per block) behind the queue
        plan = _mixed_plan(g["samprate"], 14)
        emit = range(24, 40)
    else:
        g = wl.GEOMETRY["cfg4"]
        plan = wl.channel_plan("cfg3", 14)           # FM, AM and SSB channels at cfg 4's geometry
        emit = None
    fs, L = g["samprate"], g["L"]
    olen = L // g["D"]
    nb, ncalls = 2, 12
    iq = wl.make_iq(fs, ncalls * nb * L, seed=21, emitters=emit)
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()
    am = dict(demod="am", low=-5000.0, high=5000.0, recovery_rate=50.0, hangtime=0.0, second_lo=plan[3]["second_lo"])
    cmax = len(plan) + 2
    results = []
    for drained in (False, True):
        bank = kq.Bank(fs, L, g["M"], g["D"], cmax, nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
        bank.add_channels([bank_cfg(p) for p in plan])
        audio = [_pinned(cmax * nb * 2 * olen, torch.float32) for _ in range(ncalls)]
        stat = [_pinned(cmax * nb * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(ncalls)]
        live = []

        def change(k):
            if drained:
                bank.sync()
            if k == 2:
                bank.set_filter(1, plan[1]["low"] * 0.8, plan[1]["high"] * 0.9, 4.0)
            if k == 3:
                bank.set_mode(3, bank_cfg(am))
                bank.set_n0(3, 1e-9)
            if k == 4:
                bank.remove_channel(6)
                bank.set_second_lo(2, plan[2]["second_lo"] + 410.0)
            if k == 5:
                assert bank.add_channel(bank_cfg(plan[9])) == 6          # the hole, while calls that still carry the old channel 6 run
                assert bank.add_channel(bank_cfg(plan[4])) == len(plan)
            if k == 7:
                lin = [c for c, p in enumerate(plan) if p["demod"] == "linear" and c not in (3, 6)]
                if lin:
                    bank.set_linear_options(lin[0], 1, 2)
                bank.set_filter(0, plan[0]["low"] * 0.5, plan[0]["high"] * 0.5, 2.0)
                bank.set_filter(len(plan), plan[4]["low"], plan[4]["high"] * 0.7, 3.0)
            if k == 9:
                bank.remove_channel(len(plan))
                bank.set_mode(3, bank_cfg(plan[3]))

        bank.push_iq_async(iq_pin.data_ptr(), nb * L)
        for k in range(ncalls):
            change(k)
            assert bank.process() == nb
            if k + 1 < ncalls:
                bank.push_iq_async(iq_pin.data_ptr() + 8 * (k + 1) * nb * L, nb * L)
            bank.pull_planes_async(audio[k].data_ptr(), stat[k].data_ptr())
            live.append([c for c in range(cmax) if bank.channel_active(c)])
            if drained:
                bank.sync()
        bank.host_io_wait()
        bank.sync()
        bank.close()
        st = [np.frombuffer(s.numpy().tobytes(), dtype=STATUS_DTYPE).reshape(cmax, nb) for s in stat]
        au = [a.numpy().reshape(cmax, nb, 2 * olen).copy() for a in audio]
        results.append((st, au, live))
    (sa, aa, la), (sb, ab, lb) = results
    assert la == lb
    for k in range(ncalls):
        for c in la[k]:
            for b in range(nb):
                n = int(sb[k][c, b]["nout"])
                assert int(sa[k][c, b]["nout"]) == n and n in (olen, 2 * olen), (k, c, b)
                assert np.array_equal(aa[k][c, b, :n], ab[k][c, b, :n]), (k, c, b)
                for key in STATUS_DTYPE.names:
                    x, y = sa[k][c, b][key], sb[k][c, b][key]
                    assert x == y or (np.isnan(x) and np.isnan(y)), (k, c, b, key, x, y)


def test_rtp_datagrams_from_streamed_pcm_planes(gpu):
    """kq_bank_rtp_from_planes: the datagrams of send_mono_output / send_stereo_output (audio.c:32-132) built on the host from
    the PCM and status planes kq_bank_pull_pcm_planes_async delivered -- byte for byte the oracle packetiser's output on the
    same audio, through a carrier that drops (silent packets skipped, timestamp advancing, marker on resume), for a mono FM
    and a stereo linear channel, over two streamed calls."""
    import kq_oracle as ko
    g = dict(samprate=192000, L=2048, M=2049, D=4)        # olen = 512: chunks of 480 + 32 (mono), 480 + 480 + 64 (stereo)
    fs, L = g["samprate"], g["L"]
    olen = L // g["D"]
    nb, ncalls = 4, 2
    t = np.arange(ncalls * nb * L) / fs
    sig = 0.2 * np.exp(1j * (2 * np.pi * 20000.0 * t + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
    sig[3 * L:5 * L] = 0                                    # the carrier drops for two blocks, then comes back
    rng = np.random.default_rng(18)
    iq = (sig + 1e-4 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))).astype(np.complex64)
    plan = [dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-20000.0),
            dict(demod="linear", low=-5000.0, high=5000.0, second_lo=-20000.0, hangtime=1.1, recovery_rate=6.0, channels=2)]
    C = len(plan)
    bank = kq.Bank(fs, L, g["M"], g["D"], C, nb, fwd_mode=kq.KQ_FWD_FULL)
    ssrc = [0xCAFE0001, 0x7FFFFFF2]
    ora = []
    for c, p in enumerate(plan):
        bank.add_channel(bank_cfg(p))
        bank.set_output_ssrc(c, ssrc[c])
        ora.append(ko.OutRtp(ssrc=ssrc[c]))
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()
    pcm = [_pinned(C * nb * 2 * olen, torch.int16) for _ in range(ncalls)]
    aud = [_pinned(C * nb * 2 * olen, torch.float32) for _ in range(ncalls)]
    stat = [_pinned(C * nb * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(ncalls)]
    bank.push_iq_async(iq_pin.data_ptr(), nb * L)
    sent = skipped = markers = 0
    for k in range(ncalls):
        assert bank.process() == nb
        if k + 1 < ncalls:
            bank.push_iq_async(iq_pin.data_ptr() + 8 * (k + 1) * nb * L, nb * L)
        bank.pull_pcm_planes_async(pcm[k].data_ptr(), None, stat[k].data_ptr())
        bank.pull_planes_async(aud[k].data_ptr(), None)     # the float audio of the same call, for the oracle
        bank.pull_wait(0)
        a = aud[k].numpy().reshape(C, nb, 2 * olen)
        st = np.frombuffer(stat[k].numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, nb)
        for c in range(C):
            for b in range(nb):
                n = int(st[c, b]["nout"])
                got = bank.rtp_from_planes(c, b, pcm[k].data_ptr(), stat[k].data_ptr())
                want = ora[c].packetize(a[c, b, :n], stereo=(c == 1))
                assert got == want, (k, c, b)
                sent += len(got)
                skipped += (n + 479) // 480 - len(got)
                markers += sum(1 for d in got if d[1] & 0x80)
    assert sent > 0 and skipped > 0 and markers >= 1
    bank.close()


def test_8192_mixed_channels_two_blocks_per_call_against_the_oracle(gpu):
    """cfg 3's mix (FM, AM, USB / LSB) at 8192 channels and two blocks per call: above 2048 channels the N/D = 64 demodulators
    run their one-wave forms for every mode (AM and linear AGC recurrences included), and a batch of two blocks is one pair.
    Every 64th channel against the oracle through four calls; hang counters, squelch state and sample counts exact."""
    g = wl.GEOMETRY["cfg3"]
    fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
    olen = L // D
    C, B, ncalls, every = 8192, 2, 4, 64
    plan = wl.channel_plan("cfg3", C)
    # channel c listens to emitter c mod 64: sample with a stride that is odd in emitters so that every mode is met
    sampled = list(range(0, C, every + 1))
    iq = wl.make_iq(fs, ncalls * B * L, seed=0x6B64)
    want = run_oracle([plan[c] for c in sampled], g, iq, ncalls * B, compute_n0=1)
    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, fwd_mode=kq.KQ_FWD_AUTO, pl_tone=False)
    bank.add_channels([bank_cfg(p) for p in plan])
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()
    audio = [_pinned(C * B * 2 * olen, torch.float32) for _ in range(ncalls)]
    stat = [_pinned(C * B * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(ncalls)]
    bank.push_iq_async(iq_pin.data_ptr(), B * L)
    for k in range(ncalls):
        assert bank.process() == B
        if k + 1 < ncalls:
            bank.push_iq_async(iq_pin.data_ptr() + 8 * (k + 1) * B * L, B * L)
        bank.pull_planes_async(audio[k].data_ptr(), stat[k].data_ptr())
    bank.host_io_wait()
    bank.close()
    st = [np.frombuffer(s.numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B) for s in stat]
    au = [a.numpy().reshape(C, B, 2 * olen) for a in audio]
    kinds = {}
    ties = []
    for i, c in enumerate(sampled):
        auds, sts, _ = want[i]
        p = plan[c]
        kinds[p["demod"]] = kinds.get(p["demod"], 0) + 1
        # (a linear channel's first block divides by numerically-zero start-up samples, linear.c:271-272: compared from block 1)
        first = 1 if p["demod"] == "linear" else 0
        got = np.concatenate([au[k][c, b, :olen] for k in range(ncalls) for b in range(B)][first:])
        ref = np.concatenate(auds[first:])
        err = rel_rms(got, ref)
        for k in range(ncalls):
            for b in range(B):
                s, w = st[k][c, b], sts[k * B + b]
                assert s["nout"] == w["nout"] and s["squelch_count"] == w["squelch_count"] and s["blanked"] == w["blanked"], (c, k, b)
                if s["hangcount"] != w["hangcount"] or err >= AUDIO_TOL:
                    # `gain * amplitude > headroom` with the gain riding at the limit (linear.c:271, am.c:66): a tie of the
                    # reference's own comparison; the two gain tracks stay within 1e-5
                    assert p["demod"] != "fm" and abs(s["agc_gain"] / w["agc_gain"] - 1) < 1e-4, (c, k, b, err, s["agc_gain"], w["agc_gain"])
                    ties.append((c, "agc", float(abs(s["agc_gain"] / w["agc_gain"] - 1))))
                np.testing.assert_allclose(s["bb_power"], w["bb_power"], rtol=2e-5)
        assert err < (AUDIO_TOL if not any(t[0] == c for t in ties) else 1e-4), (c, p["demod"], err)
    assert set(kinds) == {"fm", "am", "linear"}, kinds
    assert len({t[0] for t in ties}) <= 2, ties
    import conftest
    conftest.note_ties("test_8192_mixed_channels_two_blocks_per_call_against_the_oracle", ties, len(sampled))


def test_many_filter_changes_between_two_calls(gpu):
    """Responses are designed on the bank's stream in front of the call that uses them (kq_bank.cpp DesignQueue): 1300
    kq_bank_set_filter calls between two calls in flight -- more than one design launch holds, so the queue is applied early
    once -- then a second set_filter for some of those channels (the later design of a channel replaces the earlier one),
    nothing drained.  Against a bank that is drained around every single change: responses (fetched from the device when
    asked for), audio and every status word (noise_gain among them) bit for bit; and three of the responses against the oracle's
    set_filter."""
    import kq_oracle as ko
    from common import oracle_cfg
    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L = g["samprate"], g["L"]
    olen = L // g["D"]
    C, nb, ncalls = 1300, 2, 4
    base = _mixed_plan(fs, 13)
    plan = [dict(base[c % 13]) for c in range(C)]
    iq = wl.make_iq(fs, ncalls * nb * L, seed=8, emitters=range(24, 40))
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()

    def edges(c, again):
        p = plan[c]
        f = 0.55 + 0.4 * ((c * 37) % 101) / 101.0
        if again:
            f *= 0.8
        return p["low"] * f, p["high"] * (1.5 - f), 2.0 + (c % 4)

    out = []
    for drained in (False, True):
        bank = kq.Bank(fs, L, g["M"], g["D"], C, nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
        bank.add_channels([bank_cfg(p) for p in plan])
        audio = [_pinned(C * nb * 2 * olen, torch.float32) for _ in range(ncalls)]
        stat = [_pinned(C * nb * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(ncalls)]
        bank.push_iq_async(iq_pin.data_ptr(), nb * L)
        for k in range(ncalls):
            if k == 2:
                for c in range(C):
                    if drained:
                        bank.sync()
                    bank.set_filter(c, *edges(c, False))
                for c in range(0, C, 7):
                    if drained:
                        bank.sync()
                    bank.set_filter(c, *edges(c, True))
            assert bank.process() == nb
            if k + 1 < ncalls:
                bank.push_iq_async(iq_pin.data_ptr() + 8 * (k + 1) * nb * L, nb * L)
            bank.pull_planes_async(audio[k].data_ptr(), stat[k].data_ptr())
            if drained:
                bank.sync()
        resp = np.stack([bank.response(c) for c in range(C)])      # (waits for what is in flight)
        bank.host_io_wait()
        bank.sync()
        bank.close()
        st = [np.frombuffer(s.numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, nb) for s in stat]
        au = [a.numpy().reshape(C, nb, 2 * olen).copy() for a in audio]
        out.append((st, au, resp))
    (sa, aa, ra), (sb, ab, rb) = out
    assert np.array_equal(ra, rb)
    for k in range(ncalls):
        assert np.array_equal(aa[k], ab[k]), k
        assert sa[k].tobytes() == sb[k].tobytes(), k
    assert not np.array_equal(sa[1]["noise_gain"], sa[2]["noise_gain"])     # the new responses' noise gains reached the status
    for c in (0, 7, 1299):
        ch = ko.Channel(oracle_cfg(plan[c], fs, L, g["M"], g["D"], compute_n0=1))
        ch.set_filter(*edges(c, c % 7 == 0))
        np.testing.assert_allclose(ra[c], ch.response(), rtol=0, atol=2e-9)


def test_operator_thread_beside_the_receiver_thread(gpu):
    """One bank, two threads, as the reference has them (display.c / radio_status.c beside the demodulator threads): the receiver's
    loop -- process, push, queue the delivery, wait for the delivery two back -- and an operator who meanwhile changes filters and
    modes, retunes, drops channels and brings them back, and reads the bank's bookkeeping.  Every entry point takes the handle's
    lock; the ones that wait for the device let go of it while they wait, so the operator gets in between any two steps of the
    loop.  No call may fail, every delivery of an untouched channel is whole, the operator gets its work done while the loop
    runs, and the filters left at the end are the ones last asked for (against the oracle's set_filter)."""
    import threading
    import kq_oracle as ko
    from common import oracle_cfg
    g = wl.GEOMETRY["cfg4"]
    fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
    olen = L // D
    C, B, ncalls = 4096, 2, 400
    plan = wl.channel_plan("cfg3", C)                 # FM, AM and SSB
    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, pl_tone=False)
    bank.add_channels([bank_cfg(p) for p in plan])
    iq_pin = torch.from_numpy(wl.make_iq(fs, B * L, seed=3).copy()).pin_memory()
    pcm = [_pinned(C * B * 2 * olen, torch.int16) for _ in range(3)]
    mask = [_pinned(C * B, torch.int32) for _ in range(3)]
    stat = [_pinned(C * B * ctypes.sizeof(kq.ChanStatus), torch.uint8) for _ in range(3)]
    quiet = np.arange(C) < 1024                       # the operator leaves the first 1024 channels alone
    errors, done, ops = [], threading.Event(), [0]
    last_filter = {}

    def operator():
        rng = np.random.default_rng(4)
        away = None
        try:
            while not done.is_set():
                c = int(rng.integers(1024, C))
                what = int(rng.integers(0, 6))
                if c == away:
                    continue
                if what == 0:
                    f = float(rng.uniform(0.6, 1.0))
                    last_filter[c] = (plan[c]["low"] * f, plan[c]["high"] * f, 3.0)
                    bank.set_filter(c, *last_filter[c])
                elif what == 1:
                    bank.set_mode(c, bank_cfg(plan[c]))
                    last_filter.pop(c, None)          # (set_mode designs the mode's own filter again)
                elif what == 2:
                    bank.set_second_lo(c, plan[c]["second_lo"] + float(rng.integers(-2, 3)))
                elif what == 3 and away is None:
                    bank.remove_channel(c)
                    away = c
                    last_filter.pop(c, None)
                elif what == 4 and away is not None:
                    assert bank.add_channel(bank_cfg(plan[away])) == away
                    away = None
                else:
                    assert bank.num_channels == C and bank.channel_active(int(rng.integers(0, 1024)))
                ops[0] += 1
            if away is not None:
                assert bank.add_channel(bank_cfg(plan[away])) == away
        except Exception as e:                         # noqa: BLE001 -- handed to the main thread
            errors.append(e)

    th = threading.Thread(target=operator)
    bank.push_iq_async(iq_pin.data_ptr(), B * L)
    th.start()
    try:
        for k in range(ncalls):
            assert bank.process() == B
            bank.push_iq_async(iq_pin.data_ptr(), B * L)
            j = k % 3
            bank.pull_pcm_planes_async(pcm[j].data_ptr(), mask[j].data_ptr(), stat[j].data_ptr())
            bank.pull_wait(2)
            if k >= 2:
                st = np.frombuffer(stat[(k - 2) % 3].numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B)
                assert np.all(st["nout"][quiet] >= olen), k
                assert np.all(np.isfinite(st["bb_power"][quiet])), k
    finally:
        done.set()
        th.join()
    assert not errors, errors
    assert ops[0] > ncalls // 4, ops        # the operator was not locked out by the loop
    bank.host_io_wait()
    bank.sync()
    for c, args in list(last_filter.items())[:24]:
        ch = ko.Channel(oracle_cfg(plan[c], fs, L, M, D, compute_n0=1))
        ch.set_filter(*args)
        np.testing.assert_allclose(bank.response(c), ch.response(), rtol=0, atol=2e-9)
    bank.close()


def test_streaming_through_the_library_s_own_pinned_buffers(gpu):
    """kq_host_alloc / kq_host_free: the pinned buffers of the streaming entry points from the library itself (a host that
    links nothing of HIP).  One call through kq_bank_push_iq_async / kq_bank_pull_pcm_planes_async with such buffers against the
    same call through the synchronous entry points."""
    g = dict(samprate=192000, L=512, M=513, D=4)
    fs, L = g["samprate"], g["L"]
    olen = L // g["D"]
    plan = _mixed_plan(fs, 9)
    C, nb = len(plan), 3
    iq = wl.make_iq(fs, nb * L, seed=12, emitters=range(24, 40))
    hin = kq.HostBuffer(nb * L * 8)
    hpcm = kq.HostBuffer(C * nb * 2 * olen * 2)
    hmask = kq.HostBuffer(C * nb * 4)
    hst = kq.HostBuffer(C * nb * ctypes.sizeof(kq.ChanStatus))
    hin.array(np.complex64)[:] = iq
    got = []
    for streamed in (True, False):
        bank = kq.Bank(fs, L, g["M"], g["D"], C, nb, compute_n0=True, fwd_mode=kq.KQ_FWD_FULL)
        bank.add_channels([bank_cfg(p) for p in plan])
        if streamed:
            bank.push_iq_async(hin.ptr, nb * L)
            assert bank.process() == nb
            bank.pull_pcm_planes_async(hpcm.ptr, hmask.ptr, hst.ptr)
            bank.pull_wait(0)
            st = np.frombuffer(hst.array().tobytes(), dtype=STATUS_DTYPE).reshape(C, nb)
            pcm = hpcm.array(np.int16).reshape(C, nb, 2 * olen).copy()
            got.append(([[int(st[c, b]["nout"]) for b in range(nb)] for c in range(C)],
                        [[pcm[c, b, :int(st[c, b]["nout"])].tobytes() for b in range(nb)] for c in range(C)]))
        else:
            bank.enable_pcm(True)
            bank.push_iq(iq)
            assert bank.process() == nb
            nout = [[bank.status(c, b)["nout"] for b in range(nb)] for c in range(C)]
            got.append((nout, [[bank.pcm(c, b)[0].tobytes() for b in range(nb)] for c in range(C)]))
        bank.close()
    assert got[0][0] == got[1][0]
    assert got[0][1] == got[1][1]
    for h in (hin, hpcm, hmask, hst):
        h.free()
