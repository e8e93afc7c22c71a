"""GPU AFSK-1200 / HDLC decoder bank (kq_afsk_*) against the oracle: decoded frames byte-exact, HDLC
state exact, bit-clock phase to the one sample the oracle itself resolves, filter output within 1e-5 relative RMS."""
import numpy as np
import pytest

import kq_oracle as ko
from common import afsk_audio, afsk_bits, ax25_fcs, rel_rms

pytestmark = pytest.mark.gpu

FRAMES = [bytes([0x82, 0xA0, 0xA4, 0xA6, 0x40, 0x40, 0x60, 0x96, 0x82, 0x72, 0xA2, 0x40, 0x40, 0x61, 0x03, 0xF0]) +
          b"!4903.50N/07201.75W-test %d" % i for i in range(6)] + [bytes(range(1, 200)), bytes([0xFF] * 40)]


def _session_audio(k, n_total):
    rng = np.random.default_rng(100 + k)
    pick = [FRAMES[i] for i in rng.permutation(len(FRAMES))[: 2 + k % 3]]
    # the transmitter keys up after a noise-only lead and then idles on flags to the end of the stream: the bit clock
    # is a random walk on noise alone (one rounding-level decision shifts it for good), but locked on a signal, so
    # the decoder state at the end is comparable
    bits = afsk_bits(pick, lead_flags=6 + k)
    bits = bits + [0, 1, 1, 1, 1, 1, 1, 0] * (n_total // 320 + 2)
    x = afsk_audio(bits, amp=0.2 + 0.1 * (k % 4), clock_ppm=(-1) ** k * 150.0 * (k % 4))
    lead = 100 * k + 7
    x = np.concatenate([np.zeros(lead, np.float32), x])
    out = ((0.004 + 0.01 * (k % 5)) * rng.standard_normal(n_total)).astype(np.float32)
    out[: min(len(x), n_total)] += x[:n_total]
    return out, pick


def test_bank_decodes_like_oracle(gpu):
    from ka9q_sdr_amd import AfskBank
    S = 12
    n_total = 48000 * 3
    audio, want = zip(*[_session_audio(k, n_total) for k in range(S)])
    audio = np.stack(audio)
    bank = AfskBank(S, max_frames=8)
    oracles = [ko.Afsk() for _ in range(S)]
    rng = np.random.default_rng(1)
    i = 0
    while i < n_total:                      # ragged call sizes, some shorter than a block, some many blocks long
        n = int(rng.choice([1, 333, 960, 1000, 4800, 25001]))
        n = min(n, n_total - i)
        nb = bank.push(audio[:, i:i + n])
        for k in range(S):
            oracles[k].push(audio[k, i:i + n])
        i += n
        if nb:
            for k in (0, S - 1):
                assert rel_rms(bank.filter_output(k), oracles[k].filter_output()) < 1e-5
    for k in range(S):
        got = bank.frames(k)
        assert got == oracles[k].frames(), "session %d" % k
        assert got == [f + ax25_fcs(f) for f in want[k]], "session %d decodes what was sent" % k
        st, ost = bank.state(k), oracles[k].state()
        for key in ("frame_bit", "flagsync", "ones"):
            assert st[key] == ost[key], (k, key)
        # the bang-bang bit clock dithers around lock: the oracle's own final symphase moves by one sample when its
        # input is scaled by 1 +- 1e-7, so that is the resolution at which it can be compared
        assert abs(st["symphase"] - ost["symphase"]) <= 1, k
        assert st["decoded_packets"] == len(got)
        assert st["pending_samples"] == n_total % 1000
        assert st["blocks"] == n_total // 1000
        assert bank.dropped(k) == 0
    bank.close()


def test_pcm_words_and_arena_limit(gpu):
    from ka9q_sdr_amd import AfskBank
    S = 3
    x = afsk_audio(afsk_bits(FRAMES[:5], lead_flags=10), amp=0.4)
    x = np.concatenate([x, np.zeros(2000 - len(x) % 1000, np.float32)])
    pcm = np.round(x * 32767).astype(">i2")
    words = np.stack([pcm] * S)
    bank = AfskBank(S, max_frames=3)                       # arena smaller than the number of good frames
    o = ko.Afsk()
    bank.push_pcm_be(words)
    o.push_pcm_be(pcm.tobytes())
    of = o.frames()
    for k in range(S):
        assert bank.frames(k) == of[:3]
        assert bank.dropped(k) == len(of) - 3
        assert bank.state(k)["decoded_packets"] == len(of)
    assert rel_rms(bank.filter_output(1), o.filter_output()) < 1e-5
    bank.clear_frames()
    assert bank.frames(0) == [] and bank.dropped(0) == 0
    bank.close()


def test_device_resident_input(gpu):
    import ctypes as C
    from ka9q_sdr_amd import AfskBank
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    S, n = 4, 40000
    audio = np.stack([_session_audio(k, n)[0] for k in range(S)])
    padded = np.zeros((S, n + 24), np.float32)             # a stride larger than the row
    padded[:, :n] = audio
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), padded.nbytes) == 0
    assert hip.hipMemcpy(d, padded.ctypes.data, padded.nbytes, 1) == 0
    a, b = AfskBank(S), AfskBank(S)
    assert a.push_device(d, n, n + 24) == 40
    a.sync()
    assert b.push(audio) == 40
    for k in range(S):
        assert a.frames(k) == b.frames(k)
        np.testing.assert_array_equal(a.filter_output(k), b.filter_output(k))
    assert sum(len(a.frames(k)) for k in range(S)) > 0
    hip.hipFree(d)
    a.close()
    b.close()
