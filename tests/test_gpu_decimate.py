"""GPU half-band cascade (kq_decim_*) against the oracle: bit-exact, state carried across calls."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle")]
import kq_oracle as ko  # noqa: E402

pytestmark = pytest.mark.gpu


def _iq(n, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


@pytest.mark.parametrize("log_dec,thr,offset", [(6, 8, 1), (6, 3, 1), (1, 8, 0), (2, 0, 3), (4, 2, 2), (10, 8, 1),
                                               (7, 5, 1)])
def test_cascade_bit_exact(gpu, log_dec, thr, offset):
    from ka9q_sdr_amd import Decimator
    blocks = [700, 1, 513, 64]  # n_out per call: ragged, smaller than the halo, across workgroup tiles
    if log_dec >= 10:
        blocks = [96, 1, 33]
    dec = Decimator(log_dec, thr, offset, max_out=1024)
    fe = ko.FrontEndDecimator(log_dec, thr, offset)
    for i, n_out in enumerate(blocks):
        x = _iq(n_out << log_dec, 100 + i)
        y, s16, e = dec.process(x)
        wy, ws16, we = fe.process(x)
        assert np.array_equal(y.view(np.uint32), wy.view(np.uint32)), "call %d" % i
        assert np.array_equal(s16, ws16)
        np.testing.assert_allclose(e, we, rtol=2e-6)
    dec.close()


def test_reset_and_int16_clip_region(gpu):
    from ka9q_sdr_amd import Decimator
    dec = Decimator(3, 8, 1, filter_atten=1.0, max_out=256)  # no attenuation: |s| reaches several units
    fe = ko.FrontEndDecimator(3, 8, 1, filter_atten=1.0)
    x = _iq(256 << 3, 7, scale=0.2)
    y, s16, _ = dec.process(x)
    wy, ws16, _ = fe.process(x)
    assert np.array_equal(y, wy)
    ok = np.abs(np.float32(32767) * wy.real) < 32767  # outside that the C conversion is undefined (hackrf.c:310)
    assert np.array_equal(s16[ok, 0], ws16[ok, 0])
    dec.reset()
    y2, _, _ = dec.process(x)
    assert np.array_equal(y2, y)
    dec.close()


def test_device_resident_feeds_bank(gpu):
    """Decimator output on the device goes straight into the channel bank (front end -> radio, zero copy)."""
    import ctypes as C
    from ka9q_sdr_amd import Bank, Decimator, channel_config, KQ_FM_DEMOD
    hip = C.CDLL("libamdhip64.so")  # the runtime libka9q_hip.so is already bound to
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    log_dec, L, M, D = 3, 7680, 513, 8
    nblk = 4
    x = _iq((L * nblk) << log_dec, 11)
    t = np.arange(len(x))
    x += (0.5 * np.exp(2j * np.pi * (0.01 / 8) * t + 1j * 0.3 * np.sin(2 * np.pi * 1e-5 * t))).astype(np.complex64)
    xin, yout = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(xin), x.nbytes) == 0
    assert hip.hipMalloc(C.byref(yout), 8 * L * nblk) == 0
    assert hip.hipMemcpy(xin, x.ctypes.data, x.nbytes, 1) == 0
    dec = Decimator(log_dec, 8, 0, max_out=L * nblk)
    dec.process_device(xin, L * nblk, yout)
    dec.sync()
    bank = Bank(192000, L, M, D, max_channels=1, max_blocks=nblk)
    ch = bank.add_channel(channel_config(KQ_FM_DEMOD, -8000, 8000, second_lo=-0.01 * 192000))
    bank.push_iq_device(yout, L * nblk)
    assert bank.process() == nblk
    bank.sync()
    y = np.empty(L * nblk, np.complex64)
    assert hip.hipMemcpy(y.ctypes.data, yout, y.nbytes, 2) == 0
    fe = ko.FrontEndDecimator(log_dec, 8, 0)
    wy, _, _ = fe.process(x)
    assert np.array_equal(y, wy)
    # and the bank demodulates the decimated stream exactly as it does the same samples pushed from the host
    ref = Bank(192000, L, M, D, max_channels=1, max_blocks=nblk)
    ref.add_channel(channel_config(KQ_FM_DEMOD, -8000, 8000, second_lo=-0.01 * 192000))
    ref.push_iq(wy)
    ref.process()
    ref.sync()
    for b in range(nblk):
        assert np.array_equal(bank.audio(ch, b), ref.audio(0, b))
    assert np.abs(bank.audio(ch, nblk - 1)).max() > 0
    hip.hipFree(xin)
    hip.hipFree(yout)
    dec.close()
    bank.close()
    ref.close()


def test_random_cascades_bit_exact(gpu):
    """Twenty random cascades (1..11 stages, any switch point between the two filters, any rotation step, unit or
    default attenuation) over random call sizes: still bit for bit."""
    from ka9q_sdr_amd import Decimator
    rng = np.random.default_rng(2025)
    for trial in range(20):
        log_dec = int(rng.integers(1, 12))
        thr = int(rng.integers(0, 13))
        offset = int(rng.integers(0, 4))
        atten = float(rng.choice([0.0, 1.0, 0.37]))
        max_out = 4096 >> max(0, log_dec - 6)
        dec = Decimator(log_dec, thr, offset, filter_atten=atten, max_out=max_out)
        fe = ko.FrontEndDecimator(log_dec, thr, offset, filter_atten=atten if atten else None)
        for call in range(4):
            n_out = int(rng.integers(1, max_out + 1))
            x = _iq(n_out << log_dec, 1000 * trial + call, scale=0.02)
            y, s16, e = dec.process(x)
            wy, ws16, we = fe.process(x)
            assert np.array_equal(y.view(np.uint32), wy.view(np.uint32)), (trial, log_dec, thr, offset, call)
            ok = np.abs(np.float32(32767) * wy.real) < 32767
            assert np.array_equal(s16[ok, 0], ws16[ok, 0])
            np.testing.assert_allclose(e, we, rtol=5e-6)
        dec.close()


@pytest.mark.parametrize("log_dec", [4, 6])
def test_many_tiles_energy_on_device(gpu, log_dec):
    """More tiles than resident workgroups, outputs and energy left on the device, three calls of different sizes: the
    edge workgroup adds up the tagged per-workgroup energies of THIS call (a stale word of the call before must not
    count), and the outputs stay bit-exact whichever workgroup took which tile."""
    import torch
    from ka9q_sdr_amd import Decimator
    max_out = 300_000 if log_dec == 4 else 80_000
    dec = Decimator(log_dec, 8, 1, max_out=max_out, stream=torch.cuda.current_stream().cuda_stream)
    fe = ko.FrontEndDecimator(log_dec, 8, 1)
    y = torch.empty(max_out, 2, device="cuda", dtype=torch.float32)
    s16 = torch.empty(max_out, 2, device="cuda", dtype=torch.int16)
    e = torch.full((1,), -1.0, device="cuda", dtype=torch.float32)
    for call, n_out in enumerate([max_out, max_out // 3 + 5, max_out - 255]):
        x = _iq(n_out << log_dec, 40 + call)
        xd = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
        dec.process_device(xd.data_ptr(), n_out, y.data_ptr(), s16.data_ptr(), e.data_ptr())
        dec.sync()
        wy, ws16, we = fe.process(x)
        got = y[:n_out].cpu().numpy().view(np.complex64).ravel()
        assert np.array_equal(got.view(np.uint32), wy.view(np.uint32)), call
        assert np.array_equal(s16[:n_out].cpu().numpy(), ws16)
        exact = float(np.sum(np.abs(wy.astype(np.complex128)) ** 2))
        np.testing.assert_allclose(float(e.item()), exact, rtol=1e-6)
        np.testing.assert_allclose(float(e.item()), we, rtol=2e-5)  # the oracle's own sum is sequential float32
    dec.close()
