"""Half-band decimators: the oracle restatement against the reference's own decimate.c
(oracle/_ref/libref_decimate.so, built unmodified) and against committed golden vectors of that build."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle")]
import kq_oracle as ko  # noqa: E402

GOLD = os.path.join(HERE, "golden", "decimate_ref.npz")


def _impulse_response(log_dec, thr):
    x = np.zeros(64 << log_dec, np.float32)
    x[0] = 1
    y, _ = ko.halfband_cascade(x, log_dec, thr)
    return y


def test_hb3_is_1_2_1():
    # decimate.c:146-160: state holds the previous odd sample
    x = np.arange(1, 17, dtype=np.float32)
    y, _ = ko.halfband_cascade(x, 1, 0)
    want = [2 * x[2 * k] + x[2 * k + 1] + (x[2 * k - 1] if k else 0) for k in range(8)]
    assert np.array_equal(y, np.array(want, np.float32))


def test_hb15_taps_and_dc_gain():
    # impulse on an even input slot hits the unity centre tap 3 outputs later; on an odd slot the four pairs
    x = np.zeros(64, np.float32)
    x[1] = 1
    y, _ = ko.halfband_cascade(x, 1, 8)
    c = np.array([-6, 33, -116, 490, 490, -116, 33, -6], np.float32) / np.float32(802)
    np.testing.assert_allclose(y[:8], c, rtol=1e-6)
    x[:] = 0
    x[0] = 1
    y, _ = ko.halfband_cascade(x, 1, 8)
    assert y[3] == 1 and np.count_nonzero(y) == 1
    # DC gain: 2 per 15-tap stage (what hackrf.c:469 compensates), 4 per 1-2-1 stage
    y, _ = ko.halfband_cascade(np.ones(4096, np.float32), 4, 2)
    np.testing.assert_allclose(y[-1], 4.0 * 4.0 * 2.0 * 2.0, rtol=1e-5)


def test_state_carries_across_calls():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(8 << 6).astype(np.float32)
    whole, _ = ko.halfband_cascade(np.concatenate([x, x]), 6, 3)
    a, st = ko.halfband_cascade(x, 6, 3)
    b, _ = ko.halfband_cascade(x, 6, 3, st)
    assert np.array_equal(np.concatenate([a, b]), whole)


@pytest.mark.skipif(ko.ref_decimate_lib() is None, reason="oracle/_ref not built (no reference tree)")
@pytest.mark.parametrize("log_dec,thr", [(6, 8), (6, 3), (10, 8), (3, 0)])
def test_oracle_matches_reference_build(log_dec, thr):
    """Pin: the reference is built with -funsafe-math-optimizations, which may reassociate the 1-2-1 sum and the tap
    accumulation, so agreement is to a few ulp of the running magnitude rather than bitwise."""
    rng = np.random.default_rng(log_dec * 16 + thr)
    x = rng.standard_normal(32 << log_dec).astype(np.float32)
    a, sa = ko.halfband_cascade(x, log_dec, thr)
    b, sb = ko.halfband_cascade(x, log_dec, thr, use_ref=True)
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= 4e-7 * scale
    a2, _ = ko.halfband_cascade(x[::-1].copy(), log_dec, thr, sa)
    b2, _ = ko.halfband_cascade(x[::-1].copy(), log_dec, thr, sb, use_ref=True)
    assert np.abs(a2 - b2).max() <= 4e-7 * scale


def test_golden_vectors_from_reference_build():
    g = np.load(GOLD)
    for key in ("l6_t8", "l6_t3", "l4_t0"):
        log_dec, thr = int(g[key + "_cfg"][0]), int(g[key + "_cfg"][1])
        y, st = ko.halfband_cascade(g["x"][: 64 << log_dec], log_dec, thr)
        y2, _ = ko.halfband_cascade(g["x"][: 64 << log_dec], log_dec, thr, st)
        want = g[key]
        assert np.abs(np.concatenate([y, y2]) - want).max() <= 4e-7 * np.abs(want).max()


def test_front_end_block():
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(16 << 6) + 1j * rng.standard_normal(16 << 6)).astype(np.complex64)
    fe = ko.FrontEndDecimator(6, 8, offset=1)
    y, s16, e = fe.process(x)
    # rotation by +Fs/4 then decimation == decimating x * j^n
    rot = (x * (1j ** (np.arange(len(x)) & 3))).astype(np.complex64)
    fe0 = ko.FrontEndDecimator(6, 8, offset=0)
    y0, _, _ = fe0.process(rot)
    assert np.array_equal(y, y0)
    v = np.float32(32767) * y.real
    assert np.all(np.abs(s16[:, 0] - v) <= 0.5)
    np.testing.assert_allclose(e, np.sum(np.abs(y.astype(np.complex128)) ** 2), rtol=1e-5)
