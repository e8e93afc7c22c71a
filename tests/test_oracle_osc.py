"""NCO: the oracle restatement (oracle/kq_osc.c) against the reference itself.

oracle/_ref/libref_osc.so is the reference's osc.c + dsp.c compiled unmodified (oracle/Makefile), so
these tests PIN the restatement.  tests/golden/osc_*.npz hold sequences generated from that build
(tests/golden/make_golden.py) for boxes where neither the reference tree nor _ref is present.
"""
import ctypes as C
import os

import numpy as np
import pytest

import kq_oracle as ko

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-11   # double recurrences compiled with different FP contraction flags


def _run_oracle(script, n):
    L = ko.lib()
    o = ko.Osc()
    out = np.zeros((n, 2))
    for i in range(n):
        if i in script:
            f, r = script[i]
            L.kqo_set_osc(C.byref(o), f, r)
        z = L.kqo_step_osc(C.byref(o))
        out[i] = (z.re, z.im)
    return out, o.steps


def _run_ref(script, n):
    R = ko.ref_osc_lib()
    o = ko.RefOsc()
    out = np.zeros((n, 2))
    for i in range(n):
        if i in script:
            f, r = script[i]
            R.set_osc(C.byref(o), f, r)
        z = R.step_osc(C.byref(o))
        out[i] = (z.re, z.im)
    return out, o.steps


SCRIPTS = {
    "fixed": ({0: (0.0123456789, 0.0)}, 40000),                       # > 2 renormalisation periods (osc.c:11)
    "sweep": ({0: (-0.31, 3.5e-10)}, 40000),
    "frozen": ({0: (0.0, 1e-9)}, 300),                                # freq == 0 never advances (osc.c:43)
    "retune": ({0: (0.1, 0.0), 5000: (-0.2000001, 0.0), 9000: (0.0, 0.0), 9500: (0.05, 1e-9)}, 20000),
}


@pytest.mark.parametrize("name", sorted(SCRIPTS))
def test_oracle_osc_matches_reference_build(name):
    if ko.ref_osc_lib() is None:
        pytest.skip("oracle/_ref not built (no reference tree on this box); golden test covers it")
    script, n = SCRIPTS[name]
    a, sa = _run_oracle(script, n)
    b, sb = _run_ref(script, n)
    assert sa == sb
    assert np.abs(a - b).max() < TOL


@pytest.mark.parametrize("name", sorted(SCRIPTS))
def test_oracle_osc_matches_golden(name):
    script, n = SCRIPTS[name]
    g = np.load(os.path.join(GOLD, "osc_%s.npz" % name))
    a, steps = _run_oracle(script, n)
    assert steps == int(g["steps"])
    idx = g["index"]
    assert np.abs(a[idx] - g["phasor"]).max() < TOL


def test_closed_form_matches_recurrence():
    """phase(n) = f n + r n (n-1)/2 (what the GPU evaluates) against the stepped reference phasor."""
    script, n = SCRIPTS["sweep"]
    a, _ = _run_oracle(script, n)
    f, r = script[0]
    k = np.arange(n, dtype=np.float64)
    ph = f * k + r * (0.5 * k * (k - 1.0))
    z = np.exp(2j * np.pi * (ph - np.rint(ph)))
    assert np.abs(a[:, 0] + 1j * a[:, 1] - z).max() < 1e-9


def test_is_phasor_init():
    if ko.ref_osc_lib() is None:
        pytest.skip("needs oracle/_ref")
    R, L = ko.ref_osc_lib(), ko.lib()
    L.kqo_is_phasor_init.argtypes = [ko._Cplx]
    for re, im in ((1.0, 0.0), (0.0, 0.0), (0.9, 0.3), (0.6, 0.6), (float("nan"), 0.0), (0.0, float("nan"))):
        z = ko._Cplx(re, im)
        assert R.is_phasor_init(z) == L.kqo_is_phasor_init(z)
