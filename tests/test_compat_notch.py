"""Compat surface: notch_create / notch (filter.h:95-105, filter.c:549-571) -- host scalar code in libka9q_hip.so,
checked against the oracle restatement and on a known answer.  No GPU needed."""
import ctypes as C

import numpy as np

import kq_oracle as ko
from ka9q_sdr_amd import load_library


class CFloat(C.Structure):          # same x86-64 calling convention as float _Complex
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def _lib_notch(f, bw, x):
    L = load_library()
    L.notch_create.restype = C.c_void_p
    L.notch_create.argtypes = [C.c_double, C.c_float]
    L.notch.restype = CFloat
    L.notch.argtypes = [C.c_void_p, CFloat]
    nf = L.notch_create(f, bw)
    out = np.empty(len(x), np.complex64)
    for i, v in enumerate(x):
        r = L.notch(nf, CFloat(float(v.real), float(v.imag)))
        out[i] = complex(r.re, r.im)
    C.CDLL(None).free(C.c_void_p(nf))
    return out


def test_notch_matches_oracle_bit_for_bit():
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64)
    got = _lib_notch(0.0371, 0.01, x)
    want = ko.notch_run(0.0371, 0.01, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_notch_removes_the_tone_and_keeps_the_rest():
    n = np.arange(20000)
    tone = np.exp(2j * np.pi * 0.05 * n)
    other = 0.3 * np.exp(2j * np.pi * -0.21 * n)
    y = ko.notch_run(0.05, 0.002, (tone + other).astype(np.complex64))
    tail = y[10000:]
    # projection on the notched tone is gone, the other component passes
    assert abs(np.vdot(tone[10000:], tail)) / len(tail) < 1e-3
    assert abs(abs(np.vdot(other[10000:] / 0.3, tail)) / len(tail) - 0.3) < 3e-3


def test_notch_null_handle_returns_nan():
    L = load_library()
    L.notch.restype = CFloat
    L.notch.argtypes = [C.c_void_p, CFloat]
    r = L.notch(None, CFloat(1.0, 0.0))
    assert np.isnan(r.re)
