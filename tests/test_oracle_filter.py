"""Overlap-save filter restatement (oracle/kq_filter.c) against INDEPENDENT float64 formulations.

The reference's filter.c cannot be built here (needs FFTW headers) and ships no vectors, so these
tests check the restatement against mathematics the reference is known to implement:
scipy's Kaiser window, numpy's FFT, and direct time-domain convolution.
"""
import ctypes as C

import numpy as np
import pytest
from scipy.signal.windows import kaiser

import kq_oracle as ko


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 16384])
def test_fft_matches_numpy(n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(ko.fft_c2c(x, -1) - ref).max() / np.abs(ref).max() < 5e-7
    ref = np.fft.ifft(x.astype(np.complex128)) * n   # unnormalised backward, like FFTW
    assert np.abs(ko.fft_c2c(x, +1) - ref).max() / np.abs(ref).max() < 5e-7


@pytest.mark.parametrize("M,beta", [(33, 0.0), (33, 3.0), (129, 3.0), (129, 7.0), (2049, 3.0), (64, 2.0)])
def test_kaiser_matches_scipy(M, beta):
    """make_kaiser (filter.c:337-357): I0(pi*beta*sqrt(1-p^2))/I0(pi*beta) = scipy kaiser with beta*pi."""
    w = ko.make_kaiser(M, beta)
    assert np.abs(w - kaiser(M, beta * np.pi)).max() < 2e-6
    assert np.array_equal(w, w[::-1])
    if M % 2:
        assert w[M // 2] == 1.0


def _design64(N, L_dec, M_dec, low, high, beta, scale):
    """float64 restatement of set_filter + window_filter (filter.c:365-415, 500-546)."""
    n_dec = L_dec + M_dec - 1
    f = np.where(np.arange(n_dec) <= n_dec // 2, np.arange(n_dec), np.arange(n_dec) - n_dec) / n_dec
    H0 = np.where((f >= low) & (f <= high), scale / N, 0.0)
    h = np.fft.ifft(H0) * n_dec
    w = kaiser(M_dec, beta * np.pi)
    hp = np.zeros(n_dec, complex)
    idx = (np.arange(M_dec) - M_dec // 2) % n_dec
    hp[:M_dec] = h[idx] * w / n_dec
    return np.fft.fft(hp), hp


@pytest.mark.parametrize("D,low,high", [(4, -8000 / 48000, 8000 / 48000), (4, 100 / 48000, 3000 / 48000),
                                         (16, -0.2, 0.3), (1, -0.1, 0.1)])
def test_set_filter_matches_float64_design(D, low, high):
    L, M = 512, 513
    Lib = ko.lib()
    m = Lib.kqo_create_filter_input(L, M, ko.KQO_COMPLEX)
    s = Lib.kqo_create_filter_output(m, None, D, ko.KQO_COMPLEX)
    assert Lib.kqo_set_filter(s, low, high, 3.0) == 0
    so = C.cast(s, C.POINTER(_FilterOut)).contents
    n_dec = (L + M - 1) // D
    resp = np.ctypeslib.as_array(C.cast(so.response, C.POINTER(C.c_float)), (2 * n_dec,)).view(np.complex64).copy()
    want, _ = _design64(L + M - 1, L // D, (M - 1) // D + 1, np.float32(low), np.float32(high), 3.0, 1.0)
    assert np.abs(resp - want).max() / np.abs(want).max() < 2e-6
    # noise_gain = N * sum |H|^2 (filter.c:472-497)
    np.testing.assert_allclose(so.noise_gain, (L + M - 1) * np.sum(np.abs(want) ** 2), rtol=1e-5)
    # NaN edges are refused (filter.c:504-505)
    assert Lib.kqo_set_filter(s, float("nan"), 0.1, 3.0) == -1
    Lib.kqo_delete_filter_output(s)
    Lib.kqo_delete_filter_input(m)


class _FilterOut(C.Structure):
    _fields_ = [("master", C.c_void_p), ("out_type", C.c_int), ("response", C.c_void_p), ("f_fdomain", C.c_void_p),
                ("noise_gain", C.c_float), ("outbuf_c", C.c_void_p), ("outbuf_r", C.c_void_p), ("output_c", C.c_void_p),
                ("output_r", C.c_void_p), ("decimate", C.c_uint), ("olen", C.c_uint), ("n_dec", C.c_uint),
                ("blocknum", C.c_uint), ("plan", C.c_void_p)]


class _FilterIn(C.Structure):
    _fields_ = [("in_type", C.c_int), ("ilen", C.c_uint), ("impulse_length", C.c_uint), ("n", C.c_uint),
                ("fdomain", C.c_void_p), ("inbuf_c", C.c_void_p), ("inbuf_r", C.c_void_p), ("input_c", C.c_void_p),
                ("input_r", C.c_void_p), ("blocknum", C.c_uint), ("plan", C.c_void_p)]


def _as(ptr, n, dtype):
    nf = n * (2 if dtype == np.complex64 else 1)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (nf,)).view(dtype)


@pytest.mark.parametrize("D", [1, 4, 16])
@pytest.mark.parametrize("out_type", [ko.KQO_COMPLEX, ko.KQO_CROSS_CONJ, ko.KQO_REAL])
def test_overlap_save_complex_in_equals_direct_convolution(D, out_type):
    """COMPLEX in: output block = (x * h') decimated by D, h' = the designed impulse response scaled for
    the unnormalised N-point / N/D-point transform pair.  CROSS_CONJ and REAL out via their definitions
    (filter.c:232-234, 239-249) applied to the same linear convolution."""
    L, M = 256, 257
    N = L + M - 1
    Lib = ko.lib()
    m = Lib.kqo_create_filter_input(L, M, ko.KQO_COMPLEX)
    s = Lib.kqo_create_filter_output(m, None, D, out_type)
    low, high = (-0.11, 0.17) if out_type != ko.KQO_REAL else (-0.2, 0.2)
    Lib.kqo_set_filter(s, low, high, 3.0)
    mi = C.cast(m, C.POINTER(_FilterIn)).contents
    so = C.cast(s, C.POINTER(_FilterOut)).contents
    n_dec, olen = N // D, L // D
    resp = _as(so.response, n_dec, np.complex64).astype(complex)
    rng = np.random.default_rng(D * 10 + out_type)
    nblocks = 5
    x = (rng.standard_normal(nblocks * L) + 1j * rng.standard_normal(nblocks * L)).astype(np.complex64)
    got = []
    for b in range(nblocks):
        _as(mi.input_c, L, np.complex64)[:] = x[b * L:(b + 1) * L]
        Lib.kqo_execute_filter_input(m)
        Lib.kqo_execute_filter_output(s)
        if out_type == ko.KQO_REAL:
            got.append(_as(so.output_r, olen, np.float32).copy())
        else:
            got.append(_as(so.output_c, olen, np.complex64).copy())
    got = np.concatenate(got)
    # float64 model: full-rate spectrum X of each window times the response placed on the N-point grid
    xz = np.concatenate([np.zeros(M - 1, complex), x.astype(complex)])
    want = []
    for b in range(nblocks):
        X = np.fft.fft(xz[b * L:b * L + N])
        G = np.zeros(n_dec, complex)
        for k in range(n_dec):
            kk = k if k <= n_dec // 2 else k - n_dec
            G[k] = resp[k] * X[kk % N]
        if out_type == ko.KQO_REAL:
            Gr = G.copy()
            for p in range(1, n_dec // 2):
                Gr[p] = G[p] + np.conj(G[n_dec - p])
            full = np.zeros(n_dec, complex)
            full[0], full[n_dec // 2] = Gr[0].real, Gr[n_dec // 2].real
            for p in range(1, n_dec // 2):
                full[p], full[n_dec - p] = Gr[p], np.conj(Gr[p])
            y = (np.fft.ifft(full) * n_dec).real
        else:
            if out_type == ko.KQO_CROSS_CONJ:
                G2 = G.copy()
                for p in range(1, n_dec // 2):
                    pos, neg = G[p], G[n_dec - p]
                    G2[p], G2[n_dec - p] = pos + np.conj(neg), neg - np.conj(pos)
                G = G2
            y = np.fft.ifft(G) * n_dec
        want.append(y[n_dec - olen:])
    want = np.concatenate(want)
    assert np.sqrt(np.mean(np.abs(got - want) ** 2)) / np.sqrt(np.mean(np.abs(want) ** 2)) < 2e-6
    if out_type == ko.KQO_COMPLEX and D == 1:
        # and that IS a linear convolution with the windowed impulse response (overlap-save identity)
        h = np.fft.ifft(resp) * 1.0
        lin = np.convolve(x.astype(complex), h[:M])[:nblocks * L] * N
        assert np.sqrt(np.mean(np.abs(got - lin) ** 2)) / np.sqrt(np.mean(np.abs(lin) ** 2)) < 5e-6
    Lib.kqo_delete_filter_output(s)
    Lib.kqo_delete_filter_input(m)


@pytest.mark.parametrize("out_type", [ko.KQO_COMPLEX, ko.KQO_REAL])
def test_overlap_save_real_in(out_type):
    """REAL in (packet.c:272, modulate.c:131, FM audio fm.c:43): r2c master, filter.c:214-216 / 206-208."""
    L, M, D = 128, 129, 1
    N = L + M - 1
    Lib = ko.lib()
    m = Lib.kqo_create_filter_input(L, M, ko.KQO_REAL)
    nb = N // 2 + 1 if out_type == ko.KQO_REAL else N
    resp_buf = (C.c_float * (2 * N))()
    rng = np.random.default_rng(4)
    resp = (rng.standard_normal(nb) + 1j * rng.standard_normal(nb)).astype(np.complex64) / N
    # response must be malloc'ed: the filter frees it (filter.c:271)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    p = libc.malloc(8 * N)
    C.memmove(p, resp.ctypes.data, 8 * nb)
    s = Lib.kqo_create_filter_output(m, p, D, out_type)
    mi = C.cast(m, C.POINTER(_FilterIn)).contents
    so = C.cast(s, C.POINTER(_FilterOut)).contents
    x = rng.standard_normal(3 * L).astype(np.float32)
    xz = np.concatenate([np.zeros(M - 1), x.astype(float)])
    for b in range(3):
        _as(mi.input_r, L, np.float32)[:] = x[b * L:(b + 1) * L]
        Lib.kqo_execute_filter_input(m)
        Lib.kqo_execute_filter_output(s)
        X = np.fft.fft(xz[b * L:b * L + N])
        if out_type == ko.KQO_REAL:
            G = resp.astype(complex) * X[:nb]
            full = np.zeros(N, complex)
            full[0], full[N // 2] = G[0].real, G[N // 2].real
            full[1:N // 2] = G[1:N // 2]
            full[N // 2 + 1:] = np.conj(G[1:N // 2][::-1])
            want = (np.fft.ifft(full) * N).real[N - L:]
            got = _as(so.output_r, L, np.float32)
        else:
            G = np.zeros(N, complex)
            G[:N // 2 + 1] = resp[:N // 2 + 1] * X[:N // 2 + 1]
            for k in range(N // 2 + 1, N):
                G[k] = resp[k] * np.conj(X[N - k])
            want = (np.fft.ifft(G) * N)[N - L:]
            got = _as(so.output_c, L, np.complex64)
        assert np.abs(got - want).max() / np.abs(want).max() < 5e-6
    Lib.kqo_delete_filter_output(s)
    Lib.kqo_delete_filter_input(m)
    _ = resp_buf


def test_compute_n0_white_noise():
    """radio.c:383-425 on white noise: two-pass thresholded mean -> sigma^2 per component per Hz."""
    N, fs = 16384, 192000
    rng = np.random.default_rng(0)
    sigma = 1e-3
    x = sigma * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    X = np.fft.fft(x).astype(np.complex64)
    n0 = ko.compute_n0(X, fs, -8000.0, 8000.0)
    P = np.abs(X.astype(complex)) ** 2
    f = np.where(np.arange(N) <= N // 2, np.arange(N), np.arange(N) - N) * fs / N
    out = ~((f >= -8000) & (f <= 8000))
    a1 = P[out].mean()
    a2 = P[out & (P < 2 * a1)].mean()
    np.testing.assert_allclose(n0, a2 / (2.0 * N * fs), rtol=2e-4)


@pytest.mark.parametrize("log2n", [2, 3, 5, 8, 11, 12, 14, 15, 16])
def test_fast_transform_equals_the_plain_one(log2n):
    """oracle/kq_fft.c has two transforms: the plain radix-2 one the parity vectors were made with, and the radix-4
    autosort one that bench.py's cpu_baseline times (kqo_fft_set_fast).  Same conventions, both signs, odd and even
    log2 n, and the r2c / c2r wrappers on top: equal to float rounding, and both equal to numpy's float64 transform."""
    import ctypes as C
    L = ko.lib()
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    xr = rng.standard_normal(n).astype(np.float32)
    p = L.kqo_fft_create(n)
    try:
        res = {}
        for fast in (0, 1):
            L.kqo_fft_set_fast(fast)
            for sign in (-1, 1):
                out = np.zeros(n, np.complex64)
                L.kqo_fft_c2c(p, x.ctypes.data, out.ctypes.data, sign)
                res[fast, sign] = out
            half = np.zeros(n // 2 + 1, np.complex64)
            L.kqo_fft_r2c(p, xr.ctypes.data, half.ctypes.data)
            back = np.zeros(n, np.float32)
            L.kqo_fft_c2r(p, half.ctypes.data, back.ctypes.data)
            res[fast, "r2c"], res[fast, "c2r"] = half, back
    finally:
        L.kqo_fft_set_fast(0)
        L.kqo_fft_destroy(p)
    want = {-1: np.fft.fft(x.astype(np.complex128)), 1: np.fft.ifft(x.astype(np.complex128)) * n,
            "r2c": np.fft.rfft(xr.astype(np.float64)), "c2r": xr.astype(np.float64) * n}
    tol = 4e-7 * max(1, log2n) ** 0.5
    for key, w in want.items():
        scale = np.sqrt(np.mean(np.abs(w) ** 2))
        for fast in (0, 1):
            assert np.sqrt(np.mean(np.abs(res[fast, key] - w) ** 2)) / scale < tol, (key, fast)
        assert np.sqrt(np.mean(np.abs(res[1, key] - res[0, key]) ** 2)) / scale < tol, key


@pytest.mark.parametrize("n", [6, 60, 1200, 1920, 3000, 9600, 12000, 15360, 48000, 14, 1792, 7168, 13440, 2 * 7 ** 4])
def test_mixed_radix_transform_against_float64(n):
    """Sizes 2^a 3^b 5^c 7^d (round 6: FFTW plans whatever N and N / decimate come out, filter.c:78,132; decimate =
    samprate / 48000 is 5 at 240 kHz, radio_status.c:266): the oracle's mixed-radix transform against numpy's float64
    one, both directions, and the r2c / c2r pair the FM audio filter uses; a size with a factor 11 is refused."""
    L = ko.lib()
    L.kqo_fft_create.restype = C.c_void_p
    L.kqo_fft_create.argtypes = [C.c_uint]
    L.kqo_fft_destroy.argtypes = [C.c_void_p]
    L.kqo_fft_c2c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.kqo_fft_r2c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kqo_fft_c2r.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert not L.kqo_fft_create(11 * n)
    p = L.kqo_fft_create(n)
    assert p
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y = np.zeros(n, np.complex64)
    for sign, ref in ((-1, np.fft.fft(x.astype(np.complex128))), (+1, np.fft.ifft(x.astype(np.complex128)) * n)):
        L.kqo_fft_c2c(p, x.ctypes.data, y.ctypes.data, sign)
        assert np.sqrt(np.mean(np.abs(y - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 4e-7
    xr = rng.standard_normal(n).astype(np.float32)
    yc = np.zeros(n // 2 + 1, np.complex64)
    L.kqo_fft_r2c(p, xr.ctypes.data, yc.ctypes.data)
    ref = np.fft.rfft(xr.astype(np.float64))
    assert np.sqrt(np.mean(np.abs(yc - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 4e-7
    back = np.zeros(n, np.float32)
    L.kqo_fft_c2r(p, yc.ctypes.data, back.ctypes.data)
    assert np.abs(back / n - xr).max() < 3e-6
    L.kqo_fft_destroy(p)


def test_overlap_save_at_a_size_with_factors_3_and_5_is_direct_convolution():
    """N = 9600, decimate 5 (a 240 kHz front end): the oracle's master + slave against the direct convolution of the
    input with the designed impulse response, decimated -- the overlap-save identity, independent of any FFT."""
    Lb, D = 4800, 5
    M = Lb + 1
    N = Lb + M - 1
    O = ko.lib()
    om = O.kqo_create_filter_input(Lb, M, 1)
    os_ = O.kqo_create_filter_output(om, None, D, 1)
    assert om and os_
    assert O.kqo_set_filter(os_, -0.11, 0.17, 3.0) == 0
    oso = C.cast(os_, C.POINTER(_FilterOut)).contents
    omi = C.cast(om, C.POINTER(_FilterIn)).contents
    nd = N // D
    H = _as(oso.response, nd, np.complex64).astype(np.complex128)
    # the decimated response H[k] serves signed master bins k: the equivalent N-point response is zero elsewhere
    Hfull = np.zeros(N, np.complex128)
    Hfull[:nd // 2 + 1] = H[:nd // 2 + 1]
    Hfull[N - (nd - nd // 2 - 1):] = H[nd // 2 + 1:]
    h = np.fft.ifft(Hfull) * N                  # impulse response at the input rate (both transforms are unnormalised)
    rng = np.random.default_rng(3)
    nb = 4
    x = (rng.standard_normal(nb * Lb) + 1j * rng.standard_normal(nb * Lb)).astype(np.complex64)
    full = np.convolve(np.concatenate([np.zeros(M - 1), x.astype(np.complex128)]), h[:N])[M - 1:M - 1 + nb * Lb]
    for b in range(nb):
        _as(omi.input_c, Lb, np.complex64)[:] = x[b * Lb:(b + 1) * Lb]
        O.kqo_execute_filter_input(om)
        O.kqo_execute_filter_output(os_)
        got = _as(oso.output_c, Lb // D, np.complex64)
        want = full[b * Lb:(b + 1) * Lb][::D]
        if b > 0:                                       # (block 0 starts from a zero history on both sides)
            assert np.sqrt(np.mean(np.abs(got - want) ** 2) / np.mean(np.abs(want) ** 2)) < 2e-5
    O.kqo_delete_filter_output(os_)
    O.kqo_delete_filter_input(om)
