"""The reference's own one-channel filter API (include/ka9q_hip_compat.h) served by the GPU, against the
oracle: create/execute/set_filter for every in/out type combination of filter.c:206-250."""
import ctypes as C

import numpy as np
import pytest

import ka9q_sdr_amd as kq
import kq_oracle as ko
from test_oracle_filter import _FilterIn as OFilterIn, _FilterOut as OFilterOut, _as

pytestmark = pytest.mark.gpu


class FilterIn(C.Structure):      # struct filter_in of ka9q_hip_compat.h (x86-64 glibc pthread sizes)
    _fields_ = [("in_type", C.c_int), ("ilen", C.c_uint), ("impulse_length", C.c_uint), ("fdomain", C.c_void_p),
                ("input_buffer", C.c_void_p), ("input", C.c_void_p), ("fwd_plan", C.c_void_p), ("blocknum", C.c_uint),
                ("filter_mutex", C.c_byte * 40), ("filter_cond", C.c_byte * 48)]


class FilterOut(C.Structure):
    _fields_ = [("master", C.c_void_p), ("out_type", C.c_int), ("response", C.c_void_p), ("response_mutex", C.c_byte * 40),
                ("f_fdomain", C.c_void_p), ("noise_gain", C.c_float), ("output_buffer", C.c_void_p), ("output", C.c_void_p),
                ("rev_plan", C.c_void_p), ("decimate", C.c_uint), ("olen", C.c_uint), ("blocknum", C.c_uint)]


@pytest.fixture(scope="module")
def lib(gpu):
    L = kq.load_library()
    L.create_filter_input.restype = C.POINTER(FilterIn)
    L.create_filter_input.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.create_filter_output.restype = C.POINTER(FilterOut)
    L.create_filter_output.argtypes = [C.POINTER(FilterIn), C.c_void_p, C.c_uint, C.c_int]
    L.execute_filter_input.argtypes = [C.POINTER(FilterIn)]
    L.execute_filter_output.argtypes = [C.POINTER(FilterOut)]
    L.delete_filter_input.argtypes = [C.POINTER(FilterIn)]
    L.delete_filter_output.argtypes = [C.POINTER(FilterOut)]
    L.set_filter.argtypes = [C.POINTER(FilterOut), C.c_float, C.c_float, C.c_float]
    return L


# (1, 3, 4) -- COMPLEX in, REAL out -- runs first: it reads all N_dec response bins (filter.c:232-234), which a
# preceding case of the same size would otherwise have left behind in the allocator's recycled block.
# The last two are the master sizes past one LDS block: cfg 5's N = 65536 (create_filter_input has no size limit,
# filter.c:54-91) and 2^17.
# Lb = 4800 ... : sizes with factors 3 and 5 (N = 9600, 15360, 3840; 48000 = 2^7 3 5^3 past one LDS block), which FFTW plans
# like any other (filter.c:78,132): the mixed-radix transforms of kq_ldsfft.hpp.  Lb = 6720, 3584, 896, 23520: a factor 7 (336 kHz =
# 7 x 48 kHz: N = 13440; N = 7168 = 2^10 7; 1792; 47040 = 2^6 3 5 7^2 past one LDS block).
@pytest.mark.parametrize("in_type,out_type,D,Lb,M", [(a, b, c, d, None) for a, b, c, d in [(1, 3, 4, 512), (1, 1, 4, 512), (1, 2, 4, 512), (1, 1, 16, 512),
                                                    (3, 3, 1, 512), (3, 1, 1, 512), (1, 1, 512, 32768), (1, 2, 64, 65536),
                                                    (1, 3, 5, 4800), (1, 1, 5, 4800), (1, 2, 8, 7680), (3, 3, 1, 1920),
                                                    (3, 1, 2, 1920), (1, 1, 25, 24000),
                                                    (1, 1, 7, 6720), (1, 3, 4, 3584), (3, 3, 1, 896), (1, 2, 14, 23520)]] + [
    # L and M apart (filter.c:78: N = L + M - 1 whatever they are): the reference's default -L 3840 -M 4353 in every output type
    # (the history is longer than a block: filter.c:168's memmove overlaps), an impulse response of three blocks, a short one,
    # the FM audio filter of such a geometry (fm.c:64: 640 samples per block, 1409 taps), decimate 1
    (1, 1, 4, 3840, 4353), (1, 2, 4, 3840, 4353), (1, 3, 4, 3840, 4353), (1, 1, 8, 1024, 3073), (1, 1, 5, 2880, 1921),
    (3, 3, 1, 640, 1409), (3, 1, 2, 2880, 961), (1, 1, 1, 960, 961), (1, 1, 16, 49152, 16385)])
def test_compat_filter_matches_oracle(lib, in_type, out_type, D, Lb, M):
    M = Lb + 1 if M is None else M
    N = Lb + M - 1
    O = ko.lib()
    m = lib.create_filter_input(Lb, M, in_type)
    assert m and m.contents.ilen == Lb and m.contents.impulse_length == M
    s = lib.create_filter_output(m, None, D, out_type)
    assert s and s.contents.olen == Lb // D and np.isnan(s.contents.noise_gain)
    om = O.kqo_create_filter_input(Lb, M, in_type)
    os_ = O.kqo_create_filter_output(om, None, D, out_type)
    if in_type == 3 and out_type == 3:
        # REAL/REAL slaves take an N/2+1-bin response (fm.c:56-66): use the FM de-emphasis design
        from ka9q_sdr_amd import workload  # noqa: F401
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        resp = np.zeros(N // 2 + 1, np.complex64)
        f = np.arange(N // 2 + 1) * 48000.0 / N
        sel = (f >= 300) & (f <= 6000)
        resp[sel] = (10.0 / N) * 300.0 / f[sel]
        for target, setter in ((s, None), (os_, None)):
            p = libc.malloc(8 * (N // 2 + 1))
            C.memmove(p, resp.ctypes.data, 8 * (N // 2 + 1))
            if target is s:
                s.contents.response = p
            else:
                C.cast(os_, C.POINTER(OFilterOut)).contents.response = p
    else:
        assert lib.set_filter(s, -0.11, 0.17, 3.0) == 0
        assert O.kqo_set_filter(os_, -0.11, 0.17, 3.0) == 0
        oso = C.cast(os_, C.POINTER(OFilterOut)).contents
        np.testing.assert_allclose(s.contents.noise_gain, oso.noise_gain, rtol=1e-5)
    assert lib.set_filter(s, float("nan"), 0.1, 3.0) == -1      # filter.c:504-505
    assert lib.execute_filter_input(None) == -1                  # filter.c:148-149
    omi = C.cast(om, C.POINTER(OFilterIn)).contents
    oso = C.cast(os_, C.POINTER(OFilterOut)).contents
    rng = np.random.default_rng(7)
    olen = Lb // D
    for b in range(4 if M <= Lb + 1 else 7):
        if in_type == 3:
            x = rng.standard_normal(Lb).astype(np.float32)
            _as(m.contents.input, Lb, np.float32)[:] = x
            _as(omi.input_r, Lb, np.float32)[:] = x
        else:
            x = (rng.standard_normal(Lb) + 1j * rng.standard_normal(Lb)).astype(np.complex64)
            _as(m.contents.input, Lb, np.complex64)[:] = x
            _as(omi.input_c, Lb, np.complex64)[:] = x
        assert lib.execute_filter_input(m) == 0
        assert m.contents.blocknum == b + 1
        assert lib.execute_filter_output(s) == 0
        O.kqo_execute_filter_input(om)
        O.kqo_execute_filter_output(os_)
        nb = N // 2 + 1 if in_type == 3 else N
        fd_g = _as(m.contents.fdomain, nb, np.complex64)
        fd_o = _as(omi.fdomain, nb, np.complex64)
        assert np.abs(fd_g - fd_o).max() / np.abs(fd_o).max() < 2e-6          # master spectrum (radio.c:396 reads it)
        if out_type == 3:
            got, want = _as(s.contents.output, olen, np.float32), _as(oso.output_r, olen, np.float32)
        else:
            got, want = _as(s.contents.output, olen, np.complex64), _as(oso.output_c, olen, np.complex64)
        # (until the history is full the output is the leading edge of the impulse response: 1e-3 of the steady level at
        #  M = 3 L + 1, and float rounding of the transform is relative to the input)
        tol = 1e-5 if (b + 1) * Lb >= M - 1 or M <= Lb + 1 else 1e-4
        assert np.sqrt(np.mean(np.abs(got - want) ** 2)) / np.sqrt(np.mean(np.abs(want) ** 2)) < tol
    assert lib.delete_filter_output(s) == 0 and lib.delete_filter_input(m) == 0
    O.kqo_delete_filter_output(os_)
    O.kqo_delete_filter_input(om)


def test_design_entry_points_run_on_the_device(lib):
    """make_kaiser / window_filter / window_rfilter of the compat surface: the design kernels (kq_design.hip) against
    the oracle's restatement of filter.c:337-469."""
    w = np.zeros(129, np.float32)
    lib.make_kaiser.argtypes = [C.c_void_p, C.c_uint, C.c_float]
    assert lib.make_kaiser(w.ctypes.data, 129, 3.0) == 0
    np.testing.assert_allclose(w, ko.make_kaiser(129, 3.0), rtol=3e-7)
    assert w[64] == 1.0 and np.array_equal(w, w[::-1])             # exactly symmetric, centre exactly one (filter.c:354-356)
    w2 = np.zeros(64, np.float32)
    assert lib.make_kaiser(w2.ctypes.data, 64, 7.5) == 0
    np.testing.assert_allclose(w2, ko.make_kaiser(64, 7.5), rtol=3e-7)
    assert lib.make_kaiser(None, 129, 3.0) == -1
    rng = np.random.default_rng(0)
    r = (rng.standard_normal(256) + 1j * rng.standard_normal(256)).astype(np.complex64)
    r2 = r.copy()
    lib.window_filter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    assert lib.window_filter(128, 129, r.ctypes.data, 3.0) == 0
    ko.lib().kqo_window_filter(128, 129, r2.ctypes.data, 3.0)
    assert np.abs(r - r2).max() / np.abs(r2).max() < 1e-6
    h = (rng.standard_normal(129) + 1j * rng.standard_normal(129)).astype(np.complex64)
    h2 = h.copy()
    lib.window_rfilter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    assert lib.window_rfilter(128, 129, h.ctypes.data, 2.0) == 0
    ko.lib().kqo_window_rfilter(128, 129, h2.ctypes.data, 2.0)
    assert np.abs(h - h2).max() / np.abs(h2).max() < 1e-6


def test_fftw_entry_points_outside_the_filter_api(lib):
    """include/ka9q_hip_fftw.h: the FFTW names fm.c:226-283 (PL tone: 16384-point r2c), linear.c:90-317 (carrier search:
    65536-point c2c) and main.c:102-103,183-184 call outside filter.c, served by the library so that their objects link
    without libfftw3f.  Transforms against numpy's float64 ones; a response from fftwf_alloc_complex handed to
    create_filter_output and released by delete_filter_output (fm.c:56 / filter.c:271)."""
    for n in ("fftwf_alloc_real", "fftwf_alloc_complex", "fftwf_malloc"):
        getattr(lib, n).restype = C.c_void_p
        getattr(lib, n).argtypes = [C.c_size_t]
    lib.fftwf_free.argtypes = [C.c_void_p]
    lib.fftwf_plan_dft_1d.restype = C.c_void_p
    lib.fftwf_plan_dft_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    lib.fftwf_plan_dft_r2c_1d.restype = C.c_void_p
    lib.fftwf_plan_dft_r2c_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
    lib.fftwf_plan_dft_c2r_1d.restype = C.c_void_p
    lib.fftwf_plan_dft_c2r_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
    lib.fftwf_execute.argtypes = [C.c_void_p]
    lib.fftwf_destroy_plan.argtypes = [C.c_void_p]
    assert lib.fftwf_import_system_wisdom() == 1 and lib.fftwf_init_threads() == 1
    lib.fftwf_make_planner_thread_safe()
    lib.fftwf_plan_with_nthreads(4)
    rng = np.random.default_rng(12)

    def err(a, b):
        return np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2))

    # fm.c:226-228,255: r2c of (1 << 19) / 32 points, executed again and again on the same buffers
    n = 16384
    pin, pout = lib.fftwf_alloc_real(n), lib.fftwf_alloc_complex(n // 2 + 1)
    assert pin and pout and pin % 64 == 0
    plan = lib.fftwf_plan_dft_r2c_1d(n, pin, pout, 1 << 6)
    assert plan
    for _ in range(2):
        x = rng.standard_normal(n).astype(np.float32)
        _as(pin, n, np.float32)[:] = x
        lib.fftwf_execute(plan)
        assert err(_as(pout, n // 2 + 1, np.complex64), np.fft.rfft(x.astype(np.float64))) < 4e-7
    lib.fftwf_destroy_plan(plan)
    # back again: c2r (unnormalised, DC and Nyquist taken as real)
    X = np.fft.rfft(x.astype(np.float64)).astype(np.complex64)
    _as(pout, n // 2 + 1, np.complex64)[:] = X
    plan = lib.fftwf_plan_dft_c2r_1d(n, pout, pin, 1 << 6)
    lib.fftwf_execute(plan)
    assert np.abs(_as(pin, n, np.float32) / n - x).max() < 2e-6
    lib.fftwf_destroy_plan(plan)
    lib.fftwf_free(pin)
    lib.fftwf_free(pout)
    # linear.c:90-92,178: c2c forward of 65536 points (two passes through device memory); and sizes with factors 3, 5 and 7
    for n, sign in ((65536, -1), (65536, +1), (9600, -1), (48000, -1), (7 * 64, -1), (47040, +1)):
        a, b = lib.fftwf_alloc_complex(n), lib.fftwf_alloc_complex(n)
        plan = lib.fftwf_plan_dft_1d(n, a, b, sign, 1 << 6)
        assert plan
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        _as(a, n, np.complex64)[:] = x
        lib.fftwf_execute(plan)
        ref = np.fft.fft(x.astype(np.complex128)) if sign < 0 else np.fft.ifft(x.astype(np.complex128)) * n
        assert err(_as(b, n, np.complex64), ref) < 5e-7, (n, sign)
        lib.fftwf_destroy_plan(plan)
        lib.fftwf_free(a)
        lib.fftwf_free(b)
    assert not lib.fftwf_plan_dft_1d(11 * 64, lib.fftwf_alloc_complex(704), lib.fftwf_alloc_complex(704), -1, 0)   # a factor 11: refused
    # fm.c:56-66: a response allocated by fftwf_alloc_complex, owned and released by the filter (filter.c:271)
    Lb, M = 512, 513
    m = lib.create_filter_input(Lb, M, 3)
    resp = lib.fftwf_alloc_complex((Lb + M - 1) // 2 + 1)
    _as(resp, (Lb + M - 1) // 2 + 1, np.complex64)[:] = 1e-3
    s = lib.create_filter_output(m, resp, 1, 3)
    assert s and s.contents.response == resp
    assert lib.delete_filter_output(s) == 0 and lib.delete_filter_input(m) == 0
