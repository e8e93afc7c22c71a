"""The C-ABI library loads and exports every symbol that include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import ka9q_sdr_amd as kq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt):
        n = m.group(1)
        if n in ("defined", "sizeof") or n.startswith("__"):
            continue
        names.add(n)
    return names


@pytest.fixture(scope="module")
def lib():
    kq.build_library()
    return kq.load_library()


def test_bank_symbols_exported(lib):
    names = {n for n in _declared("ka9q_hip.h") if n.startswith("kq_")}
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), "libka9q_hip.so does not export %s" % n


def test_compat_symbols_exported(lib):
    want = {"create_filter_input", "create_filter_output", "execute_filter_input", "execute_filter_output",
            "delete_filter_input", "delete_filter_output", "set_filter", "window_filter", "window_rfilter",
            "make_kaiser", "noise_gain", "set_osc", "step_osc", "renorm_osc", "is_phasor_init",
            "csincosf", "csincospif", "csincos", "csincospi", "cnrmf", "cnrm"}
    assert want <= _declared("ka9q_hip_compat.h")
    for n in sorted(want):
        assert hasattr(lib, n), "libka9q_hip.so does not export %s" % n
    assert C.c_float.in_dll(lib, "Kaiser_beta").value == 3.0     # filter.c:279


def test_radio_thread_entry_points_exported(lib):
    """radio.h:235-237: the three thread entry points radio.c's Demodtab[] binds, plus the library's compute_n0."""
    decl = _declared("ka9q_hip_radio.h")
    for n in ("demod_fm", "demod_am", "demod_linear", "kq_compat_compute_n0"):
        assert n in decl
        assert hasattr(lib, n), "libka9q_hip.so does not export %s" % n
    # the two hand-off functions stay with the host program (audio.c:32,82): referenced weakly, never defined here
    assert {"send_mono_output", "send_stereo_output"} <= decl
    import subprocess
    syms = subprocess.run(["nm", "-D", kq.library_path()], capture_output=True, text=True, check=True).stdout
    for n in ("send_mono_output", "send_stereo_output"):
        assert re.search(r"^\s+w %s$" % n, syms, flags=re.M), "%s must be a weak undefined reference" % n
    # a NULL argument returns at once (no thread state, no device work)
    lib.demod_am.restype = C.c_void_p
    lib.demod_am.argtypes = [C.c_void_p]
    assert lib.demod_am(None) is None


def test_library_exports_exactly_its_headers(lib):
    """A library linked under an 18 kLoC C program with a flat namespace exports its headers' names and nothing else
    (csrc/exports.map): the kq_* surface of ka9q_hip.h, the reference names of filter.h:81-105, osc.h:21-24,
    dsp.h:20-31 and radio.h:235-237, and the one global of filter.c:279."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", kq.library_path()], capture_output=True, text=True,
                         check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = set()
    for h in ("ka9q_hip.h", "ka9q_hip_compat.h", "ka9q_hip_radio.h", "ka9q_hip_fftw.h"):
        declared |= _declared(h)
    declared |= {"Kaiser_beta"}
    declared -= {"send_mono_output", "send_stereo_output"}       # the host program's (weak references here)
    declared = {n for n in declared if not n.isupper()}           # macros with arguments
    stray = exported - declared
    assert not stray, "exported but declared by no header: %s" % sorted(stray)
    missing = {n for n in declared if n.startswith("kq_") or n in exported} - exported
    assert not missing, sorted(missing)
    # every function the headers declare is there (macros such as notch_delete aside)
    for n in sorted(declared - exported):
        assert n in ("notch_delete", "free", "kq_cfloat", "kq_cdouble"), "declared but not exported: %s" % n


def test_version_and_errors(lib):
    assert b"gfx950" in lib.kq_version()
    # argument validation happens before any device work
    assert lib.kq_bank_create(None) is None
    assert b"NULL" in lib.kq_last_error()
    cfg = kq.BankConfig(0, 192000, 512, 500, 4, 1, 1, 1.0, 0, 0, None)      # N = 1011: not a power of two
    assert lib.kq_bank_create(C.byref(cfg)) is None
    assert b"power of two" in lib.kq_last_error()
    cfg = kq.BankConfig(0, 192000, 512, 513, 3, 1, 1, 1.0, 0, 0, None)      # decimate must divide N
    assert lib.kq_bank_create(C.byref(cfg)) is None


def test_no_silent_cpu_fallback(lib):
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    if lib.kq_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(kq.KqError, match="no HIP device"):
        kq.Bank(192000, 512, 513, 4, 1, 1)
    lib.create_filter_input.restype = C.c_void_p
    assert lib.create_filter_input(512, 513, 1) is None


def test_sizes_are_checked_before_any_device_is_asked_for(lib):
    """N = L + M - 1 and N / decimate: powers of two, or even 2^a 3^b 5^c 7^d (round 6: the sizes FFTW takes and a front end at 240 kHz
    needs, filter.c:78,132, radio_status.c:266).  A prime factor beyond 7 is refused with a message that says so -- with or
    without a GPU in the box; a size that is served fails here only for want of a device."""
    for L_, M_, D_, ok in ((4800, 4801, 5, True), (7680, 7681, 8, True), (12000, 12001, 25, True), (448 * 8, 448 * 8 + 1, 4, True),
                           (11 * 256, 11 * 256 + 1, 4, False),
                           (4800, 4801, 7, False), (33 * 512, 33 * 512 + 1, 2, False), (960, 961, 1, True),
                           (8192, 8193, 1, False)):
        try:
            b = kq.Bank(240000, L_, M_, D_, 1, 1, compute_n0=False)
            b.close()
            assert ok and lib.kq_device_count() > 0
        except kq.KqError as e:
            if ok:
                assert "no HIP device" in str(e) and lib.kq_device_count() <= 0, str(e)
            else:
                assert "2^a 3^b 5^c 7^d" in str(e) or "divide" in str(e) or "decimate 1 needs" in str(e), str(e)


def test_host_nco_entry_points(lib):
    """set_osc/step_osc of the compat surface are host scalar code: check them against the oracle."""
    import kq_oracle as ko
    lib.set_osc.argtypes = [C.POINTER(ko.RefOsc), C.c_double, C.c_double]
    lib.step_osc.argtypes = [C.POINTER(ko.RefOsc)]
    lib.step_osc.restype = ko._Cplx
    O = ko.lib()
    a, b = ko.RefOsc(), ko.Osc()
    lib.set_osc(C.byref(a), 0.0371, 2e-10)
    O.kqo_set_osc(C.byref(b), 0.0371, 2e-10)
    for i in range(20000):
        x, y = lib.step_osc(C.byref(a)), O.kqo_step_osc(C.byref(b))
        assert abs(x.re - y.re) < 1e-11 and abs(x.im - y.im) < 1e-11
    assert a.steps == b.steps


def test_headers_are_valid_c():
    """The boundary is a C ABI: both headers must compile as plain C (gnu11, the reference's dialect)."""
    import subprocess
    import tempfile
    inc = os.path.join(ROOT, "include")
    for body in ('#include "ka9q_hip.h"\nint f(void){ kq_bank_config c = {0}; return c.device; }\n',
                 '#include "ka9q_hip_compat.h"\nint f(void){ struct osc o; struct notchfilter n; (void)o; (void)n; return 0; }\n',
                 '#include "ka9q_hip.h"\n#include "ka9q_hip_compat.h"\nint f(void){ return 0; }\n',
                 # the FFTW names of fm.c:226-228 / linear.c:90-92 / main.c:102-103, as a C host without <fftw3.h> sees them
                 '#include "ka9q_hip_fftw.h"\nint f(void){ fftwf_complex *a = fftwf_alloc_complex(8); float *r = fftwf_alloc_real(8);\n'
                 '  fftwf_plan p = fftwf_plan_dft_r2c_1d(8, r, a, FFTW_ESTIMATE); fftwf_execute(p); fftwf_destroy_plan(p);\n'
                 '  fftwf_free(a); fftwf_free(r); fftwf_make_planner_thread_safe(); return fftwf_import_system_wisdom(); }\n'):
        with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as t:
            t.write(body)
        r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", t.name],
                           capture_output=True, text=True)
        os.unlink(t.name)
        assert r.returncode == 0, r.stderr


def test_c_example_builds_against_the_library():
    """examples/radio_bank.c compiles and links against libka9q_hip.so with gcc (running it needs the GPU)."""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "ka9q_sdr_amd", "lib")
    out = os.path.join(tempfile.gettempdir(), "kq_radio_bank_example")
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "radio_bank.c"), "-L", lib, "-lka9q_hip",
                        "-Wl,-rpath," + lib, "-lm", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    os.unlink(out)
