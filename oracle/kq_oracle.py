"""ctypes binding of the CPU oracle (oracle/libkq_oracle.so) and of oracle/_ref/libref_osc.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under ka9q_sdr_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

KQO_NONE, KQO_COMPLEX, KQO_CROSS_CONJ, KQO_REAL = 0, 1, 2, 3
KQO_LINEAR, KQO_AM, KQO_FM = 0, 1, 2


class ChanCfg(C.Structure):
    _fields_ = [
        ("samprate", C.c_int), ("L", C.c_uint), ("M", C.c_uint), ("D", C.c_uint),
        ("demod_type", C.c_int), ("flat", C.c_int), ("isb", C.c_int), ("channels", C.c_int),
        ("low", C.c_float), ("high", C.c_float), ("kaiser_beta", C.c_float),
        ("headroom", C.c_float), ("hangtime", C.c_float), ("recovery_rate", C.c_float),
        ("gain_factor", C.c_float),
        ("lo2_hz", C.c_double), ("doppler_hz", C.c_double), ("doppler_rate", C.c_double),
        ("shift_hz", C.c_double), ("compute_n0", C.c_int), ("pll", C.c_int), ("square", C.c_int),
    ]


class Status(C.Structure):
    _fields_ = [
        ("if_power", C.c_float), ("bb_power", C.c_float), ("n0", C.c_float), ("snr", C.c_float),
        ("foffset", C.c_float), ("pdeviation", C.c_float), ("agc_gain", C.c_float), ("plfreq", C.c_float),
        ("cphase", C.c_float), ("pll_lock", C.c_int), ("lock_count", C.c_int),
        ("squelch_count", C.c_int), ("hangcount", C.c_int), ("blanked", C.c_int), ("nout", C.c_int),
        ("samples", C.c_longlong),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Osc(C.Structure):
    """kqo_osc (oracle restatement)"""
    _fields_ = [("freq", C.c_double), ("rate", C.c_double),
                ("phasor", C.c_double * 2), ("phasor_step", C.c_double * 2),
                ("phasor_step_step", C.c_double * 2), ("steps", C.c_int)]


class RefOsc(C.Structure):
    """struct osc of the reference (osc.h:9-17), x86-64 glibc layout: pthread_mutex_t is 40 bytes."""
    _fields_ = [("freq", C.c_double), ("rate", C.c_double),
                ("phasor", C.c_double * 2), ("phasor_step", C.c_double * 2),
                ("phasor_step_step", C.c_double * 2),
                ("mutex", C.c_byte * 40), ("steps", C.c_int)]


class _Cplx(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libkq_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("kq_fft.c", "kq_osc.c", "kq_filter.c", "kq_chan.c", "kq_decimate.c", "kq_packet.c", "kq_rtp.c",
                                             "kq_oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libkq_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_osc.so")
    ref_dec = os.path.join(_HERE, "_ref", "libref_ax25.so")
    if os.path.exists("/root/reference/osc.c") and (force or not os.path.exists(ref_so) or not os.path.exists(ref_dec)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    L.kqo_chan_create.restype = C.c_void_p
    L.kqo_chan_create.argtypes = [C.POINTER(ChanCfg)]
    L.kqo_chan_destroy.argtypes = [C.c_void_p]
    L.kqo_chan_block.argtypes = [C.c_void_p, fp, fp, C.POINTER(Status), fp, fp]
    L.kqo_chan_block_i16.argtypes = [C.c_void_p, C.POINTER(C.c_int16), fp, C.POINTER(Status)]
    L.kqo_chan_block_i8.argtypes = [C.c_void_p, C.POINTER(C.c_int8), fp, C.POINTER(Status)]
    L.kqo_chan_prime_history.argtypes = [C.c_void_p, fp]
    L.kqo_chan_prime_history.restype = None
    L.kqo_chan_zero_fill.argtypes = [C.c_void_p, C.c_int, fp, C.POINTER(Status)]
    L.kqo_chan_zero_fill.restype = C.c_int
    L.kqo_chan_olen.argtypes = [C.c_void_p]
    L.kqo_chan_olen.restype = C.c_uint
    L.kqo_chan_noise_gain.argtypes = [C.c_void_p]
    L.kqo_chan_noise_gain.restype = C.c_float
    L.kqo_chan_response.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    L.kqo_chan_response.restype = C.c_void_p
    L.kqo_chan_audio_response.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    L.kqo_chan_audio_response.restype = C.c_void_p
    L.kqo_chan_set_lo2.argtypes = [C.c_void_p, C.c_double]
    L.kqo_chan_set_mode.argtypes = [C.c_void_p, C.c_void_p]
    L.kqo_chan_set_shift.argtypes = [C.c_void_p, C.c_double]
    L.kqo_chan_set_filter.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    L.kqo_chan_set_doppler.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.kqo_compute_n0.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_float, C.c_float]
    L.kqo_compute_n0.restype = C.c_float
    L.kqo_make_kaiser.argtypes = [fp, C.c_uint, C.c_float]
    L.kqo_window_filter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.kqo_window_rfilter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.kqo_fft_create.restype = C.c_void_p
    L.kqo_fft_create.argtypes = [C.c_uint]
    L.kqo_fft_destroy.argtypes = [C.c_void_p]
    L.kqo_fft_c2c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.kqo_fft_r2c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kqo_fft_c2r.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kqo_set_osc.argtypes = [C.POINTER(Osc), C.c_double, C.c_double]
    L.kqo_step_osc.argtypes = [C.POINTER(Osc)]
    L.kqo_step_osc.restype = _Cplx
    L.kqo_create_filter_input.restype = C.c_void_p
    L.kqo_create_filter_input.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.kqo_create_filter_output.restype = C.c_void_p
    L.kqo_create_filter_output.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int]
    L.kqo_execute_filter_input.argtypes = [C.c_void_p]
    L.kqo_execute_filter_output.argtypes = [C.c_void_p]
    L.kqo_delete_filter_input.argtypes = [C.c_void_p]
    L.kqo_delete_filter_output.argtypes = [C.c_void_p]
    L.kqo_set_filter.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    L.kqo_notch_create.restype = C.c_void_p
    L.kqo_notch_create.argtypes = [C.c_double, C.c_float]
    L.kqo_notch_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.kqo_chan_push_raw.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, fp, C.POINTER(Status)]
    L.kqo_iq_packet.argtypes = [C.POINTER(IqIngest), C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 4
    L.kqo_crc_good.argtypes = [C.c_char_p, C.c_int]
    L.kqo_afsk_create.restype = C.c_void_p
    L.kqo_afsk_destroy.argtypes = [C.c_void_p]
    L.kqo_afsk_push.argtypes = [C.c_void_p, fp, C.c_int]
    L.kqo_afsk_push_pcm_be.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.kqo_afsk_nframes.argtypes = [C.c_void_p]
    L.kqo_afsk_frame.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.kqo_afsk_filter_output.argtypes = [C.c_void_p]
    L.kqo_afsk_filter_output.restype = C.c_void_p
    L.kqo_afsk_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4 + [fp, fp]
    L.kqo_hb15_init.argtypes = [C.POINTER(Hb15State)]
    L.kqo_hb15_block.argtypes = [C.POINTER(Hb15State), fp, fp, C.c_int]
    L.kqo_hb3_block.argtypes = [fp, fp, fp, C.c_int]
    L.kqo_pcm_block.argtypes = [fp, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]
    L.kqo_pcm_block.restype = C.c_int
    L.kqo_pcm_rtp.argtypes = [C.POINTER(OutRtp), fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_int)]
    L.kqo_bench_channels.restype = C.c_double
    L.kqo_bench_channels.argtypes = [C.POINTER(ChanCfg), C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_double)]
    L.kqo_fft_set_fast.argtypes = [C.c_int]
    _LIB = L
    return L


def ref_osc_lib():
    """The reference's own osc.c+dsp.c (oracle/_ref/libref_osc.so); None if not built."""
    global _REF
    if _REF is not None:
        return _REF
    build()
    so = os.path.join(_HERE, "_ref", "libref_osc.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    R.set_osc.argtypes = [C.POINTER(RefOsc), C.c_double, C.c_double]
    R.step_osc.argtypes = [C.POINTER(RefOsc)]
    R.step_osc.restype = _Cplx
    R.renorm_osc.argtypes = [C.POINTER(RefOsc)]
    R.is_phasor_init.argtypes = [_Cplx]
    R.is_phasor_init.restype = C.c_int
    _REF = R
    return R


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_cfg(**kw):
    """ChanCfg with the reference's defaults (main.c:113-117, modes.txt:25-38)."""
    d = dict(samprate=192000, L=8192, M=8193, D=4, demod_type=KQO_FM, flat=0, isb=0, channels=1,
             low=-8000.0, high=8000.0, kaiser_beta=3.0, headroom=10 ** (-15 / 20), hangtime=0.0,
             recovery_rate=0.0, gain_factor=1.0, lo2_hz=0.0, doppler_hz=0.0, doppler_rate=0.0,
             shift_hz=0.0, compute_n0=1, pll=0, square=0)
    d.update(kw)
    c = ChanCfg()
    for k, v in d.items():
        setattr(c, k, v)
    return c


class Channel:
    """One oracle receiver channel (kqo_chan)."""

    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.h = self.L.kqo_chan_create(C.byref(cfg))
        if not self.h:
            raise ValueError("oracle: unsupported geometry (FFT sizes must be of the form 2^a 3^b 5^c 7^d)")
        self.olen = self.L.kqo_chan_olen(self.h)
        self.N = cfg.L + cfg.M - 1

    def close(self):
        if self.h:
            self.L.kqo_chan_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def block(self, iq, want_filt=False, want_spectrum=False):
        """iq: complex64[L] -> (audio float32[nout], status dict, filt complex64[olen]|None, spectrum|None)"""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        assert iq.shape == (self.cfg.L,)
        audio = np.zeros(2 * self.olen, np.float32)
        filt = np.zeros(self.olen, np.complex64) if want_filt else None
        spec = np.zeros(self.N, np.complex64) if want_spectrum else None
        st = Status()
        self.L.kqo_chan_block(self.h, _fp(iq.view(np.float32)), _fp(audio), C.byref(st),
                              _fp(filt.view(np.float32)) if want_filt else None,
                              _fp(spec.view(np.float32)) if want_spectrum else None)
        return audio[:st.nout].copy(), st.as_dict(), filt, spec

    def block_i16(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        audio = np.zeros(2 * self.olen, np.float32)
        st = Status()
        self.L.kqo_chan_block_i16(self.h, iq.ctypes.data_as(C.POINTER(C.c_int16)), _fp(audio), C.byref(st))
        return audio[:st.nout].copy(), st.as_dict()

    def block_i8(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        audio = np.zeros(2 * self.olen, np.float32)
        st = Status()
        self.L.kqo_chan_block_i8(self.h, iq.ctypes.data_as(C.POINTER(C.c_int8)), _fp(audio), C.byref(st))
        return audio[:st.nout].copy(), st.as_dict()

    def push_raw(self, raw, count, fmt):
        """One packet payload (bytes; fmt 1 = int16, 2 = int8 interleaved I/Q) -> list of (audio, status) of the blocks
        it completes"""
        nb = count // self.cfg.L + 2
        audio = np.zeros(nb * 2 * self.olen, np.float32)
        sts = (Status * nb)()
        buf = bytes(raw)
        done = self.L.kqo_chan_push_raw(self.h, buf, count, fmt, _fp(audio), sts)
        return [(audio[b * 2 * self.olen: b * 2 * self.olen + sts[b].nout].copy(), sts[b].as_dict()) for b in range(done)]

    def prime_history(self, iq):
        """the master's M-1 history samples for a channel created while the master runs (complex64, oldest first)"""
        iq = np.ascontiguousarray(iq, np.complex64)
        assert len(iq) == self.cfg.M - 1
        self.L.kqo_chan_prime_history(self.h, _fp(iq.view(np.float32)))

    def zero_fill(self, count):
        nb = count // self.cfg.L + 2
        audio = np.zeros(nb * 2 * self.olen, np.float32)
        sts = (Status * nb)()
        done = self.L.kqo_chan_zero_fill(self.h, count, _fp(audio), sts)
        out = []
        for b in range(done):
            out.append((audio[b * 2 * self.olen: b * 2 * self.olen + sts[b].nout].copy(), sts[b].as_dict()))
        return out

    def set_lo2(self, hz):
        self.L.kqo_chan_set_lo2(self.h, hz)

    def set_mode(self, cfg):
        """radio.c:322-374 with the fields of `cfg` (a ChanCfg) as the mode table entry"""
        assert self.L.kqo_chan_set_mode(self.h, C.byref(cfg)) == 0
        self.olen = self.L.kqo_chan_olen(self.h)

    def set_shift(self, hz):
        self.L.kqo_chan_set_shift(self.h, hz)

    def set_filter(self, low, high, beta):
        self.L.kqo_chan_set_filter(self.h, low, high, beta)

    def set_doppler(self, hz, rate):
        self.L.kqo_chan_set_doppler(self.h, hz, rate)

    def response(self):
        n = C.c_uint()
        p = self.L.kqo_chan_response(self.h, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (2 * n.value,)).view(np.complex64).copy()

    def audio_response(self):
        n = C.c_uint()
        p = self.L.kqo_chan_audio_response(self.h, C.byref(n))
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (2 * n.value,)).view(np.complex64).copy()

    def noise_gain(self):
        return self.L.kqo_chan_noise_gain(self.h)


def run_chain(cfg, iq_blocks, want_filt=False):
    """Run nblocks through one channel; returns (audio list, status list, filt list)."""
    ch = Channel(cfg)
    auds, sts, filts = [], [], []
    for blk in iq_blocks:
        a, s, f, _ = ch.block(blk, want_filt=want_filt)
        auds.append(a)
        sts.append(s)
        filts.append(f)
    ch.close()
    return auds, sts, filts


def fft_c2c(x, sign=-1):
    L = lib()
    x = np.ascontiguousarray(x, np.complex64)
    p = L.kqo_fft_create(len(x))
    out = np.zeros_like(x)
    L.kqo_fft_c2c(p, x.ctypes.data, out.ctypes.data, sign)
    L.kqo_fft_destroy(p)
    return out


def make_kaiser(M, beta):
    w = np.zeros(M, np.float32)
    lib().kqo_make_kaiser(_fp(w), M, beta)
    return w


def compute_n0(spec, samprate, low, high):
    spec = np.ascontiguousarray(spec, np.complex64)
    return lib().kqo_compute_n0(spec.ctypes.data, len(spec), samprate, low, high)


class OutRtp(C.Structure):
    """demod->output.rtp + output.silent (audio.c:32-132)"""
    _fields_ = [("ssrc", C.c_uint32), ("seq", C.c_uint16), ("timestamp", C.c_uint32), ("silent", C.c_int),
                ("packets", C.c_longlong), ("bytes", C.c_longlong)]

    def packetize(self, audio, stereo):
        """-> list of datagrams send_mono_output / send_stereo_output would send for this audio block"""
        a = np.ascontiguousarray(audio, np.float32)
        buf = C.create_string_buffer(4 * len(a) + 4096)
        used = C.c_int()
        n = lib().kqo_pcm_rtp(C.byref(self), _fp(a), len(a), int(stereo), C.cast(buf, C.POINTER(C.c_ubyte)), len(buf),
                              C.byref(used))
        assert n >= 0
        return split_packets(buf.raw[:used.value])


def split_packets(blob):
    out, pos = [], 0
    while pos < len(blob):
        ln = blob[pos] | (blob[pos + 1] << 8)
        out.append(blob[pos + 2:pos + 2 + ln])
        pos += 2 + ln
    return out


class RtpState(C.Structure):
    _fields_ = [("ssrc", C.c_uint32), ("init", C.c_int), ("seq", C.c_uint16), ("timestamp", C.c_uint32),
                ("packets", C.c_longlong), ("drops", C.c_longlong), ("dupes", C.c_longlong)]


class IqIngest(C.Structure):
    """radio's packet bookkeeping: struct rtp_state + demod->input.samples"""
    _fields_ = [("rtp", RtpState), ("samples", C.c_longlong)]

    def packet(self, data):
        """-> None when ignored / dropped, else (zeros, offset, count, fmt) as kqo_iq_packet reports them"""
        z, o, n, f = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        data = bytes(data)
        ok = lib().kqo_iq_packet(C.byref(self), data, len(data), C.byref(z), C.byref(o), C.byref(n), C.byref(f))
        return (z.value, o.value, n.value, f.value) if ok else None


class Afsk:
    """packet.c session: REAL master (1000/1049), analytic slave, mark/space correlators, HDLC."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.kqo_afsk_create()

    def close(self):
        if self.h:
            self.L.kqo_afsk_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def push(self, samples):
        x = np.ascontiguousarray(samples, np.float32)
        self.L.kqo_afsk_push(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), len(x))

    def push_pcm_be(self, raw):
        raw = bytes(raw)
        self.L.kqo_afsk_push_pcm_be(self.h, raw, len(raw) // 2)

    def frames(self):
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(self.L.kqo_afsk_nframes(self.h)):
            n = self.L.kqo_afsk_frame(self.h, i, buf, 1024)
            out.append(bytes(buf.raw[:n]))
        return out

    def filter_output(self):
        p = self.L.kqo_afsk_filter_output(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2000,)).view(np.complex64).copy()

    def state(self):
        i = [C.c_int() for _ in range(4)]
        f = [C.c_float() for _ in range(2)]
        self.L.kqo_afsk_state(self.h, *[C.byref(v) for v in i], *[C.byref(v) for v in f])
        return dict(symphase=i[0].value, frame_bit=i[1].value, flagsync=i[2].value, ones=i[3].value,
                    last_val=f[0].value, mid_val=f[1].value)


def notch_run(f, bw, x):
    """filter.c:549-571 over a complex64 array (fresh filter state)."""
    L = lib()
    nf = L.kqo_notch_create(f, bw)
    x = np.ascontiguousarray(x, np.complex64)
    y = np.empty_like(x)
    L.kqo_notch_run(nf, x.ctypes.data, y.ctypes.data, len(x))
    C.CDLL(None).free(C.c_void_p(nf))
    return y


def crc_good(frame):
    return int(lib().kqo_crc_good(bytes(frame), len(frame)))


def ref_ax25_lib():
    """The reference's own ax25.c (oracle/_ref/libref_ax25.so); None if not built."""
    build()
    so = os.path.join(_HERE, "_ref", "libref_ax25.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    R.crc_good.argtypes = [C.c_char_p, C.c_int]
    return R


class Hb15State(C.Structure):
    """struct hb15_state (decimate.h:4-9) == kqo_hb15_state"""
    _fields_ = [("coeffs", C.c_float * 4), ("even_samples", C.c_float * 4), ("odd_samples", C.c_float * 4),
                ("old_odd_samples", C.c_float * 4)]


def ref_decimate_lib():
    """The reference's own decimate.c (oracle/_ref/libref_decimate.so); None if not built."""
    build()
    so = os.path.join(_HERE, "_ref", "libref_decimate.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    R.hb15_block.argtypes = [C.POINTER(Hb15State), fp, fp, C.c_int]
    R.hb3_block.argtypes = [fp, fp, fp, C.c_int]
    return R


def halfband_cascade(x, log_decimate, stage_threshold, states=None, use_ref=False):
    """hackrf.c:295-300 on one real channel: stages j = log_decimate-1 .. stage_threshold use hb3, the rest hb15.
    x: float32[n * 2**log_decimate] -> float32[n].  states = (hb15 list, hb3 array) carried across calls."""
    L = ref_decimate_lib() if use_ref else lib()
    fp = C.POINTER(C.c_float)
    if states is None:
        h15 = [Hb15State() for _ in range(log_decimate)]
        for h in h15:
            lib().kqo_hb15_init(C.byref(h))
        states = (h15, np.zeros(log_decimate, np.float32))
    h15, h3 = states
    work = np.ascontiguousarray(x, np.float32).copy()
    n_out = len(work) >> log_decimate
    for j in range(log_decimate - 1, -1, -1):
        cnt = (1 << j) * n_out
        out = np.zeros(cnt, np.float32)
        if j >= stage_threshold:
            st = h3[j:j + 1]
            (L.hb3_block if use_ref else L.kqo_hb3_block)(st.ctypes.data_as(fp), out.ctypes.data_as(fp), work.ctypes.data_as(fp), cnt)
        else:
            (L.hb15_block if use_ref else L.kqo_hb15_block)(C.byref(h15[j]), out.ctypes.data_as(fp), work.ctypes.data_as(fp), cnt)
        work = out
    return work, states


class FrontEndDecimator:
    """hackrf.c:260-330 for one stream: Fs/4 rotation, half-band cascade on I and Q, Filter_atten, int16, energy."""

    def __init__(self, log_decimate, stage_threshold=8, offset=1, filter_atten=None, use_ref=False):
        self.log, self.thr, self.offset, self.use_ref = log_decimate, stage_threshold, offset, use_ref
        self.atten = np.float32(filter_atten if filter_atten else np.float32(0.5) ** np.float32(log_decimate))
        self.phase = 0
        self.st = [None, None]

    def process(self, iq):
        iq = np.asarray(iq, np.complex64)
        n = len(iq)
        ph = (self.phase + np.arange(n) * self.offset) & 3
        self.phase = int((self.phase + n * self.offset) & 3)
        re, im = iq.real.copy(), iq.imag.copy()
        # hackrf.c:272-289
        wr = np.select([ph == 0, ph == 1, ph == 2, ph == 3], [re, -im, -re, im]).astype(np.float32)
        wi = np.select([ph == 0, ph == 1, ph == 2, ph == 3], [im, re, -im, -re]).astype(np.float32)
        outs = []
        for k, w in enumerate((wr, wi)):
            y, self.st[k] = halfband_cascade(w, self.log, self.thr, self.st[k], self.use_ref)
            outs.append((y * self.atten).astype(np.float32))
        energy = np.float32(0)
        for y in outs:  # hackrf.c:308,325 accumulate real block then imaginary block, in float
            energy = np.float32(energy + np.sum((y * y).astype(np.float32), dtype=np.float64))
        v = [np.float32(32767) * y for y in outs]
        s16 = [(np.sign(t) * np.floor(np.abs(t.astype(np.float64)) + 0.5)).astype(np.int64).astype(np.int16) for t in v]
        return (outs[0] + 1j * outs[1]).astype(np.complex64), np.stack(s16, axis=1), float(energy)


def pcm_block(audio):
    """float32 audio -> (int16 big-endian words as a raw int16 array, silent-chunk mask, chunk count)"""
    audio = np.ascontiguousarray(audio, np.float32)
    out = np.zeros(len(audio), np.int16)
    mask = C.c_uint32()
    n = lib().kqo_pcm_block(_fp(audio), len(audio), out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(mask))
    return out, mask.value, n


def cpu_baseline(cfgs, iq, nblocks_avail, warm, timed, nthreads, fast_fft=True):
    """Times nchan oracle channels over nthreads host threads: `timed` blocks per channel after `warm` untimed ones,
    channel set-up outside the clock; returns (seconds, checksum)."""
    arr = (ChanCfg * len(cfgs))(*cfgs)
    iq = np.ascontiguousarray(iq, np.complex64)
    cs = C.c_double()
    t = lib().kqo_bench_channels(arr, len(cfgs), _fp(iq.view(np.float32)), nblocks_avail, warm, timed, nthreads,
                                 int(fast_fft), C.byref(cs))
    return t, cs.value
