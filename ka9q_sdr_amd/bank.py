"""ctypes mirror of include/ka9q_hip.h (the channel-bank C ABI of libka9q_hip.so).

Names follow the reference's vocabulary: a *bank* holds the shared master half of the
overlap-save filter (create_filter_input, filter.c:54) and one slave + demodulator per
*channel* (what one `radio` process is in the reference, main.c:105).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "lib", "libka9q_hip.so")
_lib = None

KQ_LINEAR_DEMOD, KQ_AM_DEMOD, KQ_FM_DEMOD = 0, 1, 2
KQ_IQ_CF32, KQ_IQ_S16, KQ_IQ_S8 = 0, 1, 2
KQ_FWD_AUTO, KQ_FWD_FULL, KQ_FWD_PRUNED = 0, 1, 2
KQ_ABI_VERSION = 6        # include/ka9q_hip.h


class KqError(RuntimeError):
    pass


class BankConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("samprate", C.c_int), ("L", C.c_uint), ("M", C.c_uint),
        ("decimate", C.c_uint), ("max_channels", C.c_uint), ("max_blocks", C.c_uint),
        ("gain_factor", C.c_float), ("compute_n0", C.c_int), ("fwd_mode", C.c_int),
        ("stream", C.c_void_p), ("pl_tone_off", C.c_int),
    ]


class ChannelConfig(C.Structure):
    _fields_ = [
        ("demod_type", C.c_int), ("flat", C.c_int), ("isb", C.c_int), ("channels", C.c_int),
        ("low", C.c_float), ("high", C.c_float), ("kaiser_beta", C.c_float), ("headroom", C.c_float),
        ("hangtime", C.c_float), ("recovery_rate", C.c_float),
        ("second_lo", C.c_double), ("doppler", C.c_double), ("doppler_rate", C.c_double),
        ("shift", C.c_double), ("pll", C.c_int), ("square", C.c_int),
    ]


class ChanStatus(C.Structure):
    _fields_ = [
        ("if_power", C.c_float), ("bb_power", C.c_float), ("n0", C.c_float), ("snr", C.c_float),
        ("foffset", C.c_float), ("pdeviation", C.c_float), ("agc_gain", C.c_float),
        ("noise_gain", C.c_float), ("plfreq", C.c_float), ("cphase", C.c_float),
        ("pll_lock", C.c_int32), ("lock_count", C.c_int32),
        ("squelch_count", C.c_int32), ("hangcount", C.c_int32), ("blanked", C.c_int32), ("nout", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OutRtpState(C.Structure):
    """demod->output.rtp + output.silent"""
    _fields_ = [("ssrc", C.c_uint32), ("seq", C.c_uint16), ("timestamp", C.c_uint32), ("silent", C.c_int32),
                ("packets", C.c_int64), ("bytes", C.c_int64)]


class RtpCounters(C.Structure):
    """struct rtp_state (multicast.h:41-50) + demod->input.samples"""
    _fields_ = [("ssrc", C.c_uint32), ("next_seq", C.c_uint16), ("next_timestamp", C.c_uint32),
                ("packets", C.c_int64), ("drops", C.c_int64), ("dupes", C.c_int64), ("samples", C.c_int64)]


class HostTiming(C.Structure):
    """kq_host_timing: the host's own time inside the process calls"""
    _fields_ = [("call_ms", C.c_double), ("stage_ms", C.c_double), ("slot_wait_ms", C.c_double), ("calls", C.c_uint64),
                ("lock_wait_ms", C.c_double), ("lock_wait_max_ms", C.c_double), ("ctl_hold_max_ms", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("filter_ms", C.c_double), ("demod_ms", C.c_double), ("ingest_ms", C.c_double),
                ("filter_launches", C.c_uint64), ("channel_blocks", C.c_uint64), ("filter_max_ms", C.c_double),
                ("filter_max_submit_ms", C.c_double), ("filter_max_launch", C.c_uint64)]


# kq_chan_status_compact: aux = FM foffset / AM, linear agc_gain; state = FM squelch_count / AM, linear hangcount
COMPACT_STATUS_DTYPE = np.dtype([("bb_power", "f4"), ("n0", "f4"), ("snr", "f4"), ("aux", "f4"), ("state", "i4"), ("nout", "i4")])

STATUS_DTYPE = np.dtype([
    ("if_power", "f4"), ("bb_power", "f4"), ("n0", "f4"), ("snr", "f4"), ("foffset", "f4"),
    ("pdeviation", "f4"), ("agc_gain", "f4"), ("noise_gain", "f4"), ("plfreq", "f4"), ("cphase", "f4"), ("pll_lock", "i4"), ("lock_count", "i4"),
    ("squelch_count", "i4"), ("hangcount", "i4"), ("blanked", "i4"), ("nout", "i4")])


def library_path():
    return _LIBPATH


def build_library(force=False):
    """Compile every HIP source for gfx950 into ka9q_sdr_amd/lib/libka9q_hip.so (in-tree)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src_dir, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src_dir], stdout=subprocess.DEVNULL)
    if not os.path.exists(_LIBPATH):
        raise KqError("build did not produce " + _LIBPATH)
    return _LIBPATH


def load_library():
    """dlopen libka9q_hip.so.  Raises KqError when it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise KqError("libka9q_hip.so is missing (%s): run __graft_entry__.build() or "
                      "`make -C ka9q_sdr_amd/csrc`; the HIP library is the only compute path" % _LIBPATH)
    L = C.CDLL(_LIBPATH)
    L.kq_last_error.restype = C.c_char_p
    L.kq_version.restype = C.c_char_p
    L.kq_device_count.restype = C.c_int
    L.kq_bank_create.restype = C.c_void_p
    L.kq_bank_create.argtypes = [C.POINTER(BankConfig)]
    L.kq_bank_destroy.argtypes = [C.c_void_p]
    L.kq_bank_add_channel.argtypes = [C.c_void_p, C.POINTER(ChannelConfig)]
    L.kq_bank_add_channels.argtypes = [C.c_void_p, C.POINTER(ChannelConfig), C.c_uint, C.POINTER(C.c_int)]
    L.kq_bank_set_mode.argtypes = [C.c_void_p, C.c_int, C.POINTER(ChannelConfig)]
    L.kq_bank_remove_channel.argtypes = [C.c_void_p, C.c_int]
    L.kq_bank_channel_active.argtypes = [C.c_void_p, C.c_int]
    L.kq_bank_num_channels.argtypes = [C.c_void_p]
    L.kq_bank_num_channels.restype = C.c_uint
    L.kq_bank_set_linear_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.kq_bank_set_second_lo.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.kq_bank_set_doppler.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    L.kq_bank_set_shift.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.kq_bank_set_filter.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
    L.kq_bank_push_iq.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.kq_bank_push_zeros.argtypes = [C.c_void_p, C.c_size_t]
    L.kq_bank_set_output_ssrc.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.kq_bank_pull_rtp_audio.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kq_bank_rtp_from_planes.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
    L.kq_bank_output_rtp_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(OutRtpState)]
    L.kq_bank_push_rtp.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.kq_bank_rtp_counters.argtypes = [C.c_void_p, C.POINTER(RtpCounters)]
    L.kq_bank_blocks_ready.argtypes = [C.c_void_p]
    L.kq_bank_blocks_ready.restype = C.c_uint
    L.kq_bank_process.argtypes = [C.c_void_p]
    L.kq_bank_process_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.kq_bank_sync.argtypes = [C.c_void_p]
    L.kq_bank_join.argtypes = [C.c_void_p]
    L.kq_bank_olen.argtypes = [C.c_void_p]
    L.kq_bank_olen.restype = C.c_uint
    L.kq_bank_last_blocks.argtypes = [C.c_void_p]
    L.kq_bank_last_blocks.restype = C.c_uint
    L.kq_bank_pull_audio.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kq_bank_pull_status.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.POINTER(ChanStatus)]
    L.kq_bank_pull_filter_output.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t]
    L.kq_bank_pull_spectrum.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t]
    L.kq_bank_get_response.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.kq_bank_get_audio_response.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.kq_bank_audio_device_ptr.argtypes = [C.c_void_p]
    L.kq_bank_audio_device_ptr.restype = C.c_void_p
    L.kq_bank_status_device_ptr.argtypes = [C.c_void_p]
    L.kq_bank_status_device_ptr.restype = C.c_void_p
    L.kq_bank_enable_pcm.argtypes = [C.c_void_p, C.c_int]
    L.kq_bank_pull_pcm.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint32)]
    L.kq_bank_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.kq_bank_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing), C.c_int]
    L.kq_bank_fwd_mode.argtypes = [C.c_void_p]
    L.kq_bank_set_n0.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.kq_bank_push_iq_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.kq_bank_pull_planes_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.kq_bank_host_io_wait.argtypes = [C.c_void_p]
    L.kq_bank_pull_pcm_planes_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kq_bank_pull_pcm_planes_compact_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kq_bank_pull_wait.argtypes = [C.c_void_p, C.c_uint]
    L.kq_bank_get_host_timing.argtypes = [C.c_void_p, C.POINTER(HostTiming), C.c_int]
    L.kq_bank_worst_lock_holder.argtypes = [C.c_void_p]
    L.kq_bank_worst_lock_holder.restype = C.c_char_p
    L.kq_shard_range.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.kq_fanout_unique_id.argtypes = [C.c_void_p]
    L.kq_fanout_create.restype = C.c_void_p
    L.kq_fanout_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.kq_fanout_post.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    L.kq_fanout_acquire.restype = C.c_void_p
    L.kq_fanout_acquire.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    L.kq_fanout_release.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.kq_fanout_destroy.argtypes = [C.c_void_p]
    L.kq_fanout_stats.argtypes = [C.c_void_p, C.POINTER(FanoutInfo)]
    L.kq_fanout_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.kq_fanout_rccl_path.restype = C.c_char_p
    L.kq_host_alloc.restype = C.c_void_p
    L.kq_host_alloc.argtypes = [C.c_size_t]
    L.kq_host_free.argtypes = [C.c_void_p]
    L.kq_abi_version.restype = C.c_int
    if L.kq_abi_version() != KQ_ABI_VERSION:
        raise KqError("libka9q_hip.so has ABI revision %d, this mirror was written for %d" % (L.kq_abi_version(), KQ_ABI_VERSION))
    _lib = L
    return L


class FanoutInfo(C.Structure):
    """kq_fanout_info (include/ka9q_hip.h)"""
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("rccl_ranks", C.c_int), ("rccl_version", C.c_int),
                ("broadcasts", C.c_ulonglong), ("broadcast_ms", C.c_double),
                ("acquires", C.c_ulonglong), ("waits", C.c_ulonglong), ("wait_ms", C.c_double),
                ("waits_dropped", C.c_ulonglong)]


class HostBuffer:
    """Page-locked host memory from the library (kq_host_alloc): .ptr for the streaming entry points, .array(dtype) a numpy view."""

    def __init__(self, nbytes):
        L = load_library()
        self.nbytes = int(nbytes)
        self.ptr = L.kq_host_alloc(self.nbytes)
        if not self.ptr:
            raise KqError("kq_host_alloc: " + _err(L))

    def array(self, dtype=np.uint8):
        n = self.nbytes // np.dtype(dtype).itemsize
        return np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr)).view(dtype)[:n]

    def free(self):
        if self.ptr:
            load_library().kq_host_free(self.ptr)
            self.ptr = None


def device_count():
    return load_library().kq_device_count()


def _err(L):
    return (L.kq_last_error() or b"").decode()


def channel_config(demod_type=KQ_FM_DEMOD, low=-8000.0, high=8000.0, second_lo=0.0, flat=0, isb=0, channels=1,
                   kaiser_beta=3.0, headroom=10 ** (-15 / 20), hangtime=0.0, recovery_rate=0.0,
                   doppler=0.0, doppler_rate=0.0, shift=0.0, pll=0, square=0):
    """Defaults: main.c:113-117 (beta 3.0, headroom -15 dB); filter edges as modes.txt:25."""
    c = ChannelConfig()
    c.demod_type, c.flat, c.isb, c.channels = demod_type, flat, isb, channels
    c.low, c.high, c.kaiser_beta, c.headroom = low, high, kaiser_beta, headroom
    c.hangtime, c.recovery_rate = hangtime, recovery_rate
    c.second_lo, c.doppler, c.doppler_rate, c.shift = second_lo, doppler, doppler_rate, shift
    c.pll, c.square = pll, square
    return c


class Bank:
    """A bank of receiver channels sharing one front-end I/Q stream on one GPU."""

    def __init__(self, samprate, L, M, decimate, max_channels, max_blocks, device=0, gain_factor=1.0,
                 compute_n0=False, fwd_mode=KQ_FWD_AUTO, stream=None, pl_tone=True):
        self.lib = load_library()
        cfg = BankConfig(device, samprate, L, M, decimate, max_channels, max_blocks, gain_factor,
                         int(compute_n0), fwd_mode, stream, 0 if pl_tone else 1)
        self.h = self.lib.kq_bank_create(C.byref(cfg))
        if not self.h:
            raise KqError("kq_bank_create: " + _err(self.lib))
        self.samprate, self.L, self.M, self.D = samprate, L, M, decimate
        self.N = L + M - 1
        self.Ndec = self.N // decimate
        self.olen = self.lib.kq_bank_olen(self.h)
        self.max_blocks = max_blocks

    def close(self):
        if getattr(self, "h", None):
            self.lib.kq_bank_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc, what):
        if rc < 0:
            raise KqError("%s: %s" % (what, _err(self.lib)))
        return rc

    def add_channel(self, cfg):
        return self._chk(self.lib.kq_bank_add_channel(self.h, C.byref(cfg)), "kq_bank_add_channel")

    def add_channels(self, cfgs):
        """kq_bank_add_channels: a list of ChannelConfig in one call (responses designed once per distinct filter,
        planes uploaded in one piece); returns the channel numbers"""
        n = len(cfgs)
        arr = (ChannelConfig * n)(*cfgs)
        idx = (C.c_int * n)()
        self._chk(self.lib.kq_bank_add_channels(self.h, arr, n, idx), "kq_bank_add_channels")
        return list(idx)

    def remove_channel(self, ch):
        """fm.c:177-182 / am.c:80 / linear.c:319: end channel `ch`; its number is reused by a later add_channel"""
        self._chk(self.lib.kq_bank_remove_channel(self.h, ch), "kq_bank_remove_channel")

    def channel_active(self, ch):
        return bool(self.lib.kq_bank_channel_active(self.h, ch))

    def set_mode(self, ch, cfg):
        """radio.c:322-374: restart channel `ch`'s demodulator with the mode described by `cfg`"""
        self._chk(self.lib.kq_bank_set_mode(self.h, ch, C.byref(cfg)), "kq_bank_set_mode")

    @property
    def num_channels(self):
        return self.lib.kq_bank_num_channels(self.h)

    @property
    def fwd_mode(self):
        return self.lib.kq_bank_fwd_mode(self.h)

    def set_linear_options(self, ch, isb, channels):
        """linear.c:117-120, 291-300: demod->filter.isb / output.channels of a running linear channel"""
        self._chk(self.lib.kq_bank_set_linear_options(self.h, ch, int(isb), int(channels)), "kq_bank_set_linear_options")

    def set_second_lo(self, ch, hz):
        self._chk(self.lib.kq_bank_set_second_lo(self.h, ch, hz), "kq_bank_set_second_lo")

    def set_doppler(self, ch, hz, rate):
        self._chk(self.lib.kq_bank_set_doppler(self.h, ch, hz, rate), "kq_bank_set_doppler")

    def set_shift(self, ch, hz):
        self._chk(self.lib.kq_bank_set_shift(self.h, ch, hz), "kq_bank_set_shift")

    def set_filter(self, ch, low, high, beta):
        self._chk(self.lib.kq_bank_set_filter(self.h, ch, low, high, beta), "kq_bank_set_filter")

    def push_iq(self, iq):
        """iq: host numpy array -- complex64, or int16 / int8 of shape (n, 2)."""
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            fmt, n = KQ_IQ_CF32, iq.size
        elif iq.dtype == np.int16:
            fmt, n = KQ_IQ_S16, iq.size // 2
        elif iq.dtype == np.int8:
            fmt, n = KQ_IQ_S8, iq.size // 2
        else:
            raise TypeError("unsupported I/Q dtype %s" % iq.dtype)
        self._chk(self.lib.kq_bank_push_iq(self.h, iq.ctypes.data, n, fmt, 0), "kq_bank_push_iq")

    def set_n0(self, ch, n0):
        self._chk(self.lib.kq_bank_set_n0(self.h, ch, n0), "kq_bank_set_n0")

    def push_iq_async(self, ptr, nsamples, fmt=KQ_IQ_CF32):
        """ptr: PINNED host memory that stays unchanged until host_io_wait() (kq_bank_push_iq_async)"""
        self._chk(self.lib.kq_bank_push_iq_async(self.h, ptr, nsamples, fmt), "kq_bank_push_iq_async")

    def pull_planes_async(self, audio_ptr, status_ptr):
        """queues the copy of the last call's audio [C][max_blocks][2 olen] / status [C][max_blocks] planes to pinned host memory"""
        self._chk(self.lib.kq_bank_pull_planes_async(self.h, audio_ptr, status_ptr), "kq_bank_pull_planes_async")

    def pull_pcm_planes_async(self, pcm_ptr, mask_ptr, status_ptr, compact=False):
        """the last call's audio as clipped big-endian int16 words [C][max_blocks][2 olen], the silent-chunk masks
        [C][max_blocks] and the status plane to pinned host memory (kq_bank_pull_pcm_planes_async); compact: the status
        plane as 24-byte kq_chan_status_compact records (COMPACT_STATUS_DTYPE)"""
        if compact:
            self._chk(self.lib.kq_bank_pull_pcm_planes_compact_async(self.h, pcm_ptr, mask_ptr, status_ptr),
                      "kq_bank_pull_pcm_planes_compact_async")
        else:
            self._chk(self.lib.kq_bank_pull_pcm_planes_async(self.h, pcm_ptr, mask_ptr, status_ptr), "kq_bank_pull_pcm_planes_async")

    def pull_wait(self, lag=0):
        """block until the plane delivery queued `lag` deliveries before the newest has landed"""
        self._chk(self.lib.kq_bank_pull_wait(self.h, lag), "kq_bank_pull_wait")

    def worst_lock_holder(self):
        """the entry point that held the handle's lock longest since the last host_timing reset (ctl_hold_max_ms)"""
        return (self.lib.kq_bank_worst_lock_holder(self.h) or b"").decode()

    def host_timing(self, reset=True):
        t = HostTiming()
        self._chk(self.lib.kq_bank_get_host_timing(self.h, C.byref(t), int(reset)), "kq_bank_get_host_timing")
        return {n: getattr(t, n) for n, _ in t._fields_}

    def host_io_wait(self):
        self._chk(self.lib.kq_bank_host_io_wait(self.h), "kq_bank_host_io_wait")

    def push_iq_device(self, ptr, nsamples, fmt=KQ_IQ_CF32):
        self._chk(self.lib.kq_bank_push_iq(self.h, ptr, nsamples, fmt, 1), "kq_bank_push_iq")

    def set_output_ssrc(self, ch, ssrc):
        self._chk(self.lib.kq_bank_set_output_ssrc(self.h, ch, ssrc), "kq_bank_set_output_ssrc")

    def rtp_audio(self, ch, blk):
        """The PCM datagrams of one channel-block as audio.c would send them (needs enable_pcm); advances the state"""
        buf = C.create_string_buffer(8 * self.olen + 4096)
        used = C.c_size_t()
        self._chk(self.lib.kq_bank_pull_rtp_audio(self.h, ch, blk, buf, len(buf), C.byref(used)), "kq_bank_pull_rtp_audio")
        blob, out, pos = buf.raw[:used.value], [], 0
        while pos < len(blob):
            ln = blob[pos] | (blob[pos + 1] << 8)
            out.append(blob[pos + 2:pos + 2 + ln])
            pos += 2 + ln
        return out

    def rtp_from_planes(self, ch, blk, pcm_ptr, status_ptr):
        """the datagrams of one channel-block from PCM / status planes the host holds (kq_bank_rtp_from_planes)"""
        buf = C.create_string_buffer(8 * self.olen + 4096)
        used = C.c_size_t()
        self._chk(self.lib.kq_bank_rtp_from_planes(self.h, ch, blk, pcm_ptr, status_ptr, buf, len(buf), C.byref(used)),
                  "kq_bank_rtp_from_planes")
        blob, out, pos = buf.raw[:used.value], [], 0
        while pos < len(blob):
            ln = blob[pos] | (blob[pos + 1] << 8)
            out.append(blob[pos + 2:pos + 2 + ln])
            pos += 2 + ln
        return out

    def output_rtp_state(self, ch):
        st = OutRtpState()
        self._chk(self.lib.kq_bank_output_rtp_state(self.h, ch, C.byref(st)), "kq_bank_output_rtp_state")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def push_rtp(self, datagram):
        """One front-end datagram (RTP header + 24-byte status block + int16 / int8 I/Q); returns samples appended,
        or None when the ring has no room for it right now (nothing consumed: process, then push it again)"""
        d = bytes(datagram)
        rc = self.lib.kq_bank_push_rtp(self.h, d, len(d))
        if rc == -2:
            return None
        return self._chk(rc, "kq_bank_push_rtp")

    def rtp_counters(self):
        c = RtpCounters()
        self._chk(self.lib.kq_bank_rtp_counters(self.h, C.byref(c)), "kq_bank_rtp_counters")
        return {k: getattr(c, k) for k, _ in c._fields_}

    def push_zeros(self, n):
        self._chk(self.lib.kq_bank_push_zeros(self.h, n), "kq_bank_push_zeros")

    def blocks_ready(self):
        return self.lib.kq_bank_blocks_ready(self.h)

    def process(self):
        return self._chk(self.lib.kq_bank_process(self.h), "kq_bank_process")

    def process_resident(self, dev_ptr, nblocks):
        return self._chk(self.lib.kq_bank_process_resident(self.h, dev_ptr, nblocks), "kq_bank_process_resident")

    def join(self):
        self._chk(self.lib.kq_bank_join(self.h), "kq_bank_join")

    def sync(self):
        self._chk(self.lib.kq_bank_sync(self.h), "kq_bank_sync")

    def status(self, ch, blk):
        st = ChanStatus()
        self._chk(self.lib.kq_bank_pull_status(self.h, ch, blk, C.byref(st)), "kq_bank_pull_status")
        return st.as_dict()

    def audio(self, ch, blk):
        buf = np.zeros(2 * self.olen, np.float32)
        n = C.c_size_t()
        self._chk(self.lib.kq_bank_pull_audio(self.h, ch, blk, buf.ctypes.data, buf.size, C.byref(n)), "kq_bank_pull_audio")
        return buf[:n.value].copy()

    def enable_pcm(self, on=True):
        self._chk(self.lib.kq_bank_enable_pcm(self.h, int(on)), "kq_bank_enable_pcm")

    def pcm(self, ch, blk):
        """-> (int16 array holding big-endian words, silent-chunk mask)"""
        buf = np.zeros(2 * self.olen, np.int16)
        n = C.c_size_t()
        m = C.c_uint32()
        self._chk(self.lib.kq_bank_pull_pcm(self.h, ch, blk, buf.ctypes.data, buf.size, C.byref(n), C.byref(m)),
                  "kq_bank_pull_pcm")
        return buf[:n.value].copy(), m.value

    def filter_output(self, ch, blk):
        buf = np.zeros(self.olen, np.complex64)
        self._chk(self.lib.kq_bank_pull_filter_output(self.h, ch, blk, buf.ctypes.data, buf.size), "kq_bank_pull_filter_output")
        return buf

    def arm_spectrum(self, ch):
        buf = np.zeros(self.N, np.complex64)
        self.lib.kq_bank_pull_spectrum(self.h, ch, 0, buf.ctypes.data, buf.size)  # first call arms the capture

    def spectrum(self, ch, blk):
        buf = np.zeros(self.N, np.complex64)
        self._chk(self.lib.kq_bank_pull_spectrum(self.h, ch, blk, buf.ctypes.data, buf.size), "kq_bank_pull_spectrum")
        return buf

    def response(self, ch):
        buf = np.zeros(self.Ndec, np.complex64)
        self._chk(self.lib.kq_bank_get_response(self.h, ch, buf.ctypes.data, buf.size), "kq_bank_get_response")
        return buf

    def audio_response(self, ch):
        buf = np.zeros(self.Ndec // 2 + 1, np.complex64)
        self._chk(self.lib.kq_bank_get_audio_response(self.h, ch, buf.ctypes.data, buf.size), "kq_bank_get_audio_response")
        return buf

    def enable_timing(self, level=1):
        """0 off, 1 filter kernel only, 2 every kernel group"""
        self.lib.kq_bank_enable_timing(self.h, int(level))

    def timing(self, reset=True):
        t = Timing()
        self._chk(self.lib.kq_bank_get_timing(self.h, C.byref(t), int(reset)), "kq_bank_get_timing")
        return {n: getattr(t, n) for n, _ in t._fields_}

    def audio_device_ptr(self):
        return self.lib.kq_bank_audio_device_ptr(self.h)

    def status_device_ptr(self):
        return self.lib.kq_bank_status_device_ptr(self.h)
