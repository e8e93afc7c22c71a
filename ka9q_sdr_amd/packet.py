"""Python mirror of the AFSK-1200 / HDLC packet decoder bank (include/ka9q_hip.h: kq_afsk_*; packet.c:267-414).

ctypes over libka9q_hip.so; there is no CPU path.
"""
import ctypes as C

import numpy as np

from .bank import KqError, _err, load_library

KQ_PCM_F32, KQ_PCM_S16BE = 0, 1


class AfskConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("max_sessions", C.c_uint), ("max_frames", C.c_uint), ("stream", C.c_void_p)]


class AfskState(C.Structure):
    _fields_ = [("symphase", C.c_int), ("frame_bit", C.c_int), ("flagsync", C.c_int), ("ones", C.c_int),
                ("last_val", C.c_float), ("mid_val", C.c_float), ("decoded_packets", C.c_int),
                ("pending_samples", C.c_int), ("blocks", C.c_uint64)]


def _bind(L):
    if getattr(L, "_kq_afsk_bound", False):
        return L
    L.kq_afsk_create.restype = C.c_void_p
    L.kq_afsk_create.argtypes = [C.POINTER(AfskConfig)]
    L.kq_afsk_destroy.argtypes = [C.c_void_p]
    L.kq_afsk_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_int]
    L.kq_afsk_sync.argtypes = [C.c_void_p]
    L.kq_afsk_num_frames.argtypes = [C.c_void_p, C.c_uint]
    L.kq_afsk_dropped_frames.argtypes = [C.c_void_p, C.c_uint]
    L.kq_afsk_pull_frame.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t]
    L.kq_afsk_clear_frames.argtypes = [C.c_void_p]
    L.kq_afsk_pull_filter_output.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kq_afsk_pull_state.argtypes = [C.c_void_p, C.c_uint, C.POINTER(AfskState)]
    L._kq_afsk_bound = True
    return L


class AfskBank:
    """`sessions` independent packet.c sessions decoded in lock step on one GPU."""

    def __init__(self, sessions, max_frames=64, device=0, stream=None):
        self.L = _bind(load_library())
        cfg = AfskConfig(device, sessions, max_frames, stream)
        self.h = self.L.kq_afsk_create(C.byref(cfg))
        if not self.h:
            raise KqError("kq_afsk_create: " + _err(self.L))
        self.sessions = sessions

    def close(self):
        if getattr(self, "h", None):
            self.L.kq_afsk_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc, what):
        if rc < 0:
            raise KqError(what + ": " + _err(self.L))
        return rc

    def push(self, samples):
        """samples: float32[sessions, n] (host).  Returns blocks decoded per session."""
        x = np.ascontiguousarray(samples, np.float32)
        assert x.ndim == 2 and x.shape[0] == self.sessions
        return self._chk(self.L.kq_afsk_push(self.h, x.ctypes.data, KQ_PCM_F32, self.sessions, x.shape[1], x.shape[1],
                                             0), "kq_afsk_push")

    def push_pcm_be(self, words):
        """words: big-endian int16 array [sessions, n] (dtype '>i2') or equivalent bytes per row."""
        x = np.ascontiguousarray(words, ">i2")
        assert x.ndim == 2 and x.shape[0] == self.sessions
        return self._chk(self.L.kq_afsk_push(self.h, x.ctypes.data, KQ_PCM_S16BE, self.sessions, x.shape[1],
                                             x.shape[1], 0), "kq_afsk_push")

    def push_device(self, ptr, n, stride, fmt=KQ_PCM_F32):
        return self._chk(self.L.kq_afsk_push(self.h, ptr, fmt, self.sessions, n, stride, 1), "kq_afsk_push")

    def sync(self):
        self._chk(self.L.kq_afsk_sync(self.h), "kq_afsk_sync")

    def frames(self, session):
        n = self._chk(self.L.kq_afsk_num_frames(self.h, session), "kq_afsk_num_frames")
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(n):
            ln = self._chk(self.L.kq_afsk_pull_frame(self.h, session, i, buf, 1024), "kq_afsk_pull_frame")
            out.append(bytes(buf.raw[:ln]))
        return out

    def dropped(self, session):
        return self._chk(self.L.kq_afsk_dropped_frames(self.h, session), "kq_afsk_dropped_frames")

    def clear_frames(self):
        self._chk(self.L.kq_afsk_clear_frames(self.h), "kq_afsk_clear_frames")

    def filter_output(self, session):
        out = np.empty(1000, np.complex64)
        self._chk(self.L.kq_afsk_pull_filter_output(self.h, session, out.ctypes.data, 1000), "kq_afsk_pull_filter_output")
        return out

    def state(self, session):
        st = AfskState()
        self._chk(self.L.kq_afsk_pull_state(self.h, session, C.byref(st)), "kq_afsk_pull_state")
        return {k: getattr(st, k) for k, _ in st._fields_}
