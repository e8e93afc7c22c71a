"""Python mirror of the front-end half-band decimator cascade (include/ka9q_hip.h: kq_decim_*).

Mirrors what hackrf.c:260-330 does with hb3_block / hb15_block (decimate.c:108-160).  ctypes over libka9q_hip.so;
there is no CPU path.
"""
import ctypes as C

import numpy as np

from .bank import KqError, _err, load_library


class DecimConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("log_decimate", C.c_int), ("stage_threshold", C.c_int), ("offset", C.c_int),
                ("filter_atten", C.c_float), ("max_out", C.c_size_t), ("stream", C.c_void_p)]


def _bind(L):
    if getattr(L, "_kq_decim_bound", False):
        return L
    L.kq_decim_create.restype = C.c_void_p
    L.kq_decim_create.argtypes = [C.POINTER(DecimConfig)]
    L.kq_decim_destroy.argtypes = [C.c_void_p]
    L.kq_decim_set_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.kq_decim_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kq_decim_sync.argtypes = [C.c_void_p]
    L.kq_decim_reset.argtypes = [C.c_void_p]
    L._kq_decim_bound = True
    return L


class Decimator:
    """Cascade of log_decimate half-band stages with carried state (hackrf.c:211-216, 295-300)."""

    def __init__(self, log_decimate, stage_threshold=8, offset=1, filter_atten=0.0, max_out=1 << 16, device=0,
                 stream=None):
        self.L = _bind(load_library())
        cfg = DecimConfig(device, log_decimate, stage_threshold, offset, filter_atten, max_out, stream)
        self.h = self.L.kq_decim_create(C.byref(cfg))
        if not self.h:
            raise KqError("kq_decim_create: " + _err(self.L))
        self.log_decimate = log_decimate

    def close(self):
        if getattr(self, "h", None):
            self.L.kq_decim_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process(self, iq, want_s16=True):
        """iq: complex64[n_out << log_decimate] on the host -> (complex64[n_out], int16[n_out, 2] | None, energy)"""
        iq = np.ascontiguousarray(iq, np.complex64)
        n_out = len(iq) >> self.log_decimate
        if n_out << self.log_decimate != len(iq):
            raise ValueError("input length must be a multiple of the decimation ratio")
        out = np.empty(n_out, np.complex64)
        s16 = np.empty((n_out, 2), np.int16) if want_s16 else None
        energy = C.c_float(0)
        rc = self.L.kq_decim_process(self.h, iq.ctypes.data, 0, n_out, out.ctypes.data,
                                     s16.ctypes.data if want_s16 else None, C.addressof(energy))
        if rc != 0:
            raise KqError("kq_decim_process: " + _err(self.L))
        return out, s16, energy.value

    def process_device(self, in_ptr, n_out, out_ptr, s16_ptr=None, energy_ptr=None):
        """Asynchronous, device pointers (e.g. torch tensors' data_ptr()) on the handle's stream."""
        rc = self.L.kq_decim_process(self.h, in_ptr, 1, n_out, out_ptr, s16_ptr, energy_ptr)
        if rc != 0:
            raise KqError("kq_decim_process: " + _err(self.L))

    def sync(self):
        self.L.kq_decim_sync(self.h)

    def reset(self):
        self.L.kq_decim_reset(self.h)
