"""Shared helpers for the tests: plan -> oracle / bank configs, error metrics."""
import numpy as np

import kq_oracle as ko
import ka9q_sdr_amd as kq

_DEMOD_O = {"fm": ko.KQO_FM, "am": ko.KQO_AM, "linear": ko.KQO_LINEAR}


def oracle_cfg(p, samprate, L, M, D, compute_n0=0, gain_factor=1.0):
    return ko.make_cfg(samprate=samprate, L=L, M=M, D=D, demod_type=_DEMOD_O[p["demod"]], flat=p.get("flat", 0),
                       isb=p.get("isb", 0), channels=p.get("channels", 1), low=p["low"], high=p["high"],
                       kaiser_beta=p.get("kaiser_beta", 3.0), headroom=p.get("headroom", 10 ** (-15 / 20)),
                       hangtime=p.get("hangtime", 0.0), recovery_rate=p.get("recovery_rate", 0.0),
                       gain_factor=gain_factor, lo2_hz=p.get("second_lo", 0.0), doppler_hz=p.get("doppler", 0.0),
                       doppler_rate=p.get("doppler_rate", 0.0), shift_hz=p.get("shift", 0.0), compute_n0=compute_n0,
                       pll=p.get("pll", 0), square=p.get("square", 0))


def bank_cfg(p):
    from ka9q_sdr_amd import workload as wl
    return wl.bank_channel_config(p)


def rel_rms(a, b):
    """RMS of (a-b) relative to RMS of b (the reference side)."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.sqrt(np.mean(np.abs(b) ** 2))
    num = np.sqrt(np.mean(np.abs(a - b) ** 2))
    return num / den if den > 0 else num


def run_oracle(plan, geom, iq, nblocks, compute_n0=0, gain_factor=1.0):
    """-> per channel: (audio [nblocks] list, status list, filt list)"""
    L = geom["L"]
    out = []
    for p in plan:
        cfg = oracle_cfg(p, geom["samprate"], L, geom["M"], geom["D"], compute_n0, gain_factor)
        out.append(ko.run_chain(cfg, iq[:nblocks * L].reshape(nblocks, L), want_filt=True))
    return out
