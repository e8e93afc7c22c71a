"""Synthetic front-end I/Q and channel plans for the BASELINE.json configurations.

Wideband stream: white Gaussian noise (sigma 1e-3 per component) plus 64 emitters spread over
90 % of the band; emitter e is FM, FM, AM or SSB by e mod 4.  Channel c listens to one emitter with
a second-LO that is deliberately not bin aligned.  Everything here is host-side numpy used by
bench.py and the tests to build inputs; it is not part of the measured path.
"""
import numpy as np

GEOMETRY = {
    # name: samprate, L, M, D, channels per GPU, modes
    "cfg1": dict(samprate=192000, L=8192, M=8193, D=4, channels=1, modes="fm"),
    "cfg2": dict(samprate=2000000, L=8192, M=8193, D=64, channels=256, modes="fm"),
    "cfg3": dict(samprate=10000000, L=8192, M=8193, D=256, channels=1024, modes="mixed"),
    "cfg4": dict(samprate=10000000, L=8192, M=8193, D=256, channels=1024, modes="fm"),   # per-GPU share of 8192
    "cfg5": dict(samprate=20000000, L=32768, M=32769, D=512, channels=512, modes="ssb_doppler"),  # per-GPU share of 4096
}

N_EMITTERS = 64
EMITTER_AMPL = 0.0125
NOISE_SIGMA = 1e-3
HEADROOM = 10 ** (-15 / 20)          # main.c:117


def emitter_freq(e, samprate):
    return (e - 31.5) * 0.9 * samprate / N_EMITTERS


def emitter_kind(e):
    return ("fm", "fm", "am", "ssb")[e % 4]


def make_iq(samprate, nsamples, seed=0x6B613971, start=0, emitters=None):
    """complex64[nsamples]: noise + emitters, evaluated at absolute sample indices start..start+n."""
    rng = np.random.default_rng(seed + start)
    n = np.arange(start, start + nsamples, dtype=np.float64)
    t = n / samprate
    x = NOISE_SIGMA * (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples))
    for e in (range(N_EMITTERS) if emitters is None else emitters):
        fe = emitter_freq(e, samprate)
        kind = emitter_kind(e)
        carrier = 2 * np.pi * fe * t
        if kind == "fm":       # 1 kHz tone, 3 kHz peak deviation
            x += EMITTER_AMPL * np.exp(1j * (carrier + 3.0 * np.sin(2 * np.pi * 1000.0 * t)))
        elif kind == "am":     # 1 kHz tone, 50 % depth
            x += EMITTER_AMPL * (1 + 0.5 * np.cos(2 * np.pi * 1000.0 * t)) * np.exp(1j * carrier)
        else:                  # two-tone SSB, upper on even SSB emitters, lower on odd
            sgn = 1.0 if (e // 4) % 2 == 0 else -1.0
            x += 0.5 * EMITTER_AMPL * (np.exp(1j * (carrier + sgn * 2 * np.pi * 700.0 * t)) +
                                       np.exp(1j * (carrier + sgn * 2 * np.pi * 1900.0 * t)))
    return x.astype(np.complex64)


def _mode_params(kind, e):
    """Filter edges and AGC constants from modes.txt:25-38."""
    if kind == "fm":
        return dict(demod="fm", low=-8000.0, high=8000.0, hangtime=0.0, recovery_rate=0.0)
    if kind == "am":
        return dict(demod="am", low=-5000.0, high=5000.0, hangtime=0.0, recovery_rate=50.0)
    if (e // 4) % 2 == 0:
        return dict(demod="linear", low=100.0, high=3000.0, hangtime=1.1, recovery_rate=6.0)    # USB
    return dict(demod="linear", low=-3000.0, high=-100.0, hangtime=1.1, recovery_rate=6.0)       # LSB


def channel_plan(name, nchan=None, first=0):
    """List of per-channel dicts for configuration `name`; channels first..first+nchan-1 of the global plan."""
    g = GEOMETRY[name]
    fs = g["samprate"]
    nchan = g["channels"] if nchan is None else nchan
    plan = []
    for c in range(first, first + nchan):
        if g["modes"] == "fm":
            e = 4 * (c % 16) + ((c // 16) % 2)
            off = c // 32
        elif g["modes"] == "ssb_doppler":
            e = 4 * (c % 16) + 3
            off = c // 16
        else:
            e = c % N_EMITTERS
            off = c // N_EMITTERS
        p = _mode_params(emitter_kind(e), e)
        p.update(emitter=e, second_lo=-(emitter_freq(e, fs) + off * 1.0), doppler=0.0, doppler_rate=0.0,
                 kaiser_beta=3.0, headroom=HEADROOM, channels=1, flat=0, isb=0, shift=0.0)
        if g["modes"] == "ssb_doppler":
            total = 4096
            sgn = 1.0 if c % 2 == 0 else -1.0
            d = sgn * (5000.0 + 25000.0 * c / total)
            p["doppler"] = d
            p["doppler_rate"] = -sgn * ((c % 7) + 1) * 20.0
            # set_doppler mixes with exp(-j*2*pi*d*t): pre-offset LO2 so the channel sits on its emitter at t = 0
            p["second_lo"] += d
        plan.append(p)
    return plan


def algorithmic_bytes(name_or_geom, demod="fm", stereo=False):
    """Bytes per channel-block: 8N (window) + 8 N_dec (response) + output [+ FM audio response]."""
    g = GEOMETRY[name_or_geom] if isinstance(name_or_geom, str) else name_or_geom
    N = g["L"] + g["M"] - 1
    ndec = N // g["D"]
    olen = g["L"] // g["D"]
    b = 8 * N + 8 * ndec + (8 if stereo else 4) * olen
    if demod == "fm":
        b += 8 * (ndec // 2 + 1)
    return b


def algorithmic_flops(name_or_geom, demod="fm", doppler=False, compute_n0=True, pruned=False):
    """Floating-point operations per channel-block (secondary row): NCO mix 8 L (+ 8 L with Doppler),
    forward transform 5 N log2 N, response multiply 6 N_dec, inverse 5 N_dec log2 N_dec, demodulator ~ 30 olen;
    compute_n0 adds 3 N for the bin powers and 2 N per pass.  The pruned forward path computes only the N_dec bins
    the slave reads: 5 N log2 N_dec + 14 N."""
    import math
    g = GEOMETRY[name_or_geom] if isinstance(name_or_geom, str) else name_or_geom
    N = g["L"] + g["M"] - 1
    ndec = N // g["D"]
    olen = g["L"] // g["D"]
    mix = 8 * N * (2 if doppler else 1)     # the kernels mix the whole N-sample window of each block
    fwd = (5 * N * math.log2(ndec) + 14 * N) if pruned else 5 * N * math.log2(N)
    f = mix + fwd + 6 * ndec + 5 * ndec * math.log2(ndec) + 30 * olen
    if compute_n0 and not pruned:
        f += 3 * N + 2 * 2 * N
    return f


def bank_channel_config(p):
    """Plan entry -> kq_channel_config for kq_bank_add_channel."""
    from . import bank as _b
    demod = {"fm": _b.KQ_FM_DEMOD, "am": _b.KQ_AM_DEMOD, "linear": _b.KQ_LINEAR_DEMOD}[p["demod"]]
    return _b.channel_config(demod_type=demod, low=p["low"], high=p["high"], second_lo=p.get("second_lo", 0.0),
                             flat=p.get("flat", 0), isb=p.get("isb", 0), channels=p.get("channels", 1),
                             kaiser_beta=p.get("kaiser_beta", 3.0), headroom=p.get("headroom", HEADROOM),
                             hangtime=p.get("hangtime", 0.0), recovery_rate=p.get("recovery_rate", 0.0),
                             doppler=p.get("doppler", 0.0), doppler_rate=p.get("doppler_rate", 0.0),
                             shift=p.get("shift", 0.0), pll=p.get("pll", 0), square=p.get("square", 0))
