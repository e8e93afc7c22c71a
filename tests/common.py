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


def n0_float64(spec, samprate, low, high):
    """compute_n0 (radio.c:383-425) in float64 on a given N-point spectrum: bin powers from the float32 bins, the
    reference's int-wrapped k * samprate in the passband test (radio.c:407,409), mean of the bins outside the passband,
    then mean of those below twice that.  Exact-arithmetic yardstick for threshold ties: the float32 oracle sums 65536
    bins in sequence and is 1e-4 off this in the first-pass mean, the GPU kernel sums as a tree."""
    spec = np.asarray(spec, np.complex64)
    N = len(spec)
    p = spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2      # cnrmf in float, as the reference forms it
    n = np.arange(N, dtype=np.int64)
    k = np.where(n <= N // 2, n, n - N)
    prod = ((k * int(samprate)) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    f = prod.astype(np.float32) / np.float32(N)
    out = ~((f >= np.float32(low)) & (f <= np.float32(high)))
    pw = p[out].astype(np.float64)
    thr = 2.0 * pw.mean()
    return float(pw[pw < thr].mean() / (2.0 * N * samprate))


# ---- AX.25 / AFSK-1200 test signal (Bell 202: mark 1200 Hz, space 2200 Hz, NRZI, HDLC bit stuffing) ----
def ax25_fcs(payload):
    """CRC-16/X.25 of the frame body; appended low byte first, the receiver's residue is 0xf0b8 (ax25.c:138-156)."""
    crc = 0xFFFF
    for byte in payload:
        for i in range(8):
            fb = 0x8408 if (crc ^ (byte >> i)) & 1 else 0
            crc = (crc >> 1) ^ fb
    crc ^= 0xFFFF
    return bytes([crc & 0xFF, crc >> 8])


def afsk_bits(frames, lead_flags=8, gap_flags=3):
    """HDLC bit stream: flags, each frame (+FCS) LSB first with zero stuffing, flags in between and at the end."""
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    bits = flag * lead_flags
    for body in frames:
        ones = 0
        for byte in body + ax25_fcs(body):
            for i in range(8):
                b = (byte >> i) & 1
                bits.append(b)
                ones = ones + 1 if b else 0
                if ones == 5:
                    bits.append(0)
                    ones = 0
        bits += flag * gap_flags
    return bits


def afsk_audio(bits, samprate=48000.0, baud=1200.0, amp=0.5, noise=0.0, seed=0, clock_ppm=0.0):
    """NRZI + continuous-phase FSK; float32 audio."""
    import numpy as np
    spb = samprate / baud * (1 + clock_ppm * 1e-6)
    n = int(len(bits) * spb) + 1
    tone = np.empty(len(bits), np.int8)
    cur = 0
    for i, b in enumerate(bits):
        if b == 0:
            cur ^= 1
        tone[i] = cur
    idx = np.minimum((np.arange(n) / spb).astype(np.int64), len(bits) - 1)
    f = np.where(tone[idx] == 0, 1200.0, 2200.0)
    ph = 2 * np.pi * np.cumsum(f) / samprate
    x = amp * np.sin(ph)
    if noise:
        x = x + noise * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)
