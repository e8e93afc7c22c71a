"""Diagnostic (not collected): N = 65536 full-spectrum kernel against the oracle, channel by channel -- the dumped spectrum,
the oracle's compute_n0 on the GPU's own spectrum, and the status n0.  python tests/diag_n0_64k.py [swept|unswept]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "oracle"), os.path.join(HERE, "..")]
from common import bank_cfg, oracle_cfg, rel_rms  # noqa: E402
import kq_oracle as ko  # noqa: E402
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "swept"
g = wl.GEOMETRY["cfg5"]
plan = wl.channel_plan("cfg5", 9, first=300)
if variant == "unswept":
    for p in plan:
        p["second_lo"] -= p["doppler"]
        p["doppler"] = p["doppler_rate"] = 0.0
nblocks = 5
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else nblocks
fs, L = g["samprate"], g["L"]
iq = wl.make_iq(fs, nblocks * L, seed=23)
for c, p in enumerate(plan):
    ch = ko.Channel(oracle_cfg(p, fs, L, g["M"], g["D"], compute_n0=1))
    bank = kq.Bank(fs, L, g["M"], g["D"], len(plan), per_call, compute_n0=True, fwd_mode=kq.KQ_FWD_AUTO)
    for q in plan:
        bank.add_channel(bank_cfg(q))
    try:
        bank.spectrum(c, 0)
    except Exception:
        pass
    for b in range(nblocks):
        if b % per_call == 0:
            nb = min(per_call, nblocks - b)
            bank.push_iq(iq[b * L:(b + nb) * L])
            assert bank.process() == nb
        _, st, filt, spec = ch.block(iq[b * L:(b + 1) * L], want_filt=True, want_spectrum=True)
        gs = bank.spectrum(c, b % per_call)
        n0_o_on_g = ko.compute_n0(gs, fs, p["low"], p["high"])
        n0_o = ko.compute_n0(spec, fs, p["low"], p["high"])
        sg = bank.status(c, b % per_call)
        print("ch %d blk %d spec err %.2e filt err %.2e | n0 oracle(raw) %.6e oracle-on-gpu-spectrum %.6e | status gpu %.6e oracle %.6e"
              % (c, b, rel_rms(gs, spec), rel_rms(bank.filter_output(c, b % per_call), filt), n0_o, n0_o_on_g, sg["n0"], st["n0"]))
        if abs(sg["n0"] / st["n0"] - 1) > 2e-4 and b == 0:
            os.makedirs(os.path.join(HERE, "..", "gpurun_out"), exist_ok=True)
            np.save(os.path.join(HERE, "..", "gpurun_out", "spec_ch%d.npy" % c), gs)
            print("saved", c, p)
    bank.close()
    ch.close()
