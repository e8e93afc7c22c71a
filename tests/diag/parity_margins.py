"""Diagnostic (not a test): per-channel parity margins HIP vs oracle at the cfg5 geometry.  Lives under tests/ because\nit drives the oracle.  Run on the GPU box: python tests/diag/parity_margins.py"""
import sys
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, '..', '..'), os.path.join(HERE, '..', '..', 'oracle'), os.path.join(HERE, '..')]
import numpy as np
import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
from common import bank_cfg, rel_rms, run_oracle
from test_gpu_parity import _run_bank
g = wl.GEOMETRY["cfg5"]
plan = wl.channel_plan("cfg5", 11, first=200)
plan[5]["doppler"] = 0.0; plan[5]["doppler_rate"] = 0.0
nblocks = 3
iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=17)
want = run_oracle(plan, g, iq, nblocks)
for mode in (kq.KQ_FWD_PRUNED, kq.KQ_FWD_FULL):
    got, _ = _run_bank(plan, g, iq, nblocks, mode, per_call=2)
    for c in range(len(plan)):
        fe = rel_rms(np.concatenate(got[c]["filt"]), np.concatenate(want[c][2]))
        ae = rel_rms(np.concatenate(got[c]["audio"][1:]), np.concatenate(want[c][0][1:]))
        print(mode, c, "filt %.2e audio %.2e" % (fe, ae), "gain", got[c]["status"][2]["agc_gain"], want[c][1][2]["agc_gain"])
