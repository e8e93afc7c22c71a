import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
from common import rel_rms, run_oracle
from test_gpu_parity import _run_bank
for name, nchan in (("cfg3", 1024), ("cfg5", 256), ("cfg1", 1)):
    g = wl.GEOMETRY[name]
    plan = wl.channel_plan(name, nchan)
    if name == "cfg1":
        plan = [dict(wl._mode_params("ssb", 31), second_lo=-(wl.emitter_freq(31, g["samprate"]) + 1.0))]
    nblocks = 4
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=0x6B61)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
    got, _ = _run_bank(plan, g, iq, nblocks, kq.KQ_FWD_FULL, compute_n0=True, per_call=nblocks)
    e0s, bbs, bbo = [], [], []
    for c, p in enumerate(plan):
        for b in range(nblocks):
            f = got[c]["filt"][b].astype(np.complex128)
            ex = float(np.sum(np.abs(f) ** 2) / (2 * len(f)))
            bbs.append(abs(got[c]["status"][b]["bb_power"] / ex - 1))
            bbo.append(abs(got[c]["status"][b]["bb_power"] / want[c][1][b]["bb_power"] - 1))
        if p["demod"] == "linear" and np.abs(want[c][0][0]).max() > 0:
            e0s.append(rel_rms(got[c]["audio"][0], want[c][0][0]))
    print(name, "first linear block audio: max %.3g median %.3g (n=%d)" % (max(e0s), float(np.median(e0s)), len(e0s)) if e0s else "",
          "| bb_power vs float64: max %.3g | vs oracle: max %.3g" % (max(bbs), max(bbo)))
