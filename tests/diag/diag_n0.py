"""Diagnostic (not collected): per-channel n0 / audio deviations of a random plan.  python tests/diag_n0.py 54 full"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "oracle"), os.path.join(HERE, "..")]
import test_gpu_parity as T  # noqa: E402
from common import rel_rms, run_oracle  # noqa: E402
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

seed, mode = int(sys.argv[1]), sys.argv[2]
g = wl.GEOMETRY["cfg3"]
rng = np.random.default_rng(1000 + seed)
plan = T._random_plan(rng, g["samprate"], 24)
nblocks = 6
iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=100 + seed)
want = run_oracle(plan, g, iq, nblocks, compute_n0=int(mode == "full"))
fwd = kq.KQ_FWD_PRUNED if mode == "pruned" else kq.KQ_FWD_FULL
got, used = T._run_bank(plan, g, iq, nblocks, fwd, compute_n0=(mode == "full"), per_call=4)
for c, p in enumerate(plan):
    auds, sts, filts = want[c]
    sk = 1 if p["demod"] == "linear" else 0
    ea = rel_rms(np.concatenate(got[c]["audio"][sk:]), np.concatenate(auds[sk:]))
    ef = rel_rms(np.concatenate(got[c]["filt"]), np.concatenate(filts))
    en = max(abs(got[c]["status"][b]["n0"] / sts[b]["n0"] - 1) for b in range(nblocks)) if mode == "full" else 0
    flag = "  <<<" if (ea > 1e-5 or en > 2e-4) else ""
    print("ch %2d %-6s audio %.2e filt %.2e n0 %.2e%s" % (c, p["demod"], ea, ef, en, flag))
    if flag:
        print("    ", {k: v for k, v in p.items()})
        for b in range(nblocks):
            print("     blk", b, "n0", got[c]["status"][b]["n0"], sts[b]["n0"], "gain", got[c]["status"][b]["agc_gain"],
                  sts[b]["agc_gain"], "hang", got[c]["status"][b]["hangcount"], sts[b]["hangcount"],
                  "a", rel_rms(got[c]["audio"][b], auds[b]))
