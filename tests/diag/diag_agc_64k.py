"""Diagnostic (not collected): cfg 5 on the N = 65536 full-spectrum kernel, channels whose AGC state differs from the
oracle's -- per block gain / hang counter and the sample where the two gain tracks part.  python tests/diag_agc_64k.py 128"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "oracle"), os.path.join(HERE, "..")]
from common import rel_rms, run_oracle  # noqa: E402
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402
from test_gpu_parity import _run_bank  # noqa: E402

nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = wl.GEOMETRY["cfg5"]
plan = wl.channel_plan("cfg5", nchan)
nblocks = 4
iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=0x6B61)
want = run_oracle(plan, g, iq, nblocks, compute_n0=1)
for mode, n0 in ((kq.KQ_FWD_FULL, True), (kq.KQ_FWD_PRUNED, False)):
    got, _ = _run_bank(plan, g, iq, nblocks, mode, compute_n0=n0, per_call=nblocks)
    worst_f = 0
    for c in range(nchan):
        worst_f = max(worst_f, rel_rms(np.concatenate(got[c]["filt"]), np.concatenate(want[c][2])))
        bad = [b for b in range(nblocks) if got[c]["status"][b]["hangcount"] != want[c][1][b]["hangcount"]]
        if not bad:
            continue
        print("mode", mode, "ch", c, plan[c]["low"], plan[c]["high"], "blocks", bad)
        for b in range(nblocks):
            sg, sw = got[c]["status"][b], want[c][1][b]
            fg, fw = got[c]["filt"][b], want[c][2][b]
            ag, aw = got[c]["audio"][b], want[c][0][b]
            # the output is Re(s * gain): gain per sample from audio / Re(filt) where that is well conditioned
            k = np.argmax(np.abs(fw.real))
            print("   blk %d gain %.9g %.9g hang %d %d filt %.2e audio %.2e |s|max gpu %.9g oracle %.9g at %d %d" %
                  (b, sg["agc_gain"], sw["agc_gain"], sg["hangcount"], sw["hangcount"], rel_rms(fg, fw), rel_rms(ag, aw),
                   np.abs(fg).max(), np.abs(fw).max(), np.abs(fg).argmax(), np.abs(fw).argmax()))
    print("mode", mode, "worst filter error", worst_f)
