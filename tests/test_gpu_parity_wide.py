"""Parity at the sizes the bench runs, and the decision-flip count SURVEY.md 8d asks for.

* all 1024 channels of cfg 4 (the bench's default workload: FM, compute_n0 on every block, full-spectrum path), all
  1024 mixed channels of cfg 3 (256 of them once more on the pruned path) and all 256 channels of cfg 2, every channel against the oracle with the tolerances of
  test_gpu_parity._compare (filter output and audio 1e-5 relative RMS; counts, squelch / hang state, blanked samples
  exact).
* Randomised plans in bulk: the reference's algorithm holds comparisons that sit within one float rounding of their
  threshold for a few inputs (a leakage-skirt bin at compute_n0's 2 x mean cut, radio.c:414-420; `gain * amplitude >
  headroom` with the gain riding at the limit, linear.c:271) -- two correct float transforms then decide differently.
  They are counted, channel by channel, and the test fails when more than FLIP_BUDGET of the channels differ, or when
  any channel differs in something that is not one of those ties (sample counts, squelch state, filter output)."""
import numpy as np
import pytest

import ka9q_sdr_amd as kq
from common import rel_rms, run_oracle
from ka9q_sdr_amd import workload as wl
from test_gpu_parity import FILT_TOL, _compare, _compare_counting_ties, _n0_ties_are_ties, _random_plan, _run_bank

pytestmark = pytest.mark.gpu

FLIP_BUDGET = 0.004      # fraction of randomised channels allowed to differ through a threshold tie (observed: 3 of 5 760)


@pytest.mark.parametrize("name,mode,n0,nchan", [("cfg4", "full", True, 1024), ("cfg3", "full", True, 1024),
                                               ("cfg3", "pruned", False, 256), ("cfg2", "full", True, 256),
                                               ("cfg5", "pruned", False, 128), ("cfg5", "full", True, 128),
                                               ("cfg5", "full", True, 512)])
def test_channels_at_bench_geometry(gpu, name, mode, n0, nchan):
    """cfg 4, cfg 3, cfg 2 at BASELINE.json's per-GPU channel counts (1024 FM; 512 FM + 256 AM + 256 SSB; 256 FM at
    N/D = 256), every channel against the oracle; 128 of cfg 5's swept SSB channels at N = 65536 on the pruned path, and
    128 and all 512 (BASELINE.json's per-GPU share) on the full-spectrum path with compute_n0, as linear.c:123-126 runs it."""
    g = wl.GEOMETRY[name]
    plan = wl.channel_plan(name, nchan)
    nblocks = 4
    iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=0x6B61)
    want = run_oracle(plan, g, iq, nblocks, compute_n0=int(n0))
    fwd = kq.KQ_FWD_FULL if mode == "full" else kq.KQ_FWD_PRUNED
    got, used = _run_bank(plan, g, iq, nblocks, fwd, compute_n0=n0, per_call=nblocks)
    assert used == fwd
    # channel by channel, with the same allowance as test_decision_flips_stay_within_budget below: a bin that sits
    # within one float rounding of compute_n0's 2 x mean cut (radio.c:414-420) moves n0 by one bin's worth (0.2 %) in
    # a few channels per thousand; everything else of such a channel still has to agree
    flips = _compare_counting_ties(plan, got, want, nblocks, check_n0=n0)
    import conftest
    conftest.note_ties("test_channels_at_bench_geometry[%s-%s-%s-%d]" % (name, mode, n0, nchan), flips, len(plan))
    big = g["L"] + g["M"] - 1 > 16384
    # N = 65536: the float32 oracle sums its first pass over 62 000 bins in sequence and lands 1e-4 below the exact mean
    # (at 16384 bins: 3e-5), so a bin within 1e-4 of the cut is decided differently: 2 - 3 % of the channels.  Each of them
    # is then checked against float64 arithmetic on the kernel's own spectrum: the kernel's n0 is the exact one.
    budget = 0.04 if big else FLIP_BUDGET
    assert len(flips) <= budget * len(plan), flips
    if big and n0:
        _n0_ties_are_ties(plan, g, iq, nblocks, flips, 0.001)
    # the status scalars the tolerances of _compare are wide for: how close they really are at this size
    worst_if = max(abs(got[c]["status"][b]["if_power"] / want[c][1][b]["if_power"] - 1) for c in range(len(plan))
                   for b in range(nblocks))
    assert worst_if < 1e-4, worst_if      # the reference sums 8192 floats in sequence; the kernel sums them as a tree
    # ... and against float64 arithmetic on the same samples (radio.c:123,143-145: E <- (E + sum |s|^2) / 2, if_power = E / L)
    # the kernel's figure holds to 2e-6: the 1e-4 above is the float32 oracle's sequential sum
    E, L = 0.0, g["L"]
    for b in range(nblocks):
        x = iq[b * L:(b + 1) * L].astype(np.complex128)
        E = 0.5 * (E + float(np.sum(x.real ** 2 + x.imag ** 2)))
        np.testing.assert_allclose(got[0]["status"][b]["if_power"], E / L, rtol=2e-6)


def test_decision_flips_stay_within_budget(gpu):
    g = wl.GEOMETRY["cfg3"]
    nblocks, per_plan, seeds = 6, 24, range(40, 56)
    total, flips = 0, []
    for seed in seeds:
        rng = np.random.default_rng(1000 + seed)
        plan = _random_plan(rng, g["samprate"], per_plan)
        iq = wl.make_iq(g["samprate"], nblocks * g["L"], seed=100 + seed)
        for mode in ("pruned", "full"):
            n0 = mode == "full"
            want = run_oracle(plan, g, iq, nblocks, compute_n0=int(n0))
            fwd = kq.KQ_FWD_PRUNED if mode == "pruned" else kq.KQ_FWD_FULL
            got, _ = _run_bank(plan, g, iq, nblocks, fwd, compute_n0=n0, per_call=4)
            for c, p in enumerate(plan):
                total += 1
                try:
                    _compare([p], [got[c]], [want[c]], check_n0=n0)
                except AssertionError as e:
                    # a tie may move the noise estimate or a linear channel's gain track; nothing else
                    filt = rel_rms(np.concatenate(got[c]["filt"]), np.concatenate(want[c][2]))
                    same_counts = all(got[c]["status"][b]["nout"] == want[c][1][b]["nout"] and
                                      got[c]["status"][b]["squelch_count"] == want[c][1][b]["squelch_count"]
                                      for b in range(nblocks))
                    assert filt < FILT_TOL and same_counts, ("not a threshold tie", seed, mode, c, p, str(e)[:300])
                    flips.append((seed, mode, c, p["demod"], str(e)[:120]))
    assert len(flips) <= FLIP_BUDGET * total, (len(flips), total, flips)
    import conftest
    conftest.TIES.append(("test_decision_flips_stay_within_budget", len(flips), total,
                          "; ".join("seed %d %s ch %d %s" % f[:4] for f in flips[:6])))
