"""Host-side logic that needs no GPU: workload plans, algorithmic byte accounting, channel sharding."""
import numpy as np

from ka9q_sdr_amd import workload as wl
from ka9q_sdr_amd.shard import shard_range


def test_algorithmic_bytes_match_survey():
    # SURVEY.md 8d
    assert wl.algorithmic_bytes("cfg2", "fm") == 134664
    assert wl.algorithmic_bytes("cfg3", "fm") == 131976
    assert wl.algorithmic_bytes("cfg3", "am") == 131712
    assert wl.algorithmic_bytes("cfg5", "linear") == 525568


def test_geometry_is_consistent():
    for name, g in wl.GEOMETRY.items():
        N = g["L"] + g["M"] - 1
        assert N & (N - 1) == 0 and N % g["D"] == 0 and g["L"] % g["D"] == 0 and (g["M"] - 1) % g["D"] == 0


def test_channel_plans():
    p3 = wl.channel_plan("cfg3")
    assert len(p3) == 1024
    kinds = [p["demod"] for p in p3[:64]]
    assert kinds.count("fm") == 32 and kinds.count("am") == 16 and kinds.count("linear") == 16
    los = [p["second_lo"] for p in p3]
    assert len(set(los)) == len(los)                       # every channel has its own LO
    N, fs = 16384, 10e6
    assert all(abs((lo * N / fs) - round(lo * N / fs)) > 1e-3 for lo in los[64:256])   # not bin aligned
    p4 = wl.channel_plan("cfg4")
    assert all(p["demod"] == "fm" for p in p4)
    p5 = wl.channel_plan("cfg5", 16)
    assert all(p["doppler"] != 0 and p["doppler_rate"] != 0 for p in p5)
    # sharded plans are slices of the global plan
    assert wl.channel_plan("cfg4", 8, first=1024) == wl.channel_plan("cfg4", 1032)[1024:]


def test_make_iq_is_deterministic_and_bounded():
    a = wl.make_iq(10_000_000, 4096, seed=3)
    b = wl.make_iq(10_000_000, 4096, seed=3)
    assert np.array_equal(a, b) and a.dtype == np.complex64
    assert np.abs(a).max() < 1.0


def test_shard_range_covers_everything_once():
    for total, world in ((8192, 8), (4096, 8), (1000, 3), (5, 8)):
        seen = []
        for r in range(world):
            first, count = shard_range(total, world, r)
            seen += list(range(first, first + count))
        assert seen == list(range(total))
