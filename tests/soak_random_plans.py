"""Diagnostic soak (not collected by pytest): many seeds of test_random_channel_plans.  python tests/soak_random_plans.py 40"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle"), HERE]
import test_gpu_parity as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(10, 10 + n):
    for mode in ("pruned", "full"):
        try:
            T.test_random_channel_plans(None, seed, mode)
        except AssertionError as e:
            bad += 1
            print("seed", seed, mode, "FAILED:", str(e)[:300])
print("soak done:", n, "seeds,", bad, "failures")
