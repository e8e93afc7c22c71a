#!/usr/bin/env python3
"""Phase timeline of k_filter_full16k from a -DKQ_TIMELINE build (see kq_full16k.hip):

  make -C ka9q_sdr_amd/csrc clean && make -C ka9q_sdr_amd/csrc EXTRA=-DKQ_TIMELINE
  python tools/timeline.py            # on the GPU box; runs a few bench steps, prints shader cycles per phase

Per phase: mean over the sampled workgroups of the slowest wave's stamp difference, and of wave 0's."""
import ctypes
import os
import runpy
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--spinup", "100", "--no-cpu-baseline", "--no-second-row", "--no-rows", "--no-host-io"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
import ka9q_sdr_amd as kq

lib = kq.load_library()
buf = np.zeros((64, 8, 12), dtype=np.uint64)
lib.kq_debug_timeline.argtypes = [ctypes.c_void_p]
assert lib.kq_debug_timeline(buf.ctypes.data) == 0
names = ["load + mix", "pass 1 + twiddles", "transpose 1", "pass 2 + twiddles", "transpose 2", "pass 3", "slave bins + compute_n0",
         "last barrier", "epilogue (wave 0)"]
t = buf.astype(np.int64)
ok = t[:, 0, 0] > 0
t = t[ok]
print("workgroups sampled:", len(t))
start = t[:, :, 0].min(axis=1)                       # first wave in
for i, n in enumerate(names):
    if i == 8:
        d = t[:, 0, 9] - t[:, 0, 8]
        print("%-28s wave 0: %7.0f" % (n, d.mean()))
        continue
    d = t[:, :, i + 1] - t[:, :, i]
    print("%-28s mean of waves: %7.0f   slowest wave: %7.0f   fastest: %7.0f" % (n, d.mean(), d.max(axis=1).mean(), d.min(axis=1).mean()))
for a, b_, n in ((0, 10, "  start -> loads issued"), (10, 11, "  -> oscillator table, P_t"), (11, 1, "  -> all loads landed, mixed")):
    d = t[:, :, b_] - t[:, :, a]
    print("%-28s mean of waves: %7.0f   slowest wave: %7.0f   fastest: %7.0f" % (n, d.mean(), d.max(axis=1).mean(), d.min(axis=1).mean()))
print("workgroup lifetime (first stamp to wave 0's last): %.0f cycles" % (t[:, 0, 9] - start).mean())
print("spread of the waves' first stamps within a workgroup: %.0f cycles" % (t[:, :, 0].max(axis=1) - start).mean())
