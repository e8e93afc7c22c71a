#!/usr/bin/env python3
"""Stability soak on the GPU box: (1) 50 banks created, run and destroyed -- device memory must come back;
(2) one bank stepped for many thousand calls on the headline workload -- audio and status stay finite and the first
channel's audio of a given input block is the same at the end as at the start (the oscillators wrap, nothing drifts)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

g = wl.GEOMETRY["cfg4"]
fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
B = 16
iq = wl.make_iq(fs, (M - 1) + B * L, seed=7)
dev = torch.device("cuda", 0)
buf = torch.from_numpy(iq).to(dev)
plan = wl.channel_plan("cfg4", 256)

free0 = torch.cuda.mem_get_info()[0]
for i in range(50):
    bank = kq.Bank(fs, L, M, D, len(plan), B, compute_n0=True)
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))
    bank.process_resident(buf.data_ptr(), B)
    bank.sync()
    bank.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("device memory free before / after 50 banks: %.1f / %.1f MiB" % (free0 / 2**20, free1 / 2**20))
assert free0 - free1 < 64 * 2**20, "device memory leaked"

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
bank = kq.Bank(fs, L, M, D, len(plan), B, compute_n0=True)
for p in plan:
    bank.add_channel(wl.bank_channel_config(p))
t0 = time.time()
first = None
for k in range(steps):
    bank.process_resident(buf.data_ptr(), B)
    if k in (4, steps - 1):
        bank.sync()
        a = np.concatenate([bank.audio(0, b) for b in range(B)])
        st = bank.status(0, B - 1)
        assert np.isfinite(a).all() and np.isfinite(st["n0"]) and np.isfinite(st["snr"]), k
        if first is None:
            first = (a, st)
        else:
            # the same input block every call; the channel's oscillator phase differs from call to call, the FM audio of
            # a steady carrier does not care: compare statistics
            print("call %d vs call 4: audio rms %.6g vs %.6g, n0 %.6g vs %.6g, snr %.4g vs %.4g" %
                  (k, float(np.sqrt(np.mean(a * a))), float(np.sqrt(np.mean(first[0] ** 2))), st["n0"], first[1]["n0"],
                   st["snr"], first[1]["snr"]))
bank.sync()
print("%d calls of %d blocks x %d channels in %.1f s" % (steps, B, len(plan), time.time() - t0))
bank.close()
print("soak ok")
