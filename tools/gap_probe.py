#!/usr/bin/env python3
"""Where the microseconds between the kernels of a step go: the headline step timed with and without the pieces that put
markers / waits on the stream (kernel timing events, the fan-out's acquire / release / post).  python tools/gap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402
from ka9q_sdr_amd.shard import CFanout  # noqa: E402

g = wl.GEOMETRY["cfg4"]
L, M, D, fs = g["L"], g["M"], g["D"], g["samprate"]
C, B = 1024, 64
plan = wl.channel_plan("cfg4", C)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
nwin = (M - 1) + B * L
iq = torch.from_numpy(wl.make_iq(fs, nwin, seed=1)).to(dev)
lib = kq.load_library()


def run(label, timing, fanout, steps=200):
    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, stream=stream.cuda_stream)
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))
    fan = None
    if fanout:
        fan = CFanout(lib, 0, 0, 1, nwin)
        for i in range(2):
            fan.fill(i, iq.data_ptr())
    bank.enable_timing(timing)

    def step(k):
        if fan:
            i = k & 1
            p = fan.acquire(i, stream.cuda_stream)
            bank.process_resident(p, B)
            fan.release(i, stream.cuda_stream)
            fan.post(i)
        else:
            bank.process_resident(iq.data_ptr(), B)

    for k in range(300):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = bank.timing(reset=True) if timing else None
    print("%-44s %.4f ms/step%s" % (label, dt * 1e3, ("  kernel %.4f" % (tm["filter_ms"] / max(1, tm["filter_launches"]))) if tm else ""))
    bank.close()
    if fan:
        fan.close()


ov = os.environ.get("KQ_DEMOD_OVERLAP", "default")
for rep in range(2):
    run("overlap=%s  bare process_resident" % ov, 0, False)
    run("overlap=%s  + kernel timing events" % ov, 1, False)
    run("overlap=%s  + fan-out acquire/release/post" % ov, 0, True)
    run("overlap=%s  + both (bench.py's step)" % ov, 1, True)
