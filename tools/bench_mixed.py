#!/usr/bin/env python3
"""What the generic (mixed-radix) path costs: a bank at a front-end rate that is not 48 kHz x 2^k -- 240 kHz, N = 9600 = 4800 + 4801 - 1,
decimate 5 (radio_status.c:266) -- C channels (FM / AM / USB in turn), B blocks per call, input resident.  Prints ms per call, the
real-time factor of the front end (a block is 20 ms of signal) and, for scale, the same channel count at N = 8192 / decimate 4 /
192 kHz (the reference's default size, a power of two, same generic kernels).

    python tools/bench_mixed.py [channels [blocks [which [calls]]]]      which: 0 / 1 / 2 = one of the three sizes only (-1: all);
                                                                         calls: timed calls per line (100; a counter pass wants few)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
WHICH = int(sys.argv[3]) if len(sys.argv) > 3 else -1
CALLS = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda", 0)
for idx, (name, fs, L, M, D) in enumerate((("240 kHz, N = 9600 (2^7 3 5^2), N/D = 1920", 240000, 4800, 4801, 5),
                          ("384 kHz, N = 15360 (2^10 3 5), N/D = 1920", 384000, 7680, 7681, 8),
                          ("192 kHz, N = 8192, N/D = 2048 (powers of two, same kernels)", 192000, 3840, 4353, 4))):
    if WHICH >= 0 and idx != WHICH:
        continue
    for n0 in (True, False):
        plan = []
        for c in range(C):
            f = (c / C - 0.5) * 0.8 * fs
            if c % 3 == 0:
                plan.append(dict(demod="fm", low=-8000.0, high=8000.0, second_lo=-f))
            elif c % 3 == 1:
                plan.append(dict(demod="am", low=-5000.0, high=5000.0, second_lo=-f, recovery_rate=50.0))
            else:
                plan.append(dict(demod="linear", low=100.0, high=3000.0, second_lo=-f, hangtime=1.1, recovery_rate=6.0))
        bank = kq.Bank(fs, L, M, D, C, B, compute_n0=n0, fwd_mode=kq.KQ_FWD_FULL, pl_tone=False)
        bank.add_channels([wl.bank_channel_config(p) for p in plan])
        iq = wl.make_iq(fs, (M - 1) + B * L, seed=3)
        buf = torch.from_numpy(iq).to(dev)
        for _ in range(min(20, CALLS)):
            bank.process_resident(buf.data_ptr(), B)
        torch.cuda.synchronize()
        n = CALLS
        t0 = time.perf_counter()
        for _ in range(n):
            bank.process_resident(buf.data_ptr(), B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        bank.close()
        print("%-62s compute_n0 %d: %d channels x %d blocks: %.3f ms per call = %.1f us per channel-block, %.0f x real time, %.1f G channel-samples/s"
              % (name, n0, C, B, dt * 1e3, dt * 1e6 / (C * B), B * L / fs / dt, C * B * L / dt / 1e9), flush=True)
