#!/usr/bin/env python3
"""Step time of the headline workload while channels are being retuned: K set_second_lo calls before every call
(the reference's tuning knob / Doppler thread, radio.c:290-311).  python tools/bench_retune.py [K ...]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

g = wl.GEOMETRY["cfg4"]
L, M, D, fs, C, B = g["L"], g["M"], g["D"], g["samprate"], 1024, 64
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
plan = wl.channel_plan("cfg4", C)
bank = kq.Bank(fs, L, M, D, C, B, device=0, fwd_mode=kq.KQ_FWD_AUTO, stream=stream.cuda_stream)
for p in plan:
    bank.add_channel(wl.bank_channel_config(p))
buf = torch.from_numpy(wl.make_iq(fs, (M - 1) + B * L)).to(dev)
for K in [int(x) for x in sys.argv[1:]] or [0, 1, 16, 128]:
    for k in range(3):
        bank.process_resident(buf.data_ptr(), B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for k in range(steps):
        for j in range(K):
            c = (k * K + j) % C
            bank.set_second_lo(c, plan[c]["second_lo"] + 0.25 * ((k + j) % 5))
        bank.process_resident(buf.data_ptr(), B)
    torch.cuda.synchronize()
    print("retunes per call %4d: %.3f ms per step" % (K, (time.perf_counter() - t0) / steps * 1e3))
bank.close()
