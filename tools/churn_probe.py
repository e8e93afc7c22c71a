#!/usr/bin/env python3
"""What the control plane costs a bank that runs at real time: kq_bank_add_channel / remove_channel / set_filter / set_mode /
set_second_lo on a bank of C channels with calls in flight (host time of the call itself, and the length of the process
call that follows it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B = 2
g = wl.GEOMETRY["cfg4"]
fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
plan = wl.channel_plan("cfg4", C)
bank = kq.Bank(fs, L, M, D, C + 16, B, compute_n0=True, pl_tone=False)
bank.add_channels([wl.bank_channel_config(p) for p in plan])
iq = torch.from_numpy(wl.make_iq(fs, (M - 1) + B * L, seed=3)).to("cuda")


def run(n):
    for _ in range(n):
        bank.process_resident(iq.data_ptr(), B)


def timed(name, fn, reps=5):
    ts, nxt = [], []
    for _ in range(reps):
        run(6)                      # calls in flight
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        bank.process_resident(iq.data_ptr(), B)
        t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3)
        nxt.append((t2 - t1) * 1e3)
        bank.sync()
    print("%-28s host %.3f ms (min %.3f)   next process call %.3f ms" % (name, np.median(ts), min(ts), np.median(nxt)))


run(50)
bank.sync()
t0 = time.perf_counter()
run(200)
bank.sync()
print("%d channels x %d blocks: %.4f ms per call" % (C, B, (time.perf_counter() - t0) / 200 * 1e3))
timed("nothing", lambda: None)
timed("set_second_lo", lambda: bank.set_second_lo(5, plan[5]["second_lo"] + 3.0))
timed("set_filter", lambda: bank.set_filter(7, -7000.0, 7000.0, 3.0))
state = {}
timed("add_channel", lambda: state.setdefault("ch", []).append(bank.add_channel(wl.bank_channel_config(plan[9]))))
timed("remove_channel", lambda: bank.remove_channel(state["ch"].pop()))
timed("set_mode", lambda: bank.set_mode(11, wl.bank_channel_config(dict(plan[11], demod="am", low=-5000.0, high=5000.0, recovery_rate=50.0))))


def paced(name, fn, n=300):
    """n calls back to back with fn() in front of each: what the operation costs the PIPELINE (the device's pace)"""
    run(20)
    bank.sync()
    t0 = time.perf_counter()
    for k in range(n):
        fn(k)
        bank.process_resident(iq.data_ptr(), B)
    bank.sync()
    print("%-28s %.4f ms per call with one in front of every call" % (name, (time.perf_counter() - t0) / n * 1e3))


paced("nothing", lambda k: None)
paced("set_second_lo", lambda k: bank.set_second_lo(5 + k % 100, plan[5 + k % 100]["second_lo"] + 3.0))
paced("set_filter", lambda k: bank.set_filter(7 + k % 100, -7000.0 - k % 7, 7000.0, 3.0))
paced("set_filter (same edges)", lambda k: bank.set_filter(7 + k % 100, -7000.0, 7000.0, 3.0))
paced("set_n0", lambda k: bank.set_n0(7 + k % 100, 1e-9))
paced("set_mode", lambda k: bank.set_mode(200 + k % 100, wl.bank_channel_config(plan[200 + k % 100])))
bank.close()
