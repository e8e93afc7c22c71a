#!/usr/bin/env python3
"""Which half of the streaming host I/O costs the step what: the same steps with the input pushed from pinned host memory or
from device memory, and with the planes pulled to pinned host memory or not.   python tools/hostio_split.py [--config cfg2]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--blocks", type=int, default=64)
a = ap.parse_args()
g = wl.GEOMETRY[a.config]
L, M, D, fs, C = g["L"], g["M"], g["D"], g["samprate"], g["channels"]
B = a.blocks
plan = wl.channel_plan(a.config, C)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
iq = wl.make_iq(fs, B * L, seed=3)
iq_pin = torch.from_numpy(iq).pin_memory()
iq_dev = torch.from_numpy(iq).to(dev)
olen = L // D
audio_pin = torch.zeros(C * B * 2 * olen, dtype=torch.float32).pin_memory()
status_pin = torch.empty(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory()
# the same copy kernel with DEVICE memory as its destination: its workgroups and loads without the link
audio_dev = torch.zeros(C * B * 2 * olen, dtype=torch.float32, device=dev)
status_dev = torch.empty(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8, device=dev)


def run(push, pull, steps=400):
    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, stream=stream.cuda_stream, pl_tone=a.config == "cfg1")
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))

    def feed():
        if push == "host":
            bank.push_iq_async(iq_pin.data_ptr(), B * L)
        else:
            bank.push_iq_device(iq_dev.data_ptr(), B * L)

    def step():
        assert bank.process() == B
        feed()
        if pull == "dev":
            bank.pull_planes_async(audio_dev.data_ptr(), status_dev.data_ptr())
        elif pull:
            bank.pull_planes_async(audio_pin.data_ptr(), status_pin.data_ptr())

    feed()
    for _ in range(600):
        step()
    bank.host_io_wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    bank.host_io_wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    bank.close()
    return dt


for push in ("device", "host"):
    for pull in (False, "dev", True):
        what = {False: "left on the device", "dev": "copied to device memory", True: "to pinned host memory"}[pull]
        print("%s: input from %-6s memory, planes %-24s %.4f ms/step" % (a.config, push, what, run(push, pull)))
