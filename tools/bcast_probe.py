#!/usr/bin/env python3
"""What an RCCL-shaped kernel on the fan-out's side stream costs the step (VERDICT r3 #1b).  On one GPU RCCL moves
nothing, so the broadcast is stood in for by tools/side_probe.hip: `n` workgroups of 256 threads streaming one batch
(4.26 MB), launched exactly where kq_fanout_post puts ncclBroadcast -- behind the slot's `freed` marker, in front of its
`ready` marker, concurrent with the NEXT step's filter launch.  Prints ms/step against n, LDS footprint and residence time.
   python tools/bcast_probe.py [--config cfg4] [--priority]        (build: see side_probe.hip)"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg4")
ap.add_argument("--blocks", type=int, default=64)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--priority", action="store_true", help="side stream at high priority")
a = ap.parse_args()

probe = ctypes.CDLL(os.path.join(ROOT, "tools", "side_probe.bin"))
probe.side_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_double]
g = wl.GEOMETRY[a.config]
L, M, D, fs, C = g["L"], g["M"], g["D"], g["samprate"], g["channels"]
B = a.blocks
plan = wl.channel_plan(a.config, C)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev, priority=-1 if a.priority else 0)
torch.cuda.set_stream(stream)
nwin = (M - 1) + B * L
src = torch.from_numpy(wl.make_iq(fs, nwin, seed=1)).to(dev)
slots = [src.clone(), src.clone()]
bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, stream=stream.cuda_stream)
for p in plan:
    bank.add_channel(wl.bank_channel_config(p))
ready = [torch.cuda.Event() for _ in range(2)]
freed = [torch.cuda.Event() for _ in range(2)]
nbytes = nwin * 8


def run(nwg, lds, hold_us, steps):
    def post(i):
        if nwg:
            side.wait_event(freed[i])
            rc = probe.side_probe_launch(side.cuda_stream, src.data_ptr(), slots[i].data_ptr(), nbytes, nwg, 256, lds, hold_us)
            assert rc == 0, rc
            ready[i].record(side)

    def step(k):
        i = k & 1
        if nwg:
            stream.wait_event(ready[i])
        bank.process_resident(slots[i].data_ptr(), B)
        if nwg:
            freed[i].record(stream)
        post(i)

    for i in range(2):
        freed[i].record(stream)
        post(i)
    for k in range(400):
        step(k)
    torch.cuda.synchronize()
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    return best


for k in range(3000):      # sustained clocks first
    bank.process_resident(slots[0].data_ptr(), B)
torch.cuda.synchronize()
base = run(0, 0, 0.0, a.steps)
print("%s, %d blocks/step, side stream priority %s" % (a.config, B, "high" if a.priority else "default"))
print("no side kernel                                 %.4f ms/step" % base)
for lds, hold in ((0, 0.0), (65536, 0.0), (65536, 100.0), (65536, 400.0)):
    for nwg in (1, 2, 4, 8, 16, 32, 64):
        t = run(nwg, lds, hold, a.steps)
        print("side kernel %2d wg x 256, LDS %3d KiB, hold %3.0f us   %.4f ms/step  (%+.1f %%)" % (nwg, lds // 1024, hold, t, (t / base - 1) * 100))
again = run(0, 0, 0.0, a.steps)
print("no side kernel (again)                         %.4f ms/step" % again)
bank.close()
