#!/usr/bin/env python3
"""Stability soak of the round-3 additions on the GPU box: (1) N = 65536 banks with compute_n0 (four sibling workgroups
per channel-block: tagged exchange words, epilogue kernel) created, run and destroyed 30 times -- device memory must come
back; (2) one such bank stepped for thousands of calls: kq_bank_sync never reports a lost sibling, n0 and audio stay finite,
and the same input block gives the same audio at the end as at the start; (3) the streaming host I/O entry points over
thousands of steps on the headline workload, pinned buffers checked against the blocking pulls at the end."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)

g = wl.GEOMETRY["cfg5"]
fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
B = 8
iq = wl.make_iq(fs, (M - 1) + B * L, seed=7)
buf = torch.from_numpy(iq).to(dev)
plan = wl.channel_plan("cfg5", 128)
free0 = torch.cuda.mem_get_info()[0]
for i in range(30):
    bank = kq.Bank(fs, L, M, D, len(plan), B, compute_n0=True)
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))
    bank.process_resident(buf.data_ptr(), B)
    bank.sync()
    bank.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("N = 65536: device memory free before / after 30 banks: %.1f / %.1f MiB" % (free0 / 2**20, free1 / 2**20))
assert free0 - free1 < 64 * 2**20, "device memory leaked"

bank = kq.Bank(fs, L, M, D, len(plan), B, compute_n0=True)
for p in plan:
    p = dict(p, doppler_rate=0.0)       # constant LO: the same block then gives the same filter output up to the LO's phase
    bank.add_channel(wl.bank_channel_config(p))
first = None
for k in range(steps):
    bank.process_resident(buf.data_ptr(), B)
    if k % 500 == 0 or k == steps - 1:
        bank.sync()                      # raises if a sibling's word never arrived
        st = [bank.status(c, B - 1) for c in (0, 63, 127)]
        assert all(np.isfinite(s["n0"]) and s["n0"] > 0 for s in st), st
        a = bank.filter_output(5, B - 1)
        assert np.all(np.isfinite(a))
        if first is None:
            first = (a.copy(), st[0]["n0"])
drift = np.sqrt(np.mean((np.abs(a) - np.abs(first[0])) ** 2) / np.mean(np.abs(first[0]) ** 2))
print("N = 65536: %d calls of %d blocks, n0 %.6e -> %.6e (smoothed), magnitude of the last block's filter output differs by %.2e "
      "(relative RMS) between the first and the last call" % (steps, B, first[1], st[0]["n0"], drift))
assert drift < 1e-5
bank.close()

# ---- streaming host I/O on the headline workload
g = wl.GEOMETRY["cfg4"]
fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
B, C = 16, 256
plan = wl.channel_plan("cfg4", C)
iq = wl.make_iq(fs, B * L, seed=11)
olen = L // D
bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True)
ref = kq.Bank(fs, L, M, D, C, B, compute_n0=True)
for p in plan:
    bank.add_channel(wl.bank_channel_config(p))
    ref.add_channel(wl.bank_channel_config(p))
pin = torch.from_numpy(iq).pin_memory()
audio = [torch.zeros(C * B * 2 * olen, dtype=torch.float32).pin_memory() for _ in range(2)]
status = [torch.zeros(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory() for _ in range(2)]
n_io = max(200, steps // 3)
bank.push_iq_async(pin.data_ptr(), B * L)
for k in range(n_io):
    assert bank.process() == B
    bank.push_iq_async(pin.data_ptr(), B * L)
    bank.pull_planes_async(audio[k & 1].data_ptr(), status[k & 1].data_ptr())
bank.host_io_wait()
for k in range(n_io):
    ref.push_iq(iq)
    assert ref.process() == B
last = audio[(n_io - 1) & 1].numpy().reshape(C, B, 2 * olen)
worst = 0.0
for c in (0, 100, 255):
    for b in (0, B - 1):
        want = ref.audio(c, b)
        worst = max(worst, float(np.abs(last[c, b, :len(want)] - want).max()))
print("host I/O: %d steps queued without a host wait; last step's audio vs the blocking path: max abs difference %.3g" % (n_io, worst))
assert worst == 0.0
bank.close()
ref.close()
# ---- N = 65536 at real time (round 6): the sibling workgroups' exchange beside other tenants' kernels -- the plane copy kernel of the
# call before and the demodulators on their own stream fill CUs while the siblings wait for each other.  cfg 5's geometry, one
# block (1.64 ms of signal) per call, paced by the clock, PCM planes + compact status out; kq_bank_host_io_wait raises if a
# sibling's word never arrived.
from realtime_harness import measure_realtime  # noqa: E402
rt_seconds = max(4.0, min(30.0, steps / 300.0))
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
r = measure_realtime(torch, kq, wl, "cfg5", 8192, 1, 0, stream, seconds=rt_seconds, pcm=True, compact_status=True, paced=True)
print("N = 65536 at real time: %d swept SSB channels x 1 block per call, paced for %.0f s: %d of %d deliveries late (worst %.3f ms), "
      "delivery intervals %s, filter pass %.3f ms mean / %.3f max of a %.3f ms period" %
      (r["channels"], r["wall_s"], r["deadline"]["late_deliveries"], r["deadline"]["deliveries"], r["deadline"]["worst_lateness_ms"],
       r["delivery_interval_ms"], r["filter_kernel_ms"], r["filter_kernel_max_ms"], r["deadline"]["period_ms"]))
assert r["check"]["nout_sum"] == 8192 * 64, r["check"]
print("soak ok")
