import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
g = wl.GEOMETRY["cfg4"]; L, M, D, fs = g["L"], g["M"], g["D"], g["samprate"]
C, B = 1024, 64
plan = wl.channel_plan("cfg4", C)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, stream=stream.cuda_stream)
for p in plan: bank.add_channel(wl.bank_channel_config(p))
iq = wl.make_iq(fs, B * L, seed=1)
olen = L // D
iq_pin = torch.from_numpy(iq).pin_memory()
audio_pin = torch.empty(C * B * 2 * olen, dtype=torch.float32).pin_memory()
status_pin = torch.empty(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory()
for k in range(12):
    t0 = time.perf_counter(); bank.push_iq_async(iq_pin.data_ptr(), B * L)
    t1 = time.perf_counter(); bank.process()
    t2 = time.perf_counter(); bank.pull_planes_async(audio_pin.data_ptr(), status_pin.data_ptr())
    t3 = time.perf_counter()
    print("step %2d push %.3f ms process %.3f ms pull %.3f ms" % (k, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
bank.host_io_wait(); torch.cuda.synchronize()
