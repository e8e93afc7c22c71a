import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ka9q_sdr_amd as kq
from ka9q_sdr_amd import workload as wl
g = wl.GEOMETRY["cfg2"]; L, M, D, fs, C = g["L"], g["M"], g["D"], g["samprate"], g["channels"]
B = 64
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
iq = torch.from_numpy(wl.make_iq(fs, (M - 1) + B * L, seed=1)).cuda()
for pl in (False, True):
    bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, stream=stream.cuda_stream, pl_tone=pl)
    for p in wl.channel_plan("cfg2", C): bank.add_channel(wl.bank_channel_config(p))
    for _ in range(600): bank.process_resident(iq.data_ptr(), B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): bank.process_resident(iq.data_ptr(), B)
    torch.cuda.synchronize(); print("cfg2 pl_tone=%s: %.4f ms/step" % (pl, (time.perf_counter() - t0) / 400 * 1e3))
    bank.close()
