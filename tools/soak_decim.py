#!/usr/bin/env python3
"""Soak of the half-band cascade's in-kernel output energy (tagged per-workgroup words polled by the edge workgroup):
thousands of device-resident calls of random sizes, the energy of every call against the sum over the call's own output
samples, kq_decim_sync after every call (it reports a workgroup whose energy never arrived).
    python tools/soak_decim.py [calls]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ka9q_sdr_amd import Decimator  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(7)
worst = 0.0
for log_dec, max_out in ((4, 400_000), (6, 100_000), (3, 300_000)):
    dec = Decimator(log_dec, 8, 1, max_out=max_out, stream=torch.cuda.current_stream().cuda_stream)
    x = torch.randn(max_out << log_dec, 2, device="cuda") * 0.05
    y = torch.empty(max_out, 2, device="cuda")
    e = torch.zeros(1, device="cuda")
    for k in range(calls):
        n_out = int(rng.integers(1, max_out + 1)) if k % 8 else int(rng.integers(1, 600))
        dec.process_device(x.data_ptr(), n_out, y.data_ptr(), None, e.data_ptr())
        dec.sync()                                   # raises if a workgroup's share went missing
        want = float((y[:n_out].double() ** 2).sum().item())
        got = float(e.item())
        rel = abs(got - want) / max(want, 1e-30)
        worst = max(worst, rel)
        assert rel < 2e-6, (log_dec, k, n_out, got, want)
    dec.close()
    print("log_decimate %d: %d calls of 1..%d outputs, energy within %.1e of the sum over the call's samples" % (log_dec, calls, max_out, worst))
print("soak ok")
