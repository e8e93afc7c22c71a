#!/usr/bin/env python3
"""What the box's host link delivers: D2H / H2D copies between device memory and pinned host memory, alone and under a
running channel-bank step (python tools/pcie_probe.py).  Context for bench.py's with_host_io row."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = torch.device("cuda", 0)
for mb in (4, 21, 64, 256):
    n = mb * (1 << 20)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    st = torch.cuda.Stream()
    for name, src, dst in (("d2h", d, h), ("h2d", h, d)):
        with torch.cuda.stream(st):
            for _ in range(3):
                dst.copy_(src, non_blocking=True)
            st.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                dst.copy_(src, non_blocking=True)
            st.synchronize()
            dt = (time.perf_counter() - t0) / reps
        print("%s %4d MB alone: %.3f ms  %.1f GB/s" % (name, mb, dt * 1e3, n / dt / 1e9))
