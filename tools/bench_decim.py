"""Throughput of the front-end half-band cascade (kq_decim_*) on device-resident samples.

python tools/bench_decim.py [--log 6] [--thr 8] [--out 1048576] [--steps 20]
Prints one JSON line: input samples/s, algorithmic GB/s (8 B per input sample read + 8 B (+4 B int16) per output
sample written) and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", type=int, default=6)
    ap.add_argument("--thr", type=int, default=8)
    ap.add_argument("--out", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=400, help="untimed calls: enough (~60 ms) to be at sustained clocks")
    a = ap.parse_args()
    import torch
    from ka9q_sdr_amd import Decimator
    n_in = a.out << a.log
    stream = torch.cuda.Stream()
    x = torch.randn(n_in, 2, device="cuda", dtype=torch.float32) * 0.05
    y = torch.empty(a.out, 2, device="cuda", dtype=torch.float32)
    s16 = torch.empty(a.out, 2, device="cuda", dtype=torch.int16)
    e = torch.zeros(1, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    dec = Decimator(a.log, a.thr, 1, max_out=a.out, stream=stream.cuda_stream)
    for _ in range(a.warmup):
        dec.process_device(x.data_ptr(), a.out, y.data_ptr(), s16.data_ptr(), e.data_ptr())
    dec.sync()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    for _ in range(a.steps):
        dec.process_device(x.data_ptr(), a.out, y.data_ptr(), s16.data_ptr(), e.data_ptr())
    t1.record(stream)
    dec.sync()
    ms = t0.elapsed_time(t1) / a.steps
    alg = 8.0 * n_in + 12.0 * a.out
    print(json.dumps({"metric": "front-end samples decimated per second", "value": n_in / (ms * 1e-3),
                      "unit": "complex samples/s", "ms_per_call": ms, "log_decimate": a.log,
                      "stage_threshold": a.thr, "n_out": a.out,
                      "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0,
                                   "unit": "GB/s", "frac": alg / (ms * 1e-3) / 8e12}}))
    dec.close()


if __name__ == "__main__":
    main()
