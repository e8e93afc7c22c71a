#!/usr/bin/env python3
"""How many channels does one GPU carry at 1.0 x real time?  (BASELINE.json's "channels @ real-time"; the reference's
operating point is one `radio` per channel at the front end's rate, main.c:105, README.md:470-477.)

cfg 4 geometry (N = 16384, D = 256, 10 MS/s, FM, compute_n0 on), C channels on ONE GPU, a SMALL batch of B blocks per
call (B = 2: 1.64 ms of signal), the batch fed from pinned host memory (kq_bank_push_iq_async) and every channel's audio
+ status delivered to pinned host memory after every call (kq_bank_pull_planes_async) -- then once more delivering the
int16 PCM plane instead of float audio.  Prints, per (C, B): channel set-up time, ms per call, the real-time factor
(signal time / wall time), the host's own time inside kq_bank_process (everything that scales with C on the host: the
per-channel oscillator parameters of the call), and the D2H rate.

    python tools/realtime_probe.py --channels 8192,16384,32768 --blocks 2,4 --seconds 3
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", default="8192,32768")
    ap.add_argument("--blocks", default="2,4")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--no-io", action="store_true", help="resident input, no planes to the host (the kernels alone)")
    ap.add_argument("--pcm", action="store_true", help="deliver the int16 PCM plane instead of float audio")
    ap.add_argument("--retunes", type=int, default=0, help="channels retuned before every call (every call then stages on the host)")
    ap.add_argument("--swept", type=int, default=0, help="channels with a swept Doppler oscillator (rate != 0)")
    ap.add_argument("--rtp", type=int, default=0, help="input as RTP datagrams of this many int16 I/Q samples (kq_bank_push_rtp)")
    ap.add_argument("--control-plane", action="store_true", help="filter / mode changes and a channel leaving and returning around every call")
    ap.add_argument("--paced", action="store_true", help="batches arrive by the wall clock at the front end's rate: deadline accounting "
                    "(late deliveries, backlog) instead of a mean factor")
    ap.add_argument("--compact", action="store_true", help="with --pcm: 24-byte status records (kq_bank_pull_pcm_planes_compact_async)")
    a = ap.parse_args()
    import torch
    import ka9q_sdr_amd as kq
    from ka9q_sdr_amd import workload as wl
    from realtime_harness import measure_realtime
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for C in [int(x) for x in a.channels.split(",")]:
        for B in [int(x) for x in a.blocks.split(",")]:
            r = measure_realtime(torch, kq, wl, a.config, C, B, 0, stream, seconds=a.seconds, host_io=not a.no_io, pcm=a.pcm,
                                 retunes_per_call=a.retunes, swept_channels=a.swept, rtp_samples=a.rtp, control_plane=a.control_plane,
                                 paced=a.paced, compact_status=a.compact)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
