#!/usr/bin/env python3
"""Soak of the channels-at-real-time operating point with a busy control plane: a bank of C FM / AM / SSB channels at cfg 4's
geometry, two blocks per call, input from pinned host memory and PCM planes back every call -- while, between the calls, channels
are retuned (second LO, Doppler with and without a rate, shift), filters changed, modes switched, channels removed and added
again.  Checks as it goes: the library never reports an error, every delivered channel-block of a live channel carries its
samples (nout), the FM channels that sit on their emitters keep their squelch open, status words stay finite, device memory does
not creep, and the real-time factor over the run.  A second bank, drained around every change and after every call, runs the
same script for the first `--check-calls` calls: the soaked bank's planes must equal its planes bit for bit.

    python tools/soak_realtime.py --channels 16384 --seconds 30
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ka9q_sdr_amd as kq  # noqa: E402
from ka9q_sdr_amd import workload as wl  # noqa: E402
from ka9q_sdr_amd.bank import STATUS_DTYPE  # noqa: E402


ONLY = None     # --only: restrict the script to one kind of operation (where does the time go?)
KINDS = {"retune": ("set_second_lo", "set_doppler", "set_shift"), "filter": ("set_filter",), "mode": ("set_mode",),
         "churn": ("remove_channel", "add_channel"), "none": ()}


def script(rng, plan, C, k, live, holes, added_at):
    """the control-plane operations in front of call k: a list of (name, args) -- the same for both banks"""
    import heapq
    keep = None if ONLY is None else KINDS[ONLY]
    ok = lambda name: keep is None or name in keep
    ops = []
    for _ in range(int(rng.integers(0, 4))):
        c = int(rng.integers(0, C))
        if c not in live:
            continue
        what = int(rng.integers(0, 10))
        p = plan[c]
        if what < 4:
            op = ("set_second_lo", (c, p["second_lo"] + float(rng.integers(-3, 4))))
        elif what < 6:
            op = ("set_doppler", (c, float(rng.integers(-50, 51)), float(rng.choice([0.0, 0.0, -30.0, 45.0]))))
        elif what < 7:
            op = ("set_shift", (c, float(rng.choice([0.0, 150.0]))))
        elif what < 8:
            op = ("set_filter", (c, p["low"] * float(rng.uniform(0.7, 1.0)), p["high"] * float(rng.uniform(0.7, 1.0)), 3.0))
        elif what < 9:
            op = ("remove_channel", (c,))
        else:
            op = ("set_mode", (c,))
        if not ok(op[0]):
            continue
        if op[0] == "remove_channel":
            live.discard(c)
            heapq.heappush(holes, c)
        ops.append(op)
    if holes and rng.integers(0, 3) == 0 and ok("add_channel"):
        c = heapq.heappop(holes)                 # the bank hands out the lowest hole
        ops.append(("add_channel", (c,)))
        live.add(c)
        added_at[c] = k
    return ops


OP_S = {}       # host seconds inside the control-plane calls of the soaked bank, by name: (count, total, worst)


def apply(bank, plan, ops, clock=False):
    for name, args in ops:
        t = time.perf_counter()
        if name == "add_channel":
            got = bank.add_channel(wl.bank_channel_config(plan[args[0]]))
            assert got == args[0], (got, args)
        elif name == "set_mode":
            bank.set_mode(args[0], wl.bank_channel_config(plan[args[0]]))
        else:
            getattr(bank, name)(*args)
        if clock:
            t = time.perf_counter() - t
            n, tot, worst = OP_S.get(name, (0, 0.0, 0.0))
            OP_S[name] = (n + 1, tot + t, max(worst, t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=16384)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--check-calls", type=int, default=300)
    ap.add_argument("--window", type=int, default=0, help="print the pace every this many calls")
    ap.add_argument("--only", default=None, choices=["retune", "filter", "mode", "churn", "none"])
    a = ap.parse_args()
    global ONLY
    ONLY = a.only
    C, B = a.channels, 2
    g = wl.GEOMETRY["cfg3"]
    fs, L, M, D = g["samprate"], g["L"], g["M"], g["D"]
    olen = L // D
    plan = wl.channel_plan("cfg3", C)
    iq = wl.make_iq(fs, B * L, seed=5)
    iq_pin = torch.from_numpy(iq.copy()).pin_memory()
    free0 = None
    banks = []
    for drained in (False, True):
        bank = kq.Bank(fs, L, M, D, C, B, compute_n0=True, pl_tone=False)
        bank.add_channels([wl.bank_channel_config(p) for p in plan])
        banks.append(bank)
    nbuf = 3
    pcm = [[torch.zeros(C * B * 2 * olen, dtype=torch.int16).pin_memory() for _ in range(nbuf)] for _ in banks]
    stat = [[torch.zeros(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory() for _ in range(nbuf)] for _ in banks]
    rngs = [np.random.default_rng(99), np.random.default_rng(99)]
    lives = [set(range(C)), set(range(C))]
    holes = [[], []]
    added = [{}, {}]
    for bank in banks:
        bank.push_iq_async(iq_pin.data_ptr(), B * L)
    fm_on = np.array([p["demod"] == "fm" for p in plan])
    t0 = time.perf_counter()
    k = nops = 0
    swept = set()          # channels whose Doppler carries a rate right now (they run the filter kernel's per-sample oscillator variant)
    t_win, k_win = time.perf_counter(), 0
    signal_s = B * L / fs
    while True:
        for i, bank in enumerate(banks):
            if i == 1 and k >= a.check_calls:
                continue
            ops = script(rngs[i], plan, C, k, lives[i], holes[i], added[i])
            if i == 0:
                nops += len(ops)
                for name, args in ops:
                    if name == "set_doppler":
                        (swept.add if args[2] != 0.0 else swept.discard)(args[0])
                    elif name in ("remove_channel", "add_channel", "set_mode"):
                        swept.discard(args[0]) if name != "set_mode" else None
            if i == 1:
                bank.sync()
            apply(bank, plan, ops, clock=(i == 0 and k >= max(200, a.check_calls)))
            assert bank.process() == B
            bank.push_iq_async(iq_pin.data_ptr(), B * L)
            j = k % nbuf
            bank.pull_pcm_planes_async(pcm[i][j].data_ptr(), None, stat[i][j].data_ptr())
            if i == 1:
                bank.sync()
            else:
                bank.pull_wait(2)
        if k >= 2:
            j = (k - 2) % nbuf           # delivery k - 2 of the soaked bank is in hand
            st = np.frombuffer(stat[0][j].numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B)
            if k % 50 == 0 or k < a.check_calls:
                live_now = np.zeros(C, bool)
                # (channels removed during the last two calls still show their last blocks: check the ones live throughout)
                live_now[list(lives[0])] = True
                assert np.all(st["nout"][live_now] >= olen), k
                assert np.all(np.isfinite(st["bb_power"][live_now])), k
            if k < a.check_calls and k >= 2:
                banks[1].host_io_wait()
                s1 = np.frombuffer(stat[1][j].numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B)
                both = sorted(c for c in lives[0] if added[0].get(c, -10) < k - 2)   # live since before call k - 2
                w0 = pcm[0][j].numpy().reshape(C, B, 2 * olen)[both]
                w1 = pcm[1][j].numpy().reshape(C, B, 2 * olen)[both]
                # (compared two calls late, so a channel removed meanwhile is simply skipped)
                same = np.array_equal(st["nout"][both], s1["nout"][both]) and np.array_equal(w0[:, :, :olen], w1[:, :, :olen])
                assert same, ("the soaked bank differs from the drained one at delivery", k - 2)
        k += 1
        if a.window and k % a.window == 0:
            now = time.perf_counter()
            print("  calls %6d-%6d: %.4f ms per call, %d swept channels" % (k_win, k, (now - t_win) / (k - k_win) * 1e3, len(swept)), flush=True)
            t_win, k_win = now, k
        if k == max(200, a.check_calls):      # the pace is taken once the twin bank and the host's comparisons are out of the loop
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            t0 = time.perf_counter()
            k0 = k
            banks[0].host_timing(reset=True)
            banks[0].enable_timing(1)
            banks[0].timing(reset=True)
        if k > max(200, a.check_calls) and time.perf_counter() - t0 > a.seconds:
            break
    banks[0].host_io_wait()
    banks[0].sync()
    dt = (time.perf_counter() - t0) / (k - k0)
    free1 = torch.cuda.mem_get_info()[0]
    ht, tm = banks[0].host_timing(), banks[0].timing()
    print("host inside the process calls: %.4f ms per call (staging %.4f, waiting for a slot %.4f); filter launches %.4f ms per call" %
          (ht["call_ms"] / max(1, ht["calls"]), ht["stage_ms"] / max(1, ht["calls"]), ht["slot_wait_ms"] / max(1, ht["calls"]),
           tm["filter_ms"] / max(1, tm["filter_launches"])))
    for name, (n, tot, worst) in sorted(OP_S.items()):
        print("  %-16s %6d calls, host %.4f ms each (worst %.3f)" % (name, n, tot / n * 1e3, worst * 1e3))
    st = np.frombuffer(stat[0][(k - 1) % nbuf].numpy().tobytes(), dtype=STATUS_DTYPE).reshape(C, B)
    live_fm = np.array(sorted(c for c in lives[0] if fm_on[c]))
    print("soak: %d channels x %d blocks, %d calls, %d control-plane operations between them: %.4f ms per call = %.3f x real time; "
          "%d of %d live FM channels with the squelch open at the end; device memory %+.1f MiB" %
          (C, B, k, nops, dt * 1e3, signal_s / dt, int((st["squelch_count"][live_fm, -1] < 2).sum()), len(live_fm),
           (free0 - free1) / 2 ** 20))
    assert abs(free0 - free1) < 64 * 2 ** 20
    for bank in banks:
        bank.close()
    print("soak ok")


if __name__ == "__main__":
    main()
