"""Channels at 1.0 x real time on one GPU, MEASURED (BASELINE.json's "channels @ real-time").  A measurement harness, not
part of the product package: it lives beside bench.py.

The reference's operating point is one `radio` process per channel, each at the front end's rate (main.c:105,
README.md:470-477).  Here: C channels of one bank, a small batch of B blocks per call (B = 2 at cfg 4's geometry: 1.64 ms
of signal), every batch fed from pinned host memory and every channel's audio + status handed back to pinned host memory
after every call.  What scales with C rather than with C x B is exercised here and nowhere else: the per-call oscillator
parameters of every channel, the demodulator grid, the status plane, several GB/s of audio leaving the GPU.

Two ways of running the receiver's loop:

* throughput (paced=False): batch after batch as fast as the bank takes them; realtime_factor = signal time / wall time.
  A MEAN: it says the bank keeps up on average, not that every delivery is on time.
* paced (paced=True): the front end delivers a batch every `period` = B L / samprate seconds and not sooner (the loop waits
  for the wall clock, as a socket reader waits for main.c:288-365's packets).  Iteration n starts when batch n is complete
  (A_n = origin + (n + 1) period): it processes the batch pushed one iteration earlier, pushes batch n, queues the call's
  delivery and takes the delivery queued two iterations earlier.  Deadline accounting per iteration:
      start lag   = (moment the iteration starts) - A_n            0 while the loop keeps up
      backlog     = floor(start lag / period)                      whole batches waiting behind the one being taken
      late        = delivery n - 2 in hand later than A_n + period (more than one call period behind its schedule)
  plus the intervals between consecutive deliveries.  A count "holds real time" here only with ZERO late deliveries.

Python's cyclic garbage collector is frozen and switched off inside the timed loops (gc.freeze / gc.disable): with torch
imported a full collection stops the interpreter for 20-30 ms -- the unexplained maximum delivery intervals of round 5's
float-audio legs (`gc` in the result says how the run was made, `KQ_RT_GC=1` leaves it on for an A/B).

Used by bench.py (the `realtime` object of the N = 1 line), tools/realtime_probe.py and tests/test_gpu_realtime.py.
"""
import ctypes
import gc
import os
import time

import numpy as np


def build_bank(kq, wl, config, C, B, dev_index, stream, compute_n0=True):
    """A bank of C channels of `config`'s plan with room for B blocks per call; returns (bank, plan, set-up seconds)."""
    geom = dict(wl.GEOMETRY[config])
    L, M, D, fs = geom["L"], geom["M"], geom["D"], geom["samprate"]
    plan = wl.channel_plan(config, C)
    t0 = time.perf_counter()
    bank = kq.Bank(fs, L, M, D, C, B, device=dev_index, compute_n0=compute_n0, fwd_mode=kq.KQ_FWD_AUTO,
                   stream=stream.cuda_stream if stream is not None else None, pl_tone=False)
    bank.add_channels([wl.bank_channel_config(p) for p in plan])
    bank.sync()
    return bank, plan, time.perf_counter() - t0


def _pct(a, q):
    return round(float(np.percentile(a, q)), 4)


def measure_realtime(torch, kq, wl, config, C, B, dev_index, stream, seconds=10.0, host_io=True, pcm=False, warm_calls=50,
                     retunes_per_call=0, swept_channels=0, rtp_samples=0, control_plane=False, compute_n0=True, paced=False,
                     compact_status=False):
    """C channels, B blocks per call, for `seconds` of wall time.  host_io: input from pinned host memory, audio (float, or
    the int16 PCM words when pcm) + status planes to pinned host memory after every call into one of three buffer sets; the
    host waits for the planes of call k-2 once it has queued call k (it never runs more than two deliveries ahead of what it
    has in hand, so a buffer set is never overwritten before it has landed).  retunes_per_call > 0: that many channels get
    a new second LO before every call (kq_bank_set_second_lo, 1 Hz to and fro) -- every call then stages all channels'
    oscillators on the host and carries them over the link, the path a call without retunes skips.  rtp_samples > 0: the
    input arrives as the reference's front end sends it -- RTP datagrams of that many int16 I/Q samples (payload type 97
    behind the 24-byte status block, main.c:318-341), one kq_bank_push_rtp per datagram -- instead of a float batch."""
    geom = dict(wl.GEOMETRY[config])
    L, M, D, fs = geom["L"], geom["M"], geom["D"], geom["samprate"]
    olen = L // D
    bank, plan, setup_s = build_bank(kq, wl, config, C, B, dev_index, stream, compute_n0=compute_n0)
    for i in range(swept_channels):      # satellite passes: a Doppler offset with a rate (radio.c:180-184) on some channels
        c = (i * 7919) % C
        bank.set_doppler(c, 2000.0 + i, -40.0 - (i % 7))
        bank.set_second_lo(c, plan[c]["second_lo"] + 2000.0 + i)
    dev = torch.device("cuda", dev_index)
    nwin = (M - 1) + B * L
    iq_host = wl.make_iq(fs, nwin, seed=0x6B613971)
    out_bytes = 0
    if host_io:
        iq_pin = torch.from_numpy(np.ascontiguousarray(iq_host[M - 1:])).pin_memory()
        nbuf = 3
        if pcm:
            outs = [torch.zeros(C * B * 2 * olen, dtype=torch.int16).pin_memory() for _ in range(nbuf)]
            masks = [torch.zeros(C * B, dtype=torch.int32).pin_memory() for _ in range(nbuf)]
        else:
            outs = [torch.zeros(C * B * 2 * olen, dtype=torch.float32).pin_memory() for _ in range(nbuf)]
        stats = [torch.zeros(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
        nout_words = sum((2 if p.get("channels", 1) == 2 else 1) * olen for p in plan) * B
        status_bytes = C * B * (kq.bank.COMPACT_STATUS_DTYPE.itemsize if (compact_status and pcm) else ctypes.sizeof(kq.ChanStatus))
        out_bytes = nout_words * (2 if pcm else 4) + status_bytes + (4 * C * B if pcm else 0)

        pkts = []
        if rtp_samples:      # the batch as datagrams: header patched per call (sequence number, timestamp), payload fixed
            import struct
            scale = 0.9 * 32767.0 / float(np.abs(np.concatenate([iq_host.real, iq_host.imag])).max())
            new = iq_host[M - 1:]
            i16 = np.stack([np.round(new.real * scale), np.round(new.imag * scale)], axis=1).astype("<i2")
            assert (B * L) % rtp_samples == 0
            for j in range(B * L // rtp_samples):
                body = i16[j * rtp_samples:(j + 1) * rtp_samples].tobytes()
                pkts.append(bytearray(struct.pack(">BBHII", 2 << 6, 97, 0, 0, 0x6B61) + b"\0" * 24 + body))
        seq = [0]

        def push_batch():
            if not rtp_samples:
                bank.push_iq_async(iq_pin.data_ptr(), B * L)
                return
            for pk in pkts:
                struct.pack_into(">HI", pk, 2, seq[0] & 0xFFFF, (seq[0] * rtp_samples) & 0xFFFFFFFF)
                seq[0] += 1
                assert bank.push_rtp(pk) == rtp_samples

        nops = [0]
        gone = []

        def operate(k):
            c = (k * 7919) % C
            if not gone or gone[0] != c:
                f = 0.8 + 0.2 * ((k * 31) % 17) / 17.0
                bank.set_filter(c, plan[c]["low"] * f, plan[c]["high"] * f, 3.0)
                nops[0] += 1
            if k % 3 == 0:
                c = (k * 104729 + 11) % C
                if not gone or gone[0] != c:
                    bank.set_mode(c, wl.bank_channel_config(plan[c]))
                    nops[0] += 1
            if gone:
                c = gone.pop()
                assert bank.add_channel(wl.bank_channel_config(plan[c])) == c      # (the only hole)
                nops[0] += 1
            elif k % 2 == 0:
                c = (k * 15485863 + 5) % C
                bank.remove_channel(c)
                gone.append(c)
                nops[0] += 1

        def control(k):
            for i in range(retunes_per_call):
                c = (k * 7919 + i * 104729) % C
                bank.set_second_lo(c, plan[c]["second_lo"] + (1.0 if k & 1 else 0.0))
            if control_plane:
                operate(k)

        # One order of the three steps for both modes: process what was pushed before, push the batch that has just come in,
        # queue the delivery.  (Pushing FIRST looks more natural for a paced loop and is 25-60 % slower: the input copy then
        # sits behind the previous call's output copy kernel on a hardware queue they share and waits with it for that
        # call's demodulators -- 1.85 / 2.45 ms per call against 1.49 at 33 792 channels, gpurun r6b; include/ka9q_hip.h
        # "Call order for full overlap".)  Paced, the batch pushed in iteration n is the one complete at A_n and is
        # processed in iteration n + 1: one period of pipeline delay, no effect on the schedule.
        steps = np.zeros((4, 400064), np.float32)     # ms inside control + process / push / queueing the delivery / waiting for k - 2
        step_n = [0]

        def call(k):
            t_a = time.perf_counter()
            control(k)
            assert bank.process() == B
            t_b = time.perf_counter()
            push_batch()
            t_c = time.perf_counter()
            j = k % nbuf
            if pcm:
                bank.pull_pcm_planes_async(outs[j].data_ptr(), masks[j].data_ptr(), stats[j].data_ptr(), compact=compact_status)
            else:
                bank.pull_planes_async(outs[j].data_ptr(), stats[j].data_ptr())
            t_d = time.perf_counter()
            bank.pull_wait(2)     # the planes of call k-2 are in host memory now; calls k-1 and k are in flight
            i = step_n[0]
            if i < steps.shape[1]:
                steps[0, i], steps[1, i], steps[2, i], steps[3, i] = (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (t_d - t_c) * 1e3, \
                    (time.perf_counter() - t_d) * 1e3
                step_n[0] = i + 1

        call_paced = call

        push_batch()
    else:
        iq_dev = torch.from_numpy(iq_host).to(dev)
        nops = [0]
        gone = []

        steps = np.zeros((4, 1), np.float32)
        step_n = [0]

        def call(k):
            bank.process_resident(iq_dev.data_ptr(), B)

        call_paced = call

    signal_s = B * L / fs
    gc_on = os.environ.get("KQ_RT_GC", "0") == "1"
    # a receiver thread belongs in a real-time scheduling class (the long intervals left on a shared host are the thread
    # being kept off its core, tools/rt_hold.sh); an ordinary user is normally refused -- tried, and said on the line
    sched = "SCHED_OTHER"
    if paced:
        try:
            os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(1))
            sched = "SCHED_FIFO"
        except (PermissionError, OSError, AttributeError):
            pass

    def run(ncalls, k0):
        for k in range(k0, k0 + ncalls):
            call(k)

    # (before the warm-up, not between it and the timed loop: a full collection with torch loaded takes 50-100 ms, the device
    #  falls idle and the first filter passes of the timed loop then ran 20 ms instead of 1.4 -- gpurun r6k)
    gc_pauses = []
    if not gc_on:
        gc.collect()
        gc.freeze()
        gc.disable()
    else:                           # the A/B leg: every collection's generation and length
        t_gc = [0.0]

        def on_gc(phase, info):
            if phase == "start":
                t_gc[0] = time.perf_counter()
            else:
                gc_pauses.append((info.get("generation", -1), (time.perf_counter() - t_gc[0]) * 1e3))
        gc.callbacks.append(on_gc)
    run(warm_calls, 0)
    if host_io:
        bank.host_io_wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(8, warm_calls)
    if host_io:
        bank.host_io_wait()
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    bank.host_timing(reset=True)
    bank.enable_timing(1)
    bank.timing(reset=True)
    nops[0] = 0
    step_n[0] = 0
    period = signal_s
    cap = int(min(400000, max(64, 1.5 * seconds / (period if paced else max(est, 1e-5)) + 64)))
    stamps = np.zeros(cap)          # when the iteration's delivery (k - 2) was in hand
    lag = np.zeros(cap)             # paced: how long after its batch was complete the iteration started
    k = warm_calls + 8 + 1
    n = 0
    try:
        t0 = time.perf_counter()
        if paced:
            origin = t0 + 2e-4 - period          # A_0 = origin + period: the first batch completes 0.2 ms from now
            while True:
                due = origin + (n + 1) * period
                now = time.perf_counter()
                # sleep most of the way, spin the last 0.25 ms.  (A loop that only spins is a CPU hog to the scheduler: on a host
                # shared with other jobs it was taken off its core for 5-11 ms once or twice a minute -- gpurun r6i, the whole
                # stall OUTSIDE the library's calls; a thread that sleeps is woken ahead of the hogs.)
                if due - now > 4e-4:
                    time.sleep(due - now - 2.5e-4)
                    now = time.perf_counter()
                while now < due:
                    now = time.perf_counter()
                lag[n] = now - due
                call_paced(k)
                stamps[n] = time.perf_counter()
                k += 1
                n += 1
                if n >= 16 and (n & 15) == 0 and (stamps[n - 1] - t0 >= seconds or n + 16 > cap):
                    break
        else:
            while True:                          # by the clock, not by the estimate: the run lasts at least `seconds`
                call(k)
                stamps[n] = time.perf_counter()
                k += 1
                n += 1
                if n >= 16 and (n & 15) == 0 and (stamps[n - 1] - t0 >= seconds or n + 16 > cap):
                    break
    finally:
        if not gc_on:
            gc.enable()
            gc.unfreeze()
        else:
            gc.callbacks.remove(on_gc)
    ncalls = n
    t_loop = stamps[n - 1] - t0
    if host_io:
        bank.host_io_wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ncalls
    stamps, lag = stamps[:n], lag[:n]
    # pacing as the host sees it: the intervals between consecutive deliveries (a receiver's output buffer has to ride out the
    # longest of them)
    skip_iv = min(n // 4, int(0.5 / period)) if paced else 0     # paced: the first half second is warm-up of the pacing itself
    iv = np.diff(stamps[skip_iv:]) * 1e3 if n - skip_iv > 2 else np.zeros(1)
    pacing = {"p50": _pct(iv, 50), "p99": _pct(iv, 99), "p99.9": _pct(iv, 99.9), "max": round(float(iv.max()), 4),
              "longer_than_two_periods": int((iv > 2e3 * period).sum())}
    worst = None
    if host_io and n > 2 and step_n[0] >= n:
        w = int(np.argmax(iv)) + 1 + skip_iv           # the iteration that ended the longest interval: where its host time went
        worst = {"interval_ms": round(float(iv[w - 1 - skip_iv]), 4), "at_s": round(float(stamps[w] - t0), 2),
                 "process_ms": round(float(steps[0, w]), 4), "push_ms": round(float(steps[1, w]), 4),
                 "queue_delivery_ms": round(float(steps[2, w]), 4), "wait_delivery_ms": round(float(steps[3, w]), 4),
                 "spin_before_ms": round(float(iv[w - 1 - skip_iv] - steps[:, w].sum()), 4),
                 "note": "the host's four steps inside the iteration that closed the longest delivery interval; spin_before = what "
                         "is left: waiting for the clock (paced) or time outside the loop's calls (the host itself held up)"}
    deadline = None
    if paced:
        due = origin + (np.arange(n) + 1) * period
        late_by = stamps - due - period            # delivery n - 2, due with the start of iteration n; > 0: a period behind
        backlog = np.floor(lag / period)
        skip = min(n // 4, int(0.5 / period))      # the first half second is warm-up of the pacing itself
        deadline = {"period_ms": round(period * 1e3, 4), "deliveries": int(n - skip),
                    "late_deliveries": int((late_by[skip:] > 0).sum()),
                    # the reference's player rides out 100 ms (monitor.c:83 PLAYOUT = SAMPRATE / 10): deliveries beyond THAT
                    # are the ones a listener hears
                    "playout_ms": 100.0, "late_beyond_playout": int((late_by[skip:] > 0.100).sum()),
                    "worst_lateness_ms": round(float(max(0.0, late_by[skip:].max())) * 1e3, 4),
                    "backlog_calls": {"p99.9": _pct(backlog[skip:], 99.9), "max": int(backlog[skip:].max())},
                    "start_lag_ms": {"p50": _pct(lag[skip:] * 1e3, 50), "p99.9": _pct(lag[skip:] * 1e3, 99.9),
                                     "max": round(float(lag[skip:].max()) * 1e3, 4)},
                    "host_busy_fraction": round(float(1.0 - np.clip(due[skip + 1:] - stamps[skip:-1], 0, None).sum() /
                                                      max(1e-9, stamps[-1] - due[skip])), 4),
                    "definition": "batch n is complete at A_n = origin + (n + 1) period; iteration n starts then (or as soon as "
                                  "the loop is free), pushes / processes / queues batch n and takes delivery n - 2; late = that "
                                  "delivery in hand after A_n + period; backlog = whole periods the iteration started behind A_n"}
    bank_worst_holder = bank.worst_lock_holder()
    ht = bank.host_timing(reset=True)
    tm = bank.timing(reset=True)
    bank.enable_timing(0)
    checksum = None
    if host_io:
        j = (k - 1) % nbuf
        if compact_status and pcm:
            st = np.frombuffer(stats[j].numpy().tobytes()[:C * B * kq.bank.COMPACT_STATUS_DTYPE.itemsize],
                               dtype=kq.bank.COMPACT_STATUS_DTYPE).reshape(C, B)
        else:
            st = np.frombuffer(stats[j].numpy().tobytes(), dtype=kq.bank.STATUS_DTYPE).reshape(C, B)
        a = outs[j].numpy().reshape(C, B, 2 * olen)
        if gone:           # (the channel that is away right now still shows its last delivery: not counted)
            st = np.delete(st, gone[0], axis=0)
        sq = st["state"] if (compact_status and pcm) else st["squelch_count"]
        checksum = {"nout_sum": int(st["nout"].sum()), "squelch_open": int((sq < 2).sum()),
                    "audio_abs_sum": float(np.abs(a[::max(1, C // 997), :, :olen].astype(np.float64)).sum())}
    bank.close()
    if sched == "SCHED_FIFO":
        try:
            os.sched_setscheduler(0, os.SCHED_OTHER, os.sched_param(0))
        except OSError:
            pass
    return {"config": config, "channels": C, "blocks_per_call": B, "compute_n0": int(bool(compute_n0)), "signal_ms_per_call": round(signal_s * 1e3, 4),
            "ms_per_call": round(dt * 1e3, 4), "realtime_factor": round(signal_s / dt, 4), "calls": ncalls,
            "wall_s": round(dt * ncalls, 2), "paced": bool(paced), "scheduler": sched, "deadline": deadline, "delivery_interval_ms": pacing,
            "longest_interval": worst,
            "gc": ({"collections": len(gc_pauses), "longest_ms": round(max([p[1] for p in gc_pauses] or [0.0]), 3),
                    "longest_generation": max(gc_pauses or [(-1, 0.0)], key=lambda p: p[1])[0]} if gc_on else
                   "frozen and off inside the timed loop"),
            "worst_lock_holder": bank_worst_holder,
            "filter_kernel_ms": round(tm["filter_ms"] / max(1, tm["filter_launches"]), 4),
            "filter_kernel_max_ms": round(tm["filter_max_ms"], 4),
            # the longest interval's host part (wall time between queueing its two markers) and which pass it was
            "filter_max_host_submit_ms": round(tm["filter_max_submit_ms"], 4), "filter_max_pass": int(tm["filter_max_launch"]),
            "host_ms_per_call": round(ht["call_ms"] / max(1, ht["calls"]), 4),
            "host_stage_ms_per_call": round(ht["stage_ms"] / max(1, ht["calls"]), 4),
            "host_slot_wait_ms_per_call": round(ht["slot_wait_ms"] / max(1, ht["calls"]), 4),
            "lock_wait_max_ms": round(ht["lock_wait_max_ms"], 4), "ctl_hold_max_ms": round(ht["ctl_hold_max_ms"], 4),
            "host_io": (("pcm int16 + compact status (24 B)" if compact_status else "pcm int16 + status") if pcm else
                        "float audio + status") if host_io else None,
            "d2h_bytes_per_call": out_bytes, "d2h_GBps": round(out_bytes / dt / 1e9, 3),
            "h2d_bytes_per_call": B * L * 8 if host_io else 0,
            "input": ("RTP datagrams of %d int16 I/Q samples (kq_bank_push_rtp)" % rtp_samples) if rtp_samples else
                     ("float batch (kq_bank_push_iq_async)" if host_io else "resident"),
            "retunes_per_call": retunes_per_call, "swept_channels": swept_channels,
            "control_plane_ops_per_s": round(nops[0] / (dt * ncalls), 1) if host_io and control_plane else 0,
            "setup_s": round(setup_s, 2), "check": checksum}
