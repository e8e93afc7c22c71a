#!/usr/bin/env python3
"""bench.py -- throughput of the ka9q-radio per-channel DSP hot path on MI355X.

One *step* = one pass of the whole hot path (IF power, NCO mix + overlap-save filter/decimator,
FM/AM/linear demodulators) over one batch of `--blocks` overlap-save blocks for every channel of
this GPU, with the front-end I/Q already resident in HBM.

Default workload (N=1): BASELINE.json configs[3] per-GPU share = the north_star target shape:
1024 FM channels, 16384-point overlap-save (L=8192, M=8193), decimate 256, 10 MS/s synthetic I/Q,
with compute_n0 (radio.c:383-425) on every channel-block as the reference's demodulator threads run it
(fm.c:78-82) -- the full-spectrum forward path.  The same workload without the status-only noise
estimate (pruned forward transform) is reported beside it as `without_compute_n0`, never as `value`.
With --gpus N the channels are sharded (1024 per GPU, weak scaling, configs[3] at N=8) and every
batch of front-end I/Q is broadcast from rank 0 over RCCL by the product's own fan-out (kq_fanout_* of
include/ka9q_hip.h: ncclBroadcast on the library's side stream, two slots; `--fanout torch` selects the
torch.distributed twin); started without torch.distributed.run, `--gpus N` launches its own N ranks.
At N=1 the line also carries `with_host_io` (the same step fed from pinned host memory and delivering audio + status
to pinned host memory every step, copies overlapped), `cold_first_steps_ms` (no spin-up) and the CPU baseline rows.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup", type=int, default=-1,
                    help="untimed steps run before the warm-up steps so that the GPU is in its sustained state when the "
                         "timed region starts; -1 (default): as many as fill --spinup-seconds")
    ap.add_argument("--spinup-seconds", type=float, default=15.0,
                    help="length of the default spin-up: after an idle pause the power management lets the same step get "
                         "faster for about 15 s (tools/cold_spinup.sh: 1.462 ms after 0.4 s, 1.444 after 4 s, 1.428 after "
                         "14 s of load), and a receiver streams for hours")
    ap.add_argument("--config", default="cfg4", help="workload: cfg2|cfg3|cfg4|cfg5 (ka9q_sdr_amd/workload.py)")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: the config's)")
    ap.add_argument("--blocks", type=int, default=64, help="overlap-save blocks per step")
    ap.add_argument("--fwd", default="auto", choices=["auto", "full", "pruned"])
    ap.add_argument("--n0", type=int, default=1, help="1 (default): compute_n0 every block, as the reference does; 0: off")
    ap.add_argument("--no-n0-row", "--no-second-row", dest="no_second_row", action="store_true",
                    help="skip the secondary measurement (the same workload without compute_n0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of each baseline row")
    ap.add_argument("--fanout", default="c", choices=["c", "torch"],
                    help="front-end fan-out: c = the library's kq_fanout_* (ncclBroadcast, the product path); torch = "
                         "ka9q_sdr_amd/shard.py's torch.distributed twin (always used with --backend gloo)")
    ap.add_argument("--ingest", default="both", choices=["resident", "host", "both"],
                    help="N > 1 with the C fan-out: where the root takes each batch from -- its own device memory (`value`), pinned "
                         "host memory (H2D + ncclBroadcast + compute in one pipeline: the `host_ingest` object), or both")
    ap.add_argument("--rccl-max-channels", type=int, default=0,
                    help="N > 0: export NCCL_MAX_NCHANNELS=N before RCCL initialises (default 0: RCCL's own choice; the "
                         "broadcast's footprint costs the step <= 2.5 %% at any count, profiles/r04/bcast_side_kernel_probe.txt)")
    ap.add_argument("--per-rank", action="store_true",
                    help="emit the per_rank diagnostics (step, kernel, broadcast and stall time of every rank) at N = 1 too; "
                         "with --gpus N > 1 they are always on the line")
    ap.add_argument("--no-host-io", action="store_true", help="skip the with_host_io measurement")
    ap.add_argument("--no-rows", action="store_true",
                    help="skip the `rows` object: BASELINE.json's other single-GPU shapes (cfg2, cfg3, cfg5 per-GPU share), each "
                         "with 2 s of spin-up and 20 timed steps (only measured at N = 1 on the default workload)")
    ap.add_argument("--no-realtime", action="store_true",
                    help="skip the `realtime` object: the largest channel count one GPU carries at 1.0 x real time, measured "
                         "(cfg 4 geometry, 2 blocks per call, host I/O every call; about 25 s)")
    ap.add_argument("--realtime-seconds", type=float, default=60.0,
                    help="length of the paced run that has to go without a late delivery")
    ap.add_argument("--gpu-state", action="store_true",
                    help="sample shader clock and power at 50 Hz over the spin-up steps (disturbs the timed steps by about 1 %%)")
    return ap.parse_args()


def usable_cores():
    """Host threads this process can really run at once: its affinity mask, cut to the container's CPU quota (cgroup
    cpu.max / cfs_quota_us) -- a GPU box shows all 256 logical CPUs of the host to a container that may use 16 of them,
    and 256 threads on a quota of 16 spend their time being throttled.  Returns (usable, visible)."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    usable = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    return usable, visible


def cpu_baseline(name, geom, plan, iq_host, target_s, compute_n0=0):
    """Oracle ('port') timed on this box's host cores on a bounded sample of the same workload: channel set-up outside
    the clock, 20 warm-up blocks, then >= 200 timed blocks per channel."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kq_oracle as ko
    from common import oracle_cfg
    cores, visible = usable_cores()
    L = geom["L"]
    nchan = min(len(plan), 4 * cores)      # min(C, 4 x cores) channels over all cores
    cfgs = [oracle_cfg(p, geom["samprate"], L, geom["M"], geom["D"], compute_n0=compute_n0) for p in plan[:nchan]]
    avail = len(iq_host) // L
    t, _ = ko.cpu_baseline(cfgs, iq_host, avail, 2, 8, cores)            # calibration pass
    rate = nchan * 8 / t
    timed = int(max(200, target_s * rate / nchan))
    t, _ = ko.cpu_baseline(cfgs, iq_host, avail, 20, timed, cores)
    msps = nchan * timed * L / t / 1e6
    # one channel on one otherwise idle core, for scale: all threads together run far below cores x this (SMT pairs,
    # all-core clocks, 0.7 MB of working set per channel against the shared caches)
    t1, _ = ko.cpu_baseline(cfgs[:1], iq_host, avail, 2, 8, 1)
    n1 = int(max(50, 2.0 * 8 / max(t1, 1e-6)))
    t1, _ = ko.cpu_baseline(cfgs[:1], iq_host, avail, 20, n1, 1)
    single = n1 * L / t1 / 1e6
    return {"value": round(msps, 3), "unit": "Msamples/s (channel-samples)", "cores": cores, "kind": "port",
            "per_core": round(msps / cores, 3), "single_thread_alone": round(single, 3),
            "channels_at_realtime": int(msps * 1e6 / geom["samprate"]),
            "sample": "%d channels x %d timed blocks (20 warm-up, set-up outside the clock) of %s: oracle C restatement in the "
                      "reference's structure (per-sample FP64 NCO, full N-point FFT per channel, %s), own radix-4 autosort "
                      "FFT (no libfftw3f in the image), %d threads (%d logical CPUs visible, CPU quota of the container %d), %.1f s" %
                      (nchan, timed, name, "compute_n0 every block" if compute_n0 else "no compute_n0", cores, visible, cores, t)}


class GpuState:
    """Shader clock and socket power while the timed steps run, sampled from the amdgpu hwmon files every 20 ms: the same
    build differs by a few per cent from one box of the pool to the next, and this says whether the clock or the power
    cap is behind it.  One sampler per process; the card is the one whose clock is highest under load."""

    def __init__(self, pci_bus=None):
        """pci_bus: bus number of the device in use (torch's pci_bus_id).  A box of the pool holds eight GPUs, the others
        busy with somebody else's work: only the cards whose sysfs path goes through that bus are read -- when there are
        any; otherwise all, and the one drawing most power is reported.  (Until late in round 3 the card with the
        highest clock was reported, which was a neighbour's: 2404 MHz at 290 W, while this workload runs its own GPU
        at 2.2-2.4 GHz and 1.1-1.4 kW against a 1.4 kW cap.)"""
        import glob
        self.dirs = [os.path.dirname(f) for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")]
        if pci_bus is not None:
            mine = [d for d in self.dirs if (":%02x:" % pci_bus) in os.path.realpath(os.path.join(d, "..", ".."))]
            if mine:
                self.dirs = mine
        self.samples = {d: [] for d in self.dirs}
        self.stop = False
        self.thread = None

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().strip())
        except Exception:
            return None

    def _run(self):
        while not self.stop:
            for d in self.dirs:
                f, p = self._read(d + "/freq1_input"), self._read(d + "/power1_input")
                if f is not None:
                    self.samples[d].append((f / 1e6, (p or 0.0) / 1e6))
            time.sleep(0.02)

    @classmethod
    def once(cls, pci_bus=None):
        """one reading of the card(s) on the device's bus (or of every card), the one drawing most power reported"""
        g = cls(pci_bus)
        best = None
        for d in g.dirs:
            f, p = cls._read(d + "/freq1_input"), cls._read(d + "/power1_input")
            if f is not None and (best is None or (p or 0) > best[1]):
                best = (f / 1e6, (p or 0.0) / 1e6, d)
        if not best:
            return None
        cap = cls._read(best[2] + "/power1_cap")
        return {"sclk_mhz": round(best[0]), "power_w": round(best[1]), "power_cap_w": round(cap / 1e6) if cap else None,
                "samples": 1, "source": "amdgpu hwmon freq1_input / power1_input of the card drawing most power, read once "
                                        "during the last quarter of the timed steps (--gpu-state samples at 50 Hz over the spin-up)"}

    def start(self):
        import threading
        if self.dirs:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def finish(self):
        self.stop = True
        if self.thread:
            self.thread.join()
        best = max(self.dirs, key=lambda d: max([s[1] for s in self.samples[d]] or [0]), default=None)   # most power
        if not best or not self.samples[best]:
            return None
        f = [s[0] for s in self.samples[best]]
        w = [s[1] for s in self.samples[best]]
        cap = self._read(best + "/power1_cap")
        return {"sclk_mhz": {"min": round(min(f)), "mean": round(sum(f) / len(f)), "max": round(max(f))},
                "power_w": {"min": round(min(w)), "mean": round(sum(w) / len(w)), "max": round(max(w))},
                "power_cap_w": round(cap / 1e6) if cap else None, "samples": len(f),
                "source": "amdgpu hwmon freq1_input / power1_input, 20 ms apart over the untimed spin-up and warm-up steps"}


def pmc_traffic(config, channels, blocks, fwd):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 read correction: tools/pmc_summary.py).  PMC collection
    cannot run inside this process, so the figure comes from the newest profiles/*/pmc_*.json recorded for the
    same workload; null when none matches."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("config") == config and w.get("channels") == channels and w.get("blocks") == blocks and w.get("fwd") == fwd:
            best = (d.get("traffic"), os.path.relpath(path, ROOT))
    return best if best else (None, None)


def roofline_of(wl, config, geom, plan, C, B, k_ms, n0, fwd_name, demod_ms=None):
    """SURVEY 8d's HBM figure (algorithmic bytes / kernel time; the shared input makes it exceed what crosses the
    memory side) and, beside it, the bounds that do bind: vector-ALU rate against the 157.3 TFLOP/s FP32 peak,
    and the HBM traffic the PMC counters measured."""
    abytes = sum(wl.algorithmic_bytes(geom, p["demod"], p.get("channels", 1) == 2) for p in plan) * B
    fl = sum(wl.algorithmic_flops(geom, p["demod"], bool(p.get("doppler", 0.0)), n0, fwd_name == "pruned") for p in plan) * B
    traffic, src = pmc_traffic(config, C, B, fwd_name)
    ach = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
         "traffic": traffic, "traffic_source": src,
         "kernel": "pre-detection filter (NCO mix + forward FFT%s + response + IFFT), %s path" %
                   (" + compute_n0" if n0 else "", fwd_name),
         "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": abytes,
         "algorithmic_flops_per_launch": int(fl),
         "valu_tflops": round(fl / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else 0.0,
         "valu_peak_tflops": 157.3,
         "valu_frac": round(fl / (k_ms * 1e-3) / 157.3e12, 4) if k_ms > 0 else 0.0,
         "hbm_frac_measured": round(traffic / (k_ms * 1e-3) / 8e12, 4) if (traffic and k_ms > 0) else None,
         "note": "frac = algorithmic bytes (SURVEY 8d: the N-sample window counted once per channel) / kernel time / "
                 "8 TB/s; every channel reads the one shared input through L2, so the bytes that cross the memory "
                 "side (hbm_frac_measured, from the FETCH_SIZE / WRITE_SIZE passes in profiles/) are ~1 % of peak and "
                 "HBM is not what limits the kernel; valu_frac prices the same launch against the FP32 vector peak."}
    if demod_ms is not None:
        r["demod_ms"] = round(demod_ms, 4)
    return r


def measure_row(torch, kq, wl, config, blocks, dev_index, stream, pci_bus, spin_seconds=2.0, steps=20):
    """One more single-GPU shape of BASELINE.json on the driver's own line: its per-GPU channel count,
    compute_n0 on, input resident in HBM, `spin_seconds` of untimed identical steps, then `steps` timed ones."""
    geom = dict(wl.GEOMETRY[config])
    L, M, D, fs, C = geom["L"], geom["M"], geom["D"], geom["samprate"], geom["channels"]
    plan = wl.channel_plan(config, C)
    bank = kq.Bank(fs, L, M, D, C, blocks, device=dev_index, compute_n0=True, fwd_mode=kq.KQ_FWD_AUTO, stream=stream.cuda_stream,
                   pl_tone=False)     # the synthetic configs 2-5 run without pltask
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))
    nwin = (M - 1) + blocks * L
    iq = torch.from_numpy(wl.make_iq(fs, nwin, seed=0x6B613971 ^ int(config[3:]))).to(torch.device("cuda", dev_index))
    for _ in range(4):
        bank.process_resident(iq.data_ptr(), blocks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        bank.process_resident(iq.data_ptr(), blocks)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    nspin = int(min(100000, spin_seconds / max(est, 1e-6)))
    for _ in range(nspin):
        bank.process_resident(iq.data_ptr(), blocks)
    torch.cuda.synchronize()
    bank.enable_timing(1)
    bank.timing(reset=True)
    import threading
    reading = {}
    reader = threading.Thread(target=lambda: reading.update(state=GpuState.once(pci_bus)), daemon=True)
    t0 = time.perf_counter()
    for k in range(steps):
        if k == steps // 2:       # clock and power under the load, read by a side thread (the files take a millisecond or two:
            reader.start()        # read from this thread they would let the four-step queue of a 0.4 ms step run dry)
        bank.process_resident(iq.data_ptr(), blocks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    reader.join()
    state = reading.get("state")
    tm = bank.timing(reset=True)
    bank.enable_timing(2)
    for _ in range(3):
        bank.process_resident(iq.data_ptr(), blocks)
    torch.cuda.synchronize()
    tm2 = bank.timing(reset=True)
    fwd_used = {1: "full", 2: "pruned"}[bank.fwd_mode]
    bank.close()
    k_ms = tm["filter_ms"] / max(1, tm["filter_launches"])
    r = roofline_of(wl, config, geom, plan, C, blocks, k_ms, True, fwd_used, tm2["demod_ms"] / max(1, tm2["filter_launches"]))
    per_kind = {}
    for p in plan:
        per_kind[p["demod"]] = per_kind.get(p["demod"], 0) + 1
    return {"workload": "%s: %d channels (%s), N=%d, decimate %d, %.3g MS/s, %d blocks/step, fwd=%s, compute_n0=1, pltask off" %
                        (config, C, "+".join("%d %s" % (v, k) for k, v in sorted(per_kind.items())), L + M - 1, D, fs / 1e6,
                         blocks, fwd_used),
            "value": round(C * blocks * L / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 4), "steps": steps,
            "spinup_steps": nspin, "kernel_ms": r["kernel_ms"], "demod_ms": r.get("demod_ms"),
            "frac": r["frac"], "valu_frac": r["valu_frac"],
            "step_frac": round(r["algorithmic_bytes_per_launch"] / dt / 8e12, 4),
            "realtime_factor": round(blocks * L / dt / fs, 2), "gpu_state": state}


def measure_channels_at_realtime(torch, kq, wl, dev_index, stream, seconds):
    """BASELINE.json's "channels @ real-time" as a DEADLINE figure: the reference's operating point is every
    channel at 1.0 x the front end's rate (one `radio` per channel, main.c:105, README.md:470-477; the receiver loop of
    main.c:288-365 takes packets as they arrive).  cfg 4's geometry (N = 16384, D = 256, 10 MS/s, FM, compute_n0 on), ONE bank
    of C channels on this GPU, two blocks per call (1.64 ms of signal), PACED: a batch becomes available every 1.64 ms of wall
    time and not sooner, is pushed from pinned host memory, processed, and every channel's audio + status is back in pinned
    host memory two calls later (realtime_harness.py).  Two deadlines are counted per run: deliveries more than ONE CALL PERIOD
    (1.64 ms) behind schedule (`late_deliveries`, strict: one stall of the host's scheduler or of the device is enough), and
    deliveries behind by more than the reference player's playout buffer (`late_beyond_playout`; monitor.c:83: 100 ms) -- the
    ones a listener would hear.  `channels` = the largest count tried whose run of `seconds` (>= 60 by default) had none of the
    second kind; `channels_strict` the largest whose run had none of the first kind either (null when no run came through
    clean: the trials say how close).  Beside them the mean-factor figure of the rounds before and the count that leaves 5 %
    of every period idle."""
    from realtime_harness import measure_realtime

    def keep(t):
        d = t.get("deadline") or {}
        return {"channels": t["channels"], "blocks_per_call": t["blocks_per_call"], "paced": t["paced"], "wall_s": t["wall_s"],
                "realtime_factor": t["realtime_factor"], "late_deliveries": d.get("late_deliveries"),
                "late_beyond_playout": d.get("late_beyond_playout"), "worst_lateness_ms": d.get("worst_lateness_ms"),
                "backlog_max": (d.get("backlog_calls") or {}).get("max"), "interval_max_ms": t["delivery_interval_ms"]["max"]}

    trials = []
    probe = measure_realtime(torch, kq, wl, "cfg4", 32768, 2, dev_index, stream, seconds=1.5)
    trials.append(keep(probe))
    c_mean = 32768 * probe["realtime_factor"]           # where the MEAN factor crosses 1.0 (ms per call is affine in C)
    # ---- paced, short runs: down from 3 % under the mean-factor count in steps of 2 % until 5 s pass with the backlog never
    # beyond a few periods (a count too close to the mean-factor one never works a stall off)
    C = int(c_mean * 0.97) // 256 * 256
    cand = None
    for _ in range(4):
        r = measure_realtime(torch, kq, wl, "cfg4", C, 2, dev_index, stream, seconds=5.0, paced=True)
        trials.append(keep(r))
        if r["deadline"]["late_beyond_playout"] == 0 and r["deadline"]["backlog_calls"]["max"] <= 8:
            cand = C
            break
        C = int(C * 0.98) // 256 * 256
    # ---- the hold: `seconds` of paced running.  First at the short runs' count; a second one at 93 % of the mean-factor count
    # when the first had deliveries late by the strict measure (7 % of every period to spare works a stall of a few milliseconds
    # off within a few dozen calls)
    best = strict = None
    holds = []
    if cand:
        for C in (cand, min(cand - 256, int(c_mean * 0.93) // 256 * 256)):
            r = measure_realtime(torch, kq, wl, "cfg4", C, 2, dev_index, stream, seconds=seconds, paced=True)
            trials.append(keep(r))
            holds.append(r)
            if best is None and r["deadline"]["late_beyond_playout"] == 0:
                best = r
            if strict is None and r["deadline"]["late_deliveries"] == 0:
                strict = r
            if strict is not None and best is not None:
                break
    out = {"definition": "largest channel count tried of ONE bank on one GPU that ran PACED for held_seconds without a delivery later "
                         "than the reference player's playout buffer (monitor.c:83: 100 ms): cfg4 geometry (N=16384, decimate 256, "
                         "10 MS/s, FM, compute_n0=1), 2 blocks (1.64 ms of signal) per call, a batch available every 1.64 ms of wall "
                         "time, pushed from pinned host memory, audio + status of every channel in pinned host memory two calls "
                         "later; channels_strict: the same with no delivery more than ONE call period behind schedule either "
                         "(realtime_harness.py)",
           "channels": best["channels"] if best else 0, "held_seconds": best["wall_s"] if best else 0.0,
           "late_beyond_playout": best["deadline"]["late_beyond_playout"] if best else None,
           "late_deliveries": best["deadline"]["late_deliveries"] if best else None,
           "worst_lateness_ms": best["deadline"]["worst_lateness_ms"] if best else None,
           "channels_strict": strict["channels"] if strict else None,
           "mean_factor_channels": int(c_mean) // 256 * 256,
           "mean_factor_note": "the figure of rounds 4-5: the count at which throughput-mode realtime_factor (a mean) crosses 1.0",
           "float_audio": best, "trials": trials,
           "holds": [{"channels": h["channels"], "deadline": h["deadline"], "delivery_interval_ms": h["delivery_interval_ms"],
                      "longest_interval": h.get("longest_interval"), "filter_kernel_ms": h["filter_kernel_ms"],
                      "filter_kernel_max_ms": h["filter_kernel_max_ms"],
                      # whose time the longest filter interval was: the host's between queueing its two markers, the rest the device's
                      "filter_max_host_submit_ms": h.get("filter_max_host_submit_ms"), "filter_max_pass": h.get("filter_max_pass")}
                     for h in holds]}
    if best:
        Cb, short = best["channels"], min(5.0, seconds)
        # the count that leaves 5 % of every call period idle: throughput-mode factor >= 1 / 0.95 (one placement, one check)
        C5 = int(c_mean * 0.95 * 0.995) // 256 * 256
        r5 = measure_realtime(torch, kq, wl, "cfg4", C5, 2, dev_index, stream, seconds=2.0)
        if r5["realtime_factor"] < 1.0 / 0.95:
            C5 = int(C5 * r5["realtime_factor"] * 0.95 * 0.997) // 256 * 256
            r5 = measure_realtime(torch, kq, wl, "cfg4", C5, 2, dev_index, stream, seconds=2.0)
        out["five_percent_headroom"] = {"channels": C5, "realtime_factor": r5["realtime_factor"], "ms_per_call": r5["ms_per_call"],
                                        "note": "throughput mode: factor >= 1.0526 means every 1.64 ms period has >= 5 % to spare"}
        # the reference's real output format (int16 PCM, audio.c:22-28) instead of float audio; then with the 24-byte status
        # records (kq_bank_pull_pcm_planes_compact_async); each paced, and its mean factor from a short throughput run
        for name, kw in (("pcm_int16", dict(pcm=True)), ("pcm_compact_status", dict(pcm=True, compact_status=True)),
                         ("four_blocks_per_call", dict()),
                         # both ends in the reference's own formats: RTP datagrams of int16 I/Q in (radio.c:110-122,
                         # main.c:318-341; one kq_bank_push_rtp per datagram), int16 PCM planes out
                         ("rtp_in_pcm_out", dict(pcm=True, rtp_samples=1024)),
                         # a busy control plane beside the stream: a filter change before every call, mode restarts, a channel
                         # leaving and returning; none of them waits for the device or holds up the calls in flight
                         ("with_control_plane", dict(pcm=True, control_plane=True))):
            nb = 4 if name == "four_blocks_per_call" else 2
            out[name] = measure_realtime(torch, kq, wl, "cfg4", Cb, nb, dev_index, stream, seconds=short, paced=True, **kw)
        # BASELINE's cfg 3 mix (FM + AM + USB / LSB in equal shares of the 64 emitters) at the same count: the AM / SSB
        # demodulators' AGC scans instead of the FM discriminator on half of the channels
        out["mixed_fm_am_ssb"] = measure_realtime(torch, kq, wl, "cfg3", Cb, 2, dev_index, stream, seconds=short, paced=True, pcm=True)
        for name, kw in (("pcm_int16", dict(pcm=True)), ("pcm_compact_status", dict(pcm=True, compact_status=True))):
            t = measure_realtime(torch, kq, wl, "cfg4", Cb, 2, dev_index, stream, seconds=2.0, **kw)
            out[name]["throughput_realtime_factor"] = t["realtime_factor"]
            out[name]["throughput_copy_GBps"] = t["d2h_GBps"]
        # for a host that does not read the noise estimate (status.n0 = NaN): without compute_n0 the bank runs its pruned forward
        # path (the `without_compute_n0` row of this line at 1024 x 64) -- how many channels THAT holds, PCM planes out, paced
        p2 = measure_realtime(torch, kq, wl, "cfg4", 49152, 2, dev_index, stream, seconds=1.5, pcm=True, compute_n0=False)
        C2 = int(49152 * p2["realtime_factor"] * 0.97) // 256 * 256
        for _ in range(3):
            r2 = measure_realtime(torch, kq, wl, "cfg4", C2, 2, dev_index, stream, seconds=short, pcm=True, compute_n0=False, paced=True)
            if r2["deadline"]["late_beyond_playout"] == 0 and r2["deadline"]["backlog_calls"]["max"] <= 8:
                break
            C2 = int(C2 * 0.98) // 256 * 256
        out["without_compute_n0"] = dict(r2, note="secondary: the reference's demodulators run compute_n0 on every block; `channels` is "
                                                  "the count tried last, it held iff deadline.late_beyond_playout == 0")
    return out


# The fan-out's broadcast is 4.3 MB per step on a side stream beside the filter kernel.  What an RCCL-shaped kernel there costs
# the step was measured with a stand-in on one GPU (tools/bcast_probe.py, profiles/r04/bcast_side_kernel_probe.txt): 1 to 64
# workgroups of 256 threads, resident for up to 400 us, cost 0.2-2.5 % (cfg 4) and 2-3.5 % (cfg 5), of which ~1 % is the two
# stream markers -- far inside the 25 % the 6 x target leaves at ANY channel count.  So RCCL keeps its own choice (the
# configuration it is tested in); --rccl-max-channels N exports NCCL_MAX_NCHANNELS=N before any rank creates a communicator
# for whoever wants fewer workgroups beside the filter launch (an NCCL_MAX_NCHANNELS already in the environment wins).


def self_launch(a):
    """`bench.py --gpus N` started as a plain process: start the N ranks ourselves (one child per GPU through
    torch.distributed.run) BEFORE anything here touches the GPU, pass rank 0's JSON line through, exit with its code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.rccl_max_channels > 0:
        env.setdefault("NCCL_MAX_NCHANNELS", str(a.rccl_max_channels))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    if a.gpus > 1 and a.rccl_max_channels > 0:     # (started by torch.distributed.run: set here, before torch or RCCL load)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(a.rccl_max_channels))
    import torch
    import ka9q_sdr_amd as kq
    from ka9q_sdr_amd import workload as wl
    if not os.path.exists(kq.library_path()):   # sources-only checkout: build the product before anything is timed
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            kq.build_library()
        else:
            for _ in range(1200):
                if os.path.exists(kq.library_path()):
                    break
                time.sleep(0.5)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs the MI355X: there is no CPU path to measure"
    dev_index = local_rank % torch.cuda.device_count()   # == local_rank whenever there is a GPU per rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    geom = dict(wl.GEOMETRY[a.config])
    L, M, D, fs = geom["L"], geom["M"], geom["D"], geom["samprate"]
    C = a.channels or geom["channels"]
    B = a.blocks
    from ka9q_sdr_amd.shard import FrontEndFanout, shard_range
    first, count = shard_range(C * world, world, rank)    # weak scaling: C channels per GPU
    assert count == C
    plan = wl.channel_plan(a.config, C, first=first)

    fwd = {"auto": kq.KQ_FWD_AUTO, "full": kq.KQ_FWD_FULL, "pruned": kq.KQ_FWD_PRUNED}[a.fwd]
    stream = torch.cuda.Stream(device=dev)     # an explicit (non-null) HIP stream handed to the library
    torch.cuda.set_stream(stream)
    # pltask (fm.c:189-285) is part of cfg 1 only; the synthetic configs 2-5 are measured without it
    pl_tone = a.config == "cfg1"
    bank = kq.Bank(fs, L, M, D, C, B, device=dev_index, compute_n0=bool(a.n0), fwd_mode=fwd,
                   stream=stream.cuda_stream, pl_tone=pl_tone)
    for p in plan:
        bank.add_channel(wl.bank_channel_config(p))

    # Front-end I/Q: window layout [M-1 history | B*L new samples], generated once on rank 0
    nwin = (M - 1) + B * L
    bufs = [torch.zeros(nwin, dtype=torch.complex64, device=dev) for _ in range(2)]
    iq_host = None
    if rank == 0:
        iq_host = wl.make_iq(fs, nwin, seed=0x6B613971)
        bufs[0].copy_(torch.from_numpy(iq_host))
        bufs[1].copy_(bufs[0])
    # Front-end fan-out (the reference's UDP multicast, multicast.c:143-237): rank 0 -> all over RCCL,
    # double buffered on a side stream so batch k+1 travels while batch k is processed
    # The C fan-out holds its own RCCL communicator, whatever backend torch.distributed runs on: the process group only
    # carries the 128-byte identifier and the ranks' agreement.  (Two ranks sharing one GPU -- `--backend gloo` on a one-GPU
    # box -- are refused by RCCL inside kq_fanout_create, on every rank: that run exercises the fall-back to the torch twin.)
    use_c = a.fanout == "c"
    tdev = dev if a.backend == "nccl" else torch.device("cpu")     # where the process group wants its tensors
    fan_error = None
    if use_c:
        import ctypes
        from ka9q_sdr_amd.shard import CFanout, share_unique_id
        lib = kq.load_library()

        def make_id():
            buf = ctypes.create_string_buffer(128)
            if lib.kq_fanout_unique_id(buf) != 0:
                raise RuntimeError("kq_fanout_unique_id: " + (lib.kq_last_error() or b"").decode())
            return buf.raw

        fan, fan_error = None, None
        try:
            ident = share_unique_id(make_id, rank, 0, dist, tdev) if world > 1 else None
            fan = CFanout(lib, dev_index, rank, world, nwin, ident)     # collective: every rank is here
        except Exception as e:                                            # e.g. librccl refusing the topology
            fan_error = "%s: %s" % (type(e).__name__, e)
        if world > 1:      # all ranks take the same road: one failure sends everybody to the torch twin (said in the line)
            flag = torch.tensor([1 if fan is None else 0], device=tdev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                if fan is not None:
                    fan.close()
                    fan = None
                fan_error = fan_error or "another rank's kq_fanout_create failed"
        elif fan is None:
            raise RuntimeError(fan_error)
        use_c = fan is not None
    if use_c:
        torch.cuda.synchronize()                                          # the batch is in bufs[0] before another stream reads it
        for i in range(2):                                                # the batch goes into both slots once
            fan.fill(i, bufs[0].data_ptr() if rank == 0 else None)
        cs = stream.cuda_stream

        def step(k):
            i = k & 1
            p = fan.acquire(i, cs)
            bank.process_resident(p, B)
            fan.release(i, cs)
            fan.post(i)           # refill this slot for step k+2 while step k+1 computes
    else:
        fan = FrontEndFanout(bufs, src=0)

        def step(k):
            i = k & 1
            buf = fan.acquire(i, stream)
            bank.process_resident(buf.data_ptr(), B)
            fan.release(i, stream)
            fan.post(i, stream)       # refill this buffer for step k+2 while step k+1 computes

        for i in range(2):
            fan.release(i, stream)
            fan.post(i, stream)
    # the first steps of an idle GPU, one by one (no spin-up): what a receiver sees when it starts
    cold = []
    for k in range(4):
        torch.cuda.synchronize()
        tc = time.perf_counter()
        step(k)
        torch.cuda.synchronize()
        cold.append(round((time.perf_counter() - tc) * 1e3, 4))
    # Spin-up, then the W warm-up steps, then the K timed steps: one continuous sequence of identical steps.  The
    # spin-up count is even so that the double buffers are at the same parity whatever its length.
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if a.spinup < 0 and under_profiler:
        a.spinup = 300      # a trace or counter pass of ten thousand serialised launches takes minutes and gigabytes
    if a.spinup >= 0:
        spin = 2 * (a.spinup // 2)
    else:   # by time, from the cold steps just measured; every rank runs the same count (the steps post collectives)
        est_ms = max(1e-3, min(cold[1:]))
        spin = 2 * int(min(400000, a.spinup_seconds * 1e3 / est_ms) // 2)
        if dist:
            cnt = torch.tensor([spin], device=tdev, dtype=torch.int64)
            dist.broadcast(cnt, src=0)
            spin = int(cnt.item())
    # Clocks and power: by default ONE reading, taken by a side thread during the last quarter of the timed steps.  Every read of the hwmon files is a query to
    # the SMU, and sampling disturbs what it looks at (tools/gs_probe.sh): a 50 Hz sampler over the timed steps cost them
    # 0.5 %, the same sampler over the spin-up steps only, stopped before the clock starts, 0.5-1.6 %.  --gpu-state
    # asks for that sampler (over spin-up and warm-up) all the same.
    pci_bus = getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)
    gpu_state = GpuState(pci_bus) if rank == 0 and a.gpu_state else None
    if gpu_state:
        gpu_state.start()
    torch.cuda.synchronize()
    ts0 = time.perf_counter()
    for k in range(spin):
        step(4 + k)
    torch.cuda.synchronize()     # one wait at its end: the spin-up's own average sits inside the driver's clock around this run
    spin_ms = (time.perf_counter() - ts0) / max(1, spin) * 1e3
    spin_reported = spin
    spin += 4       # the four cold steps kept the slot parity
    for k in range(a.warmup):
        step(spin + k)
    torch.cuda.synchronize()
    gpu_state = gpu_state.finish() if gpu_state else None
    if dist:
        dist.barrier()
    bank.enable_timing(1)      # HIP events around the filter kernel only: two stream operations per step
    bank.timing(reset=True)
    t0 = time.perf_counter()
    reader, reading = None, {}
    for k in range(a.steps):
        if rank == 0 and not a.gpu_state and k == max(0, a.steps - max(8, a.steps // 4)):
            # one reading under load, from a thread of its own: the files take a millisecond or two to read, and the host
            # is never more than four steps ahead of the device
            import threading
            reader = threading.Thread(target=lambda: reading.update(state=GpuState.once(pci_bus)), daemon=True)
            reader.start()
        step(spin + a.warmup + k)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t1 = time.perf_counter()
    if reader:
        reader.join()
        gpu_state = reading.get("state")
    elapsed = t1 - t0
    elapsed_local = elapsed
    rank_ms = [elapsed / a.steps * 1e3] * 2      # fastest / slowest rank
    if dist:
        tt = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        tmin = tt.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed = float(tt.item())
        rank_ms = [float(tmin.item()) / a.steps * 1e3, elapsed / a.steps * 1e3]
    fan_stats = fan.stats() if use_c else None
    tm = bank.timing(reset=True)
    # Per-rank diagnostics (a sub-linear scaling result must be readable from the one line): a few more untimed steps with
    # the consumer stream's waits for its batch timed (two event records per acquire that has to wait), then every rank's
    # step time, filter-kernel time, broadcast time and stall time travel to rank 0.
    per_rank = None
    if world > 1 or a.per_rank:
        nd = 40
        if use_c:
            fan.enable_timing(True)
        for k in range(nd):
            step(spin + a.warmup + a.steps + k)
        torch.cuda.synchronize()
        if use_c:
            fan.enable_timing(False)
            fs2 = fan.stats()
        else:      # the torch twin keeps no times of its own: step and kernel time only
            fs2 = {"broadcast_ms": float("nan"), "broadcasts": 0, "wait_ms": float("nan"), "waits": 0, "waits_dropped": 0}
        mine = torch.tensor([elapsed_local / a.steps * 1e3, tm["filter_ms"] / max(1, tm["filter_launches"]),
                             fs2["broadcast_ms"] / max(1, fs2["broadcasts"]), float(fs2["broadcasts"]),
                             fs2["wait_ms"] / nd, float(fs2["waits"]), float(nd), float(fs2["waits_dropped"])],
                            device=tdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if dist:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]

        def num(x, digits=4):
            x = float(x)
            return None if x != x else round(x, digits)

        per_rank = [{"rank": r, "ms_per_step": num(v[0]), "kernel_ms": num(v[1]),
                     "bcast_ms": num(v[2]), "bcasts_timed": int(v[3]),
                     "wait_ms_per_step": num(v[4]), "waits": int(v[5]), "diag_steps": int(v[6]), "waits_untimed": int(v[7])}
                    for r, v in enumerate(allr)]
        a_steps_done = a.steps + nd
    else:
        a_steps_done = a.steps
    # N > 1: the root's batch from PINNED HOST memory (what a socket reader fills): kq_fanout_post copies it into the slot on
    # the side stream and broadcasts it from there -- H2D, ncclBroadcast and the ranks' compute in one pipeline, which the
    # resident steps above (the root re-broadcasts a slot in place) do not exercise (VERDICT r5 #6a).  Beside `value`, never it.
    host_ingest = None
    if use_c and ((world > 1 and a.ingest in ("host", "both")) or (world == 1 and a.ingest == "host")):
        iq_pin_all = torch.from_numpy(np.ascontiguousarray(iq_host)).pin_memory() if rank == 0 else None
        src = iq_pin_all.data_ptr() if rank == 0 else None

        def step_host(k):
            i = k & 1
            p = fan.acquire(i, cs)
            bank.process_resident(p, B)
            fan.release(i, cs)
            fan._chk(lib.kq_fanout_post(fan.h, i, src, nwin, 0), "kq_fanout_post (host batch)")

        k0 = spin + a.warmup + a_steps_done
        k0 += k0 & 1
        for k in range(8):
            step_host(k0 + k)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        th0 = time.perf_counter()
        for k in range(a.steps):
            step_host(k0 + 8 + k)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        th_s = time.perf_counter() - th0
        if dist:
            th = torch.tensor([th_s], device=tdev, dtype=torch.float64)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            th_s = float(th.item())
        a_steps_done += (k0 - (spin + a.warmup + a_steps_done)) + 8 + a.steps     # (the slots keep alternating)
        host_ingest = {"ms_per_step": round(th_s / a.steps * 1e3, 4), "steps": a.steps,
                       "value": round(C * world * B * L * a.steps / th_s / 1e6, 1),
                       "h2d_bytes_per_step": nwin * 8,
                       "note": "every step's batch leaves pinned host memory on the root: kq_fanout_post(src_is_device = 0) = H2D into "
                               "the slot + ncclBroadcast on the side stream, two slots deep, beside the ranks' compute"}
    # the demodulator kernels' time comes from a few extra, untimed steps with the full set of events
    bank.enable_timing(2)
    for k in range(4):       # (an even count: the slots keep their parity)
        step(spin + a.warmup + a_steps_done + k)
    torch.cuda.synchronize()
    tm2 = bank.timing(reset=True)
    bank.enable_timing(0)

    # The path to its real ends (1 GPU only): the reference's path starts at a host packet (radio.c:106-147) and ends in a
    # host float buffer per block (audio.c:82).  Same step, but the batch comes from pinned host memory
    # (kq_bank_push_iq_async -> ring -> kq_bank_process) and audio + status of every step go back to pinned host memory
    # (kq_bank_pull_planes_async); the copies ride on the bank's copy streams under the kernels.  Beside `value`, never it.
    host_io = host_io_pcm = None
    if world == 1 and not a.no_host_io:
        import ctypes
        olen = L // D
        iq_pin = torch.from_numpy(np.ascontiguousarray(iq_host[M - 1:M - 1 + B * L])).pin_memory()
        status_pin = torch.empty(C * B * ctypes.sizeof(kq.ChanStatus), dtype=torch.uint8).pin_memory()

        def host_io_row(pcm):
            """pcm: the int16 PCM plane (the reference's own output format, audio.c:22-28: clipped, network byte order)
            + silent-chunk masks instead of float audio -- half the bytes over the link"""
            if pcm:
                out_pin = torch.zeros(C * B * 2 * olen, dtype=torch.int16).pin_memory()
                mask_pin = torch.zeros(C * B, dtype=torch.int32).pin_memory()
            else:
                out_pin = torch.zeros(C * B * 2 * olen, dtype=torch.float32).pin_memory()

            def io_step():     # the call order include/ka9q_hip.h asks for: the next batch is on its way before this one's planes leave
                assert bank.process() == B
                bank.push_iq_async(iq_pin.data_ptr(), B * L)
                if pcm:
                    bank.pull_pcm_planes_async(out_pin.data_ptr(), mask_pin.data_ptr(), status_pin.data_ptr())
                else:
                    bank.pull_planes_async(out_pin.data_ptr(), status_pin.data_ptr())

            n_io = max(2, min(50, a.steps))
            bank.push_iq_async(iq_pin.data_ptr(), B * L)
            for k in range(20):
                io_step()
            bank.host_io_wait()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for k in range(n_io):
                io_step()
            bank.host_io_wait()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t3) / n_io
            bank.process()      # (the batch the last step pushed: the ring is empty for whoever comes next)
            torch.cuda.synchronize()
            # of each channel-block's 2 olen words the nout that hold samples travel (mono olen, stereo 2 olen), plus the status plane
            h2d = iq_pin.numel() * 8
            words = sum((2 if p.get("channels", 1) == 2 else 1) * olen for p in plan) * B
            d2h = words * (2 if pcm else 4) + status_pin.numel() + (4 * C * B if pcm else 0)
            return {"value": round(C * B * L / dt / 1e6, 1), "unit": "Msamples/s (channel-samples)",
                    "ms_per_step": round(dt * 1e3, 4), "steps": n_io,
                    "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": d2h,
                    "h2d_GBps": round(h2d / dt / 1e9, 2), "d2h_GBps": round(d2h / dt / 1e9, 2),
                    "audio_checksum": float(out_pin[:C * B * 2 * olen:997].to(torch.float64).abs().sum()),
                    "note": "input from pinned host memory (kq_bank_push_iq_async + kq_bank_process, ring path with its "
                            "history copy), %s + status [C][B] planes to pinned host memory every step (%s); copies on the "
                            "bank's copy streams, overlapped with the kernels" %
                            (("int16 PCM words (audio.c:22-28: clipped, network byte order; nout per channel-block) + silent-chunk masks",
                              "kq_bank_pull_pcm_planes_async") if pcm else
                             ("audio (nout floats per channel-block)", "kq_bank_pull_planes_async"))}

        host_io = host_io_row(False)
        host_io_pcm = host_io_row(True)

    # Secondary row (1 GPU only): the same workload with compute_n0 switched the other way.  The headline computes
    # the noise estimate of radio.c:383-425 on every channel-block, as the reference's demod threads do, which needs
    # all N bins of every channel's mixed spectrum (full forward transform); without it the bank prunes the forward
    # transform to the N/D bins the slave reads.  Reported beside the headline, never as `value`.
    fwd_used = {1: "full", 2: "pruned"}[bank.fwd_mode]
    second = None
    if world == 1 and not a.no_second_row and a.fwd == "auto":
        bank.close()
        bank = kq.Bank(fs, L, M, D, C, B, device=dev_index, compute_n0=not a.n0, fwd_mode=kq.KQ_FWD_AUTO,
                       stream=stream.cuda_stream, pl_tone=pl_tone)
        for p in plan:
            bank.add_channel(wl.bank_channel_config(p))
        fwd2 = {1: "full", 2: "pruned"}[bank.fwd_mode]
        n2 = max(2, min(50, a.steps))
        for k in range(2 + spin // 4):     # setting the bank up again left the GPU idle: back to sustained clocks
            bank.process_resident(bufs[0].data_ptr(), B)
        torch.cuda.synchronize()
        bank.enable_timing(1)
        bank.timing(reset=True)
        t2 = time.perf_counter()
        for k in range(n2):
            bank.process_resident(bufs[0].data_ptr(), B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t2) / n2
        tm_b = bank.timing(reset=True)
        second = {"compute_n0": int(not a.n0), "fwd": fwd2, "value": round(C * B * L / dt / 1e6, 1),
                  "unit": "Msamples/s (channel-samples)", "ms_per_step": round(dt * 1e3, 4), "steps": n2,
                  "kernel_ms": round(tm_b["filter_ms"] / max(1, tm_b["filter_launches"]), 4)}

    # BASELINE.json's other single-GPU shapes, each measured the same way on this box in this run
    rows = None
    if world == 1 and not a.no_rows and a.config == "cfg4" and not under_profiler:
        rows = {}
        for name, blocks in (("cfg2", 64), ("cfg3", 64), ("cfg5", 16)):
            rows[name] = measure_row(torch, kq, wl, name, blocks, dev_index, stream, pci_bus)

    # "channels @ real-time", the second half of BASELINE.json's metric, measured (1 GPU, default workload only)
    realtime = None
    if world == 1 and not a.no_realtime and a.config == "cfg4" and not under_profiler:
        bank.close()
        realtime = measure_channels_at_realtime(torch, kq, wl, dev_index, stream, a.realtime_seconds)

    if rank == 0:
        total_ch = C * world
        chan_samples = total_ch * B * L * a.steps
        value = chan_samples / elapsed / 1e6
        front_end_msps = B * L * a.steps / elapsed / 1e6
        per_kind = {}
        for p in plan:
            per_kind[p["demod"]] = per_kind.get(p["demod"], 0) + 1
        def roofline(k_ms, n0, fwd_name, demod_ms=None):
            return roofline_of(wl, a.config, geom, plan, C, B, k_ms, n0, fwd_name, demod_ms)

        k_ms = tm["filter_ms"] / max(1, tm["filter_launches"])
        out = {
            "metric": "input Msamples/s + channels @ real-time, 16384-pt overlap-save",
            "value": round(value, 1),
            "unit": "Msamples/s (channel-samples: front-end input samples x channels, all GPUs)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "spinup_steps": spin_reported,
            "spinup_ms_per_step": round(spin_ms, 4),
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s: %d channels/GPU (%s), N=%d (L=%d, M=%d), decimate %d, %.3g MS/s synthetic complex-float "
                            "I/Q, %d blocks/step, fwd=%s, compute_n0=%d, pltask %s" %
                            (a.config, C, "+".join("%d %s" % (v, k) for k, v in sorted(per_kind.items())), L + M - 1, L, M, D,
                             fs / 1e6, B, fwd_used, a.n0, "on" if pl_tone else "off (SURVEY 8d)"),
                "channels_total": total_ch,
                "front_end_Msamples_per_s": round(front_end_msps, 2),
                "realtime_factor": round(front_end_msps * 1e6 / fs, 2),
                "channels_at_realtime": (realtime["channels"] if realtime else None),
                "channels_at_realtime_note": "MEASURED, = realtime.channels: one bank, 2 blocks per call, host I/O, paced, zero late "
                                             "deliveries (null when the realtime legs were not run: N > 1, --no-realtime, profiler)",
                "channels_x_factor_extrapolated": int(total_ch * front_end_msps * 1e6 / fs),
                "parallelism": "channels sharded x%d, front-end I/Q broadcast over RCCL" % world if world > 1 else "1 GPU",
            },
            "roofline": roofline(k_ms, bool(a.n0), fwd_used, tm2["demod_ms"] / max(1, tm2["filter_launches"])),
            "cold_first_steps_ms": cold,
            "gpu_state": gpu_state,
            "ms_per_step_ranks": {"min": round(rank_ms[0], 4), "max": round(rank_ms[1], 4)},
            "fanout": "kq_fanout (C ABI, ncclBroadcast on the library's side stream)" if use_c else
                      "torch.distributed.broadcast (%s)%s" % (a.backend, ("; kq_fanout unavailable: " + fan_error) if fan_error else ""),
        }
        if fan_stats is not None:
            out["rccl"] = {"ranks": fan_stats["rccl_ranks"], "world": world, "version": fan_stats["rccl_version"],
                           "broadcasts": fan_stats["broadcasts"],
                           "bcast_ms": round(fan_stats["broadcast_ms"] / max(1, fan_stats["broadcasts"]), 4),
                           "bytes_per_broadcast": nwin * 8, "max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                           "library": (lib.kq_fanout_rccl_path() or b"").decode(),
                           "libraries_mapped": sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}),
                           "note": "ranks = ncclCommCount of the fan-out's communicator (0: one rank, no communicator); "
                                   "bcast_ms = HIP events around ncclBroadcast on the side stream, rank 0"}
        # the scaling curve is a statement about RCCL over xGMI: it is measured only when RCCL itself carried the batches
        # between `world` ranks (kq_fanout_stats.rccl_ranks = ncclCommCount)
        out["scaling_measured"] = bool(world == 1 or (fan_stats is not None and fan_stats["rccl_ranks"] == world))
        if host_ingest:
            host_ingest["fraction_of_value"] = round(host_ingest["value"] / max(value, 1e-9), 4)
            out["host_ingest"] = host_ingest
        out["step_frac"] = round(out["roofline"]["algorithmic_bytes_per_launch"] / (elapsed / a.steps) / 8e12, 4)
        if per_rank is not None:
            out["per_rank"] = per_rank
        if realtime:
            out["realtime"] = realtime
        if rows:
            out["rows"] = rows
        if host_io:
            host_io["fraction_of_value"] = round(host_io["value"] / max(value, 1e-9), 4)
            out["with_host_io"] = host_io
        if host_io_pcm:
            host_io_pcm["fraction_of_value"] = round(host_io_pcm["value"] / max(value, 1e-9), 4)
            out["with_host_io_pcm"] = host_io_pcm
        if second:
            second["roofline"] = roofline(second["kernel_ms"], bool(second["compute_n0"]), second["fwd"])
            second["note"] = ("the same workload %s compute_n0 (radio.c:383-425, status only): the bank then runs its %s "
                              "forward path" % ("with" if second["compute_n0"] else "without", second["fwd"]))
            out["with_compute_n0" if second["compute_n0"] else "without_compute_n0"] = second
        if not a.no_cpu_baseline and world == 1:   # the CPU leg runs on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(a.config, geom, plan, iq_host[M - 1:], a.cpu_seconds, compute_n0=a.n0)
            # SURVEY 8d's second row: compute_n0 the other way, next to the GPU row configured the same way
            other = cpu_baseline(a.config, geom, plan, iq_host[M - 1:], a.cpu_seconds / 2, compute_n0=int(not a.n0))
            out["cpu_baseline"]["with_compute_n0" if not a.n0 else "without_compute_n0"] = {
                k: other[k] for k in ("value", "unit", "cores", "per_core", "sample")}
        print(json.dumps(out), flush=True)
    bank.close()
    if use_c:
        fan.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
