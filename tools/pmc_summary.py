#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM-side bytes per launch.

  tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [kernel substring]
                       [config channels blocks fwd]      (workload tag that bench.py matches)

Units and corrections as MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-granular units of
1024 B; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so the
read side is doubled (calibrated in the same run on k_block_energy_sum, whose byte count is known:
blocks * L * 8 B).  The two counters come from separate passes (TCC slots).
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    sel = sys.argv[4] if len(sys.argv) > 4 else "k_filter"
    tag = sys.argv[5:9]
    f, w = per_kernel(fetch), per_kernel(write)
    res = {"unit": "bytes per launch", "read_correction": 2.0, "kernels": {}}
    if len(tag) == 4:
        res["workload"] = {"config": tag[0], "channels": int(tag[1]), "blocks": int(tag[2]), "fwd": tag[3]}
    for k in f:
        if "kq::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        rd = f[k] * 1024 * 2.0
        wr = w.get(k, 0.0) * 1024
        res["kernels"][short] = {"fetch_size_raw": f[k], "write_size_raw": w.get(k, 0.0), "read_bytes": rd,
                                 "write_bytes": wr, "hbm_bytes": rd + wr}
        if sel in short:
            res["dominant"] = short
            res["traffic"] = rd + wr
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
