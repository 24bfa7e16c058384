#!/usr/bin/env python3
"""
HBM traffic per launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python profiles/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced streaming read, which is how every table / state stream here is read -> doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Infinity-Cache hits are counted too (memory-side L2 requests), so at this size
(tables resident in the 256 MiB cache) the figure is an upper bound of true HBM bytes.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0].split("<")[0]


def per_kernel(dirname, counter):
    path = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main(fetch_dir, write_dir, out):
    f = per_kernel(fetch_dir, "FETCH_SIZE")
    w = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if not (k.startswith("gibbs") or k.startswith("lik") or k.startswith("pack") or k.startswith("r_thr")):
            continue
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        res[k] = {"FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
                  "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "launches": f.get(k, (0, 0))[1]}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-24s fetch %10.1f KiB  write %10.1f KiB  -> %8.2f MB per launch (%d launches)"
              % (k, v["FETCH_SIZE_KiB_per_launch"], v["WRITE_SIZE_KiB_per_launch"], v["hbm_bytes_per_launch"] / 1e6, v["launches"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
