#!/usr/bin/env python3
"""
Per-kernel averages of an SQ counter pass (rocprofv3 --pmc ... --kernel-trace --output-format csv):

    python profiles/pmc_lds_summary.py <counter_collection.csv> "<command line that was profiled>" [summary.json]

Prints, per launch and summed over the chip, the counters of the library's kernels and a few ratios derived from them.
"""
import sys

import pandas as pd

KERNELS = ["gibbs_r_step_kernel", "gibbs_r_pipe_kernel", "gibbs_f_pair_kernel", "gibbs_f_pairx_kernel", "pack_f_kernel", "gibbs_tally_kernel",
           "lik_kernel", "corr_gram_kernel"]
N_CU, N_SIMD = 256, 1024


def main():
    df = pd.read_csv(sys.argv[1])
    print(sys.argv[2] if len(sys.argv) > 2 else "", " (per launch, summed over the chip)")
    js = {}
    for k in KERNELS:
        sub = df[df["Kernel_Name"].str.contains(k, regex=False)]
        if sub.empty:
            continue
        name = sub["Kernel_Name"].iloc[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        n = sub["Dispatch_Id"].nunique()
        avg = sub.groupby("Counter_Name")["Counter_Value"].sum() / n
        print("%s   (%d launches)" % (name, n))
        for (c, v) in avg.items():
            print("    %-28s %12.0f" % (c, v))
        g = lambda c: float(avg.get(c, float("nan")))
        print("    -> bank-conflict cycles / LDS-active cycles = %.3f;  LDS-active cycles per LDS instruction = %.2f;  "
              "LDS-active cycles per CU = %.0f;  VALU instructions per SIMD = %.0f" % (
                  g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_IDX_ACTIVE") / g("SQ_INSTS_LDS"),
                  g("SQ_LDS_IDX_ACTIVE") / N_CU, g("SQ_INSTS_VALU") / N_SIMD))
        js[k] = {"launches": int(n), "per_launch": {c: float(v) for (c, v) in avg.items()}}
    if len(sys.argv) > 3:          # machine-readable copy (bench.py reads SQ_INSTS_VALU from it for `roofline_valu`)
        import json
        js["_command"] = sys.argv[2]
        with open(sys.argv[3], "w") as fh:
            json.dump(js, fh, indent=1)


if __name__ == "__main__":
    main()
