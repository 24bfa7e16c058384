#!/usr/bin/env python3
"""
Per-call time of the five parts of one variational iteration at a given shape (default cfg5's per-GPU share):

    python profiles/vb_kernels.py [Nreg H U]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fcdiff_amd  # noqa: E402


def main():
    (Nreg, H, U) = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (400, 250, 250)
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()
    parts = {"update_lq_F": fit._update_lq_F, "update_lq_R": fit._update_lq_R, "update_theta": fit._update_theta,
             "update_lps": fit._update_lps, "eval_energy": fit._eval_energy}
    for fn in parts.values():
        fn()
    torch.cuda.synchronize()
    out = {}
    for (name, fn) in parts.items():
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 10 * 1e3
    print("Nreg=%d H=%d U=%d, ms per call: %s" % (Nreg, H, U, ", ".join("%s %.3f" % kv for kv in out.items())))


if __name__ == "__main__":
    main()
