#!/usr/bin/env python3
"""
Fixed cost of one timed region of the sampler loop at cfg3: wall time of run(K sweeps) between two device
synchronisations for several K (least-squares line: per-sweep time and the constant), with and without a short
burst of work right before the timed call.

    python profiles/fixed_cost.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fcdiff_amd  # noqa: E402
from fcdiff_amd.gibbs import GibbsEngine, run_chains  # noqa: E402


def main():
    (Nreg, H, U, G) = (200, 50, 50, 1024)
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=fit._context())
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    run_chains(eng, 10, sweep0=0)
    torch.cuda.synchronize()
    s0 = 10
    for label in ("cold (synchronised, idle device)", "right after 5 untimed sweeps (no synchronisation in between)"):
        Ks, ts = [1, 2, 5, 10, 20, 40, 80], []
        for K in Ks:
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize()
                if label.startswith("right"):
                    run_chains(eng, 5, sweep0=s0)
                    s0 += 5
                    ev = torch.cuda.Event(enable_timing=True)
                    ev2 = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    run_chains(eng, K, sweep0=s0)
                    ev2.record()
                    torch.cuda.synchronize()
                    dt = ev.elapsed_time(ev2) * 1e-3
                else:
                    t0 = time.perf_counter()
                    run_chains(eng, K, sweep0=s0)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                s0 += K
                best = min(best, dt)
            ts.append(best)
        (slope, icpt) = np.polyfit(Ks, ts, 1)
        print("%s: %s" % (label, "  ".join("K=%d: %.3f ms" % (k, t * 1e3) for (k, t) in zip(Ks, ts))))
        print("    per sweep %.4f ms, constant %.3f ms" % (slope * 1e3, icpt * 1e3))


if __name__ == "__main__":
    main()
