#!/usr/bin/env python3
"""
Time of one variational iteration (the reference's own algorithm: q_F update, q_R update, pi / gamma, tables, energy;
fcdiff/fit.py:56-82 as documented in doc/methods.rst) at cfg3 on the GPU, and of the NumPy restatement of the same
iteration (oracle/fcdiff_oracle.py, one host core) on a smaller problem, scaled by the number of (edge, patient) terms.

    python profiles/vb_iter.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fcdiff_amd  # noqa: E402


def gpu_iteration(Nreg, H, U, iters=20):
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
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / iters * 1e3
    t0 = time.perf_counter()
    for _ in range(iters):
        for fn in parts.values():
            fn()
    torch.cuda.synchronize()
    out["iteration"] = (time.perf_counter() - t0) / iters * 1e3
    return out


def cpu_iteration(Nreg, H, U):
    from oracle import fcdiff_oracle as O
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    th = dict(pi=model.pi, eta=model.eta, epsilon=model.epsilon, gamma=model.gamma, mu=model.mu, sigma=model.sigma)
    t0 = time.perf_counter()
    O.vb_fit(b, bt, th, max_iters=1, check_convergence=False)
    return (time.perf_counter() - t0) * 1e3


def main():
    g = gpu_iteration(200, 50, 50)
    print("GPU, cfg3 (Nreg=200, H=U=50), ms per call: " + ", ".join("%s %.3f" % kv for kv in g.items()))
    (n, h, u) = (60, 15, 15)
    c = cpu_iteration(n, h, u)
    scale = (200 * 199 / 2 * 50) / (n * (n - 1) / 2 * u)
    print("NumPy restatement, one core, Nreg=%d H=U=%d: %.0f ms per iteration (incl. the initial energy) -> x%.0f terms at "
          "cfg3 ~ %.1f s" % (n, h, c, scale, c * scale / 1e3))


if __name__ == "__main__":
    main()
