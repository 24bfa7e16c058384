#!/usr/bin/env python3
"""
Phase ablation of the sweep kernels (diagnostic; needs `make -C fcdiff_amd/csrc ABLATE=1`).

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/ablate.py

Times the f pass and the r pass of the cfg3 workload with parts of the kernels switched off
(levels: 0 full, 2 = no draw / loads only / no in-order part, 3 = staging only).  Interleaved rounds in one
process; medians printed.  Results of ablated runs are wrong by design -- only the times matter.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fcdiff_amd  # noqa: E402
from fcdiff_amd.gibbs import GibbsEngine  # noqa: E402


def main():
    (Nreg, H, U, G) = (200, 50, 50, 1024)
    if len(sys.argv) >= 4:                       # e.g. `ablate.py 400 250 250` = the cfg5 share
        (Nreg, H, U) = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=fit._context())
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    for s in range(3):
        eng.sweeps(s, 1)
    torch.cuda.synchronize()
    cases = [("f full", "f", {}), ("f no-draw", "f", {"FCD_ABL_F": "2"}), ("f staging-only", "f", {"FCD_ABL_F": "3"}),
             ("r full", "r", {}), ("r full, one patient per panel workgroup", "r", {"r_ub": 1}), ("r full, no empty workgroups beside D", "r", {"r_nopad": 1}),
             ("r panel loads-only", "r", {"FCD_ABL_PANEL": "2"}), ("r panel staging-only", "r", {"FCD_ABL_PANEL": "3"}),
             ("r panel single rows only", "r", {"FCD_ABL_PANEL": "4"}), ("r panel empty", "r", {"FCD_ABL_PANEL": "5"}),
             ("r diag no next-thresholds", "r", {"FCD_ABL_DIAG": "1"}), ("r diag empty", "r", {"FCD_ABL_DIAG": "5"}),
             ("r panel empty + diag empty", "r", {"FCD_ABL_PANEL": "5", "FCD_ABL_DIAG": "5"}),
             ("r diag no-chain", "r", {"FCD_ABL_DIAG": "2"}), ("r diag prologue-only", "r", {"FCD_ABL_DIAG": "3"}),
             ("r panel staging-only + diag prologue-only", "r", {"FCD_ABL_PANEL": "3", "FCD_ABL_DIAG": "3"})]
    res = {name: [] for (name, _, _) in cases}
    for rnd in range(5 if Nreg <= 200 else 2):
        for (name, which, env) in cases:
            for k in ("FCD_ABL_F", "FCD_ABL_PANEL", "FCD_ABL_DIAG"):
                os.environ.pop(k, None)
            knobs = {"r_ub": 0, "r_nopad": 0, "r_path": 3}     # (fcd_ctx_set_knob; step-per-launch form)
            knobs.update({k: v for (k, v) in env.items() if not k.startswith("FCD_")})
            for (k, v) in knobs.items():
                eng.ctx.set_knob(k, v)
            os.environ.update({k: v for (k, v) in env.items() if k.startswith("FCD_")})
            fn = eng.f_step if which == "f" else eng.r_step
            fn(100)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                fn(101 + i)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    for (name, _, _) in cases:
        print("%-46s %8.1f us (min %8.1f)" % (name, float(np.median(res[name])), float(np.min(res[name]))))


if __name__ == "__main__":
    main()
