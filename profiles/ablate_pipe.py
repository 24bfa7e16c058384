#!/usr/bin/env python3
"""
Which strand bounds the pipelined r pass (the default form) at cfg3?  (diagnostic; needs `make -C fcdiff_amd/csrc ABLATE=1`)

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/ablate_pipe.py

Times the pass with the panel terms and / or the in-order terms switched off (hand-over, staging, builds and
thresholds stay).  Results of ablated runs are wrong by design -- only the times matter.
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
    cases = [("step-per-launch form, full", {"r_path": 3}), ("pipelined, full", {"r_path": 0}),
             ("pipelined, one in-order workgroup per patient", {"r_path": 0, "r_dsplit": 1}),
             ("pipelined, no panel terms", {"r_path": 0, "FCD_ABL_PANEL": "2"}),
             ("pipelined, no panel terms, one in-order wg", {"r_path": 0, "r_dsplit": 1, "FCD_ABL_PANEL": "2"})]
    # (round 3: the in-order scan carries no ablation switch any more -- one flag read inside its rows cost more than the rows)
    res = {name: [] for (name, _) in cases}
    for rnd in range(5):
        for (name, env) in cases:
            for k in ("FCD_ABL_F", "FCD_ABL_PANEL", "FCD_ABL_DIAG"):
                os.environ.pop(k, None)
            for (k, v) in env.items():
                if k.startswith("FCD_"):
                    os.environ[k] = v
                else:
                    eng.ctx.set_knob(k, v)
            if "r_dsplit" not in env:
                eng.ctx.set_knob("r_dsplit", 0)
            eng.r_step(100)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                eng.r_step(101 + i)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    for (name, _) in cases:
        print("%-40s %8.1f us (min %8.1f)   [pass incl. the packing launch]" % (name, float(np.median(res[name])), float(np.min(res[name]))))


if __name__ == "__main__":
    main()
