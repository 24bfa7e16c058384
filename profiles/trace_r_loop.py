#!/usr/bin/env python3
"""
Where the panel role's loop spends its time (diagnostic build: `make -C fcdiff_amd/csrc ABLATE=1`), step-per-launch form.

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/trace_r_loop.py [Nreg H U]      (default: cfg5's 400 250 250)

Thread 0 of the first 1024 workgroups of every launch stamps start / rows staged / records built / end (100 MHz clock) and
accumulates, per turn of the state-word loop, the shader-clock cycles between: turn start -> next group's loads issued
(issue), -> this group's words landed (wait: s_waitcnt vmcnt), -> this group's terms done (terms: LDS reads + adds).
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
    (Nreg, H, U, G) = (400, 250, 250, 1024)
    if len(sys.argv) >= 4:
        (Nreg, H, U) = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()
    ctx = fit._context()
    ctx.set_knob("r_path", 3)
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=ctx)
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    eng.sweeps(0, 2)
    torch.cuda.synchronize()
    nl = (Nreg + 15) // 16 + 2
    # (a launch's workgroups beyond the first 1024 stamp into the ranges of the launches after it, which overwrite them: margin at the end)
    buf = torch.zeros(((nl + 8) * 1024, 8), dtype=torch.int64, device="cuda")
    os.environ["FCD_TRACE_PTR"] = hex(buf.data_ptr())
    eng.r_step(100)
    torch.cuda.synchronize()
    buf.zero_()
    eng.r_step(101)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nl + 8, 1024, 8)[:nl]
    print("shape Nreg=%d H=%d U=%d, %d chains; per launch, panel workgroups among the first 1024 of the grid" % (Nreg, H, U, G))
    print("step     n   duration us [mean / p90]   stage  build  loop   | loop turns: issue  wait  terms  (us, shader clock at 2.4 GHz)")
    for L in range(nl):
        m = (t[L, :, 6] == 1) & (t[L, :, 0] > 0) & (t[L, :, 3] > 0)
        if not m.any():
            continue
        x = t[L][m].astype(np.float64)
        st, s1, s2, en = [x[:, k] / 100.0 for k in range(4)]
        wait = x[:, 4] / 2400.0
        raw = t[L][m][:, 5]
        terms = (raw & 0xffffffff).astype(np.float64) / 2400.0
        issue = (raw >> 32).astype(np.float64) / 2400.0
        dur = en - st
        print("%4d  %4d   %7.2f / %7.2f          %5.2f  %5.2f  %5.2f  |             %5.2f  %5.2f  %5.2f" % (
            L, int(m.sum()), dur.mean(), np.percentile(dur, 90), (s1 - st).mean(), (s2 - s1).mean(), (en - s2).mean(),
            issue.mean(), wait.mean(), terms.mean()))


if __name__ == "__main__":
    main()
