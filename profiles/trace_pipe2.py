#!/usr/bin/env python3
"""
Timeline of the in-order role of the pipelined r pass at cfg3 (diagnostic; needs `make -C fcdiff_amd/csrc ABLATE=1`):

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/trace_pipe2.py [alone]

Lane 0 of the first and of the last wave of every in-order workgroup stamps the 100 MHz clock at 8 points of every block:
0 block start, 1 requests issued, 2 block b-1 announced, 3 barrier passed, 4 tiles staged and built, 5 row 8 of the scan,
6 scan done, 7 stores issued.  "alone": the panel workgroups stay empty and no hand-over is waited for.
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
    alone = len(sys.argv) > 1 and sys.argv[1] == "alone"
    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()
    ctx = fit._context()
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=ctx)
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    for s in range(3):
        eng.sweeps(s, 1)
    torch.cuda.synchronize()
    if alone:
        os.environ["FCD_TRACE_ROW"] = "2"
        os.environ["FCD_ABL_PANEL"] = "7"
    nl = 32
    buf = torch.zeros((nl * 1024, 8), dtype=torch.int64, device="cuda")
    os.environ["FCD_TRACE_PTR"] = hex(buf.data_ptr())
    eng.r_step(100)
    torch.cuda.synchronize()
    buf.zero_()
    eng.r_step(101)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nl, 1024, 8).astype(np.float64)
    nD = U
    t0 = t[:, :nD, 0][t[:, :nD, 0] > 0].min()
    names = ["issue", "announce", "barrier", "stage+build", "rows 0-7", "rows 8-15", "stores"]
    for (w, off) in (("first wave", 0), ("last wave", 16)):
        print(w + ": block | start (us) | " + " | ".join("%11s" % n for n in names) + " | block total")
        for bb in range(13):
            x = t[off + bb, :nD, :]
            ok = x[:, 0] > 0
            if not ok.any():
                continue
            x = x[ok]
            d = np.diff(x, axis=1) / 100.0
            nxt = t[off + bb + 1, :nD, 0][ok] if bb < 12 else x[:, 7]
            print("%s  %2d | %9.2f | " % (" " * len(w), bb, np.median(x[:, 0] - t0) / 100.0) +
                  " | ".join("%11.2f" % v for v in np.median(d, axis=0)) + " | %8.2f" % (np.median(nxt - x[:, 0]) / 100.0))


if __name__ == "__main__":
    main()
