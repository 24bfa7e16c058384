#!/usr/bin/env python3
"""
Where a tile of the f pass spends its time (diagnostic build: `make -C fcdiff_amd/csrc ABLATE=1`), U <= 64 kernel at cfg3.

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/trace_f.py

Thread 0 of every workgroup (one tile of 8 edges) stamps start / rows staged / records built / end (100 MHz clock) and sums,
over the tile's 8 edges, the shader-clock cycles of: the terms (25 reads + packed adds), the draw (Philox every fourth edge,
exps, compares), the rest (stores, next edge's slot words and slot numbers).
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
    eng.sweeps(0, 2)
    torch.cuda.synchronize()
    buf = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
    os.environ["FCD_TRACE_PTR"] = hex(buf.data_ptr())
    eng.f_step(100)
    torch.cuda.synchronize()
    buf.zero_()
    eng.f_step(101)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    m = (t[:, 0] > 0) & (t[:, 3] > 0)
    x = t[m].astype(np.float64)
    st, s1, s2, en = [x[:, k] / 100.0 for k in range(4)]
    t0 = st.min()
    dur = en - st
    print("tiles %d; kernel %.1f us from first start to last end" % (int(m.sum()), en.max() - t0))
    print("per tile (us): total %.2f [p10 %.2f p90 %.2f] = staged %.2f + built %.2f + 8 edges %.2f" % (
        dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90), (s1 - st).mean(), (s2 - s1).mean(), (en - s2).mean()))
    print("inside the 8 edges (us, shader clock at 2.4 GHz, wave 0): terms %.2f, draw %.2f, rest (stores, next slot words) %.2f" % (
        x[:, 4].mean() / 2400.0, x[:, 5].mean() / 2400.0, x[:, 6].mean() / 2400.0))
    hw = t[m][:, 7]
    cuid = ((hw >> 16) & 0xf) * 65536 + ((hw & 0xffff) >> 8 & 0xff)
    order = np.argsort(st)
    starts = st[order] - t0
    print("start times of the workgroups (us): first 512 within %.1f; then one every %.3f us on average" % (
        starts[min(511, len(starts) - 1)], (starts[-1] - starts[min(511, len(starts) - 1)]) / max(1, len(starts) - 512)))
    (_, counts) = np.unique(cuid, return_counts=True)
    print("CUs seen %d, tiles per CU min/mean/max %d / %.1f / %d" % (len(counts), counts.min(), counts.mean(), counts.max()))


if __name__ == "__main__":
    main()
