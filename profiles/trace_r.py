#!/usr/bin/env python3
"""
Timeline of one r pass at cfg3 (diagnostic; needs `make -C fcdiff_amd/csrc ABLATE=1`).

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/trace_r.py

Every workgroup of the block step kernel stamps the 100 MHz clock at its start, after staging, after the pair
build and at its end (thread 0).  Prints, per launch: when workgroups start (dispatch skew), how long each role
runs and where, and how many workgroups share a CU.
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
    ctx = fit._context()
    ctx.set_knob("r_path", 3)            # the step-per-launch form (the pipelined one: trace_pipe.py)
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=ctx)
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    for s in range(3):
        eng.sweeps(s, 1)
    torch.cuda.synchronize()
    nl = 16
    buf = torch.zeros((nl * 1024, 8), dtype=torch.int64, device="cuda")
    os.environ["FCD_TRACE_PTR"] = hex(buf.data_ptr())
    eng.r_step(100)
    torch.cuda.synchronize()
    buf.zero_()
    eng.r_step(101)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nl, 1024, 8)
    t0_all = t[..., 0][t[..., 0] > 0].min()
    print("step    role   n   start[min..max] us    end[max] us   dur[mean/max] us  stage  build  terms  wait(mean us)")
    for L in range(nl):
        for role in (2, 3, 1):
            m = (t[L, :, 6] == role) & (t[L, :, 0] > 0)
            if not m.any():
                continue
            x = t[L][m].astype(np.float64)
            st, s1, s2, en = [(x[:, k] - t0_all) / 100.0 for k in range(4)]
            ok = x[:, 3] > 0
            wt = (x[:, 5] - x[:, 4]) / 100.0
            print("%4d    %s %4d   %8.2f .. %8.2f   %8.2f     %6.2f / %6.2f     %5.2f  %5.2f  %5.2f  %5.2f" % (
                L, {1: "P", 2: "D", 3: "F"}[role], int(m.sum()), st.min(), st.max(), en[ok].max() if ok.any() else -1,
                (en - st)[ok].mean(), (en - st)[ok].max(), (s1 - st).mean(), (s2 - s1).mean(), (en - s2)[ok].mean(),
                wt.mean()))
        m = t[L, :, 0] > 0
        if m.any():
            hw = t[L][m][:, 7]
            xcc, cu = (hw >> 16) & 0xf, hw & 0xffff
            cuid = xcc * 65536 + ((cu >> 8) & 0xff)          # (xcc, se/sh/cu bits)
            (_, counts) = np.unique(cuid, return_counts=True)
            print("        CUs used %d, workgroups per CU: %s" % (len(counts), dict(zip(*np.unique(counts, return_counts=True)))))
            if L == 3:
                idx = np.nonzero(m)[0]
                dur = (t[L][m][:, 3] - t[L][m][:, 0]) / 100.0
                role = t[L][m][:, 6]
                by = {}
                for (i, c) in zip(range(len(idx)), cuid):
                    by.setdefault(int(c), []).append(i)
                d1 = [dur[v[0]] for v in by.values() if len(v) == 1 and role[v[0]] == 1]
                d2 = [dur[i] for v in by.values() if len(v) == 2 for i in v if role[i] == 1 and all(role[j] == 1 for j in v)]
                dD = [dur[i] for v in by.values() if len(v) == 2 for i in v if role[i] == 1 and any(role[j] == 2 for j in v)]
                print("        P alone on a CU: %d, %.1f us; P beside P: %d, %.1f us; P beside D: %d, %.1f us" % (
                    len(d1), np.mean(d1) if d1 else 0, len(d2), np.mean(d2) if d2 else 0, len(dD), np.mean(dD) if dD else 0))
                pm = role == 1
                x3 = t[L][m].astype(np.float64)
                ph = [(x3[:, 1] - x3[:, 0]) / 100.0, (x3[:, 2] - x3[:, 1]) / 100.0, (x3[:, 3] - x3[:, 2]) / 100.0]
                print("        P duration percentiles 10/50/90/100: %s" % np.round(np.percentile(dur[pm], [10, 50, 90, 100]), 1))
                slow = pm & (dur > np.percentile(dur[pm], 90))
                print("        slowest 10%%: stage %.1f build %.1f terms %.1f | all: stage %.1f build %.1f terms %.1f" % (
                    ph[0][slow].mean(), ph[1][slow].mean(), ph[2][slow].mean(), ph[0][pm].mean(), ph[1][pm].mean(), ph[2][pm].mean()))
                xs = (hw >> 16) & 0xf
                print("        mean P duration by XCD: %s" % np.round([dur[pm & (xs == k)].mean() for k in range(8)], 1))
                it = idx - 50 - np.where(idx >= 306, 50, 0)   # cfg3: 50 D workgroups first, 50 fillers at 256..305
                print("        mean P duration by row of the block: %s" % np.round([dur[pm & (it % 16 == k)].mean() for k in range(16)], 1))
                print("        start of slowest vs all: %.2f vs %.2f us" % ((x3[slow, 0].mean() - x3[pm, 0].min()) / 100.0, (x3[pm, 0].mean() - x3[pm, 0].min()) / 100.0))
                offs = [abs(int(idx[v[0]]) - int(idx[v[1]])) for v in by.values() if len(v) == 2]
                print("        blockIdx distance of workgroups sharing a CU: %s" % dict(zip(*np.unique(offs, return_counts=True))))


if __name__ == "__main__":
    main()
