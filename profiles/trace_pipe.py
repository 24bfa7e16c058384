#!/usr/bin/env python3
"""
Timeline of one pipelined r pass (the default form) at cfg3 (diagnostic; needs `make -C fcdiff_amd/csrc ABLATE=1`).

    FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so python profiles/trace_pipe.py

Thread 0 of every workgroup stamps the 100 MHz clock per step: P: start, records built, rows of the next step parked,
marks seen, sums stored; D: start, barrier passed, tiles staged, tiles built, chain done, block announced.
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
    ctx.set_knob("r_path", 0)
    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=ctx)
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))
    for s in range(3):
        eng.sweeps(s, 1)
    torch.cuda.synchronize()
    nl = 32
    buf = torch.zeros((nl * 1024, 8), dtype=torch.int64, device="cuda")
    os.environ["FCD_TRACE_PTR"] = hex(buf.data_ptr())
    eng.r_step(100)
    torch.cuda.synchronize()
    buf.zero_()
    eng.r_step(101)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nl, 1024, 8).astype(np.float64)
    t0 = t[..., 0][t[..., 0] > 0].min()
    us = lambda x: (x - t0) / 100.0
    print("step role  n | start med [min..max] | P: build  park  wait  terms | D: barrier+stage+build  chain  announce | end med / max")
    for L in range(16):
        for role in (2, 1):
            m = ((t[L, :, 6].astype(np.int64) & 255) == role) & (t[L, :, 0] > 0)
            if not m.any():
                continue
            x = t[L][m].copy()
            st, en = us(x[:, 0]), us(x[:, 4])
            if role == 1:
                # steps 0 and 1 poll no mark (nothing above them has been redrawn): their 'marks seen' stamp is never
                # written -- take 'rows parked' for it (wait = 0) instead of printing the difference to an empty word
                x[:, 3] = np.where(x[:, 3] > 0, x[:, 3], x[:, 2])
                print("%3d   P %4d | %7.2f [%7.2f..%7.2f] | %5.2f %5.2f %5.2f %5.2f | %s | %7.2f / %7.2f" % (
                    L, m.sum(), np.median(st), st.min(), st.max(), np.median(x[:, 1] - x[:, 0]) / 100, np.median(x[:, 2] - x[:, 1]) / 100,
                    np.median(x[:, 3] - x[:, 2]) / 100, np.median(x[:, 4] - x[:, 3]) / 100, " " * 38, np.median(en), en.max()))
            else:
                print("%3d   D %4d | %7.2f [%7.2f..%7.2f] | %s | %4.2f+%4.2f+%4.2f %5.2f %5.2f           | %7.2f / %7.2f" % (
                    L, m.sum(), np.median(st), st.min(), st.max(), " " * 23, np.median(x[:, 3] - x[:, 0]) / 100,
                    np.median(x[:, 5] - x[:, 3]) / 100, np.median(x[:, 1] - x[:, 5]) / 100,
                    np.median(x[:, 2] - x[:, 1]) / 100, np.median(x[:, 4] - x[:, 2]) / 100,
                    np.median(en), en.max()))
    # inside row 8 of the in-order scan (wave 0 of every D workgroup): stamps 0 start, 1 f words there, 2 e there,
    # 3 tile A summed, 4 tile B summed, 5 decided
    for b in range(3, 10):
        x = t[16 + b]
        ok = x[:, 0] > 0
        if ok.any():            # (only with FCD_TRACE_ROW=1; the stamps themselves cost ~0.5 us each)
            d = np.diff(x[ok][:, :6], axis=1) / 100.0
            print("block %d, row 8: f words %.2f  e %.2f  tile A %.2f  tile B %.2f  decision %.2f  (median us; whole row %.2f)" % (
                (b,) + tuple(np.median(d, axis=0)) + (np.median((x[ok][:, 5] - x[ok][:, 0]) / 100.0),)))
    m = t[0, :, 0] > 0
    hw = t[0][m][:, 7].astype(np.int64)
    role = t[0][m][:, 6].astype(np.int64) & 255
    cuid = ((hw >> 16) & 0xf) * 65536 + ((hw & 0xffff) >> 8 & 0xff)
    by = {}
    for (c, r) in zip(cuid, role):
        by.setdefault(int(c), []).append(int(r))
    kinds = {}
    for v in by.values():
        k = tuple(sorted(v))
        kinds[k] = kinds.get(k, 0) + 1
    print("workgroups per CU by role (1 = P, 2 = D):", kinds)
    # the panel role's step (start of step L+1 minus start of step L) by what shares its CU
    idx = np.nonzero(m)[0]
    kind_of = {}
    for (i, c) in zip(idx, cuid):
        kind_of[int(i)] = tuple(sorted(by[int(c)]))
    for want in sorted(set(kind_of.values())):
        if 1 not in want:
            continue
        wg = [i for i in idx if kind_of[int(i)] == want and (int(t[0, i, 6]) & 255) == 1]
        if not wg:
            continue
        durs, terms, builds = [], [], []
        for L in range(2, 11):
            a, b2 = t[L][wg], t[L + 1][wg]
            okk = (a[:, 0] > 0) & (b2[:, 0] > 0)
            durs.append(np.median((b2[okk, 0] - a[okk, 0]) / 100.0))
            builds.append(np.median((a[okk, 1] - a[okk, 0]) / 100.0))
            terms.append(np.median((a[okk, 4] - a[okk, 2]) / 100.0))
        print("panel workgroups on CUs holding %s (%d): step %.2f us (build %.2f, sums incl. waits %.2f), median over steps 2-10" % (
            want, len(wg), np.median(durs), np.median(builds), np.median(terms)))


if __name__ == "__main__":
    main()
