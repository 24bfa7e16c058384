"""debug helper (not a test): r pass of one sweep against the C oracle, mismatches by region / patient / chain word"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import fcdiff_amd
from fcdiff_amd import _lib
from fcdiff_amd.gibbs import GibbsEngine
from oracle import c_oracle as CO

def run(N, U, G, mode="symmetric", knobs=None, sweeps=2):
    ctx = _lib.Context()
    for k, v in (knobs or {}).items():
        ctx.set_knob(k, v)
    m = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = m.sample_fast(N, 3, U, seed=N + U)
    S_B, lM = CO.lik_tables(b, bt, m.theta())
    eng = GibbsEngine(torch.as_tensor(S_B, device="cuda"), torch.as_tensor(lM, device="cuda"), N, U, G, chain0=64, seed=5 + N,
                      edge_index=mode, ctx=ctx)
    eng.set_hyper(m.gamma, m.pi2())
    eng.init(0.3)
    f_o, r_o = CO.gibbs_init(G, N, U, 0.3, 5 + N, 64)
    lng, lnpi2 = np.log(m.gamma), np.log(m.pi2())
    for s in range(sweeps):
        eng.f_step(s)
        CO.gibbs_f_step(f_o, r_o, S_B, lM, lng, 5 + N, s, 64)
        eng.r_step(s)
        CO.gibbs_r_step(f_o, r_o, lM, lnpi2, 5 + N, s, _lib.EDGE_MODES[mode], 64)
        f_g, r_g = eng.export_state()
        bad = (r_g != r_o)
        form = ctx.stat("r_form_last") if hasattr(ctx, "stat") else -1
        print("N=%d U=%d G=%d sweep %d: f ok %s, r mismatches %d of %d  (form %s, err %s)" % (
            N, U, G, s, np.array_equal(f_g, f_o), bad.sum(), bad.size, form, ctx.stat("dev_err") if hasattr(ctx, "stat") else "?"))
        if bad.any():
            print("   by region:", np.nonzero(bad.any(axis=(0, 2)))[0][:40])
            print("   by patient:", np.nonzero(bad.any(axis=(0, 1)))[0][:40])
            print("   by chain word:", np.unique(np.nonzero(bad.any(axis=(1, 2)))[0] // 64))
            print("   first region's count per patient:", bad[:, np.nonzero(bad.any(axis=(0, 2)))[0][0], :].sum(axis=0)[:16])
            return False
    return True

if __name__ == "__main__":
    shapes = [(24, 5, 64), (24, 5, 192), (40, 6, 128), (24, 70, 192), (37, 6, 1024), (97, 5, 1024)]
    for (N, U, G) in shapes:
        run(N, U, G)
