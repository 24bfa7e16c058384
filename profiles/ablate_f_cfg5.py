import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import fcdiff_amd
from fcdiff_amd.gibbs import GibbsEngine
(Nreg, H, U, G) = (400, 250, 250, 1024)
model = fcdiff_amd.UnsharedRegionModel()
(_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)
fit = fcdiff_amd.fit.UnsharedRegionFit(); fit.model, fit.b, fit.bt = model, b, bt
fit._init_lps(Nreg, H, U); fit._update_lps()
eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, seed=1, ctx=fit._context())
eng.set_hyper(model.gamma, model.pi2()); eng.init(float(model.pi))
eng.sweeps(0, 2); torch.cuda.synchronize()
for name, env in (("f full", {}), ("f staging+build only", {"FCD_ABL_F": "3"})):
    os.environ.pop("FCD_ABL_F", None); os.environ.update(env)
    eng.f_step(100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): eng.f_step(101 + i)
    e1.record(); torch.cuda.synchronize()
    print("%-24s %8.1f us" % (name, e0.elapsed_time(e1) / 5 * 1e3))
