#!/usr/bin/env python3
"""Host time of each call of the several-rank sampler loop (run_chains, lagged schedule) on an RCCL group of one rank:
which call keeps the host from running ahead of the GPU?  (cfg3; no synchronisation inside the loop)"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import fcdiff_amd
    from fcdiff_amd.gibbs import GibbsEngine
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
    eng.run(0, 20, mstep_every=1)
    torch.cuda.synchronize()
    n = 300
    t = {"run": 0.0, "wait": 0.0, "mstep": 0.0, "clone": 0.0, "all_reduce": 0.0}
    pending = None
    t_all0 = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        counts = eng.run(20 + i, 1, mstep_every=0, want_counts=True)
        t1 = time.perf_counter()
        if pending is not None:
            pending[1].wait()
            t2 = time.perf_counter()
            eng.mstep(pending[0])
        else:
            t2 = time.perf_counter()
        t3 = time.perf_counter()
        cl = counts.clone()
        t4 = time.perf_counter()
        work = dist.all_reduce(cl, op=dist.ReduceOp.SUM, async_op=True)
        t5 = time.perf_counter()
        pending = (cl, work)
        t["run"] += t1 - t0; t["wait"] += t2 - t1; t["mstep"] += t3 - t2; t["clone"] += t4 - t3; t["all_reduce"] += t5 - t4
    t_host = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all0
    print("per iteration, host: " + ", ".join("%s %.1f us" % (k, v / n * 1e6) for (k, v) in t.items()) +
          "; host loop %.1f us, with the final synchronise %.1f us" % (t_host / n * 1e6, t_all / n * 1e6))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
