#!/usr/bin/env python3
"""
bench.py -- posterior samples/sec of the many-chain Gibbs fit path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "cfg3"): Nreg=200 regions (C=19 900 edges), 100 subjects split
H=U=50 (SURVEY.md section 8 assumption), 1024 chains PER GPU (weak scaling: chains shard, tables replicate).
Synthetic data from the model's own sampler at the model defaults (fcdiff/model.py:33-38), float64.

One step = one full sweep of every chain on this GPU -- all C f_c draws, all Nreg*U r_nu draws -- plus the
pooled statistics, their all-reduce over ranks (RCCL), the (pi, gamma) M-step and the marginal
counters.  One posterior sample = one sweep of one chain; value = chains * steps / time over all ranks.
The likelihood tables depend only on (mu, sigma, eta, epsilon), which this loop holds fixed, so they are
built once before the timed region (and timed separately: "lik_tables").

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the step (the r pass or the f pass,
whichever took longer), timed with HIP events on the launch stream inside the timed region; algorithmic
bytes per launch follow SURVEY.md section 8d (u8 state).  `cpu_baseline` is the C restatement of the same
sampler (oracle/fcdiff_oracle.c, OpenMP over chains) on the host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LDS_CLOCK_GHZ = 2.4          # MI355X peak engine clock
LDS_BYTES_PER_CLK_CU = 256   # profiles/r01_ubench_lds_fp64.txt: 250 B/clk/CU with ds_read_b128 (8-byte reads top out at ~180)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nreg", type=int, default=200)
    ap.add_argument("--subjects", type=int, default=100)
    ap.add_argument("--chains-per-gpu", type=int, default=1024)
    ap.add_argument("--mstep-every", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chains", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import fcdiff_amd
    from fcdiff_amd import _lib
    from fcdiff_amd.gibbs import GibbsEngine, allreduce_counts

    (Nreg, H, U) = (args.nreg, args.subjects // 2, args.subjects - args.subjects // 2)
    G = args.chains_per_gpu
    C = fcdiff_amd.N_to_C(Nreg)
    chain0 = rank * G
    seed = 20240601
    # which BASELINE.json config this is (configs[2] = "cfg3" is the one the metric is quoted on)
    if (Nreg, H, U, G) == (200, 50, 50, 1024):
        cfg_name = "cfg3" if world == 1 else "cfg4-style (cfg3 per GPU, %d GPUs)" % world
    elif (Nreg, H, U, G) == (400, 250, 250, 1024):
        cfg_name = "cfg5 per-GPU share (1024 of the 8192 chains)"
    elif (Nreg, H, U, G) == (64, 16, 16, 256):
        cfg_name = "cfg2"
    else:
        cfg_name = "custom"

    model = fcdiff_amd.UnsharedRegionModel()
    (_r, _t, _f, _ft, b, bt) = model.sample_fast(Nreg, H, U, seed=0)     # same data on every rank

    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = model, b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()                                                     # K_lik
    torch.cuda.synchronize()

    ctx = fit._context()
    # K_lik timed on its own (it runs once per theta_sub change, outside the sweep loop): one launch, events on its stream
    ctx.prof_enable(True)
    for _ in range(10):
        fit._update_lps()
    torch.cuda.synchronize()
    (lik_tot, lik_n) = ctx.prof_collect()["lik_kernel"]
    lik_ms = lik_tot / max(lik_n, 1)
    lik_bytes = 8 * C * (H + U) + 24 * C + 72 * C * U
    ctx.prof_enable(False)

    eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], Nreg, U, G, chain0=chain0, seed=seed, edge_index="symmetric", ctx=ctx)
    eng.set_hyper(model.gamma, model.pi2())
    eng.init(float(model.pi))

    ev = {"f": [], "r": []}

    def step(s, timed):
        # timed region: the fused driver (f pass + r pass, fcd_gibbs_sweeps) exactly as fit / run_chains use it;
        # profile pass: the same two passes as separate calls with an event between them (f / r split)
        if timed:
            eng.sweeps(s, 1)
        else:
            a, b_, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            eng.f_step(s)
            b_.record()
            eng.r_step(s)
            c.record()
            ev["f"].append((a, b_))
            ev["r"].append((b_, c))
        do_m = bool(args.mstep_every and (s + 1) % args.mstep_every == 0)
        counts = eng.tally(want_counts=do_m, accumulate=True)      # pooled counts + marginal counters, one pass
        if do_m:
            eng.mstep(allreduce_counts(counts))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s, True)
    fence()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        step(s, True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # Per-kernel durations: one HIP event pair around EVERY f / step launch costs ~3.7 us per event
    # (30 per sweep), which would slow the timed region by ~20 %; so the same K steps are run once more right
    # after it with the pairs enabled, and only that second pass feeds the per-kernel numbers.
    ctx.prof_enable(True)
    for s in range(args.warmup + args.steps, args.warmup + 2 * args.steps):
        step(s, False)
    torch.cuda.synchronize()
    prof = ctx.prof_collect()
    ctx.prof_enable(False)

    f_pass_ms = float(np.mean([a.elapsed_time(b_) for (a, b_) in ev["f"]]))
    r_pass_ms = float(np.mean([a.elapsed_time(b_) for (a, b_) in ev["r"]]))
    kern = {k: {"total_ms": v[0], "launches": v[1], "avg_launch_ms": v[0] / max(v[1], 1)} for (k, v) in prof.items()
            if k != "lik_kernel"}
    # algorithmic bytes, SURVEY.md section 8d (u8 state): lM once per pass + state
    f_bytes = 72 * C * U + 24 * C + G * (C + Nreg * U)
    r_bytes = 72 * C * U + G * (C + 2 * Nreg * U)
    n_step = max(kern["gibbs_r_step_kernel"]["launches"] // max(args.steps, 1), 1)   # step launches per pass
    per_launch_bytes = {"gibbs_f_pair_kernel": f_bytes,
                        # a step launch serves 16 of the Nreg regions' rows of the r pass
                        "gibbs_r_step_kernel": r_bytes / n_step}
    dom = max(("gibbs_f_pair_kernel", "gibbs_r_step_kernel"), key=lambda k: kern[k]["total_ms"])
    dom_ms = kern[dom]["avg_launch_ms"]
    dom_bytes = per_launch_bytes[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")    # HBM bytes per launch from the PMC passes
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # What actually bounds the two sweep kernels (DESIGN.md section e): wave-wide LDS reads.  Bytes the LDS serves per
    # launch (64 lanes x 8 or 16 bytes per read, counted from the loop structure) against the measured LDS peak.
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    lds_peak = n_cu * LDS_BYTES_PER_CLK_CU * LDS_CLOCK_GHZ / 1e3                                   # TB/s
    nblk = (Nreg + 15) // 16
    gw = (G + 63) // 64
    f_lds = C * gw * ((U + 1) // 2) * 64 * 16                                     # one ds_read_b128 per (edge, word, patient pair)
    r_lds_pass = gw * U * Nreg * nblk * 8 * 64 * 8                                # one ds_read_b64 per (region, patient, word, pair of regions)
    lds = {"unit": "TB/s", "peak": lds_peak,
           "gibbs_f_pair_kernel": f_lds / (kern["gibbs_f_pair_kernel"]["avg_launch_ms"] * 1e-3) / 1e12,
           "gibbs_r_step_kernel": r_lds_pass / n_step / (kern["gibbs_r_step_kernel"]["avg_launch_ms"] * 1e-3) / 1e12}
    lds["frac"] = {k: lds[k] / lds_peak for k in ("gibbs_f_pair_kernel", "gibbs_r_step_kernel")}

    out = {
        "metric": "posterior samples/sec at R=200 ROIs, N=100 subj, 1024 chains; 1/2/4/8 GPUs",
        "value": world * G * args.steps / elapsed,
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: Nreg=%d (C=%d edges), H=%d, U=%d, %d chains/GPU, collapsed Gibbs sweep + pooled "
                               "(pi,gamma) M-step every %d sweep(s), fixed tables" % (cfg_name, Nreg, C, H, U, G, args.mstep_every),
                   "chains_per_gpu": G, "chains_total": world * G, "edge_index": "symmetric",
                   "sample_definition": "one sweep of one chain = C f-draws + Nreg*U r-draws"},
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                     "launches_timed": kern[dom]["launches"],
                     "note": "dominant kernel by total time; HIP event pair around every launch (fcd_prof_*) in a second "
                             "pass of the same K steps right after the timed region (the pairs would perturb it). The tables "
                             "are shared by all chains and stay in L2 / Infinity Cache at this size, so the sweep is bound by "
                             "wave-wide LDS reads, not by HBM (DESIGN.md section e, profiles/r01_ubench_lds_fp64.txt)"},
        "kernels": kern,
        "lds_roofline": lds,
        "passes_ms": {"f_pass": f_pass_ms, "r_pass": r_pass_ms,
                      "f_pass_GBps": f_bytes / (f_pass_ms * 1e-3) / 1e9, "r_pass_GBps": r_bytes / (r_pass_ms * 1e-3) / 1e9,
                      "f64_adds_per_s": (2 * C * U * G + 2 * C * U * G) / ((f_pass_ms + r_pass_ms) * 1e-3)},
        "lik_tables": {"kernel": "lik_kernel", "bound": "hbm", "algorithmic_bytes": lik_bytes, "avg_launch_ms": lik_ms,
                       "achieved": lik_bytes / (lik_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": lik_bytes / (lik_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "one launch; event pair on its stream, mean of %d launches" % lik_n},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle as CO
        cores = CO.max_threads()
        n_c = args.cpu_chains or max(cores, 8)
        S_B = fit._d["S_B"].cpu().numpy()
        lM = fit._d["lM"].cpu().numpy()
        lng, lnpi2 = np.log(model.gamma), np.log(model.pi2())
        f_o, r_o = CO.gibbs_init(n_c, Nreg, U, float(model.pi), seed, 0)
        CO.gibbs_f_step(f_o, r_o, S_B, lM, lng, seed, 0)                 # warm-up sweep
        CO.gibbs_r_step(f_o, r_o, lM, lnpi2, seed, 0, 1)
        n_sw, t_c = 0, 0.0
        tc0 = time.perf_counter()
        while True:
            n_sw += 1
            CO.gibbs_f_step(f_o, r_o, S_B, lM, lng, seed, n_sw)
            CO.gibbs_r_step(f_o, r_o, lM, lnpi2, seed, n_sw, 1)
            t_c = time.perf_counter() - tc0
            if t_c > 10.0 or n_sw >= 50:
                break
        out["cpu_baseline"] = {"value": n_c * n_sw / t_c, "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": "%d chains x %d sweeps of the same cfg3 tables with the C restatement "
                                         "(oracle/fcdiff_oracle.c, OpenMP over chains), %.1f s" % (n_c, n_sw, t_c)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
