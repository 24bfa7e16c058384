#!/usr/bin/env python3
"""
bench.py -- posterior samples/sec of the many-chain Gibbs fit path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default = BASELINE.json configs[2], "cfg3"): Nreg=200 regions (C=19 900 edges), 100 subjects split
H=U=50 (SURVEY.md section 8 assumption), 1024 chains PER GPU (weak scaling: chains shard, tables replicate).
`--nreg 400 --subjects 500` is the per-GPU share of configs[4] ("cfg5": 1024 of its 8192 chains).
Synthetic data from the model's own sampler at the model defaults (fcdiff/model.py:33-38), float64.

One step = one full sweep of every chain on this GPU -- all C f_c draws, all Nreg*U r_nu draws -- plus the
pooled statistics, their all-reduce over ranks (RCCL), the (pi, gamma) M-step and the marginal
counters: the loop is fcdiff_amd.gibbs.run_chains, the one UnsharedRegionFit(method='gibbs') runs.
One posterior sample = one sweep of one chain; value = chains * steps / time over all ranks.
The likelihood tables depend only on (mu, sigma, eta, epsilon), which this loop holds fixed, so they are
built once before the timed region (and timed separately: "lik_tables").
Before the W warm-up steps the chains run `--settle` untimed sweeps (default 64; config.settle_sweeps): burn-in from the
random initial state, and the device reaches its steady state -- measured on the box, the first ~25 sweeps after
start-up take ~5 % longer than all later ones (20 timed steps: 0.459 ms after 5 untimed sweeps, 0.435-0.438 ms after
50, 200 or 1000; profiles/fixed_cost.py: the loop itself has no constant cost per call).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the step (the r block step or the f pass,
whichever took longer in total), timed with HIP events on the launch stream; algorithmic bytes per launch follow
SURVEY.md section 8d (u8 state).  `cpu_baseline` is SURVEY section 8d's mode (iii): the C restatement of the same
sampler (oracle/fcdiff_oracle.c, OpenMP over chains) on the host cores, on a bounded sample.  `vb_iteration` puts the
reference's OWN algorithm (one variational iteration, fcdiff/fit.py:75-82) beside it: GPU milliseconds against the
restatement in the reference's structure (mode (i), Python loop over edges) and the whole-array NumPy form (mode (ii)).
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
LDS_BYTES_PER_CLK_CU = 256   # MI355X_MICROARCH.md, LDS: 64 dwords per clock and CU
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured float4 copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64, dense: datasheet figure; profiles/micro/mfma64.hip measures it on the box


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=64,
                    help="untimed sweeps right after the chains are initialised, before the W warm-up steps: burn-in, and the "
                         "device reaches its steady state (measured: the first ~25 sweeps after start-up run ~5 %% slower)")
    ap.add_argument("--nreg", type=int, default=200)
    ap.add_argument("--subjects", type=int, default=100)
    ap.add_argument("--chains-per-gpu", type=int, default=1024)
    ap.add_argument("--mstep-every", type=int, default=1)
    ap.add_argument("--mstep-lag", default="auto", choices=["auto", "0", "1"],
                    help="1: the M-step of sweep s is applied after sweep s+1, its all-reduce overlaps that sweep "
                         "(auto: 1 with several ranks, 0 with one)")
    ap.add_argument("--collective", default="direct", choices=["direct", "torch"],
                    help="several ranks: 'direct' = the library's own RCCL communicator, ncclAllReduce of the pooled counts queued "
                         "on the stream of the sweep kernels inside fcd_gibbs_run (round 4); 'torch' = round 3's loop through "
                         "torch.distributed (counts -> all_reduce on its stream -> fcd_gibbs_mstep)")
    ap.add_argument("--force-pg", action="store_true",
                    help="one rank only: initialise an RCCL process group of ONE rank anyway and run the several-rank loop "
                         "(counts -> all-reduce -> fcd_gibbs_mstep, lagged schedule) on it: what an 8-GPU run does, on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vb", action="store_true", help="skip the variational-iteration comparison")
    ap.add_argument("--no-corr", action="store_true", help="skip the time-series front-end (K_corr)")
    ap.add_argument("--timepoints", type=int, default=1200)
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
    use_pg = world > 1 or args.force_pg
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                         # (--force-pg without a launcher: a group of this one process)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ranks_seen = dist.get_world_size() if use_pg else 1
    direct = args.collective == "direct"
    lag = (1 if (use_pg and not direct) else 0) if args.mstep_lag == "auto" else int(args.mstep_lag)

    import fcdiff_amd
    from fcdiff_amd.gibbs import GibbsEngine, run_chains

    (Nreg, H, U) = (args.nreg, args.subjects // 2, args.subjects - args.subjects // 2)
    G = args.chains_per_gpu
    C = fcdiff_amd.N_to_C(Nreg)
    chain0 = rank * G
    seed = 20240601
    # which BASELINE.json config this is (configs[2] = "cfg3" is the one the metric is quoted on)
    if (Nreg, H, U, G) == (200, 50, 50, 1024):
        cfg_key = "cfg3"
        cfg_name = "cfg3" if world == 1 else "cfg4-style (cfg3 per GPU, %d GPUs)" % world
    elif (Nreg, H, U, G) == (400, 250, 250, 1024):
        cfg_key = "cfg5"
        cfg_name = "cfg5 per-GPU share (1024 of the 8192 chains)"
    elif (Nreg, H, U, G) == (64, 16, 16, 256):
        cfg_key = cfg_name = "cfg2"
    else:
        cfg_key = cfg_name = "custom"

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

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: W warm-up steps, then exactly K steps of the sampler loop ----
    s_w = args.settle                                                     # first warm-up sweep
    if args.settle > 0:
        run_chains(eng, args.settle, sweep0=0, mstep_every=args.mstep_every, burn_in=0, mstep_lag=lag, force_collective=use_pg, direct=direct)
    run_chains(eng, args.warmup, sweep0=s_w, mstep_every=args.mstep_every, burn_in=0, mstep_lag=lag, force_collective=use_pg, direct=direct)
    fence()
    n_alloc0 = ctx.stat("n_alloc")
    t0 = time.perf_counter()
    run_chains(eng, args.steps, sweep0=s_w + args.warmup, mstep_every=args.mstep_every, burn_in=0, mstep_lag=lag, force_collective=use_pg, direct=direct)
    fence()
    elapsed = time.perf_counter() - t0
    allocs_in_timed_region = ctx.stat("n_alloc") - n_alloc0
    ctx.check_device()                        # (a pipelined r pass that gave a wait up would have raised the error word by now)
    r_form = {1: "one launch per block step", 2: "pipelined one-launch form", 3: "one-launch form with counters"}.get(ctx.stat("r_form_last"), "generic")
    rank_ms = [elapsed / args.steps * 1e3]
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        gathered = [torch.zeros_like(tt) for _ in range(ranks_seen)]
        dist.all_gather(gathered, tt)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in gathered]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- the collective, on its own: the 8-word all-reduce of the pooled counts ----
    allreduce_us = None
    if use_pg:
        cts = eng.counts.clone()
        for _ in range(5):
            dist.all_reduce(cts)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(cts)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - ta) / 50 * 1e6
        if ctx.stat("comm_world"):              # the library's own communicator: the same 8 words, queued on the compute stream
            from fcdiff_amd import _lib as _L
            for _ in range(5):
                ctx.call("fcd_allreduce_stats", _L.dptr(cts), _L.stream_ptr())
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(50):
                ctx.call("fcd_allreduce_stats", _L.dptr(cts), _L.stream_ptr())
            torch.cuda.synchronize()
            allreduce_us = (time.perf_counter() - ta) / 50 * 1e6

    # ---- per-kernel durations: one HIP event pair around EVERY f / step / pack launch costs ~3.7 us per event
    # (30+ per sweep), which would slow the timed region by ~20 %; so the same K steps are run once more right after it,
    # pass by pass, with the pairs enabled, and only that second pass feeds the per-kernel numbers.
    ev = {"f": [], "r": [], "t": []}
    ctx.prof_enable(True)
    for s in range(s_w + args.warmup + args.steps, s_w + args.warmup + 2 * args.steps):
        a, b_, c, d = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        a.record()
        eng.f_step(s)
        b_.record()
        eng.r_step(s)
        c.record()
        eng.tally(want_counts=True, accumulate=True)
        d.record()
        ev["f"].append((a, b_))
        ev["r"].append((b_, c))
        ev["t"].append((c, d))
    torch.cuda.synchronize()
    prof = ctx.prof_collect()
    ctx.prof_enable(False)

    f_pass_ms = float(np.mean([x.elapsed_time(y) for (x, y) in ev["f"]]))
    r_pass_ms = float(np.mean([x.elapsed_time(y) for (x, y) in ev["r"]]))
    tally_ms = float(np.mean([x.elapsed_time(y) for (x, y) in ev["t"]]))
    kern = {k: {"total_ms": v[0], "launches": v[1], "avg_launch_ms": v[0] / max(v[1], 1)} for (k, v) in prof.items()
            if k != "lik_kernel" and v[1] > 0}
    f_name = "gibbs_f_pair_kernel"        # event slot of the f pass (either pair form)
    r_name = "gibbs_r_step_kernel"        # event slot of the r pass's main launches: one per block step, or ...
    if r_name in kern and kern[r_name]["launches"] == args.steps:
        kern["gibbs_r_pipe_kernel"] = kern.pop(r_name)        # ... ONE per pass: the pipelined one-launch form ran
        r_name = "gibbs_r_pipe_kernel"
    # algorithmic bytes, SURVEY.md section 8d (u8 state): lM once per pass + state
    f_bytes = 72 * C * U + 24 * C + G * (C + Nreg * U)
    r_bytes = 72 * C * U + G * (C + 2 * Nreg * U)
    n_step = max(kern[r_name]["launches"] // max(args.steps, 1), 1) if r_name in kern else 1   # step launches per pass
    per_launch_bytes = {f_name: f_bytes,
                        # a step launch serves 16 of the Nreg regions' rows of the r pass (the pipelined form: all of them)
                        r_name: r_bytes / n_step}
    have = [k for k in (f_name, r_name) if k in kern]
    roof = None
    dom = None
    if have:
        dom = max(have, key=lambda k: kern[k]["total_ms"])
        dom_ms = kern[dom]["avg_launch_ms"]
        dom_bytes = per_launch_bytes[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of that kernel: NOT measured in this run (PMC passes cannot run beside the timed region) -- read from
        # the summary of two rocprofv3 --pmc passes of this same command committed under profiles/ (pmc_traffic.py)
        traffic = traffic_raw = traffic_src = None
        for tname in ("r04_pmc_traffic_%s.json" % cfg_key, "r03_pmc_traffic_%s.json" % cfg_key, "r02_pmc_traffic_%s.json" % cfg_key):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                try:
                    rec = json.load(open(tpath)).get(dom, {})
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_raw = (rec.get("FETCH_SIZE_KiB_per_launch", 0.0) + rec.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024.0
                    traffic_src = "profiles/" + tname
                except Exception:
                    traffic = None
                if traffic is not None:
                    break
        roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "traffic_raw_counters": traffic_raw,
                "traffic_note": "from an earlier profiled run of the same command, not from this run; 'traffic' = 2 x FETCH_SIZE + "
                                "WRITE_SIZE (the guide's gfx950 correction, calibrated for 16-byte-per-lane streams; this kernel's "
                                "per-lane loads are 8 bytes wide, for which the factor is not calibrated: the truth lies between "
                                "'traffic_raw_counters' = FETCH_SIZE + WRITE_SIZE and 'traffic')",
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                "launches_timed": kern[dom]["launches"],
                "note": "dominant kernel by total time; HIP event pair around every launch (fcd_prof_*) in a second "
                        "pass of the same K steps right after the timed region (the pairs would perturb it). The tables "
                        "are shared by all chains of the GPU, so each table byte feeds every chain's terms: the sweep is "
                        "bound by VALU issue and wave-wide LDS reads, not by HBM (DESIGN.md section e)"}

    # What the two sweep kernels actually lean on (DESIGN.md section e): wave-wide LDS reads and VALU issue.  Bytes the
    # LDS serves per launch (64 lanes x 8 or 16 bytes per read, counted from the loop structure) against its peak.
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    lds_peak = n_cu * LDS_BYTES_PER_CLK_CU * LDS_CLOCK_GHZ / 1e3                                   # TB/s
    nblk = (Nreg + 15) // 16
    gw = (G + 63) // 64
    f_lds = C * gw * ((U + 1) // 2) * 64 * 8                                      # one ds_read_b64 per (edge, word, patient pair): fp32 pair records (round 4)
    r_lds_pass = gw * U * Nreg * nblk * 8 * 64 * 8                                # one ds_read_b64 per (region, patient, word, pair of regions)
    lds = {"unit": "TB/s", "peak": lds_peak}
    if f_name in kern:
        lds[f_name] = f_lds / (kern[f_name]["avg_launch_ms"] * 1e-3) / 1e12
    if r_name in kern:
        lds[r_name] = r_lds_pass / n_step / (kern[r_name]["avg_launch_ms"] * 1e-3) / 1e12
    lds["frac"] = {k: lds[k] / lds_peak for k in (f_name, r_name) if k in lds}
    # ... and against what the LDS was MEASURED to give this access pattern (round 4): a wave-wide 8-byte GATHER -- every lane
    # its own address inside a record -- leaves a CU at one read per 1.75 ns whatever the number of waves that ask and whatever
    # adds the values up (profiles/r02_ubench_lds_fp64.txt, r04_ubench_lds_gather.txt: 6.8-7.5 ns per read and SIMD slot at 4, 8
    # and 16 waves per CU; a gather costs its bytes, 128 B per clock and CU), half the streaming figure above.  The term loops
    # of both kernels are such gathers.
    gather_ns_per_cu = 1.75
    lds["gather_peak"] = n_cu * 64 * 8 / gather_ns_per_cu / 1e3                   # TB/s
    lds["gather_peak_source"] = "profiles/r02_ubench_lds_fp64.txt (b64 gather, panel pattern: 7.01 ns per SIMD slot at 16 waves per CU)"
    lds["frac_of_gather_peak"] = {k: lds[k] / lds["gather_peak"] for k in (f_name, r_name) if k in lds}

    # The bound that binds these two kernels is vector-instruction ISSUE, not bytes: SQ_INSTS_VALU per launch
    # from the committed counter pass of this same command (profiles/pmc_lds_summary.py -> JSON), priced at 2 cycles of a SIMD
    # per wave-instruction for two-source 32-bit work and 4 for fp64 / packed-fp32 / three-source / compare / select
    # (profiles/r03_ubench_valu_rate.txt: 1.42 against 2.44 issue units), the share of the 4-cycle class counted in the
    # kernel's ISA (profiles/r04_valu_mix.json, made by profiles/valu_mix.py from fcd_gibbs*.s), against the kernel's duration
    # measured in THIS run: frac = 1 would be a kernel whose SIMDs issue a vector instruction in every slot.
    valu = None
    vpath = os.path.join(ROOT, "profiles", "r04_pmc_lds_%s.json" % cfg_key)
    mpath = os.path.join(ROOT, "profiles", "r04_valu_mix.json")
    if os.path.exists(vpath):
        try:
            rec = json.load(open(vpath))
            mix = json.load(open(mpath)) if os.path.exists(mpath) else {}
            clock_hz = 2.4e9          # hipDeviceProp_t::clockRate of the box (printed by profiles/micro/vmem_rate: 2.40 GHz); torch does not expose it
            n_simd = 4 * n_cu
            valu = {"unit": "fraction of the SIMDs' issue slots", "source": "profiles/r04_pmc_lds_%s.json" % cfg_key,
                    "mix_source": "profiles/r04_valu_mix.json" if mix else None, "clock_GHz": clock_hz * 1e-9,
                    "note": "SQ_INSTS_VALU from an earlier profiled run of the same command, kernel time from this run"}
            for k in (f_name, r_name):
                if k in kern and k in rec:
                    insts = rec[k]["per_launch"].get("SQ_INSTS_VALU")
                    wide = float(mix.get(k, {}).get("frac_4_cycle", 0.5))
                    cyc = insts * (2.0 + 2.0 * wide) / n_simd
                    valu[k] = {"insts_valu_per_launch": insts, "frac_4_cycle_class": wide, "issue_cycles_per_simd": cyc,
                               "frac": cyc / (kern[k]["avg_launch_ms"] * 1e-3 * clock_hz)}
        except Exception as exc:      # (a malformed summary must not cost the bench line)
            valu = {"error": str(exc)}

    sweep_bytes = 2 * 72 * C * U + 24 * C + G * (2 * C + 3 * Nreg * U)            # per sweep of G chains
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
                   "mstep_lag": lag, "ranks_seen": ranks_seen, "allreduce_us": allreduce_us,
                   "process_group": ("nccl, %d rank(s)%s" % (ranks_seen, ", forced" if (args.force_pg and world == 1) else "")) if use_pg else None,
                   "collective": (("library-owned RCCL communicator: ncclAllReduce(8 x int64) on the stream of the sweep kernels, inside fcd_gibbs_run"
                                   if (direct and not lag) else "torch.distributed all_reduce on the collective's stream + fcd_gibbs_mstep") if use_pg else None),
                   "comm_world": ctx.stat("comm_world"),
                   "rank_ms_per_step_min": min(rank_ms), "rank_ms_per_step_max": max(rank_ms), "r_pass_form": r_form,
                   "settle_sweeps": args.settle,     # untimed, before the W warm-up steps (burn-in; device steady state)
                   "sample_definition": "one sweep of one chain = C f-draws + Nreg*U r-draws"},
        "roofline": roof,
        "sweep_hbm": {"algorithmic_bytes_per_sweep": sweep_bytes, "achieved_GBps": sweep_bytes / (elapsed / args.steps) / 1e9,
                      "frac_of_peak": sweep_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        "kernels": kern,
        "lds_roofline": lds,
        "roofline_valu": valu,
        "passes_ms": {"f_pass": f_pass_ms, "r_pass": r_pass_ms, "tally": tally_ms,
                      "f_pass_GBps": f_bytes / (f_pass_ms * 1e-3) / 1e9, "r_pass_GBps": r_bytes / (r_pass_ms * 1e-3) / 1e9,
                      "f64_adds_per_s": (2 * C * U * G + 2 * C * U * G) / ((f_pass_ms + r_pass_ms) * 1e-3)},
        "lik_tables": {"kernel": "lik_kernel", "bound": "hbm", "algorithmic_bytes": lik_bytes, "avg_launch_ms": lik_ms,
                       "achieved": lik_bytes / (lik_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": lik_bytes / (lik_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "one launch; event pair on its stream, mean of %d launches" % lik_n},
        "allocs_in_timed_region": allocs_in_timed_region,
    }

    if rank == 0 and world == 1 and not args.no_vb:
        out["vb_iteration"] = vb_iteration(np, torch, fcdiff_amd, model, b, bt, Nreg, H, U, cfg_name)

    if rank == 0 and world == 1 and not args.no_corr:
        # K_corr, the front-end north_star names: (S, Nreg, T) time series -> edge-major correlations, fp64 MFMA Gram
        from fcdiff_amd.corr import correlations
        (S_all, T) = (H + U, args.timepoints)
        ts = torch.randn((S_all, Nreg, T), dtype=torch.float64, device="cuda")
        correlations(ts, ctx=ctx, as_numpy=False)
        torch.cuda.synchronize()
        n_rep = 20
        tcr = time.perf_counter()
        for _ in range(n_rep):
            correlations(ts, ctx=ctx, as_numpy=False)
        torch.cuda.synchronize()
        corr_ms = (time.perf_counter() - tcr) / n_rep * 1e3
        flops_full = 2.0 * S_all * Nreg * Nreg * T                       # SURVEY 8d convention (full product; SYRK does half)
        flops_syrk = 2.0 * S_all * (Nreg * (Nreg + 1) / 2) * T
        corr_bytes = 8 * S_all * Nreg * T + 8 * C * S_all
        out["corr"] = {"kernel": "corr_gram_subject_kernel (+ corr_transpose_kernel)", "bound": "mfma", "dtype": "f64",
                       "workload": "S=%d subjects, Nreg=%d, T=%d" % (S_all, Nreg, T), "ms": corr_ms,
                       "flops_full_product": flops_full, "flops_lower_triangle": flops_syrk,
                       "achieved": flops_syrk / (corr_ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops_syrk / (corr_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                       "hbm_GBps": corr_bytes / (corr_ms * 1e-3) / 1e9, "hbm_frac": corr_bytes / (corr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "both launches of fcd_corr_edges, wall time of %d calls (queue kept full: the calls are "
                               "asynchronous); flops counted for the lower triangle actually needed; the input is read once "
                               "(centring by the first sample, row sums beside the Gram product)" % n_rep}
        del ts

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
                               "mode": "(iii) of SURVEY.md section 8d: C restatement of the same sampler, OpenMP over chains "
                                       "(the reference has no sampler, so modes (i)/(ii) exist for the variational "
                                       "iteration only: see vb_iteration)",
                               "sample": "%d chains x %d sweeps of the same %s tables with the C restatement "
                                         "(oracle/fcdiff_oracle.c, OpenMP over chains), %.1f s" % (n_c, n_sw, cfg_key, t_c)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        torch.cuda.synchronize()
        try:
            ctx.detach_comm()            # the library's own communicator first (nothing of it is in flight any more)
        except Exception:
            pass
        dist.destroy_process_group()


def vb_iteration(np, torch, fcdiff_amd, model0, b, bt, Nreg, H, U, cfg_name):
    """
    The reference's own algorithm at this size, unscaled, in this run: one variational iteration (q_F, q_R, pi/gamma,
    tables, energy; fcdiff/fit.py:75-82) on the GPU against the CPU restatement in the reference's structure
    (Python loop over edges, mode (i)), the whole-array NumPy form (mode (ii)) and the C/OpenMP form (mode (iii)).
    The host legs start from the same state and must land on the GPU's energy.
    """
    import copy
    from oracle import fcdiff_oracle as O
    from oracle import c_oracle as CO
    C = b.shape[0]
    fit = fcdiff_amd.fit.UnsharedRegionFit()
    fit.model, fit.b, fit.bt = copy.deepcopy(model0), b, bt
    fit._init_lps(Nreg, H, U)
    fit._update_lps()

    def one():
        fit._update_lq_F()
        fit._update_lq_R()
        fit._update_theta()
        fit._update_lps()
        return fit._eval_energy()
    e_gpu = one()                                   # first iteration from the uniform start: the one the host legs repeat
    for _ in range(2):
        one()
    torch.cuda.synchronize()
    n_it = 10
    t0 = time.perf_counter()
    for _ in range(n_it):
        one()
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / n_it * 1e3
    res = {"workload": "one variational iteration at %s (reference edge ids)" % cfg_name, "gpu_ms": gpu_ms,
           "cpu_faithful_s": None, "cpu_vectorised_s": None, "cpu_c_openmp_s": None, "energy_gpu": e_gpu}
    th = dict(pi=float(model0.pi), eta=model0.eta, epsilon=model0.epsilon, gamma=np.array(model0.gamma, dtype=np.float64),
              mu=model0.mu, sigma=model0.sigma)
    lq_R0 = np.full((Nreg, U, 2), -np.log(2))
    lq_F0 = np.full((C, 1, 3), -np.log(3))
    if C * U <= 1500000:                            # cfg3: ~4 s + ~1 s; cfg5 would take minutes in the edge loop
        t0 = time.perf_counter()
        (_a, _b, _t, e_f) = O.vb_iteration(lq_F0, lq_R0, b, bt, th, vectorised=False)
        res["cpu_faithful_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        (_a, _b, _t, e_v) = O.vb_iteration(lq_F0, lq_R0, b, bt, th, vectorised=True)
        res["cpu_vectorised_s"] = time.perf_counter() - t0
        res["energy_cpu_faithful"], res["energy_cpu_vectorised"] = float(e_f), float(e_v)
        res["energies_agree"] = bool(abs(e_f - e_gpu) <= 1e-9 * abs(e_f) and abs(e_v - e_gpu) <= 1e-9 * abs(e_v))
    # mode (iii): the C restatement's kernels of the iteration (tables, q_F, q_R, tables, energy), all host cores
    t0 = time.perf_counter()
    S_B, lM = CO.lik_tables(b, bt, model0.theta())
    lq_F = CO.update_lq_F(lq_R0, S_B, lM, th["gamma"])
    lq_R = CO.update_lq_R(lq_R0, lq_F, lM, [1 - th["pi"], th["pi"]], 0)
    CO.lik_tables(b, bt, model0.theta())
    CO.energy_terms(lq_F, lq_R, S_B, lM, th["gamma"], [1 - th["pi"], th["pi"]])
    res["cpu_c_openmp_s"] = time.perf_counter() - t0
    res["cores"] = CO.max_threads()
    res["note"] = ("cpu_faithful = oracle/fcdiff_oracle.py with the reference's Python loop over edges (one core, like the "
                   "reference); cpu_vectorised = the same arithmetic as whole-array NumPy (one core); cpu_c_openmp = "
                   "oracle/fcdiff_oracle.c on all host cores")
    return res


if __name__ == "__main__":
    main()
