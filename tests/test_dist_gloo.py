"""
The multi-process path on CPU: 2 ranks, gloo, 127.0.0.1.

The product sampler loop (fcdiff_amd.gibbs.run_chains) only needs an object with sweeps/tally/mstep;
on a GPU that is the HIP engine.  Here an oracle-backed stand-in (C restatement, tests only) takes its place so
that what runs under gloo is exactly the distributed logic: global chain ids per rank, the all-reduce of the
pooled counts, the shared (pi, gamma) M-step.  Two ranks with 5 + 4 chains must reproduce one process with 9.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, theta_dict

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fcdiff_amd.gibbs import mstep_from_counts, run_chains, shard_chains  # noqa: E402


class OracleEngine(object):
    """Same surface as fcdiff_amd.gibbs.GibbsEngine, computed by oracle/fcdiff_oracle.c (tests only)."""

    def __init__(self, S_B, lM, Nreg, U, G, chain0, seed, gamma, pi, mode=1):
        from oracle import c_oracle as CO
        self.CO = CO
        (self.S_B, self.lM, self.Nreg, self.U, self.G) = (S_B, lM, Nreg, U, G)
        (self.chain0, self.seed, self.mode) = (chain0, seed, mode)
        (self.gamma, self.pi) = (np.array(gamma, dtype=np.float64), float(pi))
        (self.f, self.r) = CO.gibbs_init(G, Nreg, U, self.pi, seed, chain0)
        self.cnt_r = np.zeros((Nreg, U), dtype=np.int64)
        self.n_acc = 0

    def sweeps(self, sweep0, n, with_counts=False):
        lng, lnpi2 = np.log(self.gamma), np.log([1 - self.pi, self.pi])
        for s in range(sweep0, sweep0 + n):
            self.CO.gibbs_f_step(self.f, self.r, self.S_B, self.lM, lng, self.seed, s, self.chain0)
            self.CO.gibbs_r_step(self.f, self.r, self.lM, lnpi2, self.seed, s, self.mode, self.chain0)

    def stats(self):
        return torch.from_numpy(self.CO.gibbs_stats(self.f, self.r).copy())

    def mstep(self, counts):
        (self.pi, self.gamma) = mstep_from_counts(counts.numpy(), self.Nreg, self.U)

    def accumulate(self):
        self.cnt_r += self.r.sum(axis=0, dtype=np.int64)
        self.n_acc += 1

    def tally(self, want_counts=True, accumulate=True):
        if accumulate:
            self.accumulate()
        return self.stats() if want_counts else None

    def run(self, sweep0, n_sweeps, mstep_every=0, accumulate_from=None, want_counts=False):
        """Restatement of fcd_gibbs_run's contract (include/fcdiff_hip.h)."""
        for j in range(n_sweeps):
            self.sweeps(sweep0 + j, 1)
            if accumulate_from is not None and sweep0 + j >= accumulate_from:
                self.accumulate()
            if mstep_every > 0 and (j + 1) % mstep_every == 0:
                self.mstep(self.stats())
        return self.stats() if want_counts else None


def problem():
    g = load_golden("G11_gibbs_conditionals_cfg1")
    th = theta_dict(g["theta"])
    (Nreg, U) = g["r_state"].shape
    return g["lp_B_g_F"].sum(axis=1), g["lM"], Nreg, U, th


def run_single(total, n_sweeps, seed, mstep_every=1, lag=0, on_sweep=None):
    (S_B, lM, Nreg, U, th) = problem()
    eng = OracleEngine(S_B, lM, Nreg, U, total, 0, seed, th["gamma"], th["pi"])
    run_chains(eng, n_sweeps, mstep_every=mstep_every, burn_in=1, mstep_lag=lag, on_sweep=on_sweep)
    return eng


def worker(rank, world, port, total, n_sweeps, seed, out_dir, mstep_every=1, lag=0, force=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (S_B, lM, Nreg, U, th) = problem()
        (chain0, n_local) = shard_chains(total, world, rank)
        eng = OracleEngine(S_B, lM, Nreg, U, n_local, chain0, seed, th["gamma"], th["pi"])
        run_chains(eng, n_sweeps, mstep_every=mstep_every, burn_in=1, mstep_lag=lag, force_collective=force)
        cnt = torch.from_numpy(eng.cnt_r.copy())
        dist.all_reduce(cnt)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), f=eng.f, r=eng.r, pi=eng.pi, gamma=eng.gamma,
                 chain0=chain0, cnt_r=cnt.numpy())
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_chains():
    assert [shard_chains(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    assert [shard_chains(4096, 4, r) for r in range(4)] == [(0, 1024), (1024, 1024), (2048, 1024), (3072, 1024)]
    tot = sum(shard_chains(8192, 8, r)[1] for r in range(8))
    assert tot == 8192


def test_run_chains_schedules_on_one_rank():
    """
    One process: the single-call form, the per-sweep form (a callback forces chunks of one sweep) and a chunked
    M-step cadence walk the same chains; the lagged schedule is a different (documented) trajectory of its own.
    """
    (total, n_sweeps, seed) = (6, 6, 5)
    a = run_single(total, n_sweeps, seed, mstep_every=2)
    seen = []
    b = run_single(total, n_sweeps, seed, mstep_every=2, on_sweep=lambda i, e: seen.append(i))
    assert seen == list(range(n_sweeps))
    np.testing.assert_array_equal(a.f, b.f)
    np.testing.assert_array_equal(a.r, b.r)
    assert a.pi == b.pi and a.n_acc == b.n_acc == n_sweeps - 1
    np.testing.assert_array_equal(a.cnt_r, b.cnt_r)
    lag = run_single(total, n_sweeps, seed, mstep_every=2, lag=1)
    lag2 = run_single(total, n_sweeps, seed, mstep_every=2, lag=1, on_sweep=lambda i, e: None)
    np.testing.assert_array_equal(lag.r, lag2.r)
    assert lag.pi == lag2.pi
    assert not np.array_equal(lag.r, a.r) or not np.array_equal(lag.f, a.f)     # a different schedule, a different path


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mstep_every,lag", [(1, 0), (2, 1)])
def test_two_ranks_equal_one_process(tmp_path, mstep_every, lag):
    (total, n_sweeps, seed) = (9, 4, 77)
    ref = run_single(total, n_sweeps, seed, mstep_every=mstep_every, lag=lag)
    port = free_port()
    mp.spawn(worker, args=(2, port, total, n_sweeps, seed, str(tmp_path), mstep_every, lag), nprocs=2, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    assert [int(p["chain0"]) for p in parts] == [0, 5]
    np.testing.assert_array_equal(np.concatenate([p["f"] for p in parts]), ref.f)
    np.testing.assert_array_equal(np.concatenate([p["r"] for p in parts]), ref.r)
    for p in parts:
        # every rank ends with the same pooled hyper-parameters, equal to the single-process ones
        assert float(p["pi"]) == ref.pi
        np.testing.assert_array_equal(p["gamma"], ref.gamma)
        np.testing.assert_array_equal(p["cnt_r"], ref.cnt_r)
    assert 0 < ref.pi < 1 and abs(ref.gamma.sum() - 1) < 1e-12


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mstep_every,lag", [(1, 0), (1, 1), (2, 1)])
def test_group_of_one_rank_forced_collective(tmp_path, mstep_every, lag):
    """
    force_collective=True: the several-rank loop (counts -> all-reduce -> M-step between calls) in a process group of ONE
    rank walks the chains of the plain loop (what bench.py --force-pg and tests/test_dist_nccl.py do with RCCL on one GPU).
    Without a process group the flag changes nothing.
    """
    (total, n_sweeps, seed) = (7, 5, 41)
    ref = run_single(total, n_sweeps, seed, mstep_every=mstep_every, lag=lag)
    port = free_port()
    mp.spawn(worker, args=(1, port, total, n_sweeps, seed, str(tmp_path), mstep_every, lag, True), nprocs=1, join=True)
    p = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    np.testing.assert_array_equal(p["f"], ref.f)
    np.testing.assert_array_equal(p["r"], ref.r)
    assert float(p["pi"]) == ref.pi
    np.testing.assert_array_equal(p["gamma"], ref.gamma)
    np.testing.assert_array_equal(p["cnt_r"], ref.cnt_r)
    (S_B, lM, Nreg, U, th) = problem()
    eng = OracleEngine(S_B, lM, Nreg, U, total, 0, seed, th["gamma"], th["pi"])
    run_chains(eng, n_sweeps, mstep_every=mstep_every, burn_in=1, mstep_lag=lag, force_collective=True)   # no group here
    np.testing.assert_array_equal(eng.r, ref.r)
