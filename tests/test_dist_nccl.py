"""
Two ranks on two GPUs over RCCL (torch.distributed backend "nccl"): the HIP GibbsEngine under run_chains, both M-step
schedules, against the single-process run of the same global chains.  Skips when fewer than two GPUs are visible (the
build box has one; the driver's 8-GPU node runs it).  Each rank is a fresh child process started BEFORE anything touches
the GPU (torch.multiprocessing spawn), one process per GPU, rendezvous on 127.0.0.1.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, total, n_sweeps, seed, mstep_every, lag, out_dir, with_group=True, force=False, direct=True,
            break_direct=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    if with_group:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import fcdiff_amd
        from fcdiff_amd.gibbs import GibbsEngine, run_chains, shard_chains
        (N, H, U) = (24, 6, 10)
        m = fcdiff_amd.UnsharedRegionModel()
        (_r, _t, _f, _ft, b, bt) = m.sample_fast(N, H, U, seed=3)
        fit = fcdiff_amd.fit.UnsharedRegionFit()
        fit.model, fit.b, fit.bt = m, b, bt
        fit._init_lps(N, H, U)
        fit._update_lps()
        (chain0, n_local) = shard_chains(total, world, rank)
        eng = GibbsEngine(fit._d["S_B"], fit._d["lM"], N, U, n_local, chain0=chain0, seed=seed, ctx=fit._context())
        if break_direct:                  # the library's communicator "cannot be made": run_chains must fall back to torch, on every rank alike
            def _fail(group=None):
                raise RuntimeError("simulated RCCL failure")
            eng.ctx.attach_comm = _fail
        eng.set_hyper(m.gamma, m.pi2())
        eng.init(0.2)
        run_chains(eng, n_sweeps, mstep_every=mstep_every, burn_in=1, mstep_lag=lag, force_collective=force, direct=direct)
        torch.cuda.synchronize()
        (f, r) = eng.export_state()
        cnt = eng.cnt_r.to(torch.int64).clone()
        if with_group:
            dist.all_reduce(cnt)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), f=f, r=r, hyper=eng.hyper.cpu().numpy(), chain0=chain0,
                 cnt_r=cnt.cpu().numpy(), world=dist.get_world_size() if with_group else 0,
                 r_form=eng.ctx.stat("r_form_last"), comm_world=eng.ctx.stat("comm_world"))
    finally:
        if with_group:
            dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mstep_every,lag,direct", [(1, 0, True), (2, 0, True), (1, 0, False), (2, 1, False)])
def test_two_gpus_equal_one_process(tmp_path, mstep_every, lag, direct):
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    (total, n_sweeps, seed) = (192, 6, 31)
    one = tmp_path / "one"
    two = tmp_path / "two"
    one.mkdir()
    two.mkdir()
    # children only: the parent never initialises the GPU
    mp.spawn(_worker, args=(1, _free_port(), total, n_sweeps, seed, mstep_every, lag, str(one)), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), total, n_sweeps, seed, mstep_every, lag, str(two), True, False, direct), nprocs=2, join=True)
    ref = np.load(os.path.join(str(one), "rank0.npz"))
    parts = [np.load(os.path.join(str(two), "rank%d.npz" % r)) for r in range(2)]
    assert [int(p["world"]) for p in parts] == [2, 2] and [int(p["chain0"]) for p in parts] == [0, 96]
    np.testing.assert_array_equal(np.concatenate([p["f"] for p in parts]), ref["f"])
    np.testing.assert_array_equal(np.concatenate([p["r"] for p in parts]), ref["r"])
    for p in parts:
        np.testing.assert_array_equal(p["hyper"], ref["hyper"])       # the same pooled (pi, gamma) on every rank
        np.testing.assert_array_equal(p["cnt_r"], ref["cnt_r"])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mstep_every,lag,direct", [(1, 0, True), (2, 0, True), (1, 0, False), (1, 1, False), (2, 1, False), (1, 0, "broken")])
def test_one_gpu_process_group_of_one_rank(tmp_path, mstep_every, lag, direct):
    """
    RCCL on the one GPU of the build box: a process group of ONE rank (backend nccl, initialised before anything touches
    the GPU), and the several-rank loop forced on it, beside the default pipelined r pass --
      direct=True   (round 4, the default): the library's own communicator (ncclGetUniqueId -> broadcast through the torch group
                    -> ncclCommInitRank), ncclAllReduce of the pooled counts queued on the stream of the sweep kernels inside
                    fcd_gibbs_run, then the one-thread M-step kernel;
      direct=False  round 3's loop: counts, torch.distributed all-reduce (blocking, or asynchronous on the collective's stream
                    under the lagged schedule), work.wait(), fcd_gibbs_mstep.
    The chains, the hyper-parameters and the marginal counters must equal those of the same schedule without any process group.
    """
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    (total, n_sweeps, seed) = (192, 6, 31)
    plain = tmp_path / "plain"
    group = tmp_path / "group"
    plain.mkdir()
    group.mkdir()
    mp.spawn(_worker, args=(1, _free_port(), total, n_sweeps, seed, mstep_every, lag, str(plain), False, False), nprocs=1, join=True)
    broken = direct == "broken"          # (direct asked for, the communicator "fails": the loop through torch.distributed must take over)
    mp.spawn(_worker, args=(1, _free_port(), total, n_sweeps, seed, mstep_every, lag, str(group), True, True, bool(direct), broken), nprocs=1,
             join=True)
    direct = bool(direct) and not broken
    a = np.load(os.path.join(str(plain), "rank0.npz"))
    b = np.load(os.path.join(str(group), "rank0.npz"))
    assert int(b["world"]) == 1 and int(b["r_form"]) == 2          # (the pipelined form ran beside the collective)
    assert int(b["comm_world"]) == (1 if direct else 0)            # (the library's own communicator was the one used -- or not)
    for k in ("f", "r", "hyper", "cnt_r"):
        np.testing.assert_array_equal(a[k], b[k])
