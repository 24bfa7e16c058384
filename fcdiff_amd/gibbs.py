"""
Many-chain collapsed Gibbs sampler for the IAR model on one MI355X, plus the multi-GPU driver.

The reference ships only the variational fitter (doc/methods.rst:236-239); this sampler is the build's
extension named by BASELINE.json.  Its conditionals are the reference's updates at one-hot q
(fcdiff/fit.py:170-173 for f, :187-194 for r), its M-step for (pi, gamma) the sample version of
fit.py:208-220 -- see include/fcdiff_hip.h for the kernel-level contract.

Multi-GPU (one process per GPU, torch.distributed, backend "nccl" = RCCL): chains are independent given
the tables, so rank k simply owns global chain ids [chain0, chain0 + G).  The only exchange is the
all-reduce of the 8-word pooled-count vector before an M-step; there is no data-path collective.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import util


def shard_chains(total_chains, world_size, rank):
    """Contiguous split of global chain ids: returns (chain0, n_local).  Earlier ranks take the remainder."""
    base, rem = divmod(int(total_chains), int(world_size))
    n_local = base + (1 if rank < rem else 0)
    chain0 = rank * base + min(rank, rem)
    return chain0, n_local


def allreduce_counts(counts, group=None):
    """Sum the pooled sufficient statistics over ranks (RCCL on GPUs, gloo on CPU); no-op for one process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def mstep_from_counts(counts, Nreg, U):
    """Host restatement of fcd_gibbs_mstep (used to report pi/gamma; the device keeps its own copy)."""
    counts = [int(x) for x in counts]
    G = counts[4]
    n_r = float(G * Nreg * U)
    pi = min(max(counts[0] / n_r, 0.5 / n_r), 1.0 - 0.5 / n_r)
    n_f = float(G * util.N_to_C(Nreg))
    gamma = np.array([max(counts[1 + k] / n_f, 0.5 / n_f) for k in range(3)])
    return pi, gamma


class GibbsEngine(object):
    """
    Device-resident state of G chains and the kernels that move it.

    S_B (C,3) and lM (C,U,3,3) are float64 CUDA tensors produced by `fcd_lik_tables`; they are shared by
    all chains.  State layout (include/fcdiff_hip.h): f_state (GW, C, 64) uint8, r_bits (GW, Nreg, U)
    uint64 bit planes (held in an int64 tensor), GW = ceil(G / 64).
    """

    def __init__(self, S_B, lM, Nreg, U, n_chains, chain0=0, seed=0, edge_index="symmetric", ctx=None,
                 region_major=True):
        """region_major=False keeps only lM: the generic f / r kernels run (any shape, several times slower)."""
        import torch
        self.torch = torch
        self.ctx = ctx if ctx is not None else _lib.Context()
        self.Nreg, self.U, self.G = int(Nreg), int(U), int(n_chains)
        self.C = util.N_to_C(self.Nreg)
        self.GW = (self.G + 63) // 64
        self.chain0, self.seed = int(chain0), int(seed)
        self.edge_mode = _lib.EDGE_MODES[edge_index]
        if tuple(lM.shape) != (self.C, self.U, 3, 3) or tuple(S_B.shape) != (self.C, 3):
            raise ValueError("tables do not match Nreg=%d, U=%d" % (self.Nreg, self.U))
        self.S_B, self.lM = S_B.contiguous(), lM.contiguous()
        dev = self.S_B.device
        self.hyper = torch.zeros(8, dtype=torch.float64, device=dev)
        self.f_state = torch.zeros((self.GW, self.C, 64), dtype=torch.uint8, device=dev)
        self.r_bits = torch.zeros((self.GW, self.Nreg, self.U), dtype=torch.int64, device=dev)
        self.counts = torch.zeros(8, dtype=torch.int64, device=dev)
        self.cnt_f = torch.zeros((self.C, 3), dtype=torch.int32, device=dev)
        self.cnt_r = torch.zeros((self.Nreg, self.U), dtype=torch.int32, device=dev)
        self.n_accumulated = 0
        self.ctx.call("fcd_ctx_reserve", self.Nreg, self.U, self.G)
        self.lMd = self.lMf = None
        if region_major:
            self.lMd = torch.empty((self.U, self.Nreg, self.Nreg, 3, 2), dtype=torch.float64, device=dev)
            self.lMf = torch.empty((self.C, self.U, 3, 2), dtype=torch.float64, device=dev)
            self.refresh_tables()

    def refresh_tables(self):
        """Re-derive the two difference tables after lM changed (a table build = a theta_sub change)."""
        if self.lMd is not None:
            self.ctx.call("fcd_gibbs_region_tables", _lib.dptr(self.lM), self.Nreg, self.U, self.edge_mode,
                          _lib.dptr(self.lMd), _lib.stream_ptr())
            self.ctx.call("fcd_gibbs_edge_tables", _lib.dptr(self.lM), self.Nreg, self.U, _lib.dptr(self.lMf),
                          _lib.stream_ptr())

    # ---- hyper-parameters ----
    def set_hyper(self, gamma, pi2):
        (g, _g) = _lib.dbl_array(gamma)
        (p, _p) = _lib.dbl_array(pi2)
        self.ctx.call("fcd_hyper_set", _lib.dptr(self.hyper), g, p, _lib.stream_ptr())

    def hyper_values(self):
        h = self.hyper.cpu().numpy()
        self.ctx.check_device()
        return np.exp(h[0:3]), float(np.exp(h[4]))

    def host(self, t):
        """Device tensor -> NumPy array; raises if a sweep before it abandoned a device-side wait (fcd_ctx_check)."""
        a = t.cpu().numpy()
        self.ctx.check_device()
        return a

    # ---- state ----
    def init(self, pi):
        self.ctx.call("fcd_gibbs_init", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G,
                      self.chain0, C.c_uint64(self.seed), float(pi), _lib.stream_ptr())

    def export_state(self):
        t = self.torch
        f = t.empty((self.G, self.C), dtype=t.uint8, device=self.f_state.device)
        r = t.empty((self.G, self.Nreg, self.U), dtype=t.uint8, device=self.f_state.device)
        self.ctx.call("fcd_gibbs_export_state", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U,
                      self.G, _lib.dptr(f), _lib.dptr(r), _lib.stream_ptr())
        (fh, rh) = (f.cpu().numpy(), r.cpu().numpy())
        self.ctx.check_device()          # (the copies above waited for every sweep before them)
        return fh, rh

    def import_state(self, f, r):
        t = self.torch
        f = t.as_tensor(np.ascontiguousarray(f, dtype=np.uint8), device=self.f_state.device)
        r = t.as_tensor(np.ascontiguousarray(r, dtype=np.uint8), device=self.f_state.device)
        if tuple(f.shape) != (self.G, self.C) or tuple(r.shape) != (self.G, self.Nreg, self.U):
            raise ValueError("state shapes must be (G, C) and (G, Nreg, U)")
        self.ctx.call("fcd_gibbs_import_state", _lib.dptr(f), _lib.dptr(r), self.Nreg, self.U, self.G,
                      _lib.dptr(self.f_state), _lib.dptr(self.r_bits), _lib.stream_ptr())

    # ---- moves ----
    def f_step(self, sweep):
        self.ctx.call("fcd_gibbs_f_step", _lib.dptr(self.S_B), _lib.dptr(self.lM), _lib.dptr(self.lMf), _lib.dptr(self.hyper),
                      _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, self.chain0,
                      C.c_uint64(self.seed), int(sweep), _lib.stream_ptr())

    def r_step(self, sweep):
        self.ctx.call("fcd_gibbs_r_step", _lib.dptr(self.lM), _lib.dptr(self.lMd), _lib.dptr(self.hyper),
                      _lib.dptr(self.f_state),
                      _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, self.chain0, C.c_uint64(self.seed),
                      int(sweep), self.edge_mode, _lib.stream_ptr())

    def sweeps(self, sweep0, n_sweeps, with_counts=False):
        self.ctx.call("fcd_gibbs_sweeps", _lib.dptr(self.S_B), _lib.dptr(self.lM), _lib.dptr(self.lMf), _lib.dptr(self.lMd),
                      _lib.dptr(self.hyper),
                      _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, self.chain0,
                      C.c_uint64(self.seed), int(sweep0), int(n_sweeps), self.edge_mode,
                      _lib.dptr(self.counts if with_counts else None), _lib.stream_ptr())
        return self.counts if with_counts else None

    def run(self, sweep0, n_sweeps, mstep_every=0, accumulate_from=None, want_counts=False):
        """
        n_sweeps x (f pass, r pass, tally) in one call (fcd_gibbs_run): the tally of each sweep feeds the marginal
        counters from sweep `accumulate_from` on (None: never), runs the (pi, gamma) M-step on this rank's pooled counts
        every `mstep_every` sweeps (0: never -- several ranks all-reduce the returned counts and call mstep()), and
        packs the r words of the next f pass.  Returns the counts tensor of the last sweep (or None).
        """
        acc = accumulate_from is not None
        self.ctx.call("fcd_gibbs_run", _lib.dptr(self.S_B), _lib.dptr(self.lM), _lib.dptr(self.lMf), _lib.dptr(self.lMd),
                      _lib.dptr(self.hyper), _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G,
                      self.chain0, C.c_uint64(self.seed), int(sweep0), int(n_sweeps), self.edge_mode, int(mstep_every),
                      int(accumulate_from) if acc else 0, _lib.dptr(self.counts if want_counts else None),
                      _lib.dptr(self.cnt_f if acc else None), _lib.dptr(self.cnt_r if acc else None), _lib.stream_ptr())
        if acc:
            self.n_accumulated += max(0, int(sweep0) + int(n_sweeps) - max(int(accumulate_from), int(sweep0)))
        return self.counts if want_counts else None

    # ---- pooled statistics / M-step ----
    def stats(self):
        self.ctx.call("fcd_gibbs_stats", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G,
                      _lib.dptr(self.counts), _lib.stream_ptr())
        return self.counts

    def mstep(self, counts):
        self.ctx.call("fcd_gibbs_mstep", _lib.dptr(counts), self.Nreg, self.U, _lib.dptr(self.hyper), _lib.stream_ptr())

    def accumulate(self):
        self.ctx.call("fcd_gibbs_accumulate", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U,
                      self.G, _lib.dptr(self.cnt_f), _lib.dptr(self.cnt_r), _lib.stream_ptr())
        self.n_accumulated += 1

    def tally(self, want_counts=True, accumulate=True):
        """stats() and accumulate() in one pass over the state; returns the counts tensor (or None)."""
        self.ctx.call("fcd_gibbs_tally", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G,
                      _lib.dptr(self.counts if want_counts else None), _lib.dptr(self.cnt_f if accumulate else None),
                      _lib.dptr(self.cnt_r if accumulate else None), _lib.stream_ptr())
        if accumulate:
            self.n_accumulated += 1
        return self.counts if want_counts else None

    def pair_counts(self, out=None, accumulate=False):
        """(C, U, 3, 3) float64: number of this rank's chains with f_c = k and mixture case l at (c, u)."""
        t = self.torch
        if out is None:
            out = t.empty((self.C, self.U, 3, 3), dtype=t.float64, device=self.f_state.device)
            accumulate = False
        self.ctx.call("fcd_gibbs_pair_counts", _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G,
                      1 if accumulate else 0, _lib.dptr(out), _lib.stream_ptr())
        return out

    # ---- diagnostics ----
    def logjoint(self):
        out = self.torch.empty(self.G, dtype=self.torch.float64, device=self.f_state.device)
        self.ctx.call("fcd_gibbs_logjoint", _lib.dptr(self.S_B), _lib.dptr(self.lM), _lib.dptr(self.hyper),
                      _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, _lib.dptr(out),
                      _lib.stream_ptr())
        return out

    def r_sums(self):
        """(G,) int32: number of anomalous (region, patient) sites of each chain (fcd_gibbs_chain_rsum)."""
        out = self.torch.empty(self.G, dtype=self.torch.int32, device=self.f_state.device)
        self.ctx.call("fcd_gibbs_chain_rsum", _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, _lib.dptr(out),
                      _lib.stream_ptr())
        return out

    def conditionals(self, want_f=True, want_r=True):
        t = self.torch
        dev = self.f_state.device
        cf = t.empty((self.G, self.C, 3), dtype=t.float64, device=dev) if want_f else None
        cr = t.empty((self.G, self.Nreg, self.U, 2), dtype=t.float64, device=dev) if want_r else None
        self.ctx.call("fcd_gibbs_conditionals", _lib.dptr(self.S_B), _lib.dptr(self.lM), _lib.dptr(self.hyper),
                      _lib.dptr(self.f_state), _lib.dptr(self.r_bits), self.Nreg, self.U, self.G, self.edge_mode,
                      _lib.dptr(cf), _lib.dptr(cr), _lib.stream_ptr())
        return cf, cr


def _world_size(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def run_chains(engine, n_sweeps, sweep0=0, mstep_every=1, burn_in=0, update_theta=True, group=None,
               on_sweep=None, mstep_lag=0, force_collective=False, direct=True):
    """
    The sampler loop shared by UnsharedRegionFit(method='gibbs') and bench.py.

    Every sweep: f pass, r pass, one tally launch (marginal counters after `burn_in` sweeps, the packed r words of the
    next f pass).  Every `mstep_every` sweeps the pooled counts give the (pi, gamma) M-step:
      * one rank, mstep_lag=0: inside the tally launch (fcd_gibbs_run), the whole loop is ONE call;
      * several ranks, mstep_lag=0: counts -> blocking all-reduce (RCCL) -> fcd_gibbs_mstep, between two calls;
      * mstep_lag=1: the M-step from the counts of period j (mstep_every sweeps) is applied after period j+1 (it
        shapes period j+2), so the all-reduce of period j runs on the collective's stream WHILE period j+1 computes.
        The schedule -- and with it every chain's path -- is the same for any number of ranks, one rank included.
    `engine` is anything with run/mstep (the HIP engine here; the CPU tests pass an oracle-backed stand-in to
    exercise the multi-process logic under gloo).
    direct=True (default; HIP engine, mstep_lag=0): several ranks pool their counts through the context's own RCCL
    communicator INSIDE fcd_gibbs_run (Context.attach_comm); direct=False keeps round 3's loop through torch.distributed.
    force_collective=True takes the several-rank path -- counts, all-reduce, fcd_gibbs_mstep between calls -- also in a
    process group of ONE rank (bench.py --force-pg, tests/test_dist_nccl.py: RCCL initialised and used on a one-GPU box;
    the chains are those of the plain loop, bit for bit).
    """
    import torch.distributed as dist
    world = _world_size(group)
    collective = world > 1 or (bool(force_collective) and dist.is_available() and dist.is_initialized())
    k = int(mstep_every) if (update_theta and mstep_every and mstep_every > 0) else 0
    acc_from = sweep0 + burn_in
    if collective and direct and not mstep_lag and hasattr(engine, "ctx") and hasattr(engine.ctx, "attach_comm"):
        # Round 4: the all-reduce of the pooled counts is RCCL called by the library itself on the stream of the sweep
        # kernels (fcd_comm_*): the several-rank loop is then the one-rank loop -- ONE fcd_gibbs_run call per chunk, the
        # tally, the all-reduce, the M-step kernel and the next f pass queued behind one another.
        # (the path has only ever run on ONE rank -- the build box has one GPU -- so it is taken only if EVERY rank could make
        # its communicator: one all-reduce of a flag through the torch group decides, all ranks alike; else round 3's loop)
        import torch
        decided = getattr(engine, "_direct_comm", None)            # (the agreement is made once per engine, not once per call)
        if decided is not None:
            collective = not decided
        ok = 1
        try:
            if decided is None:
                engine.ctx.attach_comm(group)
        except Exception as exc:      # noqa: BLE001
            import warnings
            warnings.warn("fcdiff_amd: the library's own RCCL communicator could not be made (%s); pooling the counts through "
                          "torch.distributed instead" % (exc,))
            ok = 0
        if decided is None:
            flag = torch.tensor([ok], dtype=torch.int32, device=engine.ctx.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            engine._direct_comm = int(flag.item()) == 1
            if engine._direct_comm:
                collective = False
            else:
                engine.ctx.detach_comm()
    if not collective and on_sweep is None and not mstep_lag:
        engine.run(sweep0, n_sweeps, mstep_every=k, accumulate_from=acc_from)
        return
    pending = None       # (counts clone, work handle or None): the M-step that waits for its turn (mstep_lag)

    def apply(p):
        (cl, work) = p
        if work is not None:
            work.wait()          # the compute stream waits for the collective; the host does not (RCCL)
        engine.mstep(cl)
    i = 0
    while i < n_sweeps:
        c = n_sweeps - i
        if on_sweep is not None:
            c = 1
        if k:
            c = min(c, k - (i % k))
        end = i + c
        do_m = bool(k and end % k == 0)
        local = do_m and not collective and not mstep_lag
        counts = engine.run(sweep0 + i, c, mstep_every=(c if local else 0), accumulate_from=acc_from,
                            want_counts=do_m and not local)
        if mstep_lag and pending is not None and do_m:      # one M-step PERIOD later, whatever the chunking
            apply(pending)
            pending = None
        if do_m and not local:
            if mstep_lag:
                cl = counts.clone()
                work = dist.all_reduce(cl, op=dist.ReduceOp.SUM, group=group, async_op=True) if collective else None
                pending = (cl, work)
            else:
                if collective:
                    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
                engine.mstep(counts)
        if on_sweep is not None:
            on_sweep(i, engine)
        i = end
    if pending is not None:
        apply(pending)
