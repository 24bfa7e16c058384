"""
Fits models to observed correlations -- the MI355X build of fcdiff/fit.py.

`UnsharedRegionFit` keeps the reference's surface (fcdiff/fit.py:12-54): attributes `model, b, bt,
max_iters=10, rel_tol=1e-5, energy`, private state `_lq_R (N,U,2), _lq_F (C,1,3), _lp_B_g_F (C,H,3),
_p_Bt_g_Ft (C,U,3), _lM (C,U,3,3)` readable/assignable as NumPy arrays, `run()` and the private methods
its tests call (`_init_lps, _update_lps, _update_lq_F, _update_lq_R, _update_pi, _update_gamma,
_update_theta, _eval_energy, _is_converged`).  Every one of them runs a hand-written HIP kernel
through the C ABI (include/fcdiff_hip.h) on tensors that stay on the GPU between calls.  There is no
NumPy fallback: without libfcdiff_hip.so or a GPU these methods raise.

New knobs (all optional; defaults reproduce the reference):
    method       'vb' (reference algorithm) | 'gibbs' (many-chain collapsed Gibbs + MCEM for pi, gamma)
    edge_index   'reference' (fit.py:185-186 calls nm_to_c for every ordered pair: quirk Q1) |
                 'symmetric' (doc/methods.rst:646-653).  Default: 'reference' for vb, 'symmetric' for gibbs.
    n_chains, n_sweeps, burn_in, mstep_every, mstep_lag, seed, chain0     sampler controls
    update_theta_sub, theta_sub_every                          (eta, epsilon) step: vb every iteration / gibbs every K sweeps

Differences from the reference that are deliberate and documented:
  Q4  `model.pi` may be the scalar the model defines or the 2-vector [1-pi, pi] the reference's updates
      index; both are accepted everywhere.
  Q5  tables are float64 (the reference's np.full(shape, 1) buffers turn int64 on modern NumPy).
  Q7  run() is the documented loop (doc/methods.rst:564-600); `energy` is appended to.  The reference's
      run() raises IndexError on its first energy assignment (fit.py:73-74).  The (eta, epsilon)
      optimiser step (fit.py:222-241) cannot run in the reference (it calls undefined names); here it is
      implemented (`_update_theta_sub`, analytic gradient on the device) but stays OFF by default
      (`update_theta_sub = False`) so that the default trajectory is the one the fixtures pin.
"""
import numpy as np

from . import _lib
from . import util
from .gibbs import GibbsEngine, run_chains, allreduce_counts


class UnsharedRegionFit(object):
    """
    Fits an unshared region model to correlations (fcdiff/fit.py:12-30).

    Attributes
    ----------
    model : fcdiff_amd.UnsharedRegionModel      initial model; updated in place by run()
    b : ndarray (C, H), bt : ndarray (C, U)     correlations of healthy subjects / patients
    max_iters : int, rel_tol : float            iteration cap / relative tolerance of _is_converged
    energy : list of float                      variational free energy per iteration (vb) or minus the
                                                chain-mean log-joint per recorded sweep (gibbs)
    """

    def __init__(self):
        self.model = None
        self.b = None
        self.bt = None
        self.max_iters = 10
        self.rel_tol = 1e-5
        self.energy = []

        self.method = "vb"
        self.data_check = "sample"   # how _update_lps() notices in-place edits of b / bt: 'sample' | 'full' | 'none' (_data_digest)
        self.edge_index = None
        self.update_theta_sub = False
        self.theta_sub_params = "eta_epsilon"     # 'all': mu and sigma^2 join the optimiser (the reference's commented-out intent)
        self.n_chains = 1024
        self.n_sweeps = 100
        self.burn_in = 20
        self.mstep_every = 1
        self.mstep_lag = 0         # gibbs: 1 = apply an M-step one period late so its all-reduce overlaps the next sweeps
        self.theta_sub_every = 0   # gibbs: re-fit (eta, epsilon) from pooled chain counts every this many sweeps (0: never)
        self.energy_every = 0
        self.trace_every = 0       # gibbs: keep every chain's log-joint every this many sweeps in `trace` (chains, draws)
        self.trace = None          # ... and its number of anomalous sites sum_{n,u} r_nu in `trace_r`
        self.trace_r = None
        self.seed = 0
        self.chain0 = 0
        self.sampler = None       # the GibbsEngine of the last gibbs run

        self._ctx = None
        self._d = {}              # device tensors: lq_R, lq_F, S_B, lM, lpB, pBt, hyper, b, bt
        self._hyper_key = None

    # ------------------------------------------------------------------ device plumbing
    def _context(self):
        if self._ctx is None:
            self._ctx = _lib.Context()
        return self._ctx

    def _torch(self):
        import torch
        return torch

    def _dev(self):
        return self._context().device

    def _up(self, a):
        t = self._torch()
        return t.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self._dev())

    def _edge_mode(self):
        name = self.edge_index
        if name is None:
            name = "reference" if self.method == "vb" else "symmetric"
        if name not in _lib.EDGE_MODES:
            raise ValueError("edge_index must be 'reference' or 'symmetric'")
        return name

    def _shape(self):
        lM = self._d.get("lM")
        if lM is None:
            raise ValueError("tables have not been initialized (_init_lps / _update_lps)")
        if lM.dim() != 4 or tuple(lM.shape[2:]) != (3, 3):
            raise ValueError("_lM must have shape (C, U, 3, 3), got %s" % (tuple(lM.shape),))
        C, U = int(lM.shape[0]), int(lM.shape[1])
        N = util.C_to_N(C)
        if (N % 1) != 0:
            raise ValueError("Number of connections (%u) must be a triangular number." % C)
        return int(N), C, U

    def _check_state(self, need=("lq_R", "lq_F", "S_B")):
        """
        (N, C, U) from the tables, after checking that every buffer a kernel is about to index has the shape the
        kernel assumes.  The private state is assignable as NumPy arrays (the reference's tests do it); a mismatch
        there is a NumPy broadcasting error in the reference -- here it would be an out-of-bounds device read, so it
        is refused on the host before anything is launched.
        """
        (N, C, U) = self._shape()
        want = {"lq_R": (N, U, 2), "lq_F": (C, 1, 3), "S_B": (C, 3)}
        for key in need:
            v = self._d.get(key)
            if v is None:
                raise ValueError("_%s has not been initialized (_init_lps)" % key)
            if tuple(v.shape) != want[key]:
                raise ValueError("_%s has shape %s, the tables (_lM %s) need %s"
                                 % (key if key != "S_B" else "lp_B_g_F (summed over H)", tuple(v.shape),
                                    (C, U, 3, 3), want[key]))
        return (N, C, U)

    def _hyper(self):
        """Device hyper block {ln gamma, ln(1-pi), ln pi} refreshed whenever the model's values changed."""
        t = self._torch()
        if "hyper" not in self._d:
            self._d["hyper"] = t.zeros(8, dtype=t.float64, device=self._dev())
        gamma = np.asarray(self.model.gamma, dtype=np.float64).reshape(3)
        pi2 = self._pi2()
        key = (tuple(gamma.tolist()), tuple(pi2.tolist()))
        if key != self._hyper_key:
            (g, _g) = _lib.dbl_array(gamma)
            (p, _p) = _lib.dbl_array(pi2)
            self._context().call("fcd_hyper_set", _lib.dptr(self._d["hyper"]), g, p, _lib.stream_ptr())
            self._hyper_key = key
        return self._d["hyper"]

    def _pi2(self):
        pi = np.asarray(self.model.pi, dtype=np.float64)
        if pi.ndim == 0:
            return np.array([1.0 - float(pi), float(pi)])
        return pi.reshape(2)

    # ------------------------------------------------------------------ NumPy views of the device state
    def _get(self, key, shape=None):
        v = self._d.get(key)
        if v is None:
            return None
        a = v.cpu().numpy()
        return a.reshape(shape) if shape is not None else a

    @property
    def _lq_R(self):
        return self._get("lq_R")

    @_lq_R.setter
    def _lq_R(self, a):
        self._d["lq_R"] = None if a is None else self._up(a)

    @property
    def _lq_F(self):
        return self._get("lq_F")

    @_lq_F.setter
    def _lq_F(self, a):
        self._d["lq_F"] = None if a is None else self._up(a)

    @property
    def _lM(self):
        return self._get("lM")

    @_lM.setter
    def _lM(self, a):
        self._d["lM"] = None if a is None else self._up(a)

    @property
    def _lp_B_g_F(self):
        """(C,H,3).  The fit path only ever uses its H-sum; the full table is produced on demand."""
        if self._d.get("lpB") is None and self.b is not None and self.model is not None and "S_B" in self._d:
            self._tables(full=True)
        return self._get("lpB")

    @_lp_B_g_F.setter
    def _lp_B_g_F(self, a):
        if a is None:
            self._d["lpB"] = None
            return
        self._d["lpB"] = self._up(a)
        self._d["S_B"] = self._d["lpB"].sum(dim=1).contiguous()      # fit.py:171 sums over H

    @property
    def _p_Bt_g_Ft(self):
        if self._d.get("pBt") is None and self.bt is not None and self.model is not None and "lM" in self._d:
            self._tables(full=True)
        return self._get("pBt")

    @_p_Bt_g_Ft.setter
    def _p_Bt_g_Ft(self, a):
        self._d["pBt"] = None if a is None else self._up(a)

    # ------------------------------------------------------------------ reference methods
    def run(self):
        """Runs the fitting procedure (fcdiff/fit.py:56-82; doc/methods.rst:564-600)."""
        (C, H) = self.b.shape
        U = self.bt.shape[1]
        N = util.C_to_N(C)
        if (N % 1) != 0:
            msg = "Number of connections (%u) must be a triangular number." % C
            raise ValueError(msg)
        if self.model is None:
            msg = "Model has not been initialized."
            raise ValueError(msg)
        N = int(N)
        self._init_lps(N, H, U)
        self._update_lps()
        if self.method == "vb":
            self._run_vb()
        elif self.method == "gibbs":
            self._run_gibbs(N, U)
        else:
            raise ValueError("method must be 'vb' or 'gibbs'")

    def _run_vb(self):
        self.energy = [self._eval_energy()]
        for i in range(1, self.max_iters + 1):
            self._update_lq_F()
            self._update_lq_R()
            self._update_theta()
            self._update_lps()
            self.energy.append(self._eval_energy())
            if self._is_converged(i):
                break

    def _init_lps(self, N, H, U):
        """Uniform log-probabilities and float64 tables (fit.py:84-102)."""
        t = self._torch()
        dev = self._dev()
        C = util.N_to_C(N)
        self._d["lq_R"] = t.full((N, U, 2), -np.log(2), dtype=t.float64, device=dev)
        self._d["lq_F"] = t.full((C, 1, 3), -np.log(3), dtype=t.float64, device=dev)
        self._d["S_B"] = t.full((C, 3), float(H), dtype=t.float64, device=dev)
        self._d["lM"] = t.ones((C, U, 3, 3), dtype=t.float64, device=dev)
        self._d["lpB"] = None
        self._d["pBt"] = None
        self._d.pop("data_key", None)        # a fit starts from the arrays as they are now (see _tables)
        self._HU = (H, U)

    def _update_lps(self):
        """Likelihood tables from the current parameters (fit.py:104-122): kernel K_lik."""
        self._tables(full=False)

    def _tables(self, full):
        t = self._torch()
        dev = self._dev()
        b = np.ascontiguousarray(self.b, dtype=np.float64)
        bt = np.ascontiguousarray(self.bt, dtype=np.float64)
        (C, H) = b.shape
        U = bt.shape[1]
        # the device copies of b / bt are made by _init_lps()/run() and kept across the _update_lps() calls of a fit (the
        # reference re-reads self.b / self.bt on every call, fcdiff/fit.py:111-115; here that would be a host digest or
        # a PCIe copy of 16-320 MB per variational iteration).  What is checked on EVERY call is cheap: the arrays'
        # identity, base address, shape, strides -- and, `data_check='sample'`, a strided sample of 4096 elements that
        # visits every column.  `data_check='full'` digests the whole arrays on every call (CRC-32: 8 ms at cfg3,
        # 165 ms at cfg5); invalidate_data() forces the next call to upload.
        key = (self._array_key(self.b), self._array_key(self.bt), self._data_digest(b, self.data_check),
               self._data_digest(bt, self.data_check))
        if self._d.get("data_key") != key:
            self._d["b"], self._d["bt"] = self._up(b), self._up(bt)
            self._d["data_key"] = key
        if self._d.get("S_B") is None or tuple(self._d["S_B"].shape) != (C, 3):
            self._d["S_B"] = t.empty((C, 3), dtype=t.float64, device=dev)
        if self._d.get("lM") is None or tuple(self._d["lM"].shape) != (C, U, 3, 3):
            self._d["lM"] = t.empty((C, U, 3, 3), dtype=t.float64, device=dev)
        lpB = pBt = None
        if full:
            lpB = t.empty((C, H, 3), dtype=t.float64, device=dev)
            pBt = t.empty((C, U, 3), dtype=t.float64, device=dev)
        (th, _th) = _lib.dbl_array(self.model.theta())
        self._context().call("fcd_lik_tables", _lib.dptr(self._d["b"]), _lib.dptr(self._d["bt"]), C, H, U, th,
                             _lib.dptr(self._d["S_B"]), _lib.dptr(self._d["lM"]), _lib.dptr(lpB), _lib.dptr(pBt),
                             _lib.stream_ptr())
        self._d["lpB"], self._d["pBt"] = lpB, pBt

    @staticmethod
    def _array_key(a):
        a = np.asarray(a)
        return (id(a), a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)

    @staticmethod
    def _data_digest(a, mode="full"):
        """
        Digest of an input array, by `data_check` mode.
        'full'    CRC-32 of the WHOLE array: sees any in-place edit (a single element, one patient's column), at 1-2 GB/s
                  of host time per call (round 3 ran it on every _update_lps: 8.6 ms per variational iteration at cfg3,
                  165 ms at cfg5, beside 0.5 / 2.5 ms of kernels -- VERDICT r3).
        'sample'  (default) CRC-32 of 4096 elements taken with a stride that is coprime to the row length, so that
                  every column is visited ~4096/U times: sees whole-array edits (scaling, clipping, Fisher z) and
                  whole-column edits; an edit of a few single elements between two _update_lps() calls of the same
                  fit needs invalidate_data() (documented contract; _init_lps() and run() always upload afresh).
        'none'    identity / address / shape only.
        """
        import zlib
        from math import gcd
        flat = np.ascontiguousarray(a).reshape(-1)
        if mode == "none":
            return 0
        if mode == "full" or flat.size <= 8192:
            return zlib.crc32(flat.view(np.uint8))
        if mode != "sample":
            raise ValueError("data_check must be 'sample', 'full' or 'none'")
        row = a.shape[-1] if a.ndim > 1 else 1
        step = max(1, flat.size // 4096)
        while gcd(step, row) != 1:
            step += 1
        return zlib.crc32(np.ascontiguousarray(flat[::step]).view(np.uint8)) ^ zlib.crc32(flat[-64:].view(np.uint8))

    def invalidate_data(self):
        """Forget the device copies of b / bt: the next _update_lps() uploads them again (after in-place edits)."""
        self._d.pop("data_key", None)

    def _is_converged(self, s):
        """fit.py:124-140 (quirk Q6 kept: a negative energy makes any decrease 'converged')."""
        e = self.energy[s - 1]
        e_star = self.energy[s]
        return ((e - e_star) / e) < self.rel_tol

    def _eval_energy(self):
        """Variational free energy (fit.py:142-155): kernel K_energy."""
        return float(self._energy_terms_signed().sum())

    def _energy_terms(self):
        t = self._torch()
        (N, C, U) = self._check_state()
        out = t.empty(6, dtype=t.float64, device=self._dev())
        self._context().call("fcd_vb_energy", _lib.dptr(self._d["lq_F"]), _lib.dptr(self._d["lq_R"]),
                             _lib.dptr(self._d["S_B"]), _lib.dptr(self._d["lM"]), _lib.dptr(self._hyper()), N, U,
                             _lib.dptr(out), _lib.stream_ptr())
        return out.cpu().numpy()

    def _energy_terms_signed(self):
        return self._energy_terms() * np.array([-1.0, -1.0, -1.0, -1.0, 1.0, 1.0])

    def _update_lq_F(self):
        """Probability of the typical network template (fit.py:157-174): kernel K_qF."""
        t = self._torch()
        (N, C, U) = self._check_state(need=("lq_R", "S_B"))
        out = t.empty((C, 1, 3), dtype=t.float64, device=self._dev())
        self._context().call("fcd_vb_update_qF", _lib.dptr(self._d["lq_R"]), _lib.dptr(self._d["S_B"]),
                             _lib.dptr(self._d["lM"]), _lib.dptr(self._hyper()), N, U, _lib.dptr(out),
                             _lib.stream_ptr())
        self._d["lq_F"] = out

    def _update_lq_R(self):
        """Probability of the anomalous regions (fit.py:176-198): kernel K_qR (Gauss-Seidel over regions)."""
        (N, C, U) = self._check_state(need=("lq_R", "lq_F"))
        lq_R = self._d["lq_R"].clone()
        self._context().call("fcd_vb_update_qR", _lib.dptr(self._d["lq_F"]), _lib.dptr(self._d["lM"]),
                             _lib.dptr(self._hyper()), N, U, _lib.EDGE_MODES[self._edge_mode()], _lib.dptr(lq_R),
                             _lib.stream_ptr())
        self._d["lq_R"] = lq_R

    def _update_theta(self):
        """fit.py:200-206."""
        lq_R, lq_F = self._d.get("lq_R"), self._d.get("lq_F")
        if (lq_R is not None and lq_F is not None and lq_R.dim() == 3 and lq_F.dim() == 3
                and int(lq_F.shape[0]) == util.N_to_C(int(lq_R.shape[0]))):
            # both closed forms come out of ONE launch (and one read-back): the same four numbers _update_pi() and
            # _update_gamma() would each ask the kernel for
            out = self._theta_step(lq_R=lq_R, lq_F=lq_F)
            self.model.pi = float(out[0])
            self.model.gamma = out[1:4].copy()
        else:
            self._update_pi()
            self._update_gamma()
        if self.update_theta_sub:
            self._update_theta_sub()

    def _theta_step(self, lq_R=None, lq_F=None):
        """out4 = {mean q_R[:,:,1], mean_c q_F} from the given (or the current) log-probabilities."""
        t = self._torch()
        lq_R = self._d.get("lq_R") if lq_R is None else lq_R
        lq_F = self._d.get("lq_F") if lq_F is None else lq_F
        if lq_R is None or lq_F is None:
            raise ValueError("_lq_R / _lq_F have not been initialized (_init_lps)")
        if lq_R.dim() != 3 or int(lq_R.shape[2]) != 2 or lq_F.dim() != 3 or tuple(lq_F.shape[1:]) != (1, 3):
            raise ValueError("_lq_R must have shape (N, U, 2) and _lq_F (C, 1, 3), got %s and %s"
                             % (tuple(lq_R.shape), tuple(lq_F.shape)))
        (N, U) = (int(lq_R.shape[0]), int(lq_R.shape[1]))
        if util.N_to_C(N) != int(lq_F.shape[0]):
            raise ValueError("_lq_F and _lq_R disagree on the number of regions")
        out = t.empty(4, dtype=t.float64, device=self._dev())
        self._context().call("fcd_vb_theta_step", _lib.dptr(lq_F), _lib.dptr(lq_R), N, U, _lib.dptr(out),
                             _lib.dptr(None), _lib.stream_ptr())
        return out.cpu().numpy()

    def _update_pi(self):
        """pi* = mean q_R[:, :, 1] (fit.py:208-213); a scalar, as in the reference."""
        t = self._torch()
        lq_R = self._d.get("lq_R")
        if lq_R is None:
            raise ValueError("_lq_R has not been initialized (_init_lps)")
        # pi* needs q_R only (the reference's test sets nothing else, test_fit.py:513-533): q_F is a local stand-in
        # of the matching size, never stored
        lq_F = self._d.get("lq_F")
        if lq_F is None or lq_R.dim() != 3 or int(lq_F.shape[0]) != util.N_to_C(int(lq_R.shape[0])):
            lq_F = t.full((util.N_to_C(int(lq_R.shape[0])), 1, 3), -np.log(3), dtype=t.float64, device=self._dev())
        self.model.pi = float(self._theta_step(lq_R=lq_R, lq_F=lq_F)[0])

    def _update_gamma(self):
        """gamma* = mean_c q_F (fit.py:215-220)."""
        t = self._torch()
        lq_F = self._d.get("lq_F")
        if lq_F is None:
            raise ValueError("_lq_F has not been initialized (_init_lps)")
        # gamma* needs q_F only (test_fit.py:536-557): q_R is a local (N, 1, 2) stand-in, never stored
        N = util.C_to_N(int(lq_F.shape[0]))
        if (N % 1) != 0:
            raise ValueError("Number of connections (%u) must be a triangular number." % int(lq_F.shape[0]))
        lq_R = self._d.get("lq_R")
        if lq_R is None or int(lq_R.shape[0]) != int(N):
            lq_R = t.full((int(N), 1, 2), -np.log(2), dtype=t.float64, device=self._dev())
        self.model.gamma = self._theta_step(lq_R=lq_R, lq_F=lq_F)[1:4].copy()

    def diagnostics(self):
        """
        split-R-hat / effective sample size of the per-chain traces of the last gibbs run (trace_every > 0): the
        log-joint (the summary's own keys, as before) and, under "sum_r", the number of anomalous sites sum_{n,u} r_nu.
        """
        from . import diagnostics as D
        if self.trace is None or self.trace.shape[1] < 4:
            raise ValueError("no trace: set trace_every > 0 and keep at least 4 recorded sweeps after burn_in")
        out = D.summary(self.trace)
        if self.trace_r is not None:
            out["sum_r"] = D.summary(self.trace_r)
        return out

    def _update_theta_sub(self):
        """
        Updates (eta, epsilon) (fit.py:222-241): bounded minimisation of -E_lM with the other terms fixed, bounds
        (1e-5, 1 - 1e-5) on both, L-BFGS-B.  The reference's version cannot run (it calls the undefined `opt_fun`,
        fit.py:239) and would difference the objective numerically (`jac=False`); here objective AND its two analytic
        derivatives (the formulas of `_eval_dE_dh` / `_eval_dE_de`, fit.py:600-697) come from one kernel pass over bt.
        theta_sub_params = 'eta_epsilon' (default): mu and sigma stay fixed, as in the reference as shipped (commented
        out of its optimiser, fit.py:232-237, 250-251); 'all' reads those lines in: (eta, epsilon, mu, sigma^2) jointly,
        objective E[ln p(b|f)] + E[ln p(bt|f,r)] (fit.py:282), fcd_theta_full_objective.
        """
        W = self._theta_sub_weights()
        if self.theta_sub_params == "all":
            # the reference's commented-out intent (fit.py:232-237, 250-251, 266-267, 282): mu and sigma too
            (eta, epsilon, mu, sigma, info) = minimize_theta_full(self._context(), self._d["b"], self._d["bt"], W, self.model)
            self.model.mu, self.model.sigma = mu, sigma
        elif self.theta_sub_params == "eta_epsilon":
            (eta, epsilon, info) = minimize_theta_sub(self._context(), self._d["bt"], W, self.model)
        else:
            raise ValueError("theta_sub_params must be 'eta_epsilon' or 'all'")
        self.model.eta, self.model.epsilon = eta, epsilon
        self._theta_sub_info = info

    def _theta_sub_weights(self):
        """W[c,u,k,l] = q_F[c,k] * w_l(c,u) on the device (fit.py:382-406, 508-510)."""
        t = self._torch()
        (N, _C, U) = self._check_state(need=("lq_R", "lq_F"))
        lq_R, lq_F = self._d["lq_R"], self._d["lq_F"]
        W = t.empty((util.N_to_C(N), U, 3, 3), dtype=t.float64, device=self._dev())
        self._context().call("fcd_theta_sub_weights_vb", _lib.dptr(lq_F), _lib.dptr(lq_R), N, U, _lib.dptr(W),
                             _lib.stream_ptr())
        return W

    # ------------------------------------------------------------------ gibbs
    def _run_gibbs(self, N, U):
        """
        Many-chain collapsed Gibbs over (f, r) with an MCEM step for (pi, gamma) from statistics pooled
        over all chains of all ranks.  On return `_lq_F` / `_lq_R` hold the logs of the chain-and-sweep
        averaged marginals, `model.pi` / `model.gamma` the last M-step, `energy` minus the mean
        log-joint at the recorded sweeps.
        """
        t = self._torch()
        eng = GibbsEngine(self._d["S_B"], self._d["lM"], N, U, self.n_chains, chain0=self.chain0, seed=self.seed,
                          edge_index=self._edge_mode(), ctx=self._context())
        pi2 = self._pi2()
        eng.set_hyper(np.asarray(self.model.gamma, dtype=np.float64), pi2)
        eng.init(float(pi2[1]))
        self.energy = []

        traces = []
        traces_r = []

        def record(i, e):
            want_e = bool(self.energy_every and (i + 1) % self.energy_every == 0)
            want_t = bool(self.trace_every and i >= self.burn_in and (i + 1) % self.trace_every == 0)
            if want_e or want_t:
                lj = e.logjoint()
                if want_e:
                    self.energy.append(-float(lj.mean()))
                if want_t:
                    traces.append(e.host(lj))
                    traces_r.append(e.host(e.r_sums()))
            if self.update_theta_sub and self.theta_sub_every and (i + 1) % self.theta_sub_every == 0 and i + 1 < self.n_sweeps:
                # Monte-Carlo EM for (eta, epsilon): pooled counts of (f_c, mixture case) over all chains of all ranks
                W = e.pair_counts()
                if self.theta_sub_params == "all":
                    (eta, epsilon, mu, sigma, _info) = minimize_theta_full(self._context(), self._d["b"], self._d["bt"], W,
                                                                           self.model, reduce=allreduce_counts)
                    self.model.mu, self.model.sigma = mu, sigma
                else:
                    (eta, epsilon, _info) = minimize_theta_sub(self._context(), self._d["bt"], W, self.model,
                                                               reduce=allreduce_counts)
                self.model.eta, self.model.epsilon = eta, epsilon
                self._update_lps()            # tables follow theta_sub ...
                e.refresh_tables()            # ... and so do the sampler's two difference tables
        needs_callback = bool(self.energy_every or self.trace_every or (self.update_theta_sub and self.theta_sub_every))
        run_chains(eng, self.n_sweeps, sweep0=0, mstep_every=self.mstep_every, burn_in=self.burn_in,
                   update_theta=True, on_sweep=record if needs_callback else None, mstep_lag=self.mstep_lag)
        self.sampler = eng
        self.trace = np.stack(traces, axis=1) if traces else None
        self.trace_r = np.stack(traces_r, axis=1).astype(np.float64) if traces_r else None
        # marginals pooled over this rank's chains and, when distributed, over all ranks
        cnt = t.cat([eng.cnt_f.reshape(-1).to(t.int64), eng.cnt_r.reshape(-1).to(t.int64),
                     t.tensor([eng.n_accumulated * eng.G], dtype=t.int64, device=eng.cnt_f.device)])
        cnt = allreduce_counts(cnt).cpu().numpy().astype(np.float64)
        # (that read waited for every sweep: a pipelined r pass that gave up a device-side wait has raised the
        # context's error word by now -- no marginals, pi or gamma from such a state)
        eng.ctx.check_device()
        C = util.N_to_C(N)
        total = max(cnt[-1], 1.0)
        with np.errstate(divide="ignore"):
            self._lq_F = np.log(cnt[:3 * C].reshape(C, 1, 3) / total)
            p1 = cnt[3 * C:3 * C + N * U].reshape(N, U) / total
            self._lq_R = np.log(np.stack([1.0 - p1, p1], axis=2))
        (gamma, pi) = eng.hyper_values()
        self.model.gamma = gamma
        self.model.pi = pi


def theta_sub_objective(ctx, bt_dev, W, theta, reduce=None):
    """
    (S, dS/d eta, dS/d epsilon), S = sum W ln M(bt; eta, epsilon), through fcd_theta_sub_objective.
    `reduce` (optional) sums the three numbers over ranks (multi-GPU sampler: W holds this rank's chain counts).
    """
    import torch
    out = torch.empty(3, dtype=torch.float64, device=W.device)
    (th, _th) = _lib.dbl_array(theta)
    ctx.call("fcd_theta_sub_objective", _lib.dptr(bt_dev), _lib.dptr(W), int(W.shape[0]), int(W.shape[1]), th,
             _lib.dptr(out), _lib.stream_ptr())
    if reduce is not None:
        out = reduce(out)
    return out.cpu().numpy()


def theta_full_objective(ctx, b_dev, bt_dev, W, theta, reduce=None):
    """
    {S, dS/d eta, dS/d epsilon, dS/d mu[3], dS/d sigma^2[3]} of the full theta_sub objective (fcd_theta_full_objective):
    S = E[ln p(b | f)] + E[ln p(bt | f, r)] for the weights W.  `reduce` sums the nine numbers over ranks.
    With several ranks each rank's W holds its own chain counts, but b is the same on all: pass b_dev on every rank
    (the healthy term scales with the chain counts too, so the sum over ranks is the pooled objective).
    """
    import torch
    out = torch.empty(9, dtype=torch.float64, device=W.device)
    (th, _th) = _lib.dbl_array(theta)
    H = int(b_dev.shape[1]) if b_dev is not None else 0
    ctx.call("fcd_theta_full_objective", _lib.dptr(b_dev), _lib.dptr(bt_dev), _lib.dptr(W), int(W.shape[0]), H,
             int(W.shape[1]), th, _lib.dptr(out), _lib.stream_ptr())
    if reduce is not None:
        out = reduce(out)
    return out.cpu().numpy()


def minimize_theta_full(ctx, b_dev, bt_dev, W, model, reduce=None, bound_eps=1e-5):
    """
    argmin of -(E[ln p(b | f)] + E[ln p(bt | f, r)]) over theta_sub = (eta, epsilon, mu[3], sigma^2[3]): the step the
    reference intends (fit.py:222-286 with its commented-out lines read in): pack order and sigma ** 2 as in
    fit.py:243-252, bounds of fit.py:228-237 -- eta, epsilon in (e, 1-e), mu_0 in (-1+e, -e), mu_1 in (-e, e), mu_2 in
    (e, 1-e), sigma^2 > e --, analytic gradient, L-BFGS-B.  Returns (eta, epsilon, mu, sigma, scipy result).
    """
    import scipy.optimize as spopt
    base = np.array(model.theta(), dtype=np.float64)
    e = bound_eps

    def unpack(x):
        th = base.copy()
        th[1], th[2] = float(x[0]), float(x[1])
        th[6:9] = x[2:5]
        th[9:12] = np.sqrt(x[5:8])
        return th

    def fun(x):
        o = theta_full_objective(ctx, b_dev, bt_dev, W, unpack(x), reduce)
        return (-o[0], -o[1:9])
    bnds = ((e, 1 - e), (e, 1 - e), (-1 + e, 0 - e), (0 - e, 0 + e), (0 + e, 1 - e), (0 + e, None), (0 + e, None), (0 + e, None))
    x0 = np.concatenate([[model.eta, model.epsilon], np.asarray(model.mu, dtype=np.float64),
                         np.asarray(model.sigma, dtype=np.float64) ** 2])
    lo = np.array([bd[0] for bd in bnds])
    hi = np.array([np.inf if bd[1] is None else bd[1] for bd in bnds])
    x0 = np.minimum(np.maximum(x0, lo), hi)
    res = spopt.minimize(fun, x0, jac=True, bounds=bnds, method="L-BFGS-B")
    th = unpack(res.x)
    return float(th[1]), float(th[2]), th[6:9].copy(), th[9:12].copy(), res


def minimize_theta_sub(ctx, bt_dev, W, model, reduce=None, bound_eps=1e-5):
    """
    argmin over (eta, epsilon) in [1e-5, 1-1e-5]^2 of -sum W ln M (fit.py:222-241), analytic gradient, L-BFGS-B.
    Returns (eta, epsilon, scipy result).  The model is not modified.
    """
    import scipy.optimize as spopt
    base = np.array(model.theta(), dtype=np.float64)

    def fun(x):
        th = base.copy()
        th[1], th[2] = float(x[0]), float(x[1])
        (S, dh, de) = theta_sub_objective(ctx, bt_dev, W, th, reduce)
        return (-S, np.array([-dh, -de]))
    bnds = ((bound_eps, 1 - bound_eps), (bound_eps, 1 - bound_eps))          # fit.py:228-231
    x0 = np.clip(np.array([model.eta, model.epsilon], dtype=np.float64), bound_eps, 1 - bound_eps)
    res = spopt.minimize(fun, x0, jac=True, bounds=bnds, method="L-BFGS-B")
    return float(res.x[0]), float(res.x[1]), res


# ---------------------------------------------------------------------------------------------------------
# Module-level helpers of the reference (fcdiff/fit.py:382-733).  They are small closed forms on whole arrays
# that the reference's tests call directly; the fit path itself evaluates them inside the kernels above.
# Kept as NumPy one-liners with the reference's names, argument order and return shapes.
# ---------------------------------------------------------------------------------------------------------
def _eval_q_R_w(q_R, n, m):
    """(U,3) weights of a region pair: both typical, both anomalous, discordant (fit.py:382-406)."""
    w = np.zeros((q_R.shape[1], 3))
    w[:, 0] = q_R[n, :, 0] * q_R[m, :, 0]
    w[:, 1] = q_R[n, :, 1] * q_R[m, :, 1]
    w[:, 2] = q_R[n, :, 0] * q_R[m, :, 1]
    w[:, 2] += q_R[n, :, 1] * q_R[m, :, 0]
    return w


def _eval_M_eps(eta, epsilon, l):
    """Probability of keeping the template type under mixture case l (fit.py:433-444)."""
    if l == 0:
        return 1 - epsilon
    if l == 1:
        return epsilon
    eps = eta * epsilon
    eps += (1 - eta) * (1 - epsilon)
    return eps


def _eval_M(N, eta, epsilon, k, l):
    """M_kl from the Normal densities N (C,U,3) (fit.py:409-430)."""
    eps = _eval_M_eps(eta, epsilon, l)
    others = [j for j in range(3) if j != k]
    return eps * N[:, :, k] + (1 - eps) * 0.5 * (N[:, :, others[0]] + N[:, :, others[1]])


def _eval_E_lp_F(q_F, gamma):
    return np.sum(q_F * np.log(gamma))


def _eval_E_lp_B_g_F(q_F, lp_B_g_F):
    return np.sum(q_F * lp_B_g_F)


def _eval_E_lp_R(q_R, pi):
    return np.sum(q_R * np.log(pi))


def _eval_E_lM(q_F, q_R, lM):
    """E[log p(bt | f, r)] (fit.py:489-511), vectorised over edges."""
    C = q_F.shape[0]
    N = q_R.shape[0]
    (n, m) = np.tril_indices(N, -1)       # lower-triangular row-major == util.c_to_nm order
    assert n.shape[0] == C
    w = np.stack([q_R[n, :, 0] * q_R[m, :, 0], q_R[n, :, 1] * q_R[m, :, 1],
                  q_R[n, :, 0] * q_R[m, :, 1] + q_R[n, :, 1] * q_R[m, :, 0]], axis=2)      # (C,U,3)
    return np.sum(q_F[:, 0, :] * np.einsum("cul,cukl->ck", w, lM))


def _eval_E_lq_F(q_F, lq_F):
    return np.sum(q_F * lq_F)


def _eval_E_lq_R(q_R, lq_R):
    return np.sum(q_R * lq_R)


def _eval_dlN_dm(b, mu, sigma):
    return (b - mu) / (sigma * sigma)


def _eval_dlN_ds(b, mu, sigma):
    diff = b - mu
    sigma2 = sigma * sigma
    return ((diff * diff) - sigma2) / (2 * sigma2)


def _eval_dN_dm(N, b, mu, sigma):
    return N * _eval_dlN_dm(b, mu, sigma)


def _eval_dN_ds(N, b, mu, sigma):
    return N * _eval_dlN_ds(b, mu, sigma)


def _eval_dlM_dm(norm, mix, mu, sigma, eta, epsilon, k, l):
    """fit.py:572-597, quirk Q8 included (tests k != l; passes the density array where b is expected)."""
    eps = _eval_M_eps(eta, epsilon, l)
    if k != l:
        eps = 0.5 * (1 - eps)
    return eps * _eval_dlN_dm(norm, mu, sigma) / mix


def _eval_dlM_dh(norm, mix, epsilon, k):
    """fit.py:618-641."""
    eps = (2 * epsilon) - 1
    others = [j for j in range(3) if j != k]
    s = norm[:, :, others[0]] + norm[:, :, others[1]]
    return (eps * norm[:, :, k] - 0.5 * eps * s) / mix


def _eval_dlM_de(norm, mix, eta, k, l):
    """fit.py:667-697."""
    eps = -1 if l == 0 else (1 if l == 1 else 2 * eta - 1)
    others = [j for j in range(3) if j != k]
    s = norm[:, :, others[0]] + norm[:, :, others[1]]
    return (eps * norm[:, :, k] - 0.5 * eps * s) / mix


def _pair_weights(q_R):
    N = q_R.shape[0]
    (n, m) = np.tril_indices(N, -1)
    w0 = q_R[n, :, 0] * q_R[m, :, 0]
    w1 = q_R[n, :, 1] * q_R[m, :, 1]
    w2 = q_R[n, :, 0] * q_R[m, :, 1] + q_R[n, :, 1] * q_R[m, :, 0]
    return w0, w1, w2


def _eval_dE_dm(q_F, q_R, dlN_dmj, dlM_dmj, j):
    """dE/dmu_j (fit.py:542-569), vectorised over edges."""
    (w0, w1, w2) = _pair_weights(q_R)
    w = np.stack([w0, w1, w2], axis=2)
    d = -np.sum(q_F[:, 0, j] * np.sum(dlN_dmj, axis=1))
    d -= np.sum(q_F[:, 0, :] * np.einsum("cul,cukl->ck", w, dlM_dmj))
    return d


def _eval_dE_dh(q_R, q_F, norm, mix, epsilon):
    """dE/deta (fit.py:600-615)."""
    (_w0, _w1, w2) = _pair_weights(q_R)
    d = 0.0
    for k in range(3):
        d -= np.sum(q_F[:, 0, k] * np.sum(w2 * _eval_dlM_dh(norm, mix[:, :, k, 2], epsilon, k), axis=1))
    return d


def _eval_dE_de(q_R, q_F, norm, mix, eta):
    """dE/depsilon (fit.py:644-664)."""
    (w0, w1, w2) = _pair_weights(q_R)
    d = 0.0
    for k in range(3):
        s = w0 * _eval_dlM_de(norm, mix[:, :, k, 0], eta, k, 0)
        s += w1 * _eval_dlM_de(norm, mix[:, :, k, 1], eta, k, 1)
        s += w2 * _eval_dlM_de(norm, mix[:, :, k, 2], eta, k, 2)
        d -= np.sum(q_F[:, 0, k] * np.sum(s, axis=1))
    return d
