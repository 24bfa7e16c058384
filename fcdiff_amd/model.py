"""
The individual anomalous region (IAR) model: parameter container + forward sampler.

Mirrors fcdiff/model.py:9-236: same class name, attributes, defaults, `__str__`, `sample(N, H, U)` and the
six `sample_*` methods with the same return shapes/dtypes.  The sampler is host-side NumPy exactly like
the reference's (it is the synthetic-input generator, not part of the fit path) and consumes the legacy
`RandomState` stream in the same order, so `sample` reproduces the reference draw for draw
(tests/golden/G9).  `sample_fast` is the same distribution from a `numpy.random.Generator`, vectorised,
for benchmark-sized inputs.
"""
import textwrap

import numpy as np

from . import util


class UnsharedRegionModel(object):
    """
    Attributes (fcdiff/model.py:13-38)
    ----------
    rng : numpy.random.RandomState     random number generator (seed 0)
    pi : float                         probability of an anomalous region
    eta : float                        probability of an anomalous connection btw a typical and anomalous region
    gamma : ndarray (3,)               probability of each template connection type
    epsilon : float                    probability that a typical connection differs from the template
    mu, sigma : ndarray (3,)           mean / standard deviation of the correlation of each connection type
    """

    def __init__(self):
        self.rng = np.random.RandomState(0)
        self.pi = 0.05
        self.eta = 0.3
        self.gamma = np.array([0.1, 0.8, 0.1])
        self.epsilon = 0.03
        self.mu = np.array([-0.15, 0, 0.3])
        self.sigma = np.array([0.025, 0.035, 0.05])

    def __str__(self):
        return textwrap.dedent('''\
            fcdiff.models.UnsharedRegionModel
                rng = %s
                pi = %g
                eta = %g
                gamma = %s
                epsilon = %g
                mu = %s
                sigma = %s''' % (self.rng, self.pi, self.eta, self.gamma,
                                 self.epsilon, self.mu, self.sigma))

    # ---- packing used by the C ABI: theta[12] = pi, eta, epsilon, gamma[3], mu[3], sigma[3] ----
    def pi2(self):
        """pi as the 2-vector [1-pi, pi] the fitter indexes (quirk Q4: fit.py:183, :486)."""
        pi = np.asarray(self.pi, dtype=np.float64)
        if pi.ndim == 0:
            return np.array([1.0 - float(pi), float(pi)])
        return pi.astype(np.float64).reshape(2)

    def theta(self):
        return np.concatenate([[self.pi2()[1], self.eta, self.epsilon],
                               np.asarray(self.gamma, dtype=np.float64).reshape(3),
                               np.asarray(self.mu, dtype=np.float64).reshape(3),
                               np.asarray(self.sigma, dtype=np.float64).reshape(3)]).astype(np.float64)

    # ---- forward sampling (model.py:52-236) ----
    def sample(self, N, H, U):
        """
        Returns (r (N,U) bool, t (C,U) bool, f (C,3) bool, f_tilde (C,U,3) bool, b (C,H) float64,
        b_tilde (C,U) float64), C = N(N-1)/2.
        """
        r = self.sample_R(N, U)
        t = self.sample_T(r)
        f = self.sample_F(N)
        f_tilde = self.sample_F_tilde(f, t)
        b = self.sample_B(f, H)
        b_tilde = self.sample_B_tilde(f_tilde)
        return (r, t, f, f_tilde, b, b_tilde)

    def sample_R(self, N, U):
        """Anomalous regions, Bernoulli(pi) (model.py:92-109)."""
        return self.rng.binomial(1, self.pi, (N, U)) > 0

    def sample_T(self, r):
        """
        Anomalous connections given regions (model.py:111-143): both typical -> False, both anomalous ->
        True, discordant -> Bernoulli(eta).  Edges are visited upper-triangular row-major and a variate
        is consumed only for discordant (edge, patient) pairs, in that order.
        """
        (N, U) = r.shape
        C = util.N_to_C(N)
        iu = np.triu_indices(N, 1)            # (n, m > n) row-major: the reference's loop order
        rn = r[iu[0], :]
        rm = r[iu[1], :]
        t = rn & rm
        disc = rn ^ rm
        k = int(disc.sum())
        if k:
            t[disc] = self.rng.binomial(1, self.eta, k) > 0
        assert t.shape == (C, U)
        return t

    def sample_F(self, N):
        """Connection template, one Multinomial(1, gamma) per edge (model.py:145-160)."""
        return self.rng.multinomial(1, self.gamma, util.N_to_C(N)) > 0

    def sample_F_tilde(self, f, t):
        """Patients' connection types given template and anomalous connections (model.py:162-189)."""
        (C, U) = t.shape
        f_tilde = np.zeros((C, U, 3), dtype='bool')
        e = self.epsilon
        draw = self.rng.multinomial
        for c in range(C):
            fc = f[c, :]
            p_typ = (1 - e) * fc + (e * 0.5) * (1 - fc)
            p_ano = e * fc + (1 - e) * 0.5 * (1 - fc)
            tc = t[c]
            out = f_tilde[c]
            for u in range(U):
                out[u, :] = draw(1, p_ano if tc[u] else p_typ) > 0
        return f_tilde

    def sample_B(self, f, H):
        """Healthy correlations, Normal(mu_f, sigma_f) clipped to [-1, 1] (model.py:191-213)."""
        C = f.shape[0]
        k = np.argmax(f, axis=1)
        b = np.zeros((C, H), dtype='float64')
        for c in range(C):
            b[c, :] = self.rng.normal(self.mu[k[c]], self.sigma[k[c]], H)
        return b.clip(-1, 1)

    def sample_B_tilde(self, f_tilde):
        """Patient correlations given their connection types (model.py:215-236)."""
        k = np.argmax(f_tilde, axis=2)
        # one array call draws element by element in C order: the same variates as the scalar loop
        b_tilde = self.rng.normal(np.asarray(self.mu)[k], np.asarray(self.sigma)[k])
        return b_tilde.clip(-1, 1)

    # ---- same distribution, vectorised, own stream: benchmark-sized inputs in milliseconds ----
    def sample_fast(self, N, H, U, seed=0):
        """
        (r, t, f, f_tilde, b, b_tilde) with the shapes of `sample`, drawn with numpy.random.Generator
        (PCG64).  Edge order is the FITTER's (lower-triangular, util.c_to_nm), so the output can be
        fitted as is (no quirk-Q3 re-indexing).
        """
        g = np.random.default_rng(seed)
        C = util.N_to_C(N)
        il = np.tril_indices(N, -1)           # (n, m < n) row-major == util.c_to_nm order
        r = g.random((N, U)) < self.pi
        rn, rm = r[il[0], :], r[il[1], :]
        t = np.where(rn ^ rm, g.random((C, U)) < self.eta, rn & rm)
        gam = np.asarray(self.gamma, dtype=np.float64)
        fk = g.choice(3, size=C, p=gam / gam.sum())
        f = np.zeros((C, 3), dtype=bool)
        f[np.arange(C), fk] = True
        e = self.epsilon
        keep = np.where(t, g.random((C, U)) < e, g.random((C, U)) < (1 - e))
        other = (fk[:, None] + 1 + (g.random((C, U)) < 0.5)) % 3
        ftk = np.where(keep, fk[:, None], other)
        f_tilde = np.zeros((C, U, 3), dtype=bool)
        np.put_along_axis(f_tilde, ftk[:, :, None], True, axis=2)
        mu = np.asarray(self.mu, dtype=np.float64)
        sg = np.asarray(self.sigma, dtype=np.float64)
        b = (mu[fk][:, None] + sg[fk][:, None] * g.standard_normal((C, H))).clip(-1, 1)
        b_tilde = (mu[ftk] + sg[ftk] * g.standard_normal((C, U))).clip(-1, 1)
        return (r, t, f, f_tilde, b, b_tilde)

    def sample_gpu(self, N, H, U, seed=0, ctx=None):
        """
        `sample` on the GPU (counter RNG, every variable in parallel): same return types and distribution, the
        fitter's edge order like `sample_fast`.  Needs the HIP library and a GPU (no fallback).
        """
        import ctypes as C
        import torch
        from . import _lib
        ctx = ctx if ctx is not None else _lib.Context()
        Ce = util.N_to_C(N)
        dev = ctx.device
        r = torch.empty((N, U), dtype=torch.uint8, device=dev)
        t = torch.empty((Ce, U), dtype=torch.uint8, device=dev)
        fk = torch.empty((Ce,), dtype=torch.uint8, device=dev)
        ftk = torch.empty((Ce, U), dtype=torch.uint8, device=dev)
        b = torch.empty((Ce, H), dtype=torch.float64, device=dev)
        bt = torch.empty((Ce, U), dtype=torch.float64, device=dev)
        (th, _th) = _lib.dbl_array(self.theta())
        ctx.call("fcd_model_sample", th, N, H, U, C.c_uint64(int(seed)), _lib.dptr(r), _lib.dptr(t), _lib.dptr(fk),
                 _lib.dptr(ftk), _lib.dptr(b), _lib.dptr(bt), _lib.stream_ptr())
        fk, ftk = fk.cpu().numpy().astype(np.int64), ftk.cpu().numpy().astype(np.int64)
        f = np.zeros((Ce, 3), dtype=bool)
        f[np.arange(Ce), fk] = True
        f_tilde = np.zeros((Ce, U, 3), dtype=bool)
        np.put_along_axis(f_tilde, ftk[:, :, None], True, axis=2)
        return (r.cpu().numpy() > 0, t.cpu().numpy() > 0, f, f_tilde, b.cpu().numpy(), bt.cpu().numpy())
