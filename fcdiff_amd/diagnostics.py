"""
Chain diagnostics for the many-chain sampler: split-R-hat and effective sample size of
scalar traces, e.g. the per-chain log-joint (`GibbsEngine.logjoint()`, minus the first four terms of the
reference's free energy, fcdiff/fit.py:149-152, at the chain's state).  Host NumPy on small (chains, draws) arrays:
this is analysis of a few thousand numbers, not part of the sweep.

Definitions follow Gelman et al., Bayesian Data Analysis 3rd ed., section 11.4-11.5 (split chains; Geyer's initial
positive sequence for the autocorrelation sum).
"""
import numpy as np


def _split(x):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError("trace must have shape (chains, draws)")
    n = x.shape[1] // 2
    if n < 2:
        raise ValueError("need at least 4 draws per chain")
    return np.concatenate([x[:, :n], x[:, x.shape[1] - n:]], axis=0)


def split_rhat(x):
    """Potential scale reduction of a (chains, draws) trace after splitting each chain in halves."""
    s = _split(x)
    (m, n) = s.shape
    W = np.mean(np.var(s, axis=1, ddof=1))
    B = n * np.var(np.mean(s, axis=1), ddof=1)
    if W == 0:
        return 1.0 if B == 0 else np.inf
    var_plus = (n - 1) / n * W + B / n
    return float(np.sqrt(var_plus / W))


def ess(x):
    """Effective sample size of a (chains, draws) trace (split chains, Geyer initial positive sequence)."""
    s = _split(x)
    (m, n) = s.shape
    W = np.mean(np.var(s, axis=1, ddof=1))
    B = n * np.var(np.mean(s, axis=1), ddof=1)
    var_plus = (n - 1) / n * W + B / n
    if var_plus == 0:
        return float(m * n)
    c = s - s.mean(axis=1, keepdims=True)
    # variogram-based autocorrelation estimate, averaged over chains
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(c, nfft, axis=1)
    acov = np.fft.irfft(f * np.conj(f), nfft, axis=1)[:, :n] / n          # biased autocovariance per chain
    rho = 1.0 - (W - acov.mean(axis=0) * n / (n - 1)) / var_plus
    tau = -1.0
    t = 0
    while t + 1 < n:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        tau += 2.0 * pair
        t += 2
    tau = max(tau, 1.0 / np.log10(m * n)) if m * n > 10 else max(tau, 1.0)
    return float(m * n / tau)


def summary(trace):
    """{'rhat', 'ess', 'mean', 'sd', 'chains', 'draws'} of a (chains, draws) trace."""
    x = np.asarray(trace, dtype=np.float64)
    return dict(rhat=split_rhat(x), ess=ess(x), mean=float(x.mean()), sd=float(x.std(ddof=1)),
                chains=int(x.shape[0]), draws=int(x.shape[1]))
