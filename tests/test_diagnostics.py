"""Chain diagnostics on synthetic traces with known answers."""
import numpy as np

from fcdiff_amd import diagnostics as D


def ar1(rs, chains, draws, rho):
    x = np.zeros((chains, draws))
    e = rs.standard_normal((chains, draws)) * np.sqrt(1 - rho * rho)
    x[:, 0] = rs.standard_normal(chains)
    for t in range(1, draws):
        x[:, t] = rho * x[:, t - 1] + e[:, t]
    return x


def test_iid_chains():
    rs = np.random.RandomState(0)
    x = rs.standard_normal((16, 500))
    assert abs(D.split_rhat(x) - 1.0) < 0.01
    assert 0.8 * x.size < D.ess(x) < 1.25 * x.size


def test_disagreeing_chains_raise_rhat():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((8, 400))
    x[:4] += 3.0
    assert D.split_rhat(x) > 1.5
    y = rs.standard_normal((8, 400))
    y[:, 200:] += 2.0                       # drift inside every chain: only the SPLIT statistic sees it
    assert D.split_rhat(y) > 1.3


def test_autocorrelated_chains_lose_samples():
    rs = np.random.RandomState(2)
    rho = 0.9
    x = ar1(rs, 16, 2000, rho)
    expect = x.size * (1 - rho) / (1 + rho)
    assert 0.6 * expect < D.ess(x) < 1.6 * expect
    assert abs(D.split_rhat(x) - 1.0) < 0.05
    s = D.summary(x)
    assert s["chains"] == 16 and s["draws"] == 2000 and abs(s["mean"]) < 0.2
