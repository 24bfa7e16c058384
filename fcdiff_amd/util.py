"""
Edge <-> region-pair index maps.  Mirrors fcdiff/util.py:7-84 (same names, same meaning).

Edges are numbered in lower-triangular row-major order, c = n(n-1)/2 + m with n > m: this is the
"edge-major" order of every table of the fit path.  Unlike the Python-2 original these return ints
(its `/` was integer division), except C_to_N which returns a float exactly as the reference does --
run() uses `N % 1 != 0` on it as the triangular-number check (fcdiff/fit.py:62-65).
"""
import numpy as np


def N_to_C(N):
    """Number of connections of a network with N regions (util.py:7-21)."""
    N = int(N)
    return N * (N - 1) // 2


def C_to_N(C):
    """Number of regions (float) of a network with C connections (util.py:23-38)."""
    return (np.sqrt(8 * C + 1) - 1) / 2 + 1


def nm_to_c(n, m):
    """Connection index of the region pair (n, m), n > m (util.py:40-60)."""
    return N_to_C(n) + int(m)


def c_to_nm(c):
    """Region pair (n, m), n > m, of connection c (util.py:62-84)."""
    n = int(np.floor((np.sqrt(8 * c + 1) - 1) / 2) + 1)
    while N_to_C(n) > c:          # guard the float sqrt at large c
        n -= 1
    while N_to_C(n + 1) <= c:
        n += 1
    return (n, int(c) - N_to_C(n))


def upper_to_lower_edge_order(N):
    """
    Permutation p with lower_order[c] = upper_order[p[c]].

    The forward sampler enumerates edges upper-triangular row-major (fcdiff/model.py:133-142) while the
    fitter uses the lower-triangular order above; apply `t[p]`, `bt[p]` before
    fitting data drawn from `UnsharedRegionModel.sample` if region/edge association matters.
    """
    idx = {}
    k = 0
    for n in range(N):
        for m in range(n + 1, N):
            idx[(m, n)] = k
            k += 1
    return np.array([idx[c_to_nm(c)] for c in range(N_to_C(N))], dtype=np.int64)
