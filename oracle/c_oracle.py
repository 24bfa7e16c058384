"""
ctypes view of oracle/liboracle.so (the C restatement, oracle/fcdiff_oracle.c).  TEST INFRASTRUCTURE
ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_i64, _u64, _int = C.c_int64, C.c_uint64, C.c_int
_pd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_pb = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_pi = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        L.oracle_lik_tables.argtypes = [_pd, _pd, _i64, _i64, _i64, _pd, _pd, _pd]
        L.oracle_update_lq_F.argtypes = [_pd, _pd, _pd, _pd, _i64, _i64, _pd]
        L.oracle_update_lq_R.argtypes = [_pd, _pd, _pd, _i64, _i64, _int, _pd]
        L.oracle_energy_terms.argtypes = [_pd, _pd, _pd, _pd, _pd, _pd, _i64, _i64, _pd]
        L.oracle_gibbs_init.argtypes = [_pb, _pb, _i64, _i64, _i64, _i64, _u64, C.c_double]
        L.oracle_gibbs_f_step.argtypes = [_pb, _pb, _pd, _pd, _pd, _i64, _i64, _i64, _i64, _u64, _i64, C.c_void_p, _int]
        L.oracle_gibbs_r_step.argtypes = [_pb, _pb, _pd, _pd, _i64, _i64, _i64, _i64, _u64, _i64, _int, C.c_void_p, _int]
        L.oracle_gibbs_f_step_m.argtypes = [_pb, _pb, _pd, _pd, _pd, _i64, _i64, _i64, _i64, _u64, _i64, _pd]
        L.oracle_gibbs_r_step_m.argtypes = [_pb, _pb, _pd, _pd, _i64, _i64, _i64, _i64, _u64, _i64, _int, _pd]
        L.oracle_gibbs_stats.argtypes = [_pb, _pb, _i64, _i64, _i64, _pi]
        L.oracle_gibbs_logjoint.argtypes = [_pb, _pb, _pd, _pd, _pd, _pd, _i64, _i64, _i64, _pd]
        L.oracle_max_threads.restype = _int
        for fn in ("oracle_lik_tables", "oracle_update_lq_F", "oracle_update_lq_R", "oracle_energy_terms",
                   "oracle_gibbs_init", "oracle_gibbs_f_step", "oracle_gibbs_r_step", "oracle_gibbs_f_step_m",
                   "oracle_gibbs_r_step_m", "oracle_gibbs_stats", "oracle_gibbs_logjoint"):
            getattr(L, fn).restype = None
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def n_regions(C_edges):
    n = int(round((np.sqrt(8 * C_edges + 1) - 1) / 2 + 1))
    assert n * (n - 1) // 2 == C_edges
    return n


def lik_tables(b, bt, theta):
    (Ce, H) = b.shape
    U = bt.shape[1]
    S_B = np.empty((Ce, 3))
    lM = np.empty((Ce, U, 3, 3))
    lib().oracle_lik_tables(_f64(b), _f64(bt), Ce, H, U, _f64(theta), S_B, lM)
    return S_B, lM


def update_lq_F(lq_R, S_B, lM, gamma):
    (Nreg, U) = lq_R.shape[0:2]
    out = np.empty((lM.shape[0], 1, 3))
    lib().oracle_update_lq_F(_f64(lq_R), _f64(S_B), _f64(lM), _f64(gamma), Nreg, U, out)
    return out


def update_lq_R(lq_R, lq_F, lM, pi2, mode):
    (Nreg, U) = lq_R.shape[0:2]
    out = _f64(lq_R).copy()
    lib().oracle_update_lq_R(_f64(lq_F), _f64(lM), _f64(pi2), Nreg, U, int(mode), out)
    return out


def energy_terms(lq_F, lq_R, S_B, lM, gamma, pi2):
    (Nreg, U) = lq_R.shape[0:2]
    out = np.empty(6)
    lib().oracle_energy_terms(_f64(lq_F), _f64(lq_R), _f64(S_B), _f64(lM), _f64(gamma), _f64(pi2), Nreg, U, out)
    return out


def gibbs_init(G, Nreg, U, pi, seed, chain0=0):
    f = np.zeros((G, Nreg * (Nreg - 1) // 2), dtype=np.uint8)
    r = np.zeros((G, Nreg, U), dtype=np.uint8)
    lib().oracle_gibbs_init(f, r, Nreg, U, G, chain0, seed, float(pi))
    return f, r


def gibbs_f_step(f, r, S_B, lM, lngamma, seed, sweep, chain0=0, want_cond=False, draw=True):
    (G, Nreg, U) = r.shape
    cond = np.empty((G, f.shape[1], 3)) if want_cond else None
    lib().oracle_gibbs_f_step(f, r, _f64(S_B), _f64(lM), _f64(lngamma), Nreg, U, G, chain0, seed, sweep,
                              cond.ctypes.data if want_cond else None, int(draw))
    return cond


def gibbs_r_step(f, r, lM, lnpi2, seed, sweep, mode, chain0=0, want_cond=False, draw=True):
    (G, Nreg, U) = r.shape
    cond = np.empty((G, Nreg, U, 2)) if want_cond else None
    lib().oracle_gibbs_r_step(f, r, _f64(lM), _f64(lnpi2), Nreg, U, G, chain0, seed, sweep, int(mode),
                              cond.ctypes.data if want_cond else None, int(draw))
    return cond


def gibbs_f_step_margin(f, r, S_B, lM, lngamma, seed, sweep, chain0=0):
    """gibbs_f_step, returning the smallest tie margin of its draws (distance of t = x * sum(w) to the nearer threshold,
    relative to sum(w)): a statistic for the parity tests, not part of the algorithm."""
    (G, Nreg, U) = r.shape
    out = np.zeros(1)
    lib().oracle_gibbs_f_step_m(f, r, _f64(S_B), _f64(lM), _f64(lngamma), Nreg, U, G, chain0, seed, sweep, out)
    return float(out[0])


def gibbs_r_step_margin(f, r, lM, lnpi2, seed, sweep, mode, chain0=0):
    """gibbs_r_step, returning the smallest |(s1 - s0) - logit(x)| of its draws."""
    (G, Nreg, U) = r.shape
    out = np.zeros(1)
    lib().oracle_gibbs_r_step_m(f, r, _f64(lM), _f64(lnpi2), Nreg, U, G, chain0, seed, sweep, int(mode), out)
    return float(out[0])


def gibbs_stats(f, r):
    (G, Nreg, U) = r.shape
    out = np.zeros(8, dtype=np.int64)
    lib().oracle_gibbs_stats(f, r, Nreg, U, G, out)
    return out


def gibbs_logjoint(f, r, S_B, lM, lngamma, lnpi2):
    (G, Nreg, U) = r.shape
    out = np.empty(G)
    lib().oracle_gibbs_logjoint(f, r, _f64(S_B), _f64(lM), _f64(lngamma), _f64(lnpi2), Nreg, U, G, out)
    return out


def max_threads():
    return int(lib().oracle_max_threads())
