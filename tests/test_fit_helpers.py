"""
Module-level helpers of fcdiff_amd.fit (reference names, fcdiff/fit.py:382-733) against fixtures captured
from the reference (G3, G6, G8) -- the ports of test_fcdiff/test_fit.py:170-425 and :560-1087.
"""
import numpy as np
import numpy.testing as nptest
import pytest

from conftest import load_golden
from fcdiff_amd import fit as F


def test_eval_M_nine_cases():
    g = load_golden("G3_eval_M")
    (eta, eps) = (float(g["eta"]), float(g["epsilon"]))
    for k in range(3):
        for l in range(3):
            nptest.assert_allclose(F._eval_M(g["p"], eta, eps, k, l)[0, 0], g["M"][k, l], rtol=1e-15)
    for l in range(3):
        assert F._eval_M_eps(eta, eps, l) == g["M_eps"][l]


def test_energy_term_functions():
    g = load_golden("G6_energy_terms")
    (q_F, q_R) = (g["q_F"], g["q_R"])
    got = [F._eval_E_lp_F(q_F, g["gamma"]), F._eval_E_lp_B_g_F(q_F, g["lp_B_g_F"]), F._eval_E_lp_R(q_R, g["pi2"]),
           F._eval_E_lM(q_F, q_R, g["lM"]), F._eval_E_lq_F(q_F, np.log(q_F)), F._eval_E_lq_R(q_R, np.log(q_R))]
    nptest.assert_allclose(got, g["terms"], rtol=1e-12)


def test_q_R_w():
    g = load_golden("G6_energy_terms")
    q_R = g["q_R"]
    w = F._eval_q_R_w(q_R, 3, 1)
    assert w.shape == (q_R.shape[1], 3)
    nptest.assert_allclose(w.sum(axis=1), 1.0)
    nptest.assert_allclose(w[:, 2], q_R[3, :, 0] * q_R[1, :, 1] + q_R[3, :, 1] * q_R[1, :, 0])


def test_derivative_helpers():
    g = load_golden("G8_derivatives")
    (mu, sigma, eps, eta) = (float(g["mu"]), float(g["sigma"]), float(g["epsilon"]), float(g["eta"]))
    for j in range(3):
        nptest.assert_allclose(F._eval_dE_dm(g["q_F"], g["q_R"], g["dlN_dm"], g["dlM_dm"], j), g["dE_dm"][j], rtol=1e-12)
    for k in range(3):
        nptest.assert_allclose(F._eval_dlM_dh(g["norm3"], g["mix2"], eps, k), g["dlM_dh"][k], rtol=1e-14)
        for l in range(3):
            nptest.assert_allclose(F._eval_dlM_dm(g["norm2"], g["mix2"], mu, sigma, eta, eps, k, l), g["dlM_dm_kl"][k, l], rtol=1e-14)
            nptest.assert_allclose(F._eval_dlM_de(g["norm3"], g["mix2"], eta, k, l), g["dlM_de"][k, l], rtol=1e-14)
    nptest.assert_allclose(F._eval_dE_dh(g["q_R"], g["q_F"], g["norm3"], g["mix4"], eps), g["dE_dh"], rtol=1e-12)
    nptest.assert_allclose(F._eval_dE_de(g["q_R"], g["q_F"], g["norm3"], g["mix4"], eta), g["dE_de"], rtol=1e-12)
    nptest.assert_equal(F._eval_dlN_dm(g["bb"], mu, sigma), g["dlN_dm_fn"])
    nptest.assert_equal(F._eval_dlN_ds(g["bb"], mu, sigma), g["dlN_ds_fn"])
    nptest.assert_equal(F._eval_dN_dm(g["NN"], g["bb"], mu, sigma), g["dN_dm_fn"])
    nptest.assert_equal(F._eval_dN_ds(g["NN"], g["bb"], mu, sigma), g["dN_ds_fn"])


def test_check_state_refuses_mismatched_private_state_before_any_launch():
    """
    The private state is NumPy-assignable (the reference's tests set _lq_R, _lq_F, _lM directly).  A shape that does
    not match the tables must raise on the host -- the kernels index the buffers unconditionally.  No GPU is needed:
    the check only looks at tensor shapes.
    """
    import pytest
    import torch
    from fcdiff_amd.fit import UnsharedRegionFit
    (N, U) = (6, 4)
    C = N * (N - 1) // 2
    fit = UnsharedRegionFit()
    fit._d["lM"] = torch.zeros((C, U, 3, 3), dtype=torch.float64)
    fit._d["S_B"] = torch.zeros((C, 3), dtype=torch.float64)
    fit._d["lq_R"] = torch.zeros((N, U, 2), dtype=torch.float64)
    fit._d["lq_F"] = torch.zeros((C, 1, 3), dtype=torch.float64)
    assert fit._check_state() == (N, C, U)
    for (key, bad) in (("lq_R", (N, 1, 2)), ("lq_R", (N + 1, U, 2)), ("lq_F", (C - 1, 1, 3)), ("lq_F", (C, 3)),
                       ("S_B", (C, 4)), ("lq_R", (N, U + 3, 2))):
        good = fit._d[key]
        fit._d[key] = torch.zeros(bad, dtype=torch.float64)
        with pytest.raises(ValueError):
            fit._check_state()
        for method in (fit._update_lq_F, fit._update_lq_R, fit._eval_energy, fit._theta_sub_weights):
            if key == "lq_F" and method == fit._update_lq_F:
                continue            # _update_lq_F overwrites _lq_F; it does not read it
            if key == "S_B" and method in (fit._update_lq_R, fit._theta_sub_weights):
                continue            # these do not read S_B
            with pytest.raises(ValueError):
                method()            # raises before the context (and so before any GPU) is touched
        fit._d[key] = good
    fit._d["lM"] = torch.zeros((C - 1, U, 3, 3), dtype=torch.float64)          # not a triangular number of edges
    with pytest.raises(ValueError, match="triangular"):
        fit._check_state()
    fit._d["lM"] = torch.zeros((C, U, 3), dtype=torch.float64)
    with pytest.raises(ValueError):
        fit._check_state()


def test_data_digest_sees_in_place_edits():
    """
    Contract of UnsharedRegionFit.data_check (the reference re-reads b / bt on every _update_lps, fcdiff/fit.py:111-115):
    'full' sees every in-place edit; the default 'sample' sees whole-array and whole-column edits at a cost that does
    not grow with the array (the whole-array CRC on every call made a variational iteration 16x slower);
    single-element edits between two calls need invalidate_data().
    """
    import time
    from fcdiff_amd.fit import UnsharedRegionFit
    rng = np.random.RandomState(0)
    for mode in ("full", "sample"):
        b = rng.rand(300, 7)
        d0 = UnsharedRegionFit._data_digest(b, mode)
        b *= 0.5
        assert UnsharedRegionFit._data_digest(b, mode) != d0
        d1 = UnsharedRegionFit._data_digest(b, mode)
        np.clip(b, 0.1, 0.4, out=b)
        assert UnsharedRegionFit._data_digest(b, mode) != d1
        # one patient's column replaced in place at cfg3's shape (round 2's strided sample never looked at column 3)
        bt = rng.rand(19900, 50)
        for col in (0, 3, 17, 49):
            d2 = UnsharedRegionFit._data_digest(bt, mode)
            bt[:, col] = rng.rand(19900)
            assert UnsharedRegionFit._data_digest(bt, mode) != d2
    d3 = UnsharedRegionFit._data_digest(bt, "full")
    bt[12345, 17] += 1e-12
    assert UnsharedRegionFit._data_digest(bt, "full") != d3
    assert UnsharedRegionFit._data_digest(bt, "none") == 0
    # the default costs microseconds at cfg3's shape, whatever the size
    t0 = time.perf_counter()
    for _ in range(20):
        UnsharedRegionFit._data_digest(bt, "sample")
    assert (time.perf_counter() - t0) / 20 < 1e-3
    with pytest.raises(ValueError):
        UnsharedRegionFit._data_digest(bt, "crc")
    assert UnsharedRegionFit._array_key(bt) == UnsharedRegionFit._array_key(bt)
    assert UnsharedRegionFit._array_key(bt) != UnsharedRegionFit._array_key(bt.copy())
