"""
The C oracle (oracle/fcdiff_oracle.c) against the NumPy oracle, which is itself pinned to the reference's
fixtures (test_oracle_golden.py).  The C oracle is what the GPU parity tests and the CPU baseline use at
sizes the NumPy one is too slow for.
"""
import numpy as np
import numpy.testing as nptest
import pytest

from conftest import load_golden, theta_dict
from oracle import c_oracle as CO
from oracle import fcdiff_oracle as O


def test_c_tables_match_reference_fixture():
    for name in ("G2_update_lps", "G2b_update_lps_default"):
        g = load_golden(name)
        S_B, lM = CO.lik_tables(g["b"], g["bt"], g["theta"])
        nptest.assert_allclose(lM, g["lM"], rtol=1e-13, atol=1e-15)      # libm exp/log vs NumPy: a few ulp of M
        nptest.assert_allclose(S_B, g["lp_B_g_F"].sum(axis=1), rtol=1e-13)


def test_c_vb_updates_match_reference_fixtures():
    g = load_golden("G4_update_lq_F")
    out = CO.update_lq_F(np.log(g["q_R"]), g["lp_B_g_F"].sum(axis=1), g["lM"], g["gamma"])
    nptest.assert_allclose(out, g["lq_F"], rtol=1e-12)
    g = load_golden("G5_update_lq_R")
    out = CO.update_lq_R(np.log(g["q_R"]), np.log(g["q_F"]), g["lM"], g["pi"], O.EDGE_REFERENCE)
    nptest.assert_allclose(out, g["lq_R"], rtol=1e-12)
    g = load_golden("G6_energy_terms")
    t = CO.energy_terms(np.log(g["q_F"]), np.log(g["q_R"]), g["lp_B_g_F"].sum(axis=1), g["lM"], g["gamma"], g["pi2"])
    nptest.assert_allclose(t, g["terms"], rtol=1e-12)


@pytest.mark.parametrize("mode", [O.EDGE_REFERENCE, O.EDGE_SYMMETRIC])
def test_c_gibbs_equals_numpy_gibbs(mode):
    """Same seeds -> identical chains, state for state, through init + 3 sweeps."""
    g = load_golden("G11_gibbs_conditionals_cfg1")
    th = theta_dict(g["theta"])
    S_B = g["lp_B_g_F"].sum(axis=1)
    lM = g["lM"]
    (Nreg, U) = g["r_state"].shape
    lng, lnpi2 = np.log(th["gamma"]), np.log([1 - th["pi"], th["pi"]])
    (G, seed, chain0) = (3, 0x1234567890ABCDEF, 5)
    f_np, r_np = O.gibbs_init(G, Nreg, U, 0.3, seed, chain0)
    f_c, r_c = CO.gibbs_init(G, Nreg, U, 0.3, seed, chain0)
    nptest.assert_array_equal(f_np, f_c)
    nptest.assert_array_equal(r_np, r_c)
    assert 0 < r_c.mean() < 1 and set(np.unique(f_c)) == {0, 1, 2}
    for sweep in range(3):
        O.gibbs_f_step(f_np, r_np, S_B, lM, lng, seed, sweep, chain0)
        CO.gibbs_f_step(f_c, r_c, S_B, lM, lng, seed, sweep, chain0)
        nptest.assert_array_equal(f_np, f_c)
        O.gibbs_r_step(f_np, r_np, lM, lnpi2, seed, sweep, mode, chain0)
        CO.gibbs_r_step(f_c, r_c, lM, lnpi2, seed, sweep, mode, chain0)
        nptest.assert_array_equal(r_np, r_c)
    nptest.assert_array_equal(O.gibbs_stats(f_np, r_np), CO.gibbs_stats(f_c, r_c)[:4])
    nptest.assert_allclose(CO.gibbs_logjoint(f_c, r_c, S_B, lM, lng, lnpi2),
                           O.gibbs_logjoint(f_np, r_np, S_B, lM, lng, lnpi2), rtol=1e-13)


def test_c_tie_margin_steps_are_the_same_steps():
    """The steps that also report the closest tie of their draws (a statistic of the GPU parity tests) draw the same states;
    the margins are what the conditionals say they are."""
    g = load_golden("G11_gibbs_conditionals_cfg1")
    th = theta_dict(g["theta"])
    S_B = g["lp_B_g_F"].sum(axis=1)
    lM = g["lM"]
    (Nreg, U) = g["r_state"].shape
    lng, lnpi2 = np.log(th["gamma"]), np.log([1 - th["pi"], th["pi"]])
    (G, seed) = (5, 77)
    f_a, r_a = CO.gibbs_init(G, Nreg, U, 0.3, seed, 0)
    f_b, r_b = f_a.copy(), r_a.copy()
    for sweep in range(2):
        CO.gibbs_f_step(f_a, r_a, S_B, lM, lng, seed, sweep, 0)
        mf = CO.gibbs_f_step_margin(f_b, r_b, S_B, lM, lng, seed, sweep, 0)
        nptest.assert_array_equal(f_a, f_b)
        cond = CO.gibbs_r_step(f_a, r_a, lM, lnpi2, seed, sweep, O.EDGE_SYMMETRIC, 0, want_cond=True)
        mr = CO.gibbs_r_step_margin(f_b, r_b, lM, lnpi2, seed, sweep, O.EDGE_SYMMETRIC, 0)
        nptest.assert_array_equal(r_a, r_b)
        assert 0.0 < mf <= 0.5 and 0.0 < mr < np.abs(cond[..., 1] - cond[..., 0]).max() + 50.0
        # the r margin cannot be smaller than the distance of any draw to ITS decision: flipping needs |v| to vanish
        assert mr < 1.0          # (5 x 24 x 6 draws: some draw lies within 1 of its threshold)


def test_c_conditionals_match_reference_pins():
    """C conditionals against the reference's own one-hot evaluations (G11)."""
    g = load_golden("G11_gibbs_conditionals_mid")
    th = theta_dict(g["theta"])
    S_B = g["lp_B_g_F"].sum(axis=1)
    f = g["f_state"][None].copy()
    r = g["r_state"][None].copy()
    lng, lnpi2 = np.log(th["gamma"]), np.log([1 - th["pi"], th["pi"]])
    cf = CO.gibbs_f_step(f, r, S_B, g["lM"], lng, 0, 0, want_cond=True, draw=False)[0]
    cf = cf - O.logsumexp(cf, axis=1)
    nptest.assert_allclose(cf, g["cond_f"][:, 0, :], rtol=1e-10, atol=1e-10)
    cr = CO.gibbs_r_step(f, r, g["lM"], lnpi2, 0, 0, O.EDGE_REFERENCE, want_cond=True, draw=False)[0]
    row0 = cr[0] - O.logsumexp(cr[0], axis=1)
    nptest.assert_allclose(row0, g["lq_R_after_update"][0], rtol=1e-10, atol=1e-10)
    cs = CO.gibbs_r_step(f, r, g["lM"], lnpi2, 0, 0, O.EDGE_SYMMETRIC, want_cond=True, draw=False)[0]
    nptest.assert_allclose(cs[:, :, 1] - cs[:, :, 0], g["logjoint_r"][:, :, 1] - g["logjoint_r"][:, :, 0],
                           rtol=1e-8, atol=1e-8)
    nptest.assert_array_equal(f[0], g["f_state"])       # draw=False leaves the state alone
    nptest.assert_array_equal(r[0], g["r_state"])


def test_c_philox_known_answer():
    import ctypes as C
    L = CO.lib()
    out = (C.c_uint32 * 4)()
    L.oracle_philox((C.c_uint32 * 4)(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                    (C.c_uint32 * 2)(0xa4093822, 0x299f31d0), out)
    assert tuple(out) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
