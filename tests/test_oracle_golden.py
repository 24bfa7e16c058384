"""
The CPU oracle (oracle/fcdiff_oracle.py) against fixtures captured from the reference itself
(tests/golden/G*.npz, made by oracle/capture_golden.py).  This is what pins the oracle; the GPU
parity tests then compare the HIP path with the oracle.

Tolerances: tables bit-exact (the reference's own test is assert_equal, test_fit.py:164-166);
everything that sums uses rtol 1e-12 (the reference's tests use assert_allclose's default 1e-7).
"""
import numpy as np
import numpy.testing as nptest
import pytest

from conftest import load_golden, theta_dict
from oracle import fcdiff_oracle as O


def test_G1_index_maps():
    g = load_golden("G1_index_maps")
    for N, C, Nb in zip(g["Ns"], g["Cs"], g["N_back"]):
        assert O.N_to_C(N) == C
        assert O.C_to_N(C) == Nb
    for c in range(45):
        assert tuple(g["c_to_nm_N10"][c]) == O.c_to_nm(c)
        assert O.nm_to_c(*g["c_to_nm_N10"][c]) == g["nm_to_c_N10"][c] == c
    # quirk Q1: the asymmetric id used by the q_R update for every ordered pair
    for (n, m), c in zip(g["ordered_pairs_N10"], g["nm_to_c_asym_N10"]):
        assert O.edge_id(n, m, O.EDGE_REFERENCE) == c
        assert O.edge_id(n, m, O.EDGE_SYMMETRIC) == O.edge_id(m, n, O.EDGE_SYMMETRIC)
    nptest.assert_array_equal(O.edge_endpoints(10), g["c_to_nm_N10"])


@pytest.mark.parametrize("name", ["G2_update_lps", "G2b_update_lps_default"])
def test_G2_lik_tables_bitwise(name):
    g = load_golden(name)
    th = theta_dict(g["theta"])
    lpB, pBt, lM = O.lik_tables(g["b"], g["bt"], th["mu"], th["sigma"], th["eta"], th["epsilon"])
    nptest.assert_equal(lpB, g["lp_B_g_F"])
    nptest.assert_equal(pBt, g["p_Bt_g_Ft"])
    nptest.assert_equal(lM, g["lM"])


def test_G3_eval_M():
    g = load_golden("G3_eval_M")
    for k in range(3):
        for l in range(3):
            M = O.eval_M(g["p"], float(g["eta"]), float(g["epsilon"]), k, l)
            nptest.assert_equal(M[0, 0], g["M"][k, l])
    for l in range(3):
        assert O.eval_M_eps(float(g["eta"]), float(g["epsilon"]), l) == g["M_eps"][l]


def test_G4_update_lq_F():
    g = load_golden("G4_update_lq_F")
    S_B = O.sum_lp_B(g["lp_B_g_F"])
    lq_F = O.update_lq_F(np.log(g["q_R"]), S_B, g["lM"], g["gamma"])
    nptest.assert_allclose(lq_F, g["lq_F"], rtol=1e-12)


def test_G5_update_lq_R_reference_edge_ids():
    g = load_golden("G5_update_lq_R")
    lq_R = O.update_lq_R(np.log(g["q_R"]), np.log(g["q_F"]), g["lM"], g["pi"], O.EDGE_REFERENCE)
    nptest.assert_allclose(lq_R, g["lq_R"], rtol=1e-12)
    # the documented (symmetric) pairing is a different function: must NOT reproduce the fixture
    sym = O.update_lq_R(np.log(g["q_R"]), np.log(g["q_F"]), g["lM"], g["pi"], O.EDGE_SYMMETRIC)
    assert np.max(np.abs(sym - g["lq_R"])) > 1e-3


def test_G6_energy_terms():
    g = load_golden("G6_energy_terms")
    S_B = O.sum_lp_B(g["lp_B_g_F"])
    t = O.energy_terms(np.log(g["q_F"]), np.log(g["q_R"]), S_B, g["lM"], g["gamma"], g["pi2"])
    nptest.assert_allclose(t, g["terms"], rtol=1e-12)
    e = O.eval_energy(np.log(g["q_F"]), np.log(g["q_R"]), S_B, g["lM"], g["gamma"], g["pi2"])
    nptest.assert_allclose(e, g["energy"], rtol=1e-12)


def test_G7_pi_gamma():
    g = load_golden("G7_pi_gamma")
    nptest.assert_allclose(O.update_pi(np.log(g["q_R"])), g["pi"], rtol=1e-14)
    nptest.assert_allclose(O.update_gamma(np.log(g["q_F"])), g["gamma"], rtol=1e-14)


def test_G8_derivative_helpers():
    g = load_golden("G8_derivatives")
    (mu, sigma, eps, eta) = (float(g["mu"]), float(g["sigma"]), float(g["epsilon"]), float(g["eta"]))
    for j in range(3):
        nptest.assert_allclose(O.eval_dE_dm(g["q_F"], g["q_R"], g["dlN_dm"], g["dlM_dm"], j),
                               g["dE_dm"][j], rtol=1e-12)
    for k in range(3):
        nptest.assert_allclose(O.eval_dlM_dh(g["norm3"], g["mix2"], eps, k), g["dlM_dh"][k], rtol=1e-14)
        for l in range(3):
            nptest.assert_allclose(O.eval_dlM_dm(g["norm2"], g["mix2"], mu, sigma, eta, eps, k, l),
                                   g["dlM_dm_kl"][k, l], rtol=1e-14)
            nptest.assert_allclose(O.eval_dlM_de(g["norm3"], g["mix2"], eta, k, l),
                                   g["dlM_de"][k, l], rtol=1e-14)
    nptest.assert_allclose(O.eval_dE_dh(g["q_R"], g["q_F"], g["norm3"], g["mix4"], eps), g["dE_dh"], rtol=1e-12)
    nptest.assert_allclose(O.eval_dE_de(g["q_R"], g["q_F"], g["norm3"], g["mix4"], eta), g["dE_de"], rtol=1e-12)
    nptest.assert_equal(O.eval_dlN_dm(g["bb"], mu, sigma), g["dlN_dm_fn"])
    nptest.assert_equal(O.eval_dlN_ds(g["bb"], mu, sigma), g["dlN_ds_fn"])
    nptest.assert_equal(O.eval_dN_dm(g["NN"], g["bb"], mu, sigma), g["dN_dm_fn"])
    nptest.assert_equal(O.eval_dN_ds(g["NN"], g["bb"], mu, sigma), g["dN_ds_fn"])


@pytest.mark.parametrize("name,iters", [("G10_vb_trajectory_cfg1", 4), ("G10b_vb_trajectory_cfg1_ideal", 4),
                                        ("G12_vb_trajectory_mid", 3)])
def test_G10_vb_trajectory(name, iters):
    g = load_golden(name)
    res = O.vb_fit(g["b"], g["bt"], theta_dict(g["theta0"]), max_iters=iters, check_convergence=False)
    nptest.assert_allclose(res["energy"], g["energy"], rtol=1e-11)
    for i, (lq_F, lq_R, pi, gamma) in enumerate(res["hist"]):
        nptest.assert_allclose(lq_F, g["lq_F"][i], rtol=1e-9, atol=1e-11)
        nptest.assert_allclose(lq_R, g["lq_R"][i], rtol=1e-9, atol=1e-11)
        nptest.assert_allclose(pi, g["pi"][i + 1], rtol=1e-11)
        nptest.assert_allclose(gamma, g["gamma"][i + 1], rtol=1e-11)


def test_vectorised_forms_against_reference_fixtures():
    """The whole-array restatements (bench.py's CPU baseline mode (ii)) against the reference's own outputs G4, G5, G6."""
    g = load_golden("G4_update_lq_F")
    lq_F = O.update_lq_F_vec(np.log(g["q_R"]), O.sum_lp_B(g["lp_B_g_F"]), g["lM"], g["gamma"])
    nptest.assert_allclose(lq_F, g["lq_F"], rtol=1e-12)
    g = load_golden("G5_update_lq_R")
    lq_R = O.update_lq_R_vec(np.log(g["q_R"]), np.log(g["q_F"]), g["lM"], g["pi"], O.EDGE_REFERENCE)
    nptest.assert_allclose(lq_R, g["lq_R"], rtol=1e-12)
    g = load_golden("G6_energy_terms")
    nptest.assert_allclose(O.eval_E_lM_vec(g["q_F"], g["q_R"], g["lM"]), g["terms"][3], rtol=1e-12)


@pytest.mark.parametrize("vectorised", [False, True])
def test_vb_iteration_walks_the_reference_trajectory(vectorised):
    """vb_iteration (what bench.py times on the host, both modes) reproduces G10 iteration by iteration."""
    g = load_golden("G10_vb_trajectory_cfg1")
    th = theta_dict(g["theta0"])
    (C, U) = (g["bt"].shape[0], g["bt"].shape[1])
    N = int(O.C_to_N(C))
    lq_R = np.full((N, U, 2), -np.log(2))
    lq_F = np.full((C, 1, 3), -np.log(3))
    for i in range(3):
        (lq_F, lq_R, th, e) = O.vb_iteration(lq_F, lq_R, g["b"], g["bt"], th, vectorised=vectorised)
        nptest.assert_allclose(e, g["energy"][i + 1], rtol=1e-11)
        nptest.assert_allclose(lq_R, g["lq_R"][i], rtol=1e-9, atol=1e-11)


def test_G10_survey_anchor():
    """SURVEY.md section 8c / BASELINE.md section 2 quote these three energies."""
    g = load_golden("G10_vb_trajectory_cfg1")
    nptest.assert_allclose(g["energy"][:3], [2745.2108470540657, -620.1368815243751, -620.5198335492853],
                           rtol=1e-13)


def test_is_converged_cases():
    """test_fcdiff/test_fit.py:86-129."""
    for (e, expect) in (([1, 1.25], True), ([1, 1], True), ([1, 0.501], True), ([1, 0.5], False),
                        ([1, 0.499], False)):
        assert bool(O.is_converged(e, 1, 0.5)) is expect


@pytest.mark.parametrize("tag", ["cfg1", "mid"])
def test_G11_gibbs_conditionals(tag):
    """The sampler's two conditionals against the reference's own functions at one-hot q."""
    g = load_golden("G11_gibbs_conditionals_" + tag)
    th = theta_dict(g["theta"])
    (r, f, lM) = (g["r_state"], g["f_state"], g["lM"])
    (Nreg, U) = r.shape
    C = f.shape[0]
    S_B = O.sum_lp_B(g["lp_B_g_F"])
    lng = np.log(th["gamma"])
    lnpi2 = np.log([1 - th["pi"], th["pi"]])
    # (a) f conditional == _update_lq_F at one-hot q_R
    for c in range(C):
        a = O.f_conditional_logits(c, r, S_B, lM, lng)
        a = a - O.logsumexp(a, axis=0)
        nptest.assert_allclose(a, g["cond_f"][c, 0], rtol=1e-10, atol=1e-10)
    # (b) r conditional of region 0 == row 0 of _update_lq_R (reference edge ids) at one-hot q
    for u in range(U):
        s = np.array(O.r_conditional_logits(0, u, f, r, lM, lnpi2, O.EDGE_REFERENCE))
        s = s - O.logsumexp(s, axis=0)
        nptest.assert_allclose(s, g["lq_R_after_update"][0, u], rtol=1e-10, atol=1e-10)
    #     ... and the whole softened Gauss-Seidel result through the oracle's VB update
    q_R = np.stack([1.0 - r, 1.0 * r], axis=2)
    q_F = np.zeros((C, 1, 3))
    q_F[np.arange(C), 0, f] = 1
    with np.errstate(divide="ignore"):
        full = O.update_lq_R(np.log(q_R), np.log(q_F), lM, [1 - th["pi"], th["pi"]], O.EDGE_REFERENCE)
    nptest.assert_allclose(full, g["lq_R_after_update"], rtol=1e-9, atol=1e-10)
    # (c) symmetric-mode conditionals == differences of the reference's log-joint
    lj = O.gibbs_logjoint(f[None], r[None], S_B, lM, lng, lnpi2)[0]
    nptest.assert_allclose(lj, g["logjoint_base"], rtol=1e-12)
    for n in range(Nreg):
        for u in range(U):
            (s0, s1) = O.r_conditional_logits(n, u, f, r, lM, lnpi2, O.EDGE_SYMMETRIC)
            nptest.assert_allclose(s1 - s0, g["logjoint_r"][n, u, 1] - g["logjoint_r"][n, u, 0],
                                   rtol=1e-8, atol=1e-8)
    for c in range(C):
        a = O.f_conditional_logits(c, r, S_B, lM, lng)
        nptest.assert_allclose(a - a[0], g["logjoint_f"][c] - g["logjoint_f"][c, 0], rtol=1e-8, atol=1e-8)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert O.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert O.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert O.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
