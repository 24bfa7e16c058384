#!/usr/bin/env python3
"""
Golden-vector capture for the fcdiff fit path.  TEST INFRASTRUCTURE, never shipped logic.

Run ONLY in the build container, where the read-only reference checkout is
mounted at /root/reference:

    python oracle/capture_golden.py            # writes tests/golden/*.npz

The script contains no reference source.  It loads the reference's three
modules *from where they lie* at run time (importlib, by path), calls them on
seeded inputs and stores inputs + outputs as small .npz fixtures.  Only the
fixtures travel to the GPU box (the reference itself never does), so on any
machine without /root/reference this script just exits with a message.

The reference is Python-2 era code.  The non-invasive load recipe (SURVEY.md
section 8c) is:
  1. register an empty package object "fcdiff" whose __path__ points at the
     reference package directory and load util.py / model.py / fit.py by path
     (this skips fcdiff/__init__.py, whose implicit relative import is the only
     thing that fails on Python 3);
  2. scipy.misc.logsumexp (removed from SciPy) -> scipy.special.logsumexp;
  3. restore Python-2 integer semantics of the three index helpers
     (fcdiff/util.py:21,60,82-84 rely on int '/' and return floats on Py3);
  4. after _init_lps cast the three table buffers to float64
     (fcdiff/fit.py:100-102 np.full(shape, 1) is int64 on modern NumPy);
  5. pass pi as the 2-vector [1-pi, pi] where the reference indexes it
     (fcdiff/fit.py:183, :486) -- its own tests do the same
     (test_fcdiff/test_fit.py:208, 477-487).

Fixture ids follow SURVEY.md section 8c (G1..G11) plus G12 (a cfg2-sized VB
trajectory used by the GPU parity tests).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy
import scipy.special

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference():
    sys.dont_write_bytecode = True
    import scipy.misc  # noqa: F401  (still importable, lacks logsumexp)
    if not hasattr(scipy.misc, "logsumexp"):
        scipy.misc.logsumexp = scipy.special.logsumexp
    pkg = types.ModuleType("fcdiff")
    pkg.__path__ = [os.path.join(REF, "fcdiff")]
    sys.modules["fcdiff"] = pkg
    mods = {}
    for name in ("util", "model", "fit"):
        spec = importlib.util.spec_from_file_location(
            "fcdiff." + name, os.path.join(REF, "fcdiff", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["fcdiff." + name] = mod
        if name == "util":
            spec.loader.exec_module(mod)
            util = mod
            # Python-2 integer semantics of the index helpers.
            util.N_to_C = lambda N: int(N) * (int(N) - 1) // 2
            util.nm_to_c = lambda n, m: util.N_to_C(n) + int(m)

            def c_to_nm(c):
                n = int(np.floor((np.sqrt(8 * c + 1) - 1) / 2) + 1)
                return (n, int(c) - util.N_to_C(n))
            util.c_to_nm = c_to_nm
            pkg.N_to_C = util.N_to_C
            pkg.nm_to_c = util.nm_to_c
            pkg.c_to_nm = util.c_to_nm
            pkg.util = util
        else:
            spec.loader.exec_module(mod)
        mods[name] = mod
        setattr(pkg, name, mod)
    pkg.UnsharedRegionModel = mods["model"].UnsharedRegionModel
    return pkg


# --- the reference tests' input helpers (test_fcdiff/test_fit.py:11-64), restated ---
def rand(lower, upper, shape, seed=0):
    return np.random.RandomState(seed).uniform(lower, upper, size=shape)


def rand_prob(shape, seed=0):
    return rand(1e-7, 1, shape, seed=seed)


def rand_prob_vector(shape, seed=0):
    p = rand_prob(shape, seed=seed)
    p /= np.sum(p, axis=-1, keepdims=True)
    return p


def ideal_model(fcdiff):
    m = fcdiff.UnsharedRegionModel()
    m.pi = 0.1
    m.epsilon = 0.01
    m.eta = 0.3
    m.gamma = np.ones((3,)) / 3
    m.mu = np.array([-0.5, 0, 0.5])
    m.sigma = np.ones((3,)) * 0.05
    return m


def theta_of(model):
    """theta[12] = pi, eta, epsilon, gamma[3], mu[3], sigma[3] (scalar pi)."""
    pi = np.atleast_1d(np.asarray(model.pi, dtype=np.float64))
    pi = float(pi[-1])
    return np.concatenate([[pi, model.eta, model.epsilon], model.gamma, model.mu, model.sigma]).astype(np.float64)


def init_lps(fit, N, H, U):
    fit._init_lps(N, H, U)
    fit._lp_B_g_F = fit._lp_B_g_F.astype(np.float64)
    fit._p_Bt_g_Ft = fit._p_Bt_g_Ft.astype(np.float64)
    fit._lM = fit._lM.astype(np.float64)


def meta():
    return dict(numpy_version=np.__version__, scipy_version=scipy.__version__,
                python_version=sys.version.split()[0])


def save(name, **arrs):
    arrs.update({"meta_" + k: np.array(v) for k, v in meta().items()})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def vb_trajectory(fcdiff, b, bt, model, iters):
    """The documented loop of fcdiff/fit.py:56-82 without the broken theta_sub step."""
    (C, H) = b.shape
    U = bt.shape[1]
    N = int(round(float(fcdiff.util.C_to_N(C))))
    fit = fcdiff.fit.UnsharedRegionFit()
    fit.b, fit.bt, fit.model = b, bt, model
    init_lps(fit, N, H, U)
    fit._update_lps()

    def energy():
        pi = model.pi
        model.pi = np.array([1 - pi, pi])
        e = fit._eval_energy()
        model.pi = pi
        return e
    energies = [energy()]
    pis = [float(model.pi)]
    gammas = [np.array(model.gamma, dtype=np.float64)]
    lqF, lqR = [], []
    for _ in range(iters):
        fit._update_lq_F()
        pi = model.pi
        model.pi = np.array([1 - pi, pi])
        fit._update_lq_R()
        model.pi = pi
        fit._update_pi()
        fit._update_gamma()
        fit._update_lps()
        energies.append(energy())
        pis.append(float(model.pi))
        gammas.append(np.array(model.gamma, dtype=np.float64))
        lqF.append(fit._lq_F.copy())
        lqR.append(fit._lq_R.copy())
    return dict(energy=np.array(energies), pi=np.array(pis), gamma=np.array(gammas),
                lq_F=np.array(lqF), lq_R=np.array(lqR), lM_final=fit._lM.copy())


def main():
    if not os.path.isdir(REF):
        print("no %s on this machine: nothing to capture (fixtures are committed)" % REF)
        return 0
    os.makedirs(OUT, exist_ok=True)
    fcdiff = load_reference()
    F = fcdiff.fit

    # ---- G1 index maps (test_fcdiff/test_util.py) ----
    Ns = np.arange(2, 11)
    Cs = np.array([fcdiff.N_to_C(int(n)) for n in Ns])
    N_back = np.array([float(fcdiff.util.C_to_N(int(c))) for c in Cs])
    nm = np.array([fcdiff.c_to_nm(c) for c in range(45)])
    c_of = np.array([fcdiff.nm_to_c(n, m) for (n, m) in nm])
    # the asymmetric use of nm_to_c by _update_lq_R (fit.py:185-186), all ordered pairs, N=10
    pairs = np.array([(n, m) for n in range(10) for m in range(10) if m != n])
    c_asym = np.array([fcdiff.nm_to_c(n, m) for (n, m) in pairs])
    save("G1_index_maps", Ns=Ns, Cs=Cs, N_back=N_back, c_to_nm_N10=nm, nm_to_c_N10=c_of,
         ordered_pairs_N10=pairs, nm_to_c_asym_N10=c_asym)

    # ---- G2 likelihood tables (test_fit.py:131-166) ----
    (N, C, H, U) = (4, 6, 7, 5)
    fit = F.UnsharedRegionFit()
    fit.b = 1 - 2 * rand_prob((C, H), seed=0)
    fit.bt = 1 - 2 * rand_prob((C, U), seed=1)
    fit.model = ideal_model(fcdiff)
    init_lps(fit, N, H, U)
    fit._update_lps()
    save("G2_update_lps", b=fit.b, bt=fit.bt, theta=theta_of(fit.model),
         lp_B_g_F=fit._lp_B_g_F, p_Bt_g_Ft=fit._p_Bt_g_Ft, lM=fit._lM)
    # G2b: same shapes with the model defaults and with data from the model's own sampler
    m = fcdiff.UnsharedRegionModel()
    (r, t, f, ft, b, bt) = m.sample(8, 6, 5)
    fit = F.UnsharedRegionFit()
    fit.b, fit.bt, fit.model = b, bt, fcdiff.UnsharedRegionModel()
    init_lps(fit, 8, 6, 5)
    with np.errstate(divide="ignore"):
        fit._update_lps()
    save("G2b_update_lps_default", b=b, bt=bt, theta=theta_of(fit.model),
         lp_B_g_F=fit._lp_B_g_F, p_Bt_g_Ft=fit._p_Bt_g_Ft, lM=fit._lM)

    # ---- G3 mixture densities (test_fit.py:233-386) ----
    mdl = ideal_model(fcdiff)
    p = rand_prob_vector((1, 1, 3))
    M = np.zeros((3, 3))
    for k in range(3):
        for l in range(3):
            M[k, l] = F._eval_M(p, mdl.eta, mdl.epsilon, k, l)[0, 0]
    eps = np.array([F._eval_M_eps(mdl.eta, mdl.epsilon, l) for l in range(3)])
    save("G3_eval_M", p=p, eta=mdl.eta, epsilon=mdl.epsilon, M=M, M_eps=eps)

    # ---- G4 q_F update (test_fit.py:428-467) ----
    (N, H, U) = (6, 5, 4)
    C = fcdiff.N_to_C(N)
    q_R = rand_prob_vector((N, U, 2))
    lpB = np.log(rand_prob((C, H, 3)))
    lM = np.log(rand_prob((C, U, 3, 3)))
    fit = F.UnsharedRegionFit()
    fit._lq_R = np.log(q_R)
    fit._lp_B_g_F = lpB
    fit._lM = lM
    fit.model = fcdiff.UnsharedRegionModel()
    fit.model.gamma = rand_prob_vector((3,))
    fit._update_lq_F()
    save("G4_update_lq_F", q_R=q_R, lp_B_g_F=lpB, lM=lM, gamma=fit.model.gamma, lq_F=fit._lq_F)

    # ---- G5 q_R update (test_fit.py:470-510): the reference's asymmetric edge id ----
    (N, U) = (6, 4)
    C = fcdiff.N_to_C(N)
    pi = rand_prob_vector((2,))
    q_R = rand_prob_vector((N, U, 2))
    q_F = rand_prob_vector((C, 1, 3))
    lM = np.log(rand_prob((C, U, 3, 3)))
    fit = F.UnsharedRegionFit()
    fit._lq_R = np.log(q_R)
    fit._lq_F = np.log(q_F)
    fit._lM = lM
    fit.model = fcdiff.UnsharedRegionModel()
    fit.model.pi = pi
    fit._update_lq_R()
    save("G5_update_lq_R", pi=pi, q_R=q_R, q_F=q_F, lM=lM, lq_R=fit._lq_R)

    # ---- G6 energy terms (test_fit.py:170-230, 389-425) and the assembled energy ----
    (N, H, U) = (5, 4, 3)
    C = fcdiff.N_to_C(N)
    q_F = rand_prob_vector((C, 1, 3), seed=3)
    q_R = rand_prob_vector((N, U, 2), seed=4)
    lpB = np.log(rand_prob((C, H, 3), seed=5))
    lM = np.log(rand_prob((C, U, 3, 3), seed=6))
    gamma = rand_prob_vector((3,), seed=7)
    pi2 = rand_prob_vector((2,), seed=8)
    terms = np.array([
        F._eval_E_lp_F(q_F, gamma),
        F._eval_E_lp_B_g_F(q_F, lpB),
        F._eval_E_lp_R(q_R, pi2),
        F._eval_E_lM(q_F, q_R, lM),
        F._eval_E_lq_F(q_F, np.log(q_F)),
        F._eval_E_lq_R(q_R, np.log(q_R)),
    ])
    fit = F.UnsharedRegionFit()
    fit.model = fcdiff.UnsharedRegionModel()
    fit.model.gamma, fit.model.pi = gamma, pi2
    fit._lq_F, fit._lq_R, fit._lp_B_g_F, fit._lM = np.log(q_F), np.log(q_R), lpB, lM
    save("G6_energy_terms", q_F=q_F, q_R=q_R, lp_B_g_F=lpB, lM=lM, gamma=gamma, pi2=pi2,
         terms=terms, energy=fit._eval_energy())

    # ---- G7 pi / gamma (test_fit.py:513-557) ----
    (N, U) = (6, 4)
    C = fcdiff.N_to_C(N)
    q_R = rand_prob_vector((N, U, 2))
    q_F = rand_prob_vector((C, 1, 3))
    fit = F.UnsharedRegionFit()
    fit.model = fcdiff.UnsharedRegionModel()
    fit._lq_R, fit._lq_F = np.log(q_R), np.log(q_F)
    fit._update_pi()
    fit._update_gamma()
    save("G7_pi_gamma", q_R=q_R, q_F=q_F, pi=fit.model.pi, gamma=fit.model.gamma)

    # ---- G8 derivative helpers (test_fit.py:560-1087) ----
    (N, H, U) = (6, 4, 5)
    C = fcdiff.N_to_C(N)
    dlN_dm = rand(-10, 10, (C, H))
    dlM_dm = rand(-10, 10, (C, U, 3, 3))
    q_F = rand_prob_vector((C, 1, 3))
    q_R = rand_prob_vector((N, U, 2))
    dE_dm = np.array([F._eval_dE_dm(q_F, q_R, dlN_dm, dlM_dm, j) for j in range(3)])
    norm2 = rand_prob((C, U))
    mix2 = rand_prob((C, U))
    (mu, sigma, epsilon, eta) = (0.14, 0.02, 0.07, 0.29)
    dlM_dm_kl = np.array([[F._eval_dlM_dm(norm2, mix2, mu, sigma, eta, epsilon, k, l)
                           for l in range(3)] for k in range(3)])
    norm3 = rand_prob((C, U, 3))
    dlM_dh = np.array([F._eval_dlM_dh(norm3, mix2, epsilon, k) for k in range(3)])
    dlM_de = np.array([[F._eval_dlM_de(norm3, mix2, eta, k, l) for l in range(3)] for k in range(3)])
    mix4 = rand_prob((C, U, 3, 3), seed=2)
    dE_dh = F._eval_dE_dh(q_R, q_F, norm3, mix4, epsilon)
    dE_de = F._eval_dE_de(q_R, q_F, norm3, mix4, eta)
    bb = rand(-1, 1, (C, H), seed=9)
    NN = rand_prob((C, H), seed=10)
    save("G8_derivatives", dlN_dm=dlN_dm, dlM_dm=dlM_dm, q_F=q_F, q_R=q_R, dE_dm=dE_dm,
         norm2=norm2, mix2=mix2, mu=mu, sigma=sigma, epsilon=epsilon, eta=eta,
         dlM_dm_kl=dlM_dm_kl, norm3=norm3, dlM_dh=dlM_dh, dlM_de=dlM_de, mix4=mix4,
         dE_dh=dE_dh, dE_de=dE_de, bb=bb, NN=NN,
         dlN_dm_fn=F._eval_dlN_dm(bb, mu, sigma), dlN_ds_fn=F._eval_dlN_ds(bb, mu, sigma),
         dN_dm_fn=F._eval_dN_dm(NN, bb, mu, sigma), dN_ds_fn=F._eval_dN_ds(NN, bb, mu, sigma))

    # ---- G9 forward sampler (model.py:52-236), RandomState(0), two sizes, two models ----
    m = fcdiff.UnsharedRegionModel()
    (r, t, f, ft, b, bt) = m.sample(10, 5, 4)
    mi = ideal_model(fcdiff)
    (r2, t2, f2, ft2, b2, bt2) = mi.sample(7, 3, 6)
    save("G9_model_sample", r=r, t=t, f=f, f_tilde=ft, b=b, b_tilde=bt,
         r_ideal=r2, t_ideal=t2, f_ideal=f2, f_tilde_ideal=ft2, b_ideal=b2, b_tilde_ideal=bt2,
         str_default=np.array(str(fcdiff.UnsharedRegionModel()).split("rng = ")[0]))

    # ---- G10 VB trajectory at the cfg-1 shape (Nreg=10, H=U=4), default model, seed 0 ----
    m = fcdiff.UnsharedRegionModel()
    (r, t, f, ft, b, bt) = m.sample(10, 4, 4)
    traj = vb_trajectory(fcdiff, b, bt, fcdiff.UnsharedRegionModel(), 4)
    save("G10_vb_trajectory_cfg1", b=b, bt=bt, r_true=r, f_true=f,
         theta0=theta_of(fcdiff.UnsharedRegionModel()), **traj)
    print("G10 energies:", traj["energy"])
    #   ideal model, data drawn from it (well separated case)
    mi = ideal_model(fcdiff)
    (r, t, f, ft, b, bt) = mi.sample(10, 4, 4)
    traj = vb_trajectory(fcdiff, b, bt, ideal_model(fcdiff), 4)
    save("G10b_vb_trajectory_cfg1_ideal", b=b, bt=bt, r_true=r, f_true=f,
         theta0=theta_of(ideal_model(fcdiff)), **traj)

    # ---- G12 VB trajectory at a mid shape for the GPU parity tests (Nreg=24, H=7, U=9) ----
    mi = ideal_model(fcdiff)
    mi.rng = np.random.RandomState(12)
    (r, t, f, ft, b, bt) = mi.sample(24, 7, 9)
    traj = vb_trajectory(fcdiff, b, bt, ideal_model(fcdiff), 3)
    traj.pop("lM_final")
    save("G12_vb_trajectory_mid", b=b, bt=bt, theta0=theta_of(ideal_model(fcdiff)), **traj)

    # ---- G11 Gibbs-conditional pins: the reference's updates / log-joint at ONE-HOT q ----
    for tag, (N, H, U), mk, seed in (("cfg1", (10, 4, 4), ideal_model, 0),
                                     ("mid", (13, 6, 7), lambda fc: fc.UnsharedRegionModel(), 5)):
        C = fcdiff.N_to_C(N)
        gen = mk(fcdiff)
        gen.rng = np.random.RandomState(seed)
        (r, t, f, ft, b, bt) = gen.sample(N, H, U)
        model = mk(fcdiff)
        fit = F.UnsharedRegionFit()
        fit.b, fit.bt, fit.model = b, bt, model
        init_lps(fit, N, H, U)
        with np.errstate(divide="ignore"):
            fit._update_lps()
        rs = np.random.RandomState(100 + seed)
        r_state = (rs.uniform(size=(N, U)) < 0.3).astype(np.uint8)
        f_state = rs.randint(0, 3, size=C).astype(np.uint8)
        q_R = np.zeros((N, U, 2))
        q_R[:, :, 0] = 1 - r_state
        q_R[:, :, 1] = r_state
        q_F = np.zeros((C, 1, 3))
        q_F[np.arange(C), 0, f_state] = 1
        with np.errstate(divide="ignore"):
            # (a) f conditionals = _update_lq_F at one-hot q_R (fit.py:157-174)
            fit._lq_R = np.log(q_R)
            fit._update_lq_F()
            cond_f = fit._lq_F.copy()
            # (b) r conditional of region 0 = row 0 of _update_lq_R at one-hot q (fit.py:176-198);
            #     later rows see softened q_R[m<n] (Gauss-Seidel) so only row 0 is a pure conditional
            fit._lq_R = np.log(q_R)
            fit._lq_F = np.log(q_F)
            pi = model.pi
            model.pi = np.array([1 - pi, pi])
            fit._update_lq_R()
            model.pi = pi
            cond_r_full = fit._lq_R.copy()

        # (c) log-joint = first four terms of the free energy (fit.py:149-152) at one-hot q,
        #     at the base state and at every single-site change
        def logjoint(qF, qR):
            return (F._eval_E_lp_F(qF, model.gamma) + F._eval_E_lp_B_g_F(qF, fit._lp_B_g_F)
                    + F._eval_E_lp_R(qR, np.array([1 - model.pi, model.pi]))
                    + F._eval_E_lM(qF, qR, fit._lM))
        lj_base = logjoint(q_F, q_R)
        lj_r = np.zeros((N, U, 2))
        for n in range(N):
            for u in range(U):
                for j in range(2):
                    q2 = q_R.copy()
                    q2[n, u, :] = 0
                    q2[n, u, j] = 1
                    lj_r[n, u, j] = logjoint(q_F, q2)
        lj_f = np.zeros((C, 3))
        for c in range(C):
            for k in range(3):
                q2 = q_F.copy()
                q2[c, 0, :] = 0
                q2[c, 0, k] = 1
                lj_f[c, k] = logjoint(q2, q_R)
        save("G11_gibbs_conditionals_" + tag, b=b, bt=bt, theta=theta_of(model),
             r_state=r_state, f_state=f_state, lM=fit._lM, lp_B_g_F=fit._lp_B_g_F,
             cond_f=cond_f, lq_R_after_update=cond_r_full, logjoint_base=lj_base,
             logjoint_r=lj_r, logjoint_f=lj_f)
    return 0


if __name__ == "__main__":
    sys.exit(main())
