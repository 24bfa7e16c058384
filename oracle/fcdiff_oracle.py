"""
CPU oracle for the fcdiff fit path (NumPy).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (fcdiff_amd/) never does and fails loudly when
its HIP library is missing.

What is restated here, each function citing the reference lines it follows
(paths relative to the reference checkout):

  * index maps                        fcdiff/util.py:7-84
  * likelihood tables (_update_lps)   fcdiff/fit.py:104-122, 409-444
  * variational updates q_F, q_R      fcdiff/fit.py:157-198, 382-406
  * free energy and its six terms     fcdiff/fit.py:142-155, 447-539
  * pi / gamma closed forms           fcdiff/fit.py:208-220
  * convergence test                  fcdiff/fit.py:124-140
  * derivative helpers                fcdiff/fit.py:542-733
  * documented fit loop               fcdiff/fit.py:56-82, doc/methods.rst:564-600

and the build-defined many-chain collapsed Gibbs sampler (no counterpart in the
reference) whose two conditionals ARE the reference's q_F /
q_R updates evaluated at one-hot q:

  * Philox4x32-10 counter RNG (Salmon et al., SC'11; Random123 1.09 constants)
  * gibbs_init / gibbs_f_step / gibbs_r_step / gibbs_stats / gibbs_logjoint

Parity status: every function of the first group is pinned against fixtures
captured from the reference itself (tests/golden/G1..G12, made by
oracle/capture_golden.py).  Gibbs *conditionals* are pinned by G11 (reference
_update_lq_F, row 0 of _update_lq_R and reference log-joint differences at
one-hot states); Gibbs *trajectories* have no reference and are pinned only
against this restatement ("parity unpinned" w.r.t. the reference).

Third-party arithmetic the reference calls (versions unpinned there; this
container has NumPy 2.2.6 / SciPy 1.15.3):
  scipy.stats.norm.logpdf(x, m, s) = -z*z/2 - log(sqrt(2*pi)) - log(s), z=(x-m)/s
  scipy.stats.norm.pdf(x, m, s)    = exp(-z*z/2) / sqrt(2*pi) / s
  scipy.special.logsumexp(a)       = log(sum(exp(a - amax))) + amax
"""
import numpy as np

_LOG_SQRT_2PI = np.log(np.sqrt(2 * np.pi))
_SQRT_2PI = np.sqrt(2 * np.pi)

EDGE_REFERENCE = 0   # quirk Q1: nm_to_c(n, m) = n(n-1)/2 + m for every ordered pair (fit.py:185-186)
EDGE_SYMMETRIC = 1   # documented maths: the unordered pair's edge (doc/methods.rst:646-653)


# ----------------------------------------------------------------------------------------
# index maps -- fcdiff/util.py
# ----------------------------------------------------------------------------------------
def N_to_C(N):
    """util.py:7-21 (Python-2 integer division)."""
    return int(N) * (int(N) - 1) // 2


def C_to_N(C):
    """util.py:23-38; returns a float exactly like the reference."""
    return (np.sqrt(8 * C + 1) - 1) / 2 + 1


def nm_to_c(n, m):
    """util.py:40-60; valid for n > m only, used for any ordered pair by fit.py:186."""
    return N_to_C(n) + int(m)


def c_to_nm(c):
    """util.py:62-84 with integer results; n > m."""
    n = int(np.floor((np.sqrt(8 * c + 1) - 1) / 2) + 1)
    return (n, int(c) - N_to_C(n))


def edge_id(n, m, mode):
    """Edge used by the q_R / r update for the ordered pair (n, m), m != n."""
    if mode == EDGE_REFERENCE:
        return nm_to_c(n, m)
    return nm_to_c(max(n, m), min(n, m))


def edge_endpoints(Nreg):
    """(C, 2) array of (n, m), n > m, in edge order."""
    C = N_to_C(Nreg)
    out = np.zeros((C, 2), dtype=np.int64)
    c = 0
    for n in range(1, Nreg):
        for m in range(n):
            out[c] = (n, m)
            c += 1
    return out


# ----------------------------------------------------------------------------------------
# likelihood tables -- fcdiff/fit.py:104-122 with _eval_M / _eval_M_eps (409-444)
# ----------------------------------------------------------------------------------------
def norm_logpdf(x, mu, sigma):
    z = (x - mu) / sigma
    return -(z * z) / 2.0 - _LOG_SQRT_2PI - np.log(sigma)


def norm_pdf(x, mu, sigma):
    z = (x - mu) / sigma
    return np.exp(-(z * z) / 2.0) / _SQRT_2PI / sigma


def eval_M_eps(eta, epsilon, l):
    """fit.py:433-444."""
    if l == 0:
        return 1 - epsilon
    if l == 1:
        return epsilon
    e = eta * epsilon
    e += (1 - eta) * (1 - epsilon)
    return e


def eval_M(Nd, eta, epsilon, k, l):
    """fit.py:409-430: eps*N_k + (1-eps)*0.5*(sum of the other two densities)."""
    eps = eval_M_eps(eta, epsilon, l)
    js = [j for j in range(3) if j != k]
    sum_N = Nd[..., js[0]] + Nd[..., js[1]]
    return eps * Nd[..., k] + (1 - eps) * 0.5 * sum_N


def lik_tables(b, bt, mu, sigma, eta, epsilon):
    """fit.py:111-122.  Returns (lp_B_g_F (C,H,3), p_Bt_g_Ft (C,U,3), lM (C,U,3,3))."""
    (C, H) = b.shape
    U = bt.shape[1]
    lpB = np.zeros((C, H, 3))
    pBt = np.zeros((C, U, 3))
    for k in range(3):
        lpB[:, :, k] = norm_logpdf(b, mu[k], sigma[k])
        pBt[:, :, k] = norm_pdf(bt, mu[k], sigma[k])
    lM = np.zeros((C, U, 3, 3))
    with np.errstate(divide="ignore"):
        for k in range(3):
            for l in range(3):
                lM[:, :, k, l] = np.log(eval_M(pBt, eta, epsilon, k, l))
    return lpB, pBt, lM


def sum_lp_B(lpB):
    """S_B[c,k] = sum_h lp_B_g_F[c,h,k]: the only thing fit.py:171 and :472 consume."""
    return np.sum(lpB, axis=1)


# ----------------------------------------------------------------------------------------
# variational updates
# ----------------------------------------------------------------------------------------
def logsumexp(a, axis):
    amax = np.max(a, axis=axis, keepdims=True)
    amax = np.where(np.isfinite(amax), amax, 0.0)
    return np.log(np.sum(np.exp(a - amax), axis=axis, keepdims=True)) + amax


def eval_q_R_w(q_R, n, m):
    """fit.py:382-406."""
    U = q_R.shape[1]
    w = np.zeros((U, 3))
    w[:, 0] = q_R[n, :, 0] * q_R[m, :, 0]
    w[:, 1] = q_R[n, :, 1] * q_R[m, :, 1]
    w[:, 2] = q_R[n, :, 0] * q_R[m, :, 1]
    w[:, 2] += q_R[n, :, 1] * q_R[m, :, 0]
    return w


def update_lq_F(lq_R, S_B, lM, gamma):
    """fit.py:157-174 (edge loop kept).  S_B replaces the (C,H,3) table by its H-sums."""
    C = lM.shape[0]
    lq_F = np.tile(np.log(gamma), (C, 1, 1)).astype(np.float64)
    with np.errstate(divide="ignore"):
        q_R = np.exp(lq_R)
    for c in range(C):
        (n, m) = c_to_nm(c)
        w = eval_q_R_w(q_R, n, m)
        for k in range(3):
            lq_F[c, :, k] += S_B[c, k] + np.sum(w * lM[c, :, k, :])
    return lq_F - logsumexp(lq_F, axis=2)


def lq_R_row(n, q_R, q_F, lM, lnpi2, mode):
    """One region's two unnormalised log-weights, fit.py:184-194."""
    (Nreg, U) = q_R.shape[0:2]
    row = np.tile(lnpi2, (U, 1)).astype(np.float64)
    for m in range(Nreg):
        if m == n:
            continue
        c = edge_id(n, m, mode)
        for k in range(3):
            lM_00 = q_R[m, :, 0] * lM[c, :, k, 0]
            lM_1neq = q_R[m, :, 1] * lM[c, :, k, 2]
            row[:, 0] += q_F[c, 0, k] * (lM_00 + lM_1neq)
            lM_11 = q_R[m, :, 1] * lM[c, :, k, 1]
            lM_0neq = q_R[m, :, 0] * lM[c, :, k, 2]
            row[:, 1] += q_F[c, 0, k] * (lM_11 + lM_0neq)
    return row


def update_lq_R(lq_R, lq_F, lM, pi2, mode=EDGE_REFERENCE):
    """fit.py:176-198: Gauss-Seidel over regions, q_R[n] refreshed before region n+1.
    pi2 = [1-pi, pi] (quirk Q4)."""
    (Nreg, U) = lq_R.shape[0:2]
    with np.errstate(divide="ignore"):
        q_R = np.exp(lq_R)
        q_F = np.exp(lq_F)
        lnpi2 = np.log(np.asarray(pi2, dtype=np.float64))
    out = np.zeros((Nreg, U, 2))
    for n in range(Nreg):
        row = lq_R_row(n, q_R, q_F, lM, lnpi2, mode)
        row = row - logsumexp(row, axis=1)
        out[n] = row
        q_R[n] = np.exp(row)
    return out


def update_pi(lq_R):
    """fit.py:208-213."""
    return np.mean(np.exp(lq_R)[:, :, 1])


def update_gamma(lq_F):
    """fit.py:215-220."""
    return np.mean(np.exp(lq_F), axis=(0, 1))


def is_converged(energy, s, rel_tol):
    """fit.py:124-140 (quirk Q6 kept: sign of e not considered)."""
    e = energy[s - 1]
    e_star = energy[s]
    return ((e - e_star) / e) < rel_tol


# ----------------------------------------------------------------------------------------
# free energy -- fcdiff/fit.py:142-155, 447-539
# ----------------------------------------------------------------------------------------
def _xlogy0(q, lq):
    """q * lq with the convention 0 * (-inf) = 0 (only reachable at one-hot q)."""
    with np.errstate(invalid="ignore"):
        t = q * lq
    return np.where(q == 0, 0.0, t)


def eval_E_lp_F(q_F, gamma):
    return np.sum(q_F * np.log(gamma))


def eval_E_lp_B_g_F(q_F, S_B):
    """fit.py:461-472 with the H axis pre-summed: sum_c sum_k q_F[c,k] * S_B[c,k]."""
    return np.sum(q_F[:, 0, :] * S_B)


def eval_E_lp_R(q_R, pi2):
    return np.sum(q_R * np.log(np.asarray(pi2, dtype=np.float64)))


def eval_E_lM(q_F, q_R, lM):
    """fit.py:489-511 (edge loop kept)."""
    C = q_F.shape[0]
    e = 0.0
    for c in range(C):
        (n, m) = c_to_nm(c)
        w = eval_q_R_w(q_R, n, m)
        for k in range(3):
            e += q_F[c, 0, k] * np.sum(w * lM[c, :, k, :])
    return e


def eval_E_lq_F(q_F, lq_F):
    return np.sum(_xlogy0(q_F, lq_F))


def eval_E_lq_R(q_R, lq_R):
    return np.sum(_xlogy0(q_R, lq_R))


def energy_terms(lq_F, lq_R, S_B, lM, gamma, pi2):
    with np.errstate(divide="ignore"):
        q_F = np.exp(lq_F)
        q_R = np.exp(lq_R)
    return np.array([
        eval_E_lp_F(q_F, gamma),
        eval_E_lp_B_g_F(q_F, S_B),
        eval_E_lp_R(q_R, pi2),
        eval_E_lM(q_F, q_R, lM),
        eval_E_lq_F(q_F, lq_F),
        eval_E_lq_R(q_R, lq_R),
    ])


def eval_energy(lq_F, lq_R, S_B, lM, gamma, pi2):
    """fit.py:142-155."""
    t = energy_terms(lq_F, lq_R, S_B, lM, gamma, pi2)
    return -t[0] - t[1] - t[2] - t[3] + t[4] + t[5]


# ----------------------------------------------------------------------------------------
# the documented fit loop (run() is broken as shipped: quirk Q7) -- fit.py:56-82
# ----------------------------------------------------------------------------------------
def vb_fit(b, bt, theta, max_iters=10, rel_tol=1e-5, mode=EDGE_REFERENCE, check_convergence=True):
    """theta = dict(pi, eta, epsilon, gamma, mu, sigma).  The (eta, epsilon) optimiser step of
    fit.py:222-241 cannot run in the reference and is left out."""
    (C, H) = b.shape
    U = bt.shape[1]
    N = C_to_N(C)
    if (N % 1) != 0:
        raise ValueError("Number of connections (%u) must be a triangular number." % C)
    N = int(N)
    th = dict(theta)
    lq_R = np.full((N, U, 2), -np.log(2))
    lq_F = np.full((C, 1, 3), -np.log(3))

    def tables():
        lpB, _, lM = lik_tables(b, bt, th["mu"], th["sigma"], th["eta"], th["epsilon"])
        return sum_lp_B(lpB), lM

    def energy():
        return eval_energy(lq_F, lq_R, S_B, lM, th["gamma"], [1 - th["pi"], th["pi"]])
    S_B, lM = tables()
    energies = [energy()]
    hist = []
    for i in range(1, max_iters + 1):
        lq_F = update_lq_F(lq_R, S_B, lM, th["gamma"])
        lq_R = update_lq_R(lq_R, lq_F, lM, [1 - th["pi"], th["pi"]], mode)
        th["pi"] = update_pi(lq_R)
        th["gamma"] = update_gamma(lq_F)
        S_B, lM = tables()
        energies.append(energy())
        hist.append((lq_F.copy(), lq_R.copy(), float(th["pi"]), np.array(th["gamma"])))
        if check_convergence and is_converged(energies, i, rel_tol):
            break
    return dict(energy=np.array(energies), lq_F=lq_F, lq_R=lq_R, theta=th, hist=hist)


# ----------------------------------------------------------------------------------------
# The same variational updates WITHOUT the Python loop over edges (CPU baseline mode (ii):
# "vectorised NumPy ... the honest strong baseline").  Same arithmetic, whole-array expressions; pinned to the
# edge-loop forms above by tests/test_oracle_golden.py.  bench.py times both beside the GPU iteration.
# ----------------------------------------------------------------------------------------
def update_lq_F_vec(lq_R, S_B, lM, gamma):
    """fit.py:157-174 over all edges at once."""
    ep = edge_endpoints(lq_R.shape[0])
    (n, m) = (ep[:, 0], ep[:, 1])
    with np.errstate(divide="ignore"):
        q_R = np.exp(lq_R)
    w = np.stack([q_R[n, :, 0] * q_R[m, :, 0], q_R[n, :, 1] * q_R[m, :, 1],
                  q_R[n, :, 0] * q_R[m, :, 1] + q_R[n, :, 1] * q_R[m, :, 0]], axis=2)          # (C,U,3)
    lq_F = (np.log(gamma)[None, :] + S_B + np.einsum("cul,cukl->ck", w, lM))[:, None, :]
    return lq_F - logsumexp(lq_F, axis=2)


def update_lq_R_vec(lq_R, lq_F, lM, pi2, mode=EDGE_REFERENCE):
    """fit.py:176-198: regions still in order (Gauss-Seidel), each region's sum over (m, k) as array expressions."""
    (Nreg, U) = lq_R.shape[0:2]
    with np.errstate(divide="ignore"):
        q_R = np.exp(lq_R)
        q_F = np.exp(lq_F)[:, 0, :]
        lnpi2 = np.log(np.asarray(pi2, dtype=np.float64))
    out = np.zeros((Nreg, U, 2))
    ms_all = np.arange(Nreg)
    for n in range(Nreg):
        ms = ms_all[ms_all != n]
        cs = np.array([edge_id(n, int(mm), mode) for mm in ms])
        t = lM[cs]                                           # (Nreg-1, U, 3, 3)
        qf = q_F[cs][:, None, :]                             # (Nreg-1, 1, 3)
        q0, q1 = q_R[ms, :, 0][:, :, None], q_R[ms, :, 1][:, :, None]
        s0 = np.sum(qf * (q0 * t[:, :, :, 0] + q1 * t[:, :, :, 2]), axis=(0, 2))
        s1 = np.sum(qf * (q1 * t[:, :, :, 1] + q0 * t[:, :, :, 2]), axis=(0, 2))
        row = np.stack([lnpi2[0] + s0, lnpi2[1] + s1], axis=1)
        row = row - logsumexp(row, axis=1)
        out[n] = row
        q_R[n] = np.exp(row)
    return out


def eval_E_lM_vec(q_F, q_R, lM):
    ep = edge_endpoints(q_R.shape[0])
    (n, m) = (ep[:, 0], ep[:, 1])
    w = np.stack([q_R[n, :, 0] * q_R[m, :, 0], q_R[n, :, 1] * q_R[m, :, 1],
                  q_R[n, :, 0] * q_R[m, :, 1] + q_R[n, :, 1] * q_R[m, :, 0]], axis=2)
    return np.sum(q_F[:, 0, :] * np.einsum("cul,cukl->ck", w, lM))


def vb_iteration(lq_F, lq_R, b, bt, th, mode=EDGE_REFERENCE, vectorised=False):
    """
    ONE iteration of the documented loop (doc/methods.rst:564-600; fit.py:75-82): q_F, q_R, pi / gamma, tables,
    energy.  vectorised=False is the reference's structure (Python loop over edges, mode (i) of SURVEY.md section 8d),
    True the whole-array form (mode (ii)).  Returns (lq_F, lq_R, th, energy).
    """
    th = dict(th)
    lpB, _, lM = lik_tables(b, bt, th["mu"], th["sigma"], th["eta"], th["epsilon"])
    S_B = sum_lp_B(lpB)
    pi2 = [1 - th["pi"], th["pi"]]
    if vectorised:
        lq_F = update_lq_F_vec(lq_R, S_B, lM, th["gamma"])
        lq_R = update_lq_R_vec(lq_R, lq_F, lM, pi2, mode)
    else:
        lq_F = update_lq_F(lq_R, S_B, lM, th["gamma"])
        lq_R = update_lq_R(lq_R, lq_F, lM, pi2, mode)
    th["pi"] = update_pi(lq_R)
    th["gamma"] = update_gamma(lq_F)
    lpB, _, lM = lik_tables(b, bt, th["mu"], th["sigma"], th["eta"], th["epsilon"])
    S_B = sum_lp_B(lpB)
    pi2 = [1 - th["pi"], th["pi"]]
    if vectorised:
        with np.errstate(divide="ignore"):
            q_F, q_R = np.exp(lq_F), np.exp(lq_R)
        t = [eval_E_lp_F(q_F, th["gamma"]), eval_E_lp_B_g_F(q_F, S_B), eval_E_lp_R(q_R, pi2), eval_E_lM_vec(q_F, q_R, lM),
             eval_E_lq_F(q_F, lq_F), eval_E_lq_R(q_R, lq_R)]
        e = -t[0] - t[1] - t[2] - t[3] + t[4] + t[5]
    else:
        e = eval_energy(lq_F, lq_R, S_B, lM, th["gamma"], pi2)
    return lq_F, lq_R, th, e


# ----------------------------------------------------------------------------------------
# derivative helpers -- fcdiff/fit.py:542-733 (only the runnable module-level ones)
# ----------------------------------------------------------------------------------------
def eval_dlN_dm(b, mu, sigma):
    return (b - mu) / (sigma * sigma)


def eval_dlN_ds(b, mu, sigma):
    d = b - mu
    s2 = sigma * sigma
    return ((d * d) - s2) / (2 * s2)


def eval_dN_dm(N, b, mu, sigma):
    return N * eval_dlN_dm(b, mu, sigma)


def eval_dN_ds(N, b, mu, sigma):
    return N * eval_dlN_ds(b, mu, sigma)


def eval_dlM_dm(norm, mix, mu, sigma, eta, epsilon, k, l):
    """fit.py:572-597 including quirk Q8 (tests k != l; feeds the density to dlN_dm)."""
    eps = eval_M_eps(eta, epsilon, l)
    if k != l:
        eps = 0.5 * (1 - eps)
    return eps * eval_dlN_dm(norm, mu, sigma) / mix


def eval_dlM_dh(norm, mix, epsilon, k):
    """fit.py:618-641."""
    eps = (2 * epsilon) - 1
    ls = [j for j in range(3) if j != k]
    s = norm[:, :, ls[0]] + norm[:, :, ls[1]]
    return (eps * norm[:, :, k] - 0.5 * eps * s) / mix


def eval_dlM_de(norm, mix, eta, k, l):
    """fit.py:667-697."""
    if l == 0:
        eps = -1
    elif l == 1:
        eps = 1
    else:
        eps = 2 * eta - 1
    ls = [j for j in range(3) if j != k]
    s = norm[:, :, ls[0]] + norm[:, :, ls[1]]
    return (eps * norm[:, :, k] - 0.5 * eps * s) / mix


def eval_dE_dm(q_F, q_R, dlN_dmj, dlM_dmj, j):
    """fit.py:542-569."""
    C = dlN_dmj.shape[0]
    d = 0.0
    for c in range(C):
        (n, m) = c_to_nm(c)
        d -= q_F[c, 0, j] * np.sum(dlN_dmj[c, :])
        w = eval_q_R_w(q_R, n, m)
        for k in range(3):
            d -= q_F[c, 0, k] * np.sum(w * dlM_dmj[c, :, k, :])
    return d


def eval_dE_dh(q_R, q_F, norm, mix, epsilon):
    """fit.py:600-615."""
    C = q_F.shape[0]
    d = 0.0
    for k in range(3):
        dl = eval_dlM_dh(norm, mix[:, :, k, 2], epsilon, k)
        for c in range(C):
            (n, m) = c_to_nm(c)
            neq = q_R[n, :, 0] * q_R[m, :, 1]
            neq += q_R[n, :, 1] * q_R[m, :, 0]
            d -= q_F[c, 0, k] * np.sum(neq * dl[c, :])
    return d


def eval_dE_de(q_R, q_F, norm, mix, eta):
    """fit.py:644-664."""
    C = q_F.shape[0]
    d = 0.0
    for k in range(3):
        d0 = eval_dlM_de(norm, mix[:, :, k, 0], eta, k, 0)
        d1 = eval_dlM_de(norm, mix[:, :, k, 1], eta, k, 1)
        d2 = eval_dlM_de(norm, mix[:, :, k, 2], eta, k, 2)
        for c in range(C):
            (n, m) = c_to_nm(c)
            s = q_R[n, :, 0] * q_R[m, :, 0] * d0[c, :]
            s += q_R[n, :, 1] * q_R[m, :, 1] * d1[c, :]
            neq = q_R[n, :, 0] * q_R[m, :, 1]
            neq += q_R[n, :, 1] * q_R[m, :, 0]
            s += neq * d2[c, :]
            d -= q_F[c, 0, k] * np.sum(s)
    return d


# ----------------------------------------------------------------------------------------
# Philox4x32-10 (Random123).  Counter (c0,c1,c2,c3), key (k0,k1), all uint32.
# ----------------------------------------------------------------------------------------
_PH_M0 = 0xD2511F53
_PH_M1 = 0xCD9E8D57
_PH_W0 = 0x9E3779B9
_PH_W1 = 0xBB67AE85
_M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    (c0, c1, c2, c3) = [int(x) & _M32 for x in ctr]
    (k0, k1) = [int(x) & _M32 for x in key]
    for rnd in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        (hi0, lo0) = (p0 >> 32, p0 & _M32)
        (hi1, lo1) = (p1 >> 32, p1 & _M32)
        (c0, c1, c2, c3) = (hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0)
        k0 = (k0 + _PH_W0) & _M32
        k1 = (k1 + _PH_W1) & _M32
    return (c0, c1, c2, c3)


KIND_INIT_F, KIND_INIT_R, KIND_F, KIND_R = 0, 1, 2, 3


def site_uniform(seed, idx, chain, sweep, kind, half):
    """The double in [0,1) a site draws: words (2*half, 2*half+1) of
    philox(ctr=(idx, chain, sweep, kind), key=(seed_lo, seed_hi)); 53 high bits."""
    x = philox4x32_10((idx, chain, sweep, kind), (seed & _M32, (seed >> 32) & _M32))
    w = (x[2 * half] << 32) | x[2 * half + 1]
    return (w >> 11) * (1.0 / 9007199254740992.0)


def f_site(c):
    """initial state: one counter block per two edges, 53-bit uniforms"""
    return (c >> 1, c & 1)


def f_draw_site(c):
    """sweeps: one counter block per FOUR edges, one 32-bit word each (a 3-way draw does not need 53 bits)"""
    return (c >> 2, c & 3)


def site_uniform32(seed, idx, chain, sweep, kind, word):
    """The double in [0,1) a site draws from ONE word of philox(ctr=(idx, chain, sweep, kind), key=seed): word * 2^-32."""
    x = philox4x32_10((idx, chain, sweep, kind), (seed & _M32, (seed >> 32) & _M32))
    return x[word] * (1.0 / 4294967296.0)


def r_site(n, u, U):
    """initial state: one counter block per (pair of regions, patient)"""
    return ((n >> 1) * U + u, n & 1)


def r_draw_site(n, u, U):
    """sweeps: one counter block per (region, pair of patients) -- the two patients a panel workgroup serves"""
    return (n * ((U + 1) >> 1) + (u >> 1), u & 1)


# ----------------------------------------------------------------------------------------
# many-chain collapsed Gibbs sampler (build-defined; conditionals = reference updates at one-hot q)
#   f: (G, C) uint8 in {0,1,2};  r: (G, Nreg, U) uint8 in {0,1};  chain ids chain0..chain0+G-1
# ----------------------------------------------------------------------------------------
def gibbs_init(G, Nreg, U, pi, seed, chain0=0):
    C = N_to_C(Nreg)
    f = np.zeros((G, C), dtype=np.uint8)
    r = np.zeros((G, Nreg, U), dtype=np.uint8)
    for g in range(G):
        for c in range(C):
            (idx, half) = f_site(c)
            x = site_uniform(seed, idx, chain0 + g, 0, KIND_INIT_F, half)
            f[g, c] = min(int(x * 3.0), 2)
        for n in range(Nreg):
            for u in range(U):
                (idx, half) = r_site(n, u, U)
                r[g, n, u] = site_uniform(seed, idx, chain0 + g, 0, KIND_INIT_R, half) < pi
    return f, r


def mix_index(rn, rm):
    """l of lM[c,u,k,l]: 0 both typical, 1 both anomalous, 2 discordant (fit.py:402-405, 437-443)."""
    return np.where(rn & rm, 1, np.where(rn ^ rm, 2, 0))


def f_conditional_logits(c, r_g, S_B, lM, lngamma):
    """Unnormalised log p(f_c = k | r, b, bt): fit.py:170-173 at one-hot q_R."""
    (n, m) = c_to_nm(c)
    U = lM.shape[1]
    l = mix_index(r_g[n], r_g[m])
    a = np.zeros(3)
    for k in range(3):
        acc = 0.0
        for u in range(U):
            acc += lM[c, u, k, l[u]]
        a[k] = lngamma[k] + (S_B[c, k] + acc)
    return a


def r_conditional_logits(n, u, f_g, r_g, lM, lnpi2, mode):
    """Unnormalised (log p(r_nu=0|..), log p(r_nu=1|..)): fit.py:187-194 at one-hot q_F, q_R."""
    Nreg = r_g.shape[0]
    s0 = 0.0
    s1 = 0.0
    for m in range(Nreg):
        if m == n:
            continue
        c = edge_id(n, m, mode)
        k = f_g[c]
        if r_g[m, u]:
            s0 += lM[c, u, k, 2]
            s1 += lM[c, u, k, 1]
        else:
            s0 += lM[c, u, k, 0]
            s1 += lM[c, u, k, 2]
    return (lnpi2[0] + s0, lnpi2[1] + s1)


def draw_f(a, x):
    mx = np.max(a)
    e = np.exp(a - mx)
    t = x * (e[0] + e[1] + e[2])
    if t < e[0]:
        return 0
    if t < e[0] + e[1]:
        return 1
    return 2


def draw_r(s0, s1, x):
    """r = 1 with probability sigmoid(s1 - s0): logit(x) < s1 - s0 (same inequality as x < 1/(1+exp(s0-s1)))."""
    with np.errstate(divide="ignore"):
        t = np.log(x / (1.0 - x))
    return 1 if t < (s1 - s0) else 0


def gibbs_f_step(f, r, S_B, lM, lngamma, seed, sweep, chain0=0):
    (G, C) = f.shape
    for g in range(G):
        for c in range(C):
            a = f_conditional_logits(c, r[g], S_B, lM, lngamma)
            (idx, word) = f_draw_site(c)
            f[g, c] = draw_f(a, site_uniform32(seed, idx, chain0 + g, sweep, KIND_F, word))


def gibbs_r_step(f, r, lM, lnpi2, seed, sweep, mode=EDGE_SYMMETRIC, chain0=0):
    (G, Nreg, U) = r.shape
    for g in range(G):
        for n in range(Nreg):
            for u in range(U):
                (s0, s1) = r_conditional_logits(n, u, f[g], r[g], lM, lnpi2, mode)
                (idx, half) = r_draw_site(n, u, U)
                r[g, n, u] = draw_r(s0, s1, site_uniform(seed, idx, chain0 + g, sweep, KIND_R, half))


def gibbs_stats(f, r):
    """Pooled sufficient statistics of fit.py:208-220 over chains: [sum r, #f=0, #f=1, #f=2]."""
    return np.array([int(r.sum()), int((f == 0).sum()), int((f == 1).sum()), int((f == 2).sum())],
                    dtype=np.int64)


def gibbs_logjoint(f, r, S_B, lM, lngamma, lnpi2):
    """log p(f, r, b, bt; theta) per chain = minus the first four energy terms at one-hot q
    (fit.py:149-152)."""
    (G, C) = f.shape
    ends = edge_endpoints(r.shape[1])
    out = np.zeros(G)
    cs = np.arange(C)
    for g in range(G):
        fg = f[g].astype(np.int64)
        lj = np.sum(lngamma[fg] + S_B[cs, fg])
        lj += np.sum(np.where(r[g] != 0, lnpi2[1], lnpi2[0]))
        l = mix_index(r[g][ends[:, 0]], r[g][ends[:, 1]])           # (C, U)
        lj += np.sum(lM[cs[:, None], np.arange(lM.shape[1])[None, :], fg[:, None], l])
        out[g] = lj
    return out


# ----------------------------------------------------------------------------------------
# front-end oracle (third party: numpy.corrcoef; no reference code exists -> "parity unpinned")
# ----------------------------------------------------------------------------------------
def corr_edges(ts, fisher_z=False):
    """(S, Nreg, T) -> (C, S): numpy.corrcoef per subject, lower-triangular edge order (util.py:62-84)."""
    (S, Nreg, _T) = ts.shape
    ends = edge_endpoints(Nreg)
    out = np.zeros((ends.shape[0], S))
    for s in range(S):
        cc = np.corrcoef(ts[s])
        out[:, s] = cc[ends[:, 0], ends[:, 1]]
    return np.arctanh(out) if fisher_z else out


# ----------------------------------------------------------------------------------------
# (eta, epsilon) step: objective and gradient for general weights W (C,U,3,3)
#   W = q_F[c,k] * w_l(c,u)  ->  (E_lM, -dE/dh, -dE/de) of fit.py:489-511, 600-664
# ----------------------------------------------------------------------------------------
def vb_weights(q_F, q_R):
    C = q_F.shape[0]
    U = q_R.shape[1]
    W = np.zeros((C, U, 3, 3))
    for c in range(C):
        (n, m) = c_to_nm(c)
        w = eval_q_R_w(q_R, n, m)
        for k in range(3):
            W[c, :, k, :] = q_F[c, 0, k] * w
    return W


def theta_sub_objective(bt, W, mu, sigma, eta, epsilon):
    (C, U) = bt.shape
    norm = np.zeros((C, U, 3))
    for k in range(3):
        norm[:, :, k] = norm_pdf(bt, mu[k], sigma[k])
    S = dh = de = 0.0
    for k in range(3):
        for l in range(3):
            M = eval_M(norm, eta, epsilon, k, l)
            w = W[:, :, k, l]
            nz = w != 0
            S += np.sum(w[nz] * np.log(M[nz]))
            de += np.sum(w[nz] * eval_dlM_de(norm, M, eta, k, l)[nz])
            if l == 2:
                dh += np.sum(w[nz] * eval_dlM_dh(norm, M, epsilon, k)[nz])
    return S, dh, de


def theta_full_objective(b, bt, W, mu, sigma, eta, epsilon):
    """
    The full theta_sub objective the reference comments out (fit.py:232-237, 250-251, 266-267, 282) and its gradient:
        S = sum_{c,k} wF[c,k] sum_h ln N(b_ch; mu_k, sigma_k) + sum W[c,u,k,l] ln M_kl(bt_cu),  wF[c,k] = sum_l W[c,0,k,l]
    Returns (S, dS/d eta, dS/d epsilon, dS/d mu (3,), dS/d sigma^2 (3,)).  Built from the reference's helper forms (pinned
    by fixture G8): eval_dlN_dm / eval_dN_dm (fit.py:709-719), eval_dlN_ds / eval_dN_ds (fit.py:721-733) -- the latter
    are sigma^2 times the derivative in sigma^2 (i.e. the derivative in ln sigma^2), hence the division below -- and the
    coefficient of fit.py:700-707: eps_l for j == k, (1 - eps_l)/2 otherwise.  b=None leaves the first sum out.
    """
    (C, U) = bt.shape
    mu, sigma = np.asarray(mu, dtype=np.float64), np.asarray(sigma, dtype=np.float64)
    norm = np.zeros((C, U, 3))
    for k in range(3):
        norm[:, :, k] = norm_pdf(bt, mu[k], sigma[k])
    (S, dh, de) = theta_sub_objective(bt, W, mu, sigma, eta, epsilon)
    dm, ds = np.zeros(3), np.zeros(3)
    for k in range(3):
        for l in range(3):
            M = eval_M(norm, eta, epsilon, k, l)
            w = W[:, :, k, l]
            nz = w != 0
            e = eval_M_eps(eta, epsilon, l)
            for j in range(3):
                cf = e if j == k else (1 - e) / 2
                dm[j] += np.sum((w * cf * eval_dN_dm(norm[:, :, j], bt, mu[j], sigma[j]) / M)[nz])
                ds[j] += np.sum((w * cf * eval_dN_ds(norm[:, :, j], bt, mu[j], sigma[j]) / M)[nz]) / (sigma[j] * sigma[j])
    if b is not None:
        wF = np.sum(W[:, 0, :, :], axis=2)                  # (C, 3)
        for k in range(3):
            S += np.sum(wF[:, k] * np.sum(norm_logpdf(b, mu[k], sigma[k]), axis=1))
            dm[k] += np.sum(wF[:, k] * np.sum(eval_dlN_dm(b, mu[k], sigma[k]), axis=1))
            ds[k] += np.sum(wF[:, k] * np.sum(eval_dlN_ds(b, mu[k], sigma[k]), axis=1)) / (sigma[k] * sigma[k])
    return S, dh, de, dm, ds


def pair_counts(f, r):
    """(C,U,3,3) counts over chains of (f_c, mixture case at (c,u))."""
    (G, C) = f.shape
    (Nreg, U) = r.shape[1:]
    ends = edge_endpoints(Nreg)
    W = np.zeros((C, U, 3, 3))
    cs = np.arange(C)[:, None]
    us = np.arange(U)[None, :]
    for g in range(G):
        l = mix_index(r[g][ends[:, 0]], r[g][ends[:, 1]])
        np.add.at(W, (cs, us, f[g].astype(np.int64)[:, None], l), 1.0)
    return W
