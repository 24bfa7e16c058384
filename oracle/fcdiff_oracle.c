/*
 * CPU oracle for the fcdiff fit path, plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (oracle/liboracle.so); the product (fcdiff_amd/) never does.
 *
 * It restates, for sizes the NumPy oracle (oracle/fcdiff_oracle.py) is too slow for:
 *   - the likelihood tables of UnsharedRegionFit._update_lps    fcdiff/fit.py:104-122, 409-444
 *   - the variational updates q_F, q_R and the free energy      fcdiff/fit.py:142-198, 447-539
 *   - the build-defined many-chain collapsed Gibbs sampler whose conditionals are those updates at
 *     one-hot q (fit.py:170-173, :187-194), with the Philox4x32-10 counter RNG.
 * It is itself checked against the NumPy oracle, which is pinned to fixtures captured from the
 * reference (tests/golden/G1..G12, tests/test_oracle_golden.py, tests/test_oracle_c.py).
 *
 * State layout here is the plain one: f (G, C) uint8, r (G, Nreg, U) uint8.  Chains run in
 * parallel with OpenMP when compiled with -fopenmp (each chain is independent given the tables).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EDGE_REFERENCE 0
#define EDGE_SYMMETRIC 1

static const double SQRT_2PI = 2.5066282746310002;     /* numpy sqrt(2*pi) */
static const double LOG_SQRT_2PI = 0.9189385332046727; /* numpy log(sqrt(2*pi)) */

static inline int64_t tri(int64_t n) { return n * (n - 1) / 2; }

/* fcdiff/util.py:62-84 with integer results */
static inline void c_to_nm(int64_t c, int *n, int *m) {
    int64_t nn = (int64_t)floor((sqrt(8.0 * (double)c + 1.0) - 1.0) / 2.0) + 1;
    while (tri(nn) > c) --nn;
    while (tri(nn + 1) <= c) ++nn;
    *n = (int)nn;
    *m = (int)(c - tri(nn));
}

/* fit.py:186 calls nm_to_c(n, m) = n(n-1)/2 + m for every ordered pair (quirk Q1) */
static inline int64_t edge_id(int n, int m, int mode) {
    if (mode == EDGE_REFERENCE || n > m) return tri(n) + m;
    return tri(m) + n;
}

/* ------------------------------------------------------------------ tables: fit.py:104-122 */
void oracle_lik_tables(const double *b, const double *bt, int64_t C, int64_t H, int64_t U, const double *theta,
                       double *S_B, double *lM) {
    const double eta = theta[1], epsilon = theta[2];
    const double *mu = theta + 6, *sigma = theta + 9;
    double eps[3];
    eps[0] = 1 - epsilon;
    eps[1] = epsilon;
    eps[2] = eta * epsilon;
    eps[2] += (1 - eta) * (1 - epsilon);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        for (int k = 0; k < 3; ++k) {
            double s = 0.0;
            const double lns = log(sigma[k]);
            for (int64_t h = 0; h < H; ++h) {
                const double z = (b[c * H + h] - mu[k]) / sigma[k];
                s += -(z * z) / 2.0 - LOG_SQRT_2PI - lns;
            }
            S_B[c * 3 + k] = s;
        }
        for (int64_t u = 0; u < U; ++u) {
            double N[3];
            for (int k = 0; k < 3; ++k) {
                const double z = (bt[c * U + u] - mu[k]) / sigma[k];
                N[k] = exp(-(z * z) / 2.0) / SQRT_2PI / sigma[k];
            }
            const double others[3] = {N[1] + N[2], N[0] + N[2], N[0] + N[1]};
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 3; ++l)
                    lM[((c * U + u) * 3 + k) * 3 + l] = log(eps[l] * N[k] + (1 - eps[l]) * 0.5 * others[k]);
        }
    }
}

/* ------------------------------------------------------------------ VB: fit.py:157-198 */
static double lse(const double *a, int n) {
    double mx = a[0];
    for (int i = 1; i < n; ++i) mx = a[i] > mx ? a[i] : mx;
    if (!isfinite(mx)) mx = 0.0;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(a[i] - mx);
    return log(s) + mx;
}

void oracle_update_lq_F(const double *lq_R, const double *S_B, const double *lM, const double *gamma, int64_t Nreg,
                        int64_t U, double *lq_F) {
    const int64_t C = tri(Nreg);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        int n, m;
        c_to_nm(c, &n, &m);
        double a[3];
        for (int k = 0; k < 3; ++k) {
            double s = 0.0;
            for (int64_t u = 0; u < U; ++u) {
                const double q0n = exp(lq_R[(n * U + u) * 2]), q1n = exp(lq_R[(n * U + u) * 2 + 1]);
                const double q0m = exp(lq_R[(m * U + u) * 2]), q1m = exp(lq_R[(m * U + u) * 2 + 1]);
                const double *p = lM + ((c * U + u) * 3 + k) * 3;
                double w2 = q0n * q1m;
                w2 += q1n * q0m;
                s += (q0n * q0m) * p[0] + (q1n * q1m) * p[1] + w2 * p[2];
            }
            a[k] = log(gamma[k]) + (S_B[c * 3 + k] + s);
        }
        const double z = lse(a, 3);
        for (int k = 0; k < 3; ++k) lq_F[c * 3 + k] = a[k] - z;
    }
}

void oracle_update_lq_R(const double *lq_F, const double *lM, const double *pi2, int64_t Nreg, int64_t U, int mode,
                        double *lq_R) {
#pragma omp parallel for schedule(static)
    for (int64_t u = 0; u < U; ++u) {
        double *q0 = (double *)malloc(sizeof(double) * 2 * Nreg), *q1 = q0 + Nreg;
        for (int64_t n = 0; n < Nreg; ++n) {
            q0[n] = exp(lq_R[(n * U + u) * 2]);
            q1[n] = exp(lq_R[(n * U + u) * 2 + 1]);
        }
        for (int n = 0; n < Nreg; ++n) {
            double s[2] = {log(pi2[0]), log(pi2[1])};
            for (int m = 0; m < Nreg; ++m) {
                if (m == n) continue;
                const int64_t c = edge_id(n, m, mode);
                for (int k = 0; k < 3; ++k) {
                    const double qF = exp(lq_F[c * 3 + k]);
                    const double *p = lM + ((c * U + u) * 3 + k) * 3;
                    s[0] += qF * (q0[m] * p[0] + q1[m] * p[2]);
                    s[1] += qF * (q1[m] * p[1] + q0[m] * p[2]);
                }
            }
            const double z = lse(s, 2);
            lq_R[(n * U + u) * 2] = s[0] - z;
            lq_R[(n * U + u) * 2 + 1] = s[1] - z;
            q0[n] = exp(s[0] - z);
            q1[n] = exp(s[1] - z);
        }
        free(q0);
    }
}

static double xlogy0(double q, double lq) { return q == 0.0 ? 0.0 : q * lq; }

/* six terms of fit.py:142-155 in its order */
void oracle_energy_terms(const double *lq_F, const double *lq_R, const double *S_B, const double *lM,
                         const double *gamma, const double *pi2, int64_t Nreg, int64_t U, double *t6) {
    const int64_t C = tri(Nreg);
    double t[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t c = 0; c < C; ++c) {
        int n, m;
        c_to_nm(c, &n, &m);
        for (int k = 0; k < 3; ++k) {
            const double lq = lq_F[c * 3 + k], q = exp(lq);
            double s = 0.0;
            for (int64_t u = 0; u < U; ++u) {
                const double q0n = exp(lq_R[(n * U + u) * 2]), q1n = exp(lq_R[(n * U + u) * 2 + 1]);
                const double q0m = exp(lq_R[(m * U + u) * 2]), q1m = exp(lq_R[(m * U + u) * 2 + 1]);
                const double *p = lM + ((c * U + u) * 3 + k) * 3;
                s += (q0n * q0m) * p[0] + (q1n * q1m) * p[1] + (q0n * q1m + q1n * q0m) * p[2];
            }
            t[0] += q * log(gamma[k]);
            t[1] += q * S_B[c * 3 + k];
            t[3] += q * s;
            t[4] += xlogy0(q, lq);
        }
    }
    for (int64_t i = 0; i < Nreg * U; ++i) {
        const double l0 = lq_R[i * 2], l1 = lq_R[i * 2 + 1];
        t[2] += exp(l0) * log(pi2[0]) + exp(l1) * log(pi2[1]);
        t[5] += xlogy0(exp(l0), l0) + xlogy0(exp(l1), l1);
    }
    memcpy(t6, t, sizeof(t));
}

/* ------------------------------------------------------------------ Philox4x32-10 (Random123) */
static inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

static inline double u53(uint32_t hi, uint32_t lo) {
    const uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0);
}

static inline double site_uniform(uint64_t seed, uint32_t idx, uint32_t chain, uint32_t sweep, uint32_t kind, int half) {
    uint32_t x[4];
    philox(idx, chain, sweep, kind, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    return half ? u53(x[2], x[3]) : u53(x[0], x[1]);
}

/* sweeps of f: one counter block per four edges, one 32-bit word each */
static inline double site_uniform32(uint64_t seed, uint32_t idx, uint32_t chain, uint32_t sweep, uint32_t kind, int word) {
    uint32_t x[4];
    philox(idx, chain, sweep, kind, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    return (double)x[word] * (1.0 / 4294967296.0);
}

enum { KIND_INIT_F = 0, KIND_INIT_R = 1, KIND_F = 2, KIND_R = 3 };

/* ------------------------------------------------------------------ Gibbs */
void oracle_gibbs_init(uint8_t *f, uint8_t *r, int64_t Nreg, int64_t U, int64_t G, int64_t chain0, uint64_t seed,
                       double pi) {
    const int64_t C = tri(Nreg);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < G; ++g) {
        const uint32_t chain = (uint32_t)(chain0 + g);
        for (int64_t c = 0; c < C; ++c) {
            int v = (int)(site_uniform(seed, (uint32_t)(c >> 1), chain, 0, KIND_INIT_F, (int)(c & 1)) * 3.0);
            f[g * C + c] = (uint8_t)(v > 2 ? 2 : v);
        }
        for (int64_t n = 0; n < Nreg; ++n)
            for (int64_t u = 0; u < U; ++u)
                r[(g * Nreg + n) * U + u] =
                    site_uniform(seed, (uint32_t)((n >> 1) * U + u), chain, 0, KIND_INIT_R, (int)(n & 1)) < pi;
    }
}

static inline int draw_f(double a0, double a1, double a2, double x) {
    double mx = a0 > a1 ? a0 : a1;
    mx = mx > a2 ? mx : a2;
    const double e0 = exp(a0 - mx), e1 = exp(a1 - mx), e2 = exp(a2 - mx);
    const double t = x * ((e0 + e1) + e2);
    return (t < e0) ? 0 : ((t < e0 + e1) ? 1 : 2);
}
/* TEST STATISTIC (not part of the algorithm): how close the draw came to a tie -- the distance of t to the nearer of its
   two thresholds, relative to the sum of the weights */
static inline double draw_f_margin(double a0, double a1, double a2, double x) {
    double mx = a0 > a1 ? a0 : a1;
    mx = mx > a2 ? mx : a2;
    const double e0 = exp(a0 - mx), e1 = exp(a1 - mx), e2 = exp(a2 - mx);
    const double s = (e0 + e1) + e2, t = x * s;
    const double d0 = fabs(t - e0), d1 = fabs(t - (e0 + e1));
    return (d0 < d1 ? d0 : d1) / s;
}

/* f conditionals: fit.py:170-173 at one-hot q_R.  cond (G,C,3) optional: unnormalised log-weights. */
static void gibbs_f_step_impl(uint8_t *f, const uint8_t *r, const double *S_B, const double *lM, const double *lngamma,
                              int64_t Nreg, int64_t U, int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, double *cond,
                              int draw, double *margin) {
    const int64_t C = tri(Nreg);
    double mm = 1.0;
#pragma omp parallel for schedule(dynamic, 1) reduction(min : mm)
    for (int64_t g = 0; g < G; ++g) {
        const uint8_t *rg = r + g * Nreg * U;
        for (int64_t c = 0; c < C; ++c) {
            int n, m;
            c_to_nm(c, &n, &m);
            double a[3] = {0.0, 0.0, 0.0};
            for (int64_t u = 0; u < U; ++u) {
                const int rn = rg[n * U + u], rm = rg[m * U + u];
                const int l = (rn & rm) ? 1 : ((rn ^ rm) ? 2 : 0);
                const double *p = lM + (c * U + u) * 9 + l;
                a[0] += p[0];
                a[1] += p[3];
                a[2] += p[6];
            }
            for (int k = 0; k < 3; ++k) a[k] = lngamma[k] + (S_B[c * 3 + k] + a[k]);
            if (cond) memcpy(cond + (g * C + c) * 3, a, sizeof(a));
            if (draw) {
                const double x = site_uniform32(seed, (uint32_t)(c >> 2), (uint32_t)(chain0 + g), (uint32_t)sweep, KIND_F, (int)(c & 3));
                f[g * C + c] = (uint8_t)draw_f(a[0], a[1], a[2], x);
                if (margin) {
                    const double d = draw_f_margin(a[0], a[1], a[2], x);
                    if (d < mm) mm = d;
                }
            }
        }
    }
    if (margin) *margin = mm;
}
void oracle_gibbs_f_step(uint8_t *f, const uint8_t *r, const double *S_B, const double *lM, const double *lngamma,
                         int64_t Nreg, int64_t U, int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, double *cond,
                         int draw) {
    gibbs_f_step_impl(f, r, S_B, lM, lngamma, Nreg, U, G, chain0, seed, sweep, cond, draw, NULL);
}
/* the same step; *margin = the smallest tie margin of its draws (draw_f_margin) -- a statistic for the parity tests */
void oracle_gibbs_f_step_m(uint8_t *f, const uint8_t *r, const double *S_B, const double *lM, const double *lngamma,
                           int64_t Nreg, int64_t U, int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, double *margin) {
    gibbs_f_step_impl(f, r, S_B, lM, lngamma, Nreg, U, G, chain0, seed, sweep, NULL, 1, margin);
}

/* r conditionals: fit.py:187-194 at one-hot q_F, q_R; regions in order, state refreshed in place. */
static void gibbs_r_step_impl(const uint8_t *f, uint8_t *r, const double *lM, const double *lnpi2, int64_t Nreg, int64_t U,
                              int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, int mode, double *cond, int draw,
                              double *margin) {
    const int64_t C = tri(Nreg);
    double mm = 1e300;
#pragma omp parallel for schedule(dynamic, 1) reduction(min : mm)
    for (int64_t g = 0; g < G; ++g) {
        const uint8_t *fg = f + g * C;
        uint8_t *rg = r + g * Nreg * U;
        for (int n = 0; n < Nreg; ++n) {
            for (int64_t u = 0; u < U; ++u) {
                double s0 = 0.0, s1 = 0.0;
                for (int m = 0; m < Nreg; ++m) {
                    if (m == n) continue;
                    const int64_t c = edge_id(n, m, mode);
                    const double *p = lM + (c * U + u) * 9 + fg[c] * 3;
                    if (rg[m * U + u]) {
                        s0 += p[2];
                        s1 += p[1];
                    } else {
                        s0 += p[0];
                        s1 += p[2];
                    }
                }
                s0 += lnpi2[0];
                s1 += lnpi2[1];
                if (cond) {
                    cond[((g * Nreg + n) * U + u) * 2] = s0;
                    cond[((g * Nreg + n) * U + u) * 2 + 1] = s1;
                }
                if (draw) {
                    /* sweeps: one counter block per (region, pair of patients) */
                    const double x = site_uniform(seed, (uint32_t)(n * ((U + 1) >> 1) + (u >> 1)), (uint32_t)(chain0 + g),
                                                  (uint32_t)sweep, KIND_R, (int)(u & 1));
                    /* r = 1 w.p. sigmoid(s1 - s0): logit(x) < s1 - s0 */
                    const double lx = log(x / (1.0 - x));
                    rg[n * U + u] = lx < (s1 - s0);
                    if (margin) {       /* TEST STATISTIC: |(s1 - s0) - logit(x)|, how close the draw came to a tie */
                        const double d = fabs((s1 - s0) - lx);
                        if (d < mm) mm = d;
                    }
                }
            }
        }
    }
    if (margin) *margin = mm;
}
void oracle_gibbs_r_step(const uint8_t *f, uint8_t *r, const double *lM, const double *lnpi2, int64_t Nreg, int64_t U,
                         int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, int mode, double *cond, int draw) {
    gibbs_r_step_impl(f, r, lM, lnpi2, Nreg, U, G, chain0, seed, sweep, mode, cond, draw, NULL);
}
/* the same step; *margin = the smallest |(s1 - s0) - logit(x)| of its draws -- a statistic for the parity tests */
void oracle_gibbs_r_step_m(const uint8_t *f, uint8_t *r, const double *lM, const double *lnpi2, int64_t Nreg, int64_t U,
                           int64_t G, int64_t chain0, uint64_t seed, int64_t sweep, int mode, double *margin) {
    gibbs_r_step_impl(f, r, lM, lnpi2, Nreg, U, G, chain0, seed, sweep, mode, NULL, 1, margin);
}

void oracle_gibbs_stats(const uint8_t *f, const uint8_t *r, int64_t Nreg, int64_t U, int64_t G, int64_t *counts) {
    const int64_t C = tri(Nreg);
    int64_t c4[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < G * Nreg * U; ++i) c4[0] += r[i];
    for (int64_t i = 0; i < G * C; ++i) c4[1 + f[i]] += 1;
    for (int k = 0; k < 4; ++k) counts[k] = c4[k];
    counts[4] = G;
    counts[5] = counts[6] = counts[7] = 0;
}

/* log p(f, r, b, bt) per chain = minus the first four terms of fit.py:149-152 at one-hot q */
void oracle_gibbs_logjoint(const uint8_t *f, const uint8_t *r, const double *S_B, const double *lM, const double *lngamma,
                           const double *lnpi2, int64_t Nreg, int64_t U, int64_t G, double *out) {
    const int64_t C = tri(Nreg);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < G; ++g) {
        const uint8_t *fg = f + g * C, *rg = r + g * Nreg * U;
        double lj = 0.0;
        for (int64_t c = 0; c < C; ++c) {
            int n, m;
            c_to_nm(c, &n, &m);
            const int k = fg[c];
            double e = lngamma[k] + S_B[c * 3 + k];
            for (int64_t u = 0; u < U; ++u) {
                const int rn = rg[n * U + u], rm = rg[m * U + u];
                const int l = (rn & rm) ? 1 : ((rn ^ rm) ? 2 : 0);
                e += lM[(c * U + u) * 9 + k * 3 + l];
            }
            lj += e;
        }
        int64_t ones = 0;
        for (int64_t i = 0; i < Nreg * U; ++i) ones += rg[i];
        lj += (double)ones * lnpi2[1] + (double)(Nreg * U - ones) * lnpi2[0];
        out[g] = lj;
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
