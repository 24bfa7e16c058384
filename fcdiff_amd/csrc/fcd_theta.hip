// theta_sub M-step support: objective and analytic gradient of the pooled E[ln p(bt | f, r; eta, epsilon)].
//
// The reference sketches this step (fcdiff/fit.py:222-286: bounded minimisation of -E_lM over (eta, epsilon), bounds
// (1e-5, 1-1e-5)) but cannot run it (fit.py:239 calls an undefined name); its derivative helpers are complete and
// unit-tested (fit.py:600-697, test_fcdiff/test_fit.py:790-1087).  Here, for weights W[c,u,k,l] >= 0:
//   S     = sum W[c,u,k,l] * ln M_kl(bt_cu)                         E_lM, fit.py:489-511
//   dS/dh = sum W[c,u,k,2] * (2e-1) (N_k - 0.5 sum_{j!=k} N_j) / M_k2        _eval_dlM_dh, fit.py:618-641
//   dS/de = sum W[c,u,k,l] * c_l (N_k - 0.5 sum_{j!=k} N_j) / M_kl,  c = (-1, +1, 2h-1)   _eval_dlM_de, fit.py:667-697
// W = q_F[c,k] * w_l(c,u) for the variational fit (w of fit.py:382-406), or the pooled chain counts
// #{chains: f_c = k, mixture case l at (c,u)} for the sampler (MCEM).  Deterministic (fixed reduction order).
#include "fcd_common.h"

namespace {

constexpr double kSqrt2Pi = 2.5066282746310002;

struct SubTheta {
    double mu[3], sigma[3];
    double eps[3];        // _eval_M_eps(eta, epsilon, l)
    double deps_de[3];    // d eps_l / d epsilon = (-1, 1, 2 eta - 1)
    double deps_dh;       // d eps_2 / d eta = 2 epsilon - 1
};

// W = q_F[c,k] * w_l(c,u) from the log-probabilities (fit.py:382-406, 508-510); one thread per (c,u)
__global__ __launch_bounds__(256) void weights_vb_kernel(const double *__restrict__ lq_F, const double *__restrict__ lq_R,
                                                         int64_t C, int U, double *__restrict__ W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * U) return;
    const int64_t c = i / U;
    const int u = (int)(i - c * U);
    int n, m;
    fcd_edge_to_pair(c, n, m);
    const double q0n = exp(lq_R[((int64_t)n * U + u) * 2]), q1n = exp(lq_R[((int64_t)n * U + u) * 2 + 1]);
    const double q0m = exp(lq_R[((int64_t)m * U + u) * 2]), q1m = exp(lq_R[((int64_t)m * U + u) * 2 + 1]);
    double w[3];
    w[0] = q0n * q0m;
    w[1] = q1n * q1m;
    w[2] = q0n * q1m;
    w[2] += q1n * q0m;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double qF = exp(lq_F[c * 3 + k]);
#pragma unroll
        for (int l = 0; l < 3; ++l) W[i * 9 + k * 3 + l] = qF * w[l];
    }
}

// pooled counts of the sampler: W[c,u,k,l] (+)= #{chains of this rank with f_c = k and mixture case l at (c,u)}
// one wave per (c,u); lanes = chains of a word; ballots over the nine (k,l) combinations
__global__ __launch_bounds__(256) void pair_counts_kernel(const uint8_t *__restrict__ f_state, const uint64_t *__restrict__ r_bits,
                                                          int Nreg, int U, int64_t C, int GW, int64_t G, int accumulate,
                                                          double *__restrict__ W) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= C * U) return;
    const int64_t c = item / U;
    const int u = (int)(item - c * U);
    int n, m;
    fcd_edge_to_pair(c, n, m);
    uint32_t cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < GW; ++w) {
        const uint64_t act = fcd_active_mask(w, G);
        const uint64_t rn = r_bits[((int64_t)w * Nreg + n) * U + u], rm = r_bits[((int64_t)w * Nreg + m) * U + u];
        const uint64_t lm[3] = {~(rn | rm), rn & rm, rn ^ rm};       // typical, both anomalous, discordant
        const int k = f_state[((int64_t)w * C + c) * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const uint64_t fk = __ballot(k == kk) & act;
#pragma unroll
            for (int l = 0; l < 3; ++l) cnt[kk * 3 + l] += __popcll(fk & lm[l]);
        }
    }
    if (lane < 9) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) v = (lane == j) ? (double)cnt[j] : v;
        W[item * 9 + lane] = accumulate ? W[item * 9 + lane] + v : v;
    }
}

constexpr int OBJ_BLOCK = 256;
__global__ __launch_bounds__(OBJ_BLOCK) void theta_sub_kernel(const double *__restrict__ bt, const double *__restrict__ W,
                                                              int64_t n_items, SubTheta th, double *__restrict__ partial) {
    __shared__ double red[OBJ_BLOCK / 64][3];
    double S = 0.0, gh = 0.0, ge = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * OBJ_BLOCK + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * OBJ_BLOCK) {
        const double x = bt[i];
        double N[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double z = (x - th.mu[k]) / th.sigma[k];
            N[k] = exp(-(z * z) / 2.0) / kSqrt2Pi / th.sigma[k];                    // fit.py:115
        }
        const double others[3] = {N[1] + N[2], N[0] + N[2], N[0] + N[1]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double slope = N[k] - 0.5 * others[k];                            // dM/d eps
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double w = W[i * 9 + k * 3 + l];
                if (w != 0.0) {                                                     // a zero weight never touches ln 0
                    const double M = th.eps[l] * N[k] + (1 - th.eps[l]) * 0.5 * others[k];      // fit.py:430
                    S += w * log(M);
                    const double r = w * slope / M;
                    ge += th.deps_de[l] * r;
                    if (l == 2) gh += th.deps_dh * r;
                }
            }
        }
    }
    S = fcd_wave_sum(S);
    gh = fcd_wave_sum(gh);
    ge = fcd_wave_sum(ge);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = S; red[wave][1] = gh; red[wave][2] = ge;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int q = 0; q < OBJ_BLOCK / 64; ++q) t += red[q][threadIdx.x];
        partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// The FULL theta_sub objective the reference intends (commented out at fit.py:232-237, 250-251, 266-267, 282):
//   S(eta, epsilon, mu, sigma^2) = E[ln p(b | f)] + E[ln p(bt | f, r)]
//                               = sum_{c,k} wF[c,k] sum_h ln N(b_ch; mu_k, sigma_k) + sum W[c,u,k,l] ln M_kl(bt_cu)
// with wF[c,k] = sum_l W[c,0,k,l] (= q_F[c,k], or the number of chains with f_c = k), and its gradient
//   out9 = {S, dS/d eta, dS/d epsilon, dS/d mu_0..2, dS/d sigma^2_0..2}.
// Derivative forms: d ln N/d mu = (b - mu)/sigma^2 (_eval_dlN_dm, fit.py:715-719), d ln N/d sigma^2 =
// ((b - mu)^2 - sigma^2)/(2 sigma^4) -- the reference's _eval_dlN_ds (fit.py:727-733) is sigma^2 times this (the
// derivative in ln sigma^2) --, dM_kl/d theta_j = (eps_l if j == k else (1 - eps_l)/2) dN_j/d theta_j (fit.py:700-707;
// fit.py:572-597 has the same form up to quirk Q8; doc/methods.rst:715-944): the TRUE gradient of S, which is what
// L-BFGS-B needs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OBJ_BLOCK) void theta_full_kernel(const double *__restrict__ b, const double *__restrict__ bt,
                                                               const double *__restrict__ W, int64_t C, int H, int U,
                                                               SubTheta th, double *__restrict__ partial) {
    __shared__ double red[OBJ_BLOCK / 64][9];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // S, gh, ge, gm[3], gs[3]
    double s2[3], ls[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s2[k] = th.sigma[k] * th.sigma[k];
        ls[k] = log(th.sigma[k]);
    }
    const int64_t n_bt = C * U;
    for (int64_t i = (int64_t)blockIdx.x * OBJ_BLOCK + threadIdx.x; i < n_bt; i += (int64_t)gridDim.x * OBJ_BLOCK) {
        const double x = bt[i];
        double N[3], dNm[3], dNs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double d = x - th.mu[k];
            const double z = d / th.sigma[k];
            N[k] = exp(-(z * z) / 2.0) / kSqrt2Pi / th.sigma[k];                    // fit.py:115
            dNm[k] = N[k] * (d / s2[k]);                                            // _eval_dN_dm
            dNs[k] = N[k] * ((d * d - s2[k]) / (2.0 * s2[k] * s2[k]));              // d N / d sigma^2
        }
        const double others[3] = {N[1] + N[2], N[0] + N[2], N[0] + N[1]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double slope = N[k] - 0.5 * others[k];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double w = W[i * 9 + k * 3 + l];
                if (w != 0.0) {
                    const double M = th.eps[l] * N[k] + (1 - th.eps[l]) * 0.5 * others[k];
                    acc[0] += w * log(M);
                    const double r = w / M;
                    acc[2] += th.deps_de[l] * (r * slope);
                    if (l == 2) acc[1] += th.deps_dh * (r * slope);
                    const double co = 0.5 * (1 - th.eps[l]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double cf = (j == k) ? th.eps[l] : co;
                        acc[3 + j] += r * (cf * dNm[j]);
                        acc[6 + j] += r * (cf * dNs[j]);
                    }
                }
            }
        }
    }
    if (b) {
        const int64_t n_b = C * H;
        for (int64_t i = (int64_t)blockIdx.x * OBJ_BLOCK + threadIdx.x; i < n_b; i += (int64_t)gridDim.x * OBJ_BLOCK) {
            const int64_t c = i / H;
            const double x = b[i];
            const double *w0 = W + c * U * 9;              // patient 0 of the edge: sum over l = weight of f_c = k
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double wF = (w0[k * 3 + 0] + w0[k * 3 + 1]) + w0[k * 3 + 2];
                if (wF != 0.0) {
                    const double d = x - th.mu[k];
                    const double z = d / th.sigma[k];
                    acc[0] += wF * ((-(z * z) / 2.0 - 0.91893853320467274178) - ls[k]);     // norm.logpdf, fit.py:114
                    acc[3 + k] += wF * (d / s2[k]);                                         // _eval_dlN_dm
                    acc[6 + k] += wF * ((d * d - s2[k]) / (2.0 * s2[k] * s2[k]));
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const double v = fcd_wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double t = 0.0;
        for (int q = 0; q < OBJ_BLOCK / 64; ++q) t += red[q][threadIdx.x];
        partial[(int64_t)blockIdx.x * 12 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void theta_full_fold(const double *__restrict__ partial, int n_blocks, double *__restrict__ out9) {
    __shared__ double red[4][9];
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int bl = threadIdx.x; bl < n_blocks; bl += 256)
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] += partial[(int64_t)bl * 12 + j];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) out9[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void theta_sub_fold(const double *__restrict__ partial, int n_blocks, double *__restrict__ out3) {
    __shared__ double red[4][3];
    double v[3] = {0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
        v[0] += partial[(int64_t)b * 4 + 0];
        v[1] += partial[(int64_t)b * 4 + 1];
        v[2] += partial[(int64_t)b * 4 + 2];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) out3[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

}  // namespace

extern "C" int fcd_theta_sub_weights_vb(fcd_ctx *ctx, const double *lq_F, const double *lq_R, int64_t Nreg, int64_t U, double *W,
                                        fcd_stream stream) {
    if (!ctx || !lq_F || !lq_R || !W) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_sub_weights_vb: null pointer");
    if (Nreg < 2 || U < 1 || U > INT32_MAX) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    const int64_t C = fcd_tri(Nreg);
    hipLaunchKernelGGL(weights_vb_kernel, dim3((unsigned)((C * U + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lq_F, lq_R, C,
                       (int)U, W);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_pair_counts(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                                     int64_t G, int accumulate, double *W, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits || !W) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_pair_counts: null pointer");
    const int64_t items = g.C * U;
    if ((items + 3) / 4 > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_gibbs_pair_counts: C*U too large");
    hipLaunchKernelGGL(pair_counts_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f_state, r_bits,
                       (int)Nreg, (int)U, g.C, g.GW, G, accumulate ? 1 : 0, W);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_theta_sub_objective(fcd_ctx *ctx, const double *bt, const double *W, int64_t C, int64_t U,
                                       const double *theta, double *out3, fcd_stream stream) {
    if (!ctx || !bt || !W || !theta || !out3) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_sub_objective: null pointer");
    if (C < 1 || U < 1) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_sub_objective: C=%lld U=%lld must be >= 1", C, U);
    SubTheta th;
    const double eta = theta[1], epsilon = theta[2];
    for (int k = 0; k < 3; ++k) {
        th.mu[k] = theta[6 + k];
        th.sigma[k] = theta[9 + k];
    }
    th.eps[0] = 1 - epsilon;                       // _eval_M_eps, fit.py:433-444
    th.eps[1] = epsilon;
    double e2 = eta * epsilon;
    e2 += (1 - eta) * (1 - epsilon);
    th.eps[2] = e2;
    th.deps_de[0] = -1;                            // fit.py:689-694
    th.deps_de[1] = 1;
    th.deps_de[2] = 2 * eta - 1;
    th.deps_dh = (2 * epsilon) - 1;                // fit.py:638
    const int64_t n_items = C * U;
    int64_t blocks = (n_items + OBJ_BLOCK - 1) / OBJ_BLOCK;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    int rc = fcd_ws_reserve(ctx, (size_t)blocks * 4 * sizeof(double));
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(theta_sub_kernel, dim3((unsigned)blocks), dim3(OBJ_BLOCK), 0, s, bt, W, n_items, th, (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(theta_sub_fold, dim3(1), dim3(256), 0, s, (const double *)ctx->ws, (int)blocks, out3);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_theta_full_objective(fcd_ctx *ctx, const double *b, const double *bt, const double *W, int64_t C, int64_t H,
                                        int64_t U, const double *theta, double *out9, fcd_stream stream) {
    if (!ctx || !bt || !W || !theta || !out9) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_full_objective: null pointer");
    if (C < 1 || U < 1 || (b && H < 1) || H > INT32_MAX || U > INT32_MAX)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_full_objective: C=%lld U=%lld must be >= 1", C, U);
    SubTheta th;
    const double eta = theta[1], epsilon = theta[2];
    for (int k = 0; k < 3; ++k) {
        th.mu[k] = theta[6 + k];
        th.sigma[k] = theta[9 + k];
        if (!(th.sigma[k] > 0.0)) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_theta_full_objective: sigma must be positive");
    }
    th.eps[0] = 1 - epsilon;                       // _eval_M_eps, fit.py:433-444
    th.eps[1] = epsilon;
    double e2 = eta * epsilon;
    e2 += (1 - eta) * (1 - epsilon);
    th.eps[2] = e2;
    th.deps_de[0] = -1;
    th.deps_de[1] = 1;
    th.deps_de[2] = 2 * eta - 1;
    th.deps_dh = (2 * epsilon) - 1;
    const int64_t n_items = C * (U > H ? U : H);
    int64_t blocks = (n_items + OBJ_BLOCK - 1) / OBJ_BLOCK;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    int rc = fcd_ws_reserve(ctx, (size_t)blocks * 12 * sizeof(double));
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(theta_full_kernel, dim3((unsigned)blocks), dim3(OBJ_BLOCK), 0, s, b, bt, W, C, (int)H, (int)U, th,
                       (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(theta_full_fold, dim3(1), dim3(256), 0, s, (const double *)ctx->ws, (int)blocks, out9);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
