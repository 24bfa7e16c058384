// Forward (ancestral) sampling of the IAR model on the device: UnsharedRegionModel.sample, fcdiff/model.py:52-236.
// Counter-based (Philox4x32-10), so every variable of every (edge, subject) is drawn independently in parallel; the
// MT19937 stream of the reference is NOT reproduced (the host sampler fcdiff_amd.model does that, fixture G9) --
// parity here is statistical, in the style of test_fcdiff/test_model.py.  Edges are in the FITTER's order
// (c = n(n-1)/2 + m, fcdiff/util.py:62-84), so the output can be fitted as is (quirk Q3 does not arise).
#include "fcd_common.h"

namespace {

enum : uint32_t { K_R = 16, K_F = 17, K_T = 18, K_FT = 19, K_B = 20, K_BT = 21 };

struct SampTheta {
    double pi, eta, epsilon, g0, g01;   // g0 = gamma_0 / sum, g01 = (gamma_0 + gamma_1) / sum
    double mu[3], sigma[3];
};

__device__ inline double unif(uint64_t seed, uint32_t idx, uint32_t hi, uint32_t kind, int half) {
    return fcd_site_uniform(seed, idx, hi, 0u, kind, half);
}

__device__ inline int draw_r(const SampTheta &th, uint64_t seed, int n, int u, int U) {
    const int64_t i = (int64_t)n * U + u;                       // model.py:108
    return unif(seed, (uint32_t)i, (uint32_t)(i >> 32), K_R, 0) < th.pi;
}

__device__ inline int draw_f(const SampTheta &th, uint64_t seed, int64_t c) {
    const double x = unif(seed, (uint32_t)c, (uint32_t)(c >> 32), K_F, 0);      // model.py:160
    return x < th.g0 ? 0 : (x < th.g01 ? 1 : 2);
}

// standard normal from two uniforms (Box-Muller); u1 in (0, 1]
__device__ inline double normal(uint64_t seed, int64_t i, uint32_t kind) {
    const fcd_u4 x = fcd_philox((uint32_t)i, (uint32_t)(i >> 32), 0u, kind, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = 1.0 - fcd_u53(x.x, x.y), u2 = fcd_u53(x.z, x.w);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ __launch_bounds__(256) void sample_patients_kernel(SampTheta th, uint64_t seed, int Nreg, int U, int64_t C,
                                                              uint8_t *__restrict__ r, uint8_t *__restrict__ t,
                                                              uint8_t *__restrict__ f, uint8_t *__restrict__ ft,
                                                              double *__restrict__ bt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Nreg * U) r[i] = (uint8_t)draw_r(th, seed, (int)(i / U), (int)(i % U), U);
    if (i >= C * U) return;
    const int64_t c = i / U;
    const int u = (int)(i - c * U);
    int n, m;
    fcd_edge_to_pair(c, n, m);
    const int rn = draw_r(th, seed, n, u, U), rm = draw_r(th, seed, m, u, U);
    // anomalous connection: both typical -> no, both anomalous -> yes, discordant -> Bernoulli(eta)   model.py:136-141
    int tt = rn & rm;
    if (rn ^ rm) tt = unif(seed, (uint32_t)i, (uint32_t)(i >> 32), K_T, 0) < th.eta;
    const int fc = draw_f(th, seed, c);
    if (u == 0) f[c] = (uint8_t)fc;
    // patient's connection type: keeps the template type w.p. 1-eps (t = 0) or eps (t = 1), else one of the other two
    const double keep = tt ? th.epsilon : 1.0 - th.epsilon;                                          // model.py:182-188
    const double x = unif(seed, (uint32_t)i, (uint32_t)(i >> 32), K_FT, 0);
    int k = fc;
    if (x >= keep) k = (fc + 1 + ((x - keep) >= 0.5 * (1.0 - keep) ? 1 : 0)) % 3;
    t[i] = (uint8_t)tt;
    ft[i] = (uint8_t)k;
    const double v = th.mu[k] + th.sigma[k] * normal(seed, i, K_BT);                                 // model.py:233-236
    bt[i] = fmin(fmax(v, -1.0), 1.0);
}

__global__ __launch_bounds__(256) void sample_healthy_kernel(SampTheta th, uint64_t seed, int H, int64_t C, double *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * H) return;
    const int k = draw_f(th, seed, i / H);
    const double v = th.mu[k] + th.sigma[k] * normal(seed, i, K_B);                                  // model.py:210-213
    b[i] = fmin(fmax(v, -1.0), 1.0);
}

}  // namespace

extern "C" int fcd_model_sample(fcd_ctx *ctx, const double *theta, int64_t Nreg, int64_t H, int64_t U, uint64_t seed, uint8_t *r,
                                uint8_t *t, uint8_t *f, uint8_t *f_tilde, double *b, double *b_tilde, fcd_stream stream) {
    if (!ctx || !theta || !r || !t || !f || !f_tilde || !b || !b_tilde) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_model_sample: null pointer");
    if (Nreg < 2 || H < 1 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2, H >= 1, U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (Nreg > 46340 || U > INT32_MAX || H > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_model_sample: shape too large");
    SampTheta th;
    th.pi = theta[0];
    th.eta = theta[1];
    th.epsilon = theta[2];
    const double gs = theta[3] + theta[4] + theta[5];
    th.g0 = theta[3] / gs;
    th.g01 = (theta[3] + theta[4]) / gs;
    for (int k = 0; k < 3; ++k) {
        th.mu[k] = theta[6 + k];
        th.sigma[k] = theta[9 + k];
    }
    const int64_t C = fcd_tri(Nreg);
    int64_t items = C * U;
    if (Nreg * U > items) items = Nreg * U;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sample_patients_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, th, seed, (int)Nreg, (int)U, C,
                       r, t, f, f_tilde, b_tilde);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sample_healthy_kernel, dim3((unsigned)((C * H + 255) / 256)), dim3(256), 0, s, th, seed, (int)H, C, b);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
