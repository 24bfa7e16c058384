// K_corr: per-subject region x time sample correlation -> edge-major correlations, optional Fisher z.
//
// Not in the reference (its inputs are already correlations, fcdiff/fit.py:20-23): this is the front-end that
// BASELINE.json's north_star names; oracle = numpy.corrcoef ("parity unpinned").
// Fisher z (atanh) is OFF by default: the model's Normal components live on raw correlations clipped to
// [-1, 1] (fcdiff/model.py:213, 236), so every default of the reference assumes untransformed values.
//
// numpy.corrcoef, restated:  X -= mean(X, axis=1);  c = (X X^T) * (1/(T-1));  s = sqrt(diag c);
//                            c /= s[:, None];  c /= s[None, :];  clip to [-1, 1].
// The Gram matrix is an fp64 MFMA product (v_mfma_f64_16x16x4_f64): one wave per 16x16 tile of the lower
// triangle, K = T in steps of 4.  Output out[c][s], c = n(n-1)/2 + m (n > m): the layout of b / bt.
#include "fcd_common.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// mean and 1-sigma of every (subject, region) row: one wave per row
__global__ __launch_bounds__(256) void corr_moments_kernel(const double *__restrict__ ts, int64_t rows, int T,
                                                           double *__restrict__ mean, double *__restrict__ sdev) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double *x = ts + row * T;
    double s = 0.0;
    for (int k = lane; k < T; k += 64) s += x[k];
    s = fcd_wave_sum(s);
    const double mu = s / (double)T;
    double q = 0.0;
    for (int k = lane; k < T; k += 64) {
        const double d = x[k] - mu;
        q += d * d;
    }
    q = fcd_wave_sum(q);
    if (lane == 0) {
        mean[row] = mu;
        sdev[row] = sqrt(q * (1.0 / (double)(T - 1)));     // sqrt(diag(cov)), cov = X X^T * (1/(T-1))
    }
}

// grid = (lower-triangle tiles, subjects); block = one wave.
// A[i][k] = xc[16 I + i][k0 + k] sits in lane (i = l & 15, k = l >> 4); B[k][j] = xc[16 J + j][k0 + k] in lane
// (j = l & 15, k = l >> 4); D[i][j] comes back as 4 doubles per lane: col = l & 15, row = (l >> 4) + 4 r.
__global__ __launch_bounds__(64) void corr_tiles_kernel(const double *__restrict__ ts, const double *__restrict__ mean,
                                                        const double *__restrict__ sdev, int Nreg, int T, int64_t S,
                                                        int fisher_z, double *__restrict__ out) {
    // tile index -> (I, J), I >= J, lower-triangular row-major like the edges themselves
    const int t = blockIdx.x;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)I * (I + 1) / 2 > t) --I;
    while ((int64_t)(I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const int64_t s = blockIdx.y;
    const int l = threadIdx.x;
    const int ra = I * 16 + (l & 15), rb = J * 16 + (l & 15), kq = l >> 4;
    const bool va = ra < Nreg, vb = rb < Nreg;
    const double *xa = ts + (s * Nreg + (va ? ra : 0)) * T;
    const double *xb = ts + (s * Nreg + (vb ? rb : 0)) * T;
    const double ma = va ? mean[s * Nreg + ra] : 0.0, mb = vb ? mean[s * Nreg + rb] : 0.0;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < T; k0 += 4) {
        const int k = k0 + kq;
        const bool vk = k < T;
        const double a = (va && vk) ? xa[k] - ma : 0.0;
        const double b = (vb && vk) ? xb[k] - mb : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const double inv = 1.0 / (double)(T - 1);
    const int m = J * 16 + (l & 15);                      // column = the smaller region index of the pair
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = I * 16 + (l >> 4) + 4 * r;          // row
        if (n < Nreg && m < n) {
            double c = acc[r] * inv;
            c /= sdev[s * Nreg + n];
            c /= sdev[s * Nreg + m];
            c = fmin(fmax(c, -1.0), 1.0);
            if (fisher_z) c = atanh(c);
            out[(fcd_tri(n) + m) * S + s] = c;
        }
    }
}

}  // namespace

extern "C" int fcd_corr_edges(fcd_ctx *ctx, const double *ts, int64_t S, int64_t Nreg, int64_t T, int fisher_z, double *out,
                              fcd_stream stream) {
    if (!ctx || !ts || !out) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_corr_edges: null pointer");
    if (S < 1 || Nreg < 2 || T < 2) return fcd_fail(ctx, FCD_ERR_SHAPE, "need S >= 1, Nreg >= 2, T >= 2 (Nreg=%lld, T=%lld)", Nreg, T);
    if (S > 65535 || Nreg > 46340 || T > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: shape too large");
    const int64_t rows = S * Nreg;
    int rc = fcd_ws_reserve(ctx, (size_t)rows * 2 * sizeof(double));
    if (rc) return rc;
    double *mean = (double *)ctx->ws, *sdev = mean + rows;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_moments_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ts, rows, (int)T, mean, sdev);
    FCD_LAUNCH_CHECK();
    const int64_t nt = (Nreg + 15) / 16;
    const int64_t tiles = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(corr_tiles_kernel, dim3((unsigned)tiles, (unsigned)S), dim3(64), 0, s, ts, mean, sdev, (int)Nreg, (int)T, S,
                       fisher_z ? 1 : 0, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
