// K_lik: the per-edge likelihood tables of UnsharedRegionFit._update_lps (fcdiff/fit.py:104-122)
// with _eval_M / _eval_M_eps (fit.py:409-444) fused in.
//
// Streaming, HBM-bound: reads b (C,H) and bt (C,U) once, writes S_B (C,3) and lM (C,U,3,3) once.
// Algorithmic bytes per call = 8*C*(H+U) + 24*C + 72*C*U  (SURVEY.md section 8d).
//
// The (C,H,3) table of the reference is never consumed except through its H-sum (fit.py:171, :472),
// so only S_B[c,k] = sum_h lp_B_g_F[c,h,k] is produced unless the caller asks for the full table.
#include "fcd_common.h"

namespace {

struct LikTheta {
    double mu[3];
    double sigma[3];
    double lnsigma[3];
    double eps[3];       // _eval_M_eps(eta, epsilon, l), l = 0,1,2
    double omeps_half[3];  // (1 - eps_l) * 0.5, the reference's evaluation order
};

constexpr double kSqrt2Pi = 2.5066282746310002;      // numpy: sqrt(2*pi)
constexpr double kLogSqrt2Pi = 0.9189385332046727;   // numpy: log(sqrt(2*pi))

constexpr int LIK_BLOCK = 256;

// One thread per (c,u) item; the block's 256 x 9 results are transposed through LDS so that the
// 72-byte records leave as fully coalesced 16-byte-per-lane stores.
__global__ __launch_bounds__(LIK_BLOCK) void lik_bt_kernel(const double *__restrict__ bt, int64_t n_items,
                                                           LikTheta th, double *__restrict__ lM,
                                                           double *__restrict__ pBt) {
    __shared__ double stage[LIK_BLOCK * 9];
    const int tid = threadIdx.x;
    const int64_t n_tiles = (n_items + LIK_BLOCK - 1) / LIK_BLOCK;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * LIK_BLOCK;
        const int64_t i = base + tid;
        if (i < n_items) {
            const double x = bt[i];
            double N[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // scipy.stats.norm.pdf: exp(-z*z/2) / sqrt(2 pi) / sigma   (fit.py:115)
                const double z = (x - th.mu[k]) / th.sigma[k];
                N[k] = exp(-(z * z) / 2.0) / kSqrt2Pi / th.sigma[k];
            }
            if (pBt) {
                pBt[i * 3 + 0] = N[0];
                pBt[i * 3 + 1] = N[1];
                pBt[i * 3 + 2] = N[2];
            }
            // js = the two other components in ascending order (fit.py:428-429)
            const double others[3] = {N[1] + N[2], N[0] + N[2], N[0] + N[1]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const double M = th.eps[l] * N[k] + th.omeps_half[l] * others[k];  // fit.py:430
                    stage[tid * 9 + k * 3 + l] = log(M);                                // fit.py:122
                }
            }
        }
        __syncthreads();
        const int64_t n_here = (n_items - base < LIK_BLOCK) ? (n_items - base) : LIK_BLOCK;
        const int64_t n_dbl = n_here * 9;
        double *dst = lM + base * 9;
        // base*9*8 bytes is a multiple of 16 (LIK_BLOCK*72), so double2 stores are aligned.
        const int64_t n_d2 = n_dbl >> 1;
        const double2 *s2 = reinterpret_cast<const double2 *>(stage);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (int64_t j = tid; j < n_d2; j += LIK_BLOCK) d2[j] = s2[j];
        if ((n_dbl & 1) && tid == 0) dst[n_dbl - 1] = stage[n_dbl - 1];
        __syncthreads();
    }
}

// 16 lanes per edge: S_B[c,k] = sum_h ( -z*z/2 - log(sqrt(2 pi)) - log(sigma_k) )  (fit.py:114, :171)
__global__ __launch_bounds__(256) void lik_b_kernel(const double *__restrict__ b, int64_t C, int H, LikTheta th,
                                                    double *__restrict__ S_B, double *__restrict__ lpB) {
    const int sub = threadIdx.x & 15;
    const int64_t c = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const double *row = b + c * H;
        for (int h = sub; h < H; h += 16) {
            const double x = row[h];
            const double z0 = (x - th.mu[0]) / th.sigma[0];
            const double z1 = (x - th.mu[1]) / th.sigma[1];
            const double z2 = (x - th.mu[2]) / th.sigma[2];
            const double l0 = -(z0 * z0) / 2.0 - kLogSqrt2Pi - th.lnsigma[0];
            const double l1 = -(z1 * z1) / 2.0 - kLogSqrt2Pi - th.lnsigma[1];
            const double l2 = -(z2 * z2) / 2.0 - kLogSqrt2Pi - th.lnsigma[2];
            if (lpB) {
                double *o = lpB + (c * H + h) * 3;
                o[0] = l0; o[1] = l1; o[2] = l2;
            }
            s0 += l0; s1 += l1; s2 += l2;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 16);
        s1 += __shfl_xor(s1, o, 16);
        s2 += __shfl_xor(s2, o, 16);
    }
    if (c < C && sub == 0) {
        S_B[c * 3 + 0] = s0;
        S_B[c * 3 + 1] = s1;
        S_B[c * 3 + 2] = s2;
    }
}

}  // namespace

extern "C" int fcd_lik_tables(fcd_ctx *ctx, const double *b, const double *bt, int64_t C, int64_t H, int64_t U,
                              const double *theta, double *S_B, double *lM, double *lp_B_g_F,
                              double *p_Bt_g_Ft, fcd_stream stream) {
    if (!ctx || !b || !bt || !theta || !S_B || !lM) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_lik_tables: null pointer");
    if (C < 1 || H < 1 || U < 1) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_lik_tables: C=%lld H=%lld must be >= 1", C, H);
    if (fcd_C_to_N(C) < 0) return fcd_fail(ctx, FCD_ERR_SHAPE, "Number of connections (%lld) must be a triangular number.", C);
    if (H > INT32_MAX || U > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_lik_tables: H/U too large");
    LikTheta th;
    const double eta = theta[1], epsilon = theta[2];
    for (int k = 0; k < 3; ++k) {
        th.mu[k] = theta[6 + k];
        th.sigma[k] = theta[9 + k];
        th.lnsigma[k] = log(theta[9 + k]);
    }
    // _eval_M_eps, fit.py:433-444 (same operation order)
    th.eps[0] = 1 - epsilon;
    th.eps[1] = epsilon;
    double e2 = eta * epsilon;
    e2 += (1 - eta) * (1 - epsilon);
    th.eps[2] = e2;
    for (int l = 0; l < 3; ++l) th.omeps_half[l] = (1 - th.eps[l]) * 0.5;

    hipStream_t s = (hipStream_t)stream;
    const int64_t n_items = C * U;
    const int64_t n_tiles = (n_items + LIK_BLOCK - 1) / LIK_BLOCK;
    int64_t grid = n_tiles;
    const int64_t cap = (int64_t)ctx->num_cu * 8;  // 8 blocks of 256 threads per CU, grid-stride the rest
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(lik_bt_kernel, dim3((unsigned)grid), dim3(LIK_BLOCK), 0, s, bt, n_items, th, lM, p_Bt_g_Ft);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(lik_b_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, b, C, (int)H, th, S_B,
                       lp_B_g_F);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
