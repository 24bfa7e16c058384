// K_lik: the per-edge likelihood tables of UnsharedRegionFit._update_lps (fcdiff/fit.py:104-122)
// with _eval_M / _eval_M_eps (fit.py:409-444) fused in.
//
// Streaming: reads b (C,H) and bt (C,U) once, writes S_B (C,3) and lM (C,U,3,3) once, in ONE launch.
// Algorithmic bytes per call = 8*C*(H+U) + 24*C + 72*C*U  (SURVEY.md section 8d).
//
// The (C,H,3) table of the reference is never consumed except through its H-sum (fit.py:171, :472),
// so only S_B[c,k] = sum_h lp_B_g_F[c,h,k] is produced unless the caller asks for the full table.
#include "fcd_common.h"

namespace {

struct LikTheta {
    double mu[3];
    double sigma[3];
    double inv_sigma[3];   // 1 / sigma_k
    double pdf_scale[3];   // 1 / sqrt(2 pi) / sigma_k  (the reference's two divisions applied to 1.0)
    double lnsigma[3];
    double eps[3];       // _eval_M_eps(eta, epsilon, l), l = 0,1,2
    double omeps_half[3];  // (1 - eps_l) * 0.5, the reference's evaluation order
};

constexpr double kSqrt2Pi = 2.5066282746310002;      // numpy: sqrt(2*pi)
constexpr double kLogSqrt2Pi = 0.9189385332046727;   // numpy: log(sqrt(2*pi))

constexpr int LIK_BLOCK = 256;

// ---------------------------------------------------------------------------------------------
// log for this kernel.  9 of the ~12 transcendental calls per (edge, patient) item are logs and the kernel
// is ALU-bound on them (the ocml log is ~70 fp64 instructions), so a table-driven one (Tang 1990 style):
//   x = 2^e * m, m in [0.75, 1.5);  i = top 6 mantissa bits;  r = fma(m, 1/m_i, -1), |r| <= 2^-6;
//   log x = e*ln2 + log(m_i) + log1p(r),   log1p(r) by a degree-10 Taylor polynomial.
// m_i = 1 for the two cells around 1, so there is no cancellation for x near 1.  Error <= ~1.5 ulp over the
// whole positive range (subnormals included); log(0) = -inf as the reference's np.log gives for an
// underflowed mixture density (fit.py:115, 121-122).  The 64-entry table {1/m_i, log m_i} lives in LDS.
// ---------------------------------------------------------------------------------------------
struct LogTab {
    double inv[64];
    double lg[64];
};

__device__ inline double fast_log(double x, const double2 *__restrict__ tab) {
    if (!(x > 0.0)) return (x == 0.0) ? -__builtin_inf() : __builtin_nan("");
    int eadj = 0;
    if (x < 2.2250738585072014e-308) {   // subnormal: scale by 2^54
        x *= 18014398509481984.0;
        eadj = -54;
    }
    if (x == __builtin_inf()) return x;
    const uint64_t bits = (uint64_t)__double_as_longlong(x);
    const int idx = (int)((bits >> 46) & 63);            // top 6 mantissa bits
    int e = (int)((bits >> 52) & 0x7FF) - 1023 + eadj;
    // mantissa in [1, 2); cells 32..63 (m >= 1.5) are halved into [0.75, 1) and the exponent bumped
    uint64_t mb = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    if (idx >= 32) {
        mb -= 0x0010000000000000ull;
        e += 1;
    }
    const double m = __longlong_as_double((long long)mb);
    const double2 t = tab[idx];
    const double r = fma(m, t.x, -1.0);
    // log1p(r) = r - r^2/2 + r^3/3 - ... - r^10/10
    double p = -0.1;
    p = fma(p, r, 1.0 / 9.0);
    p = fma(p, r, -0.125);
    p = fma(p, r, 1.0 / 7.0);
    p = fma(p, r, -1.0 / 6.0);
    p = fma(p, r, 0.2);
    p = fma(p, r, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    const double r2 = r * r;
    const double de = (double)e;
    const double hi = fma(de, 6.93147180369123816490e-01, t.y);        // e*ln2_hi + log m_i (ln2_hi has 32 trailing zero bits)
    const double lo = fma(de, 1.90821492927058770002e-10, fma(p, r2, r));   // e*ln2_lo + log1p(r)
    return hi + lo;
}

// One launch, two kinds of blocks.
// Blocks [0, n_bt_blocks): one thread per (c,u) item; the block's 256 x 9 results are transposed through LDS so that
//   the 72-byte records leave as fully coalesced 16-byte-per-lane stores (grid-stride over tiles of 256 items).
// Blocks [n_bt_blocks, ...): 16 lanes per edge, S_B[c,k] = sum_h ( -z*z/2 - log(sqrt(2 pi)) - log(sigma_k) )  (fit.py:114, :171)
__global__ __launch_bounds__(LIK_BLOCK) void lik_kernel(const double *__restrict__ bt, int64_t n_items, LikTheta th,
                                                        const LogTab *__restrict__ logtab, double *__restrict__ lM,
                                                        double *__restrict__ pBt, int n_bt_blocks,
                                                        const double *__restrict__ b, int64_t C, int H,
                                                        double *__restrict__ S_B, double *__restrict__ lpB) {
    __shared__ double stage[LIK_BLOCK * 9];
    __shared__ double2 tab[64];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n_bt_blocks) {
        const int sub = tid & 15;
        const int64_t c = (int64_t)(blockIdx.x - n_bt_blocks) * 16 + (tid >> 4);
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
        return;
    }
    if (tid < 64) tab[tid] = make_double2(logtab->inv[tid], logtab->lg[tid]);
    __syncthreads();
    const int64_t n_tiles = (n_items + LIK_BLOCK - 1) / LIK_BLOCK;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += n_bt_blocks) {
        const int64_t base = tile * LIK_BLOCK;
        const int64_t i = base + tid;
        if (i < n_items) {
            const double x = bt[i];
            double N[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // scipy.stats.norm.pdf: exp(-z*z/2) / sqrt(2 pi) / sigma   (fit.py:115).  The kernel is ALU-bound and
                // an fp64 division is ~15 instructions, so the three divisions per component are multiplications by
                // host-side reciprocals (<= 2 ulp of the density, ~1e-16 relative in log M) ...
                const double d = x - th.mu[k];
                const double z = d * th.inv_sigma[k];
                N[k] = exp(-(z * z) / 2.0) * th.pdf_scale[k];
                // ... except at the edge of the double range, where the reference's own roundings decide whether the
                // density is 0 (lM = -inf) or a subnormal: there its operations are redone exactly (rare branch)
                if (N[k] < 1e-290) {
                    const double ze = d / th.sigma[k];
                    N[k] = exp(-(ze * ze) / 2.0) / kSqrt2Pi / th.sigma[k];
                }
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
                    stage[tid * 9 + k * 3 + l] = fast_log(M, tab);                     // fit.py:122
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
        th.inv_sigma[k] = 1.0 / theta[9 + k];
        th.pdf_scale[k] = 1.0 / kSqrt2Pi / theta[9 + k];
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
    int64_t grid = n_tiles;                          // one tile per block up to 16 blocks per CU, grid-stride beyond
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (grid > cap) grid = cap;
    const int64_t n_b_blocks = (C + 15) / 16;
    fcd_prof_begin(ctx, FCD_PROF_LIK, s);
    hipLaunchKernelGGL(lik_kernel, dim3((unsigned)(grid + n_b_blocks)), dim3(LIK_BLOCK), 0, s, bt, n_items, th,
                       reinterpret_cast<const LogTab *>(ctx->log_tab), lM, p_Bt_g_Ft, (int)grid, b, C, (int)H, S_B, lp_B_g_F);
    fcd_prof_end(ctx, FCD_PROF_LIK, s);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
