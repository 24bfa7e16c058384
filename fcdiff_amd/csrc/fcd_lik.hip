// K_lik: the per-edge likelihood tables of UnsharedRegionFit._update_lps (fcdiff/fit.py:104-122)
// with _eval_M / _eval_M_eps (fit.py:409-444) fused in.
//
// Streaming: reads b (C,H) and bt (C,U) once, writes S_B (C,3) and lM (C,U,3,3) once, in ONE launch.
// Algorithmic bytes per call = 8*C*(H+U) + 24*C + 72*C*U  (SURVEY.md section 8d).
//
// The (C,H,3) table of the reference is never consumed except through its H-sum (fit.py:171, :472),
// so only S_B[c,k] = sum_h lp_B_g_F[c,h,k] is produced unless the caller asks for the full table.
#include "fcd_common.h"
#include "fcd_fastmath.h"

namespace {

struct LikTheta {
    double mu[3];
    double sigma[3];
    double inv_sigma[3];   // 1 / sigma_k
    double pdf_scale[3];   // 1 / sqrt(2 pi) / sigma_k  (the reference's two divisions applied to 1.0)
    double lnsigma[3];
    double eps[3];       // _eval_M_eps(eta, epsilon, l), l = 0,1,2
    double omeps_half[3];  // (1 - eps_l) * 0.5, the reference's evaluation order
    double cmin;           // min over l of min(eps_l, omeps_half_l): every M_kl >= cmin * (N_0 + N_1 + N_2)
};

constexpr double kSqrt2Pi = 2.5066282746310002;      // numpy: sqrt(2*pi)
constexpr double kLogSqrt2Pi = 0.9189385332046727;   // numpy: log(sqrt(2*pi))

constexpr int LIK_BLOCK = 256;

// ---------------------------------------------------------------------------------------------
// exp and log of this kernel: fcd_fastmath.h.  3 of the ~12 transcendental calls per (edge, patient) item are
// exponentials of -z*z/2 <= 0 and 9 are logarithms of mixture densities; with ocml's exp and log (~35 and ~70 fp64
// instructions) the kernel is ALU-bound far below the memory system.  fcd_exp_neg (64-entry table of 2^(-j/64) + a
// degree-6 polynomial, <= 1.0 ulp) and fcd_log_normal (512-entry table {1/m_i, log m_i} + a degree-7 polynomial, <= 1.3
// ulp, no special cases) take ~18 and ~22.  The logs' special cases are decided ONCE per item: all nine M_kl are
// >= cmin * (N_0 + N_1 + N_2), so one comparison shows them positive, finite and normal; the rare item at the edge of
// the double range goes through ocml's log, which gives -inf for an underflowed mixture density exactly like the
// reference's np.log (fit.py:115, 121-122).  Both tables live in LDS.
// ---------------------------------------------------------------------------------------------
struct LikTabs {
    double exp_tab[FCD_EXP_CELLS];
    fcd_log_cell log_tab[FCD_LOG_CELLS];
};

// One launch, two kinds of blocks.
// Blocks [0, n_bt_blocks): one thread per (c,u) item; the block's 256 x 9 results are transposed through LDS so that
//   the 72-byte records leave as fully coalesced 16-byte-per-lane stores (grid-stride over tiles of 256 items).
// Blocks [n_bt_blocks, ...): 16 lanes per edge, S_B[c,k] = sum_h ( -z*z/2 - log(sqrt(2 pi)) - log(sigma_k) )  (fit.py:114, :171)
__global__ __launch_bounds__(LIK_BLOCK) void lik_kernel(const double *__restrict__ bt, int64_t n_items, LikTheta th,
                                                        const LikTabs *__restrict__ tabs, double *__restrict__ lM,
                                                        double *__restrict__ pBt, int n_bt_blocks,
                                                        const double *__restrict__ b, int64_t C, int H,
                                                        double *__restrict__ S_B, double *__restrict__ lpB) {
    __shared__ double stage[LIK_BLOCK * 9];
    __shared__ __attribute__((aligned(16))) fcd_log_cell ltab[FCD_LOG_CELLS];
    __shared__ double etab[FCD_EXP_CELLS];
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
    for (int t = tid; t < FCD_LOG_CELLS; t += LIK_BLOCK) ltab[t] = tabs->log_tab[t];
    if (tid < FCD_EXP_CELLS) etab[tid] = tabs->exp_tab[tid];
    __syncthreads();
    const int64_t n_tiles = (n_items + LIK_BLOCK - 1) / LIK_BLOCK;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += n_bt_blocks) {
        const int64_t base = tile * LIK_BLOCK;
        const int64_t i = base + tid;
        if (i < n_items) {
            const double x = bt[i];        // (a non-temporal load here: 27.6 against 24.4 us at cfg3, 425 against 434 us at cfg5)
            double N[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // scipy.stats.norm.pdf: exp(-z*z/2) / sqrt(2 pi) / sigma   (fit.py:115).  The kernel is ALU-bound and
                // an fp64 division is ~15 instructions, so the three divisions per component are multiplications by
                // host-side reciprocals (<= 2 ulp of the density, ~1e-16 relative in log M) ...
                const double d = x - th.mu[k];
                const double z = d * th.inv_sigma[k];
                N[k] = fcd_exp_neg((z * z) / 2.0, etab) * th.pdf_scale[k];
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
            const double floor_M = th.cmin * (N[0] + others[0]);
            if (floor_M >= 4.5e-308 && floor_M < __builtin_inf()) {
                // every M_kl is positive, finite and normal: the branch-free log
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const double M = th.eps[l] * N[k] + th.omeps_half[l] * others[k];  // fit.py:430
                        stage[tid * 9 + k * 3 + l] = fcd_log_normal(M, ltab);              // fit.py:122
                    }
                }
            } else {
                // edge of the double range (or eps in {0, 1}): the general log, -inf for an underflowed density
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const double M = th.eps[l] * N[k] + th.omeps_half[l] * others[k];
                        stage[tid * 9 + k * 3 + l] = log(M);
                    }
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
        {
            // non-temporal: the table is written once and read by other kernels much later -- keeping it out of the L2 on
            // the way out is worth 8 % of the launch (26.6 -> 24.4 us at cfg3, 455 -> 435 us at cfg5)
            typedef double d2v __attribute__((ext_vector_type(2)));
            for (int64_t j = tid; j < n_d2; j += LIK_BLOCK)
                __builtin_nontemporal_store(*reinterpret_cast<const d2v *>(&s2[j]), reinterpret_cast<d2v *>(&d2[j]));
        }
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
    th.cmin = th.eps[0];
    for (int l = 0; l < 3; ++l) {
        if (!(th.eps[l] >= th.cmin)) th.cmin = th.eps[l];
        if (!(th.omeps_half[l] >= th.cmin)) th.cmin = th.omeps_half[l];
    }
    if (!(th.cmin > 0.0)) th.cmin = 0.0;          // eps outside (0, 1): no floor, every item takes the general log

    hipStream_t s = (hipStream_t)stream;
    const int64_t n_items = C * U;
    const int64_t n_tiles = (n_items + LIK_BLOCK - 1) / LIK_BLOCK;
    int64_t grid = n_tiles;                          // one tile per block up to 16 blocks per CU, grid-stride beyond
    const int64_t cap = (int64_t)ctx->num_cu * 16;   // (5 per CU, all resident, each copying the tables once: the same 26 us
    if (grid > cap) grid = cap;                      //  at cfg3 and 4 % slower at cfg5 -- queued blocks even out the tail)
    const int64_t n_b_blocks = (C + 15) / 16;
    fcd_prof_begin(ctx, FCD_PROF_LIK, s);
    hipLaunchKernelGGL(lik_kernel, dim3((unsigned)(grid + n_b_blocks)), dim3(LIK_BLOCK), 0, s, bt, n_items, th,
                       reinterpret_cast<const LikTabs *>(ctx->log_tab), lM, p_Bt_g_Ft, (int)grid, b, C, (int)H, S_B, lp_B_g_F);
    fcd_prof_end(ctx, FCD_PROF_LIK, s);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
