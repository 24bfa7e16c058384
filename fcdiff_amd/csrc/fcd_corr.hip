// K_corr: per-subject region x time sample correlation -> edge-major correlations, optional Fisher z.
//
// Not in the reference (its inputs are already correlations, fcdiff/fit.py:20-23): this is the front-end that
// BASELINE.json's north_star names; oracle = numpy.corrcoef ("parity unpinned").
// Fisher z (atanh) is OFF by default: the model's Normal components live on raw correlations clipped to
// [-1, 1] (fcdiff/model.py:213, 236), so every default of the reference assumes untransformed values.
//
// numpy.corrcoef, restated:  X -= mean(X, axis=1);  c = (X X^T) * (1/(T-1));  s = sqrt(diag c);
//                            c /= s[:, None];  c /= s[None, :];  clip to [-1, 1].
// The Gram matrix is an fp64 MFMA product (v_mfma_f64_16x16x4_f64) over 64 x 64 blocks of the lower triangle with
// LDS-staged, double-buffered 64 x 16 panels.  Output out[c][s], c = n(n-1)/2 + m (n > m): the layout of b / bt.
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

// Gram blocks.  grid = subjects x lower-triangle blocks of 64 x 64 regions (dealt per XCD, see below); block = 4 waves, wave (wr, wc) owns the
// 2 x 2 MFMA tiles (2 wr + {0,1}, 2 wc + {0,1}) of the block.  K = T in steps of 16: the two 64 x 16 panels (rows of the
// block's row regions / column regions, centred on the way in) are staged in LDS -- 128 contiguous bytes per region and
// step, each element loaded once per block instead of once per tile -- double-buffered so that the loads of step k+1 fly
// while the 16 MFMAs per wave of step k run.  Panel rows are padded to 18 doubles: the fragment reads (lane = (row i,
// k-quarter q): P[16 t + i][4 g + q]) then fall on 32 distinct bank pairs.
// Fragments of v_mfma_f64_16x16x4_f64: A[i][k] in lane (i = l & 15, k = l >> 4); B[k][j] in lane (j = l & 15, k = l >> 4);
// D[i][j] comes back as 4 doubles per lane: col = l & 15, row = (l >> 4) + 4 r.
constexpr int CB = 64, CK = 16, CLD = 18;
__global__ __launch_bounds__(256) void corr_gram_kernel(const double *__restrict__ ts, const double *__restrict__ mean,
                                                        const double *__restrict__ sdev, int Nreg, int T, int64_t S,
                                                        int n_blocks, int fisher_z, int64_t C, double *__restrict__ tmp) {
    __shared__ __attribute__((aligned(16))) double pa[2][CB * CLD], pb[2][CB * CLD];
    // Workgroup -> (subject, block): consecutive workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 names the
    // XCD class), so ALL blocks of a subject take the same class: the subject's rows (1.9 MB at cfg3, 3.8 MB at cfg5) are
    // fetched into ONE 4 MB L2 and re-read from there by its other blocks, instead of once per XCD (a speed hint only).
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int t = jj % n_blocks;                        // block index -> (I, J), I >= J, lower-triangular row-major
    const int64_t s = (int64_t)(jj / n_blocks) * 8 + xcd;
    if (s >= S) return;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)I * (I + 1) / 2 > t) --I;
    while ((int64_t)(I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const bool diag = I == J;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    // staging: thread -> (row r of the panel, 16-byte piece p of its 128 bytes), rows r and r + 32: eight lanes cover one
    // whole 128-byte line per load instruction (T even: rows are 16-byte aligned; odd T takes the 8-byte path)
    const int prow = tid >> 3, pp = tid & 7;
    const bool even = (T & 1) == 0;
    int ga[2], gb[2];
    bool va[2], vb[2];
    const double *xa[2], *xb[2];
    double ma[2], mb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ga[h] = I * CB + prow + 32 * h;
        gb[h] = J * CB + prow + 32 * h;
        va[h] = ga[h] < Nreg;
        vb[h] = gb[h] < Nreg && !diag;
        xa[h] = ts + (s * Nreg + (va[h] ? ga[h] : 0)) * T;
        xb[h] = ts + (s * Nreg + (vb[h] ? gb[h] : 0)) * T;
        ma[h] = va[h] ? mean[s * Nreg + ga[h]] : 0.0;
        mb[h] = vb[h] ? mean[s * Nreg + gb[h]] : 0.0;
    }
    double2 ra[2], rb[2];
    auto fetch = [&](int k0) {
        const int k = k0 + 2 * pp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 a, b;
            if (even) {
                const int kc = k < T ? k : T - 2;              // clamped: no branch around the load
                a = *reinterpret_cast<const double2 *>(xa[h] + kc);
                b = *reinterpret_cast<const double2 *>(xb[h] + kc);
            } else {
                const int k0c = k < T ? k : T - 1, k1c = k + 1 < T ? k + 1 : T - 1;
                a = make_double2(xa[h][k0c], xa[h][k1c]);
                b = make_double2(xb[h][k0c], xb[h][k1c]);
            }
            ra[h].x = (va[h] && k < T) ? a.x - ma[h] : 0.0;
            ra[h].y = (va[h] && k + 1 < T) ? a.y - ma[h] : 0.0;
            rb[h].x = (vb[h] && k < T) ? b.x - mb[h] : 0.0;
            rb[h].y = (vb[h] && k + 1 < T) ? b.y - mb[h] : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<double2 *>(&pa[buf][(prow + 32 * h) * CLD + 2 * pp]) = ra[h];
            if (!diag) *reinterpret_cast<double2 *>(&pb[buf][(prow + 32 * h) * CLD + 2 * pp]) = rb[h];
        }
    };
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
    // tiles of this wave that hold anything: rows below Nreg, and (diagonal block) not strictly above the diagonal
    const int i16 = l & 15, kq = l >> 4;
    bool on[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tr = 2 * wr + a, tc = 2 * wc + b;
            on[a][b] = (I * CB + tr * 16 < Nreg) && (J * CB + tc * 16 < Nreg) && (!diag || tc <= tr);
        }
    const bool any_on = on[0][0] || on[0][1] || on[1][0] || on[1][1];
    fetch(0);
    put(0);
    __syncthreads();
    const int steps = (T + CK - 1) / CK;
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        if (st + 1 < steps) fetch((st + 1) * CK);
        if (any_on) {
            const double *A = pa[cur];
            const double *B = diag ? pa[cur] : pb[cur];
#pragma unroll
            for (int g = 0; g < CK / 4; ++g) {
                double fa[2], fb[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = A[((2 * wr + a) * 16 + i16) * CLD + g * 4 + kq];
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[b] = B[((2 * wc + b) * 16 + i16) * CLD + g * 4 + kq];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (on[a][b]) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (st + 1 < steps) put(cur ^ 1);
        __syncthreads();
    }
    const double inv = 1.0 / (double)(T - 1);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!on[a][b]) continue;
            const int m = J * CB + (2 * wc + b) * 16 + i16;                  // column = the smaller region index of the pair
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = I * CB + (2 * wr + a) * 16 + kq + 4 * r;       // row
                if (n < Nreg && m < n) {
                    double c = acc[a][b][r] * inv;
                    c /= sdev[s * Nreg + n];
                    c /= sdev[s * Nreg + m];
                    c = (c != c) ? c : fmin(fmax(c, -1.0), 1.0);      // (a constant series: 0 / 0 = NaN like numpy.corrcoef; fmin / fmax would drop it)
                    if (fisher_z) c = atanh(c);
                    tmp[s * C + (fcd_tri(n) + m)] = c;                       // 16 lanes = 16 consecutive edges = 128 bytes
                }
            }
        }
}

// tmp (S, C) -> out (C, S): the layout of b / bt (edge-major, subjects fastest).  The Gram kernel writes subject-major
// rows (128 contiguous bytes per 16 lanes); written straight into (C, S) every value would be an 8-byte store 8 S bytes
// from the next one -- 40 M partial-line writes at cfg5.  32 x 32 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void corr_transpose_kernel(const double *__restrict__ tmp, int64_t S, int64_t C,
                                                             double *__restrict__ out) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, s0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8 threads
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t s = s0 + ty + j, c = c0 + tx;
        if (s < S && c < C) tile[ty + j][tx] = tmp[s * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t c = c0 + ty + j, s = s0 + tx;
        if (s < S && c < C) out[c * S + s] = tile[tx][ty + j];
    }
}

}  // namespace

extern "C" int fcd_corr_edges(fcd_ctx *ctx, const double *ts, int64_t S, int64_t Nreg, int64_t T, int fisher_z, double *out,
                              fcd_stream stream) {
    if (!ctx || !ts || !out) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_corr_edges: null pointer");
    if (S < 1 || Nreg < 2 || T < 2) return fcd_fail(ctx, FCD_ERR_SHAPE, "need S >= 1, Nreg >= 2, T >= 2 (Nreg=%lld, T=%lld)", Nreg, T);
    if (S > 65535 || Nreg > 46340 || T > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: shape too large");
    const int64_t rows = S * Nreg, C = fcd_tri(Nreg);
    // workspace: means and deviations of the rows, then the subject-major copy of the result (grows the context's
    // scratch, which synchronises, the first time a shape needs it)
    int rc = fcd_ws_reserve(ctx, ((size_t)rows * 2 + (size_t)S * C) * sizeof(double));
    if (rc) return rc;
    double *mean = (double *)ctx->ws, *sdev = mean + rows, *tmp = sdev + rows;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_moments_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ts, rows, (int)T, mean, sdev);
    FCD_LAUNCH_CHECK();
    const int64_t nb = (Nreg + CB - 1) / CB;
    const int64_t blocks = nb * (nb + 1) / 2;
    const int64_t grid = (S + 7) / 8 * 8 * blocks;
    if (grid > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: S * blocks too large");
    hipLaunchKernelGGL(corr_gram_kernel, dim3((unsigned)grid), dim3(256), 0, s, ts, mean, sdev, (int)Nreg, (int)T, S, (int)blocks,
                       fisher_z ? 1 : 0, C, tmp);
    FCD_LAUNCH_CHECK();
    if ((C + 31) / 32 > INT32_MAX || (S + 31) / 32 > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: grid too large");
    hipLaunchKernelGGL(corr_transpose_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((S + 31) / 32)), dim3(256), 0, s, tmp, S, C, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
