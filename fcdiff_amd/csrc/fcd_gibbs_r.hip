// r step of the collapsed Gibbs sampler: redraw every r_nu given f, regions in order 0..Nreg-1.
// Conditional = fcdiff/fit.py:187-194 at one-hot q_F, q_R:
//   s0 = ln(1-pi) + sum_{m != n} lM[c, u, f_c, r_mu ? 2 : 0],  s1 = ln pi + sum_{m != n} lM[c, u, f_c, r_mu ? 1 : 2]
//
// The scan over n is a Gauss-Seidel sweep: r_n sees the NEW r_m for m < n and the OLD r_m for m > n.
// It is organised like a blocked forward substitution.  Regions are cut into blocks of R_NB; for a block
// B every term with m outside B is already decided when B starts (new below B, old above B), so
//   * the PANEL kernel computes, for every n in B, the sum over all m outside B -- fully parallel over
//     (n, patient, chain); it streams the region-major table rows lMr[u][n][:] (contiguous, staged in LDS
//     and shared by all chain words of the workgroup), so the table is read once per pass;
//   * the DIAGONAL kernel walks the R_NB regions of B in order for each (patient, chain word), adding the
//     few within-block terms from an LDS copy of the diagonal tile and drawing r_n; one wave per
//     (patient, chain word), no barriers: the 64 chains' new r_n is a ballot that stays in SGPRs.
// 2 * ceil(Nreg / R_NB) launches per pass; >98 % of the arithmetic is in the panel kernels.
//
// lMr (U, Nreg, Nreg, 3, 3) is a region-major re-layout of lM made once per table build:
// lMr[u][n][m] = lM[edge(n, m)][u] with edge() the SAME ordered-pair edge id the reference uses
// (fit.py:186: nm_to_c(n, m) for every ordered pair in 'reference' mode), so the quirk is baked in there.
#include "fcd_common.h"

namespace {

constexpr int R_NB = 16;   // regions per diagonal block (even: both halves of a Philox block stay inside)

// ---------------------------------------------------------------------------------------------
// region-major re-layout: one thread per double of lMr
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void region_tables_kernel(const double *__restrict__ lM, int Nreg, int U, int mode,
                                                            double *__restrict__ lMr) {
    const int64_t total = (int64_t)U * Nreg * Nreg * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % 9);
        const int64_t rec = i / 9;
        const int m = (int)(rec % Nreg);
        const int n = (int)((rec / Nreg) % Nreg);
        const int u = (int)(rec / ((int64_t)Nreg * Nreg));
        double v = 0.0;
        if (m != n) v = lM[(fcd_pair_to_edge(n, m, mode) * U + u) * 9 + j];
        lMr[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// panel kernel.  grid = (regions of the block, patient chunks of UB, groups of chain words);
// block = 64 * (chain words per group).  LDS: UB rows of Nreg*72 bytes.
// P[((w*U + u)*R_NB + i)*2 + j][lane], i = n - B0.
// ---------------------------------------------------------------------------------------------
constexpr int P_MC = 4;   // regions m per unrolled chunk
template <int UB>
__global__ __launch_bounds__(1024) void gibbs_r_panel(const double *__restrict__ lMr, const uint8_t *__restrict__ f_state,
                                                      const uint64_t *__restrict__ r_bits, double *__restrict__ P, int Nreg,
                                                      int U, int64_t C, int GW, int B0, int nb, int mode) {
    extern __shared__ double rows[];   // [UB][Nreg*9]
    const int n = B0 + blockIdx.x;
    const int u0 = blockIdx.y * UB;
    const int nu = (U - u0 < UB) ? (U - u0) : UB;
    const int row_dbl = Nreg * 9;
    for (int u = 0; u < nu; ++u) {
        const double *src = lMr + ((int64_t)(u0 + u) * Nreg + n) * row_dbl;
        for (int i = threadIdx.x; i < row_dbl; i += blockDim.x) rows[u * row_dbl + i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.z * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= GW) return;
    const uint8_t *__restrict__ fw = f_state + (int64_t)w * C * 64 + lane;
    const uint64_t *__restrict__ rw = r_bits + (int64_t)w * Nreg * U + u0;
    const char *rb = reinterpret_cast<const char *>(rows);
    const int row_bytes = row_dbl * 8;
    double s0[UB], s1[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) s0[u] = s1[u] = 0.0;

    // one region m: the chain's f_c picks the k row, the 64 chains' r_mu (a scalar mask) the two columns
    auto term = [&](int m) {
        const int64_t c = fcd_pair_to_edge(n, m, mode);
        const uint32_t kb = (uint32_t)fw[c * 64] * 24u + (uint32_t)m * 72u;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int uu = u < nu ? u : nu - 1;            // tail chunk: recompute the last patient, never stored
            const uint64_t mk = rw[(int64_t)m * U + uu];
            const uint32_t o0 = fcd_sel_mask(0u, 16u, mk);   // r_m = 0 -> lM[k,0], r_m = 1 -> lM[k,2]
            const uint32_t o1 = fcd_sel_mask(16u, 8u, mk);   // r_m = 0 -> lM[k,2], r_m = 1 -> lM[k,1]
            const char *base = rb + uu * row_bytes + kb;
            s0[u] += *reinterpret_cast<const double *>(base + o0);
            s1[u] += *reinterpret_cast<const double *>(base + o1);
        }
    };
    // m outside the block [B0, B0 + nb): two plain ranges, chunks of P_MC with a fixed trip count
    int m = 0;
    for (; m + P_MC <= B0; m += P_MC) {
#pragma unroll
        for (int j = 0; j < P_MC; ++j) term(m + j);
    }
    for (; m < B0; ++m) term(m);
    m = B0 + nb;
    for (; m + P_MC <= Nreg; m += P_MC) {
#pragma unroll
        for (int j = 0; j < P_MC; ++j) term(m + j);
    }
    for (; m < Nreg; ++m) term(m);

    const int i = n - B0;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (u < nu) {
            double *o = P + ((((int64_t)w * U + u0 + u) * R_NB + i) * 2) * 64 + lane;
            o[0] = s0[u];
            o[64] = s1[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// diagonal kernel.  grid = (U, groups of chain words); block = 64 * (words per group); one wave per
// (patient, chain word).  LDS: the diagonal tile lMr[u][B0+i][B0+j] (R_NB*R_NB*72 B, shared) and, per wave,
// the f bytes of the within-block pairs.
// ---------------------------------------------------------------------------------------------
constexpr int D_WPB = 2;   // 16 KiB of f bytes per wave + the 18 KiB tile
__global__ __launch_bounds__(64 * D_WPB) void gibbs_r_diag(const double *__restrict__ lMr, const double *__restrict__ hyper,
                                                           const uint8_t *__restrict__ f_state, uint64_t *__restrict__ r_bits,
                                                           const double *__restrict__ P, int Nreg, int U, int64_t C, int GW,
                                                           int B0, int nb, int mode, uint32_t chain0, uint64_t seed,
                                                           uint32_t sweep) {
    __shared__ double tile[R_NB * R_NB * 9];
    __shared__ uint8_t fb[D_WPB][R_NB * R_NB][64];
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * D_WPB + wave));
    // diagonal tile of patient u: rows B0+i, columns B0 .. B0+nb-1
    for (int t = threadIdx.x; t < nb * nb * 9; t += blockDim.x) {
        const int j9 = t % (nb * 9), i = t / (nb * 9);
        tile[i * R_NB * 9 + j9] = lMr[(((int64_t)u * Nreg + B0 + i) * Nreg + B0) * 9 + j9];
    }
    if (w < GW) {
        const uint8_t *__restrict__ fw = f_state + (int64_t)w * C * 64 + lane;
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < nb; ++j)
                if (i != j) fb[wave][i * R_NB + j][lane] = fw[fcd_pair_to_edge(B0 + i, B0 + j, mode) * 64];
    }
    __syncthreads();
    if (w >= GW) return;

    uint64_t *__restrict__ rw = r_bits + (int64_t)w * Nreg * U + u;
    uint64_t cur[R_NB];                       // r of the block's regions for the 64 chains: old, then new
#pragma unroll
    for (int i = 0; i < R_NB; ++i) cur[i] = (i < nb) ? rw[(int64_t)(B0 + i) * U] : 0ull;
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double *__restrict__ Pw = P + (((int64_t)w * U + u) * R_NB * 2) * 64 + lane;
    const char *tb = reinterpret_cast<const char *>(tile);
    fcd_u4 rnd = {0, 0, 0, 0};

#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        if (i < nb) {
            const int n = B0 + i;
            double s0 = Pw[(i * 2 + 0) * 64], s1 = Pw[(i * 2 + 1) * 64];
#pragma unroll
            for (int j = 0; j < R_NB; ++j) {
                if (j != i && j < nb) {
                    const uint32_t kb = (uint32_t)fb[wave][i * R_NB + j][lane] * 24u + (uint32_t)((i * R_NB + j) * 72);
                    const uint32_t o0 = fcd_sel_mask(0u, 16u, cur[j]);
                    const uint32_t o1 = fcd_sel_mask(16u, 8u, cur[j]);
                    s0 += *reinterpret_cast<const double *>(tb + kb + o0);
                    s1 += *reinterpret_cast<const double *>(tb + kb + o1);
                }
            }
            if ((n & 1) == 0) rnd = fcd_philox((uint32_t)((n >> 1) * U + u), chain, sweep, FCD_KIND_R, k0, k1);
            const double x = (n & 1) ? fcd_u53(rnd.z, rnd.w) : fcd_u53(rnd.x, rnd.y);
            cur[i] = __ballot(fcd_draw_r(lnpi0 + s0, lnpi1 + s1, x));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < R_NB; ++i)
            if (i < nb) rw[(int64_t)(B0 + i) * U] = cur[i];
    }
}

// ---------------------------------------------------------------------------------------------
// generic fallback (no region-major table, or a row that does not fit the LDS): one block per
// (patient, chain word), regions strictly in order, the four waves split the sum over m, direct gathers.
// ---------------------------------------------------------------------------------------------
constexpr int R_WAVES = 4;
__global__ __launch_bounds__(64 * R_WAVES) void gibbs_r_simple(const double *__restrict__ lM, const double *__restrict__ hyper,
                                                               const uint8_t *__restrict__ f_state,
                                                               uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                               uint32_t chain0, uint64_t seed, uint32_t sweep, int mode) {
    extern __shared__ uint64_t sh_r[];
    uint64_t *mask = sh_r;                                         // [Nreg]
    double *part = reinterpret_cast<double *>(sh_r + Nreg);        // [R_WAVES][2][64]
    const int u = blockIdx.x, w = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t *__restrict__ rcol = r_bits + (int64_t)w * Nreg * U + u;
    for (int n = tid; n < Nreg; n += 64 * R_WAVES) mask[n] = rcol[(int64_t)n * U];
    __syncthreads();
    const uint8_t *__restrict__ fw = f_state + (int64_t)w * C * 64 + lane;
    const double *__restrict__ lMu = lM + (int64_t)u * 9;
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    fcd_u4 rnd = {0, 0, 0, 0};

    for (int n = 0; n < Nreg; ++n) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
        for (int m = wave; m < Nreg; m += R_WAVES) {
            const bool valid = (m != n);
            const int64_t c = valid ? fcd_pair_to_edge(n, m, mode) : 0;
            const int k = fw[c * 64];
            const uint32_t bit = (uint32_t)((mask[m] >> lane) & 1ull);
            const double *p = lMu + (c * U) * 9 + k * 3;
            const double v0 = p[bit * 2];      // r_m = 0: lM[k,0];  r_m = 1: lM[k,2]      fit.py:188-190
            const double v1 = p[2 - bit];      // r_m = 0: lM[k,2];  r_m = 1: lM[k,1]      fit.py:192-194
            s0 += valid ? v0 : 0.0;
            s1 += valid ? v1 : 0.0;
        }
        part[(wave * 2 + 0) * 64 + lane] = s0;
        part[(wave * 2 + 1) * 64 + lane] = s1;
        __syncthreads();
        if (wave == 0) {
            double t0 = part[0 * 64 + lane], t1 = part[1 * 64 + lane];
#pragma unroll
            for (int j = 1; j < R_WAVES; ++j) {
                t0 += part[(j * 2 + 0) * 64 + lane];
                t1 += part[(j * 2 + 1) * 64 + lane];
            }
            if ((n & 1) == 0) rnd = fcd_philox((uint32_t)((n >> 1) * U + u), chain, sweep, FCD_KIND_R, k0, k1);
            const double x = (n & 1) ? fcd_u53(rnd.z, rnd.w) : fcd_u53(rnd.x, rnd.y);
            const uint64_t ball = __ballot(fcd_draw_r(lnpi0 + t0, lnpi1 + t1, x));
            if (lane == 0) mask[n] = ball;
        }
        __syncthreads();
    }
    for (int n = tid; n < Nreg; n += 64 * R_WAVES) rcol[(int64_t)n * U] = mask[n];
}

template <int UB>
int launch_panel(const double *lMr, const uint8_t *f_state, const uint64_t *r_bits, double *P, int64_t Nreg, int64_t U,
                 const fcd_geo &g, int B0, int nb, int mode, hipStream_t s) {
    const int wpb = g.GW < 16 ? g.GW : 16;
    const size_t shmem = (size_t)UB * Nreg * 72;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gibbs_r_panel<UB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)nb, (unsigned)((U + UB - 1) / UB), (unsigned)((g.GW + wpb - 1) / wpb));
    hipLaunchKernelGGL(gibbs_r_panel<UB>, grid, dim3(64 * wpb), shmem, s, lMr, f_state, r_bits, P, (int)Nreg, (int)U, g.C,
                       g.GW, B0, nb, mode);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

}  // namespace

extern "C" int fcd_gibbs_region_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, int edge_mode, double *lMr,
                                       fcd_stream stream) {
    if (!ctx || !lM || !lMr) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: null pointer");
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    const int64_t total = U * Nreg * Nreg * 9;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(region_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lM, (int)Nreg, (int)U,
                       edge_mode, lMr);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_r_step(fcd_ctx *ctx, const double *lM, const double *lMr, const double *hyper,
                                const uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                                int64_t chain0, uint64_t seed, int64_t sweep, int edge_mode, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, chain0, g);
    if (rc) return rc;
    if (!lM || !hyper || !f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_r_step: null pointer");
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_r_step: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = (size_t)Nreg * 72;
    if (!lMr || row_bytes > 160 * 1024 || U > 65535) {
        // generic path: direct gathers from the edge-major table
        const size_t shmem = (size_t)Nreg * 8 + (size_t)R_WAVES * 2 * 64 * 8;
        if (shmem > 64 * 1024) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: Nreg=%lld exceeds the LDS mask array", Nreg);
        if (U > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: U=%lld exceeds the grid", U);
        hipLaunchKernelGGL(gibbs_r_simple, dim3((unsigned)U, (unsigned)g.GW), dim3(64 * R_WAVES), shmem, s, lM, hyper, f_state,
                           r_bits, (int)Nreg, (int)U, g.C, (uint32_t)chain0, seed, (uint32_t)sweep, edge_mode);
        FCD_LAUNCH_CHECK();
        return FCD_OK;
    }
    // blocked path
    const size_t p_bytes = (size_t)g.GW * U * R_NB * 2 * 64 * sizeof(double);
    rc = fcd_ws_reserve(ctx, p_bytes);
    if (rc) return rc;
    double *P = (double *)ctx->ws;
    int ub = 1;
    while (ub < 4 && (size_t)(ub * 2) * row_bytes <= 60 * 1024 && ub * 2 <= U) ub *= 2;
    for (int B0 = 0; B0 < Nreg; B0 += R_NB) {
        const int nb = (Nreg - B0 < R_NB) ? (int)(Nreg - B0) : R_NB;
        if (nb < Nreg) {   // something outside the block
            if (ub == 4) rc = launch_panel<4>(lMr, f_state, r_bits, P, Nreg, U, g, B0, nb, edge_mode, s);
            else if (ub == 2) rc = launch_panel<2>(lMr, f_state, r_bits, P, Nreg, U, g, B0, nb, edge_mode, s);
            else rc = launch_panel<1>(lMr, f_state, r_bits, P, Nreg, U, g, B0, nb, edge_mode, s);
            if (rc) return rc;
        } else {
            FCD_HIP_TRY(hipMemsetAsync(P, 0, p_bytes, s));
        }
        dim3 grid((unsigned)U, (unsigned)((g.GW + D_WPB - 1) / D_WPB));
        hipLaunchKernelGGL(gibbs_r_diag, grid, dim3(64 * D_WPB), 0, s, lMr, hyper, f_state, r_bits, P, (int)Nreg, (int)U, g.C,
                           g.GW, B0, nb, edge_mode, (uint32_t)chain0, seed, (uint32_t)sweep);
        FCD_LAUNCH_CHECK();
    }
    return FCD_OK;
}
