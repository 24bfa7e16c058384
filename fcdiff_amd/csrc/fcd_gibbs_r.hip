// r step of the collapsed Gibbs sampler: redraw every r_nu given f, regions in order 0..Nreg-1.
// Conditional = fcdiff/fit.py:187-194 at one-hot q_F, q_R:
//   s0 = ln(1-pi) + sum_{m != n} lM[c, u, f_c, r_mu ? 2 : 0],  s1 = ln pi + sum_{m != n} lM[c, u, f_c, r_mu ? 1 : 2]
// and only d = s1 - s0 enters the draw:  r_nu = 1  <=>  logit(x) < d.
//
// The scan over n is a Gauss-Seidel sweep: r_n sees the NEW r_m for m < n and the OLD r_m for m > n.
// It is organised like a blocked forward substitution.  Regions are cut into blocks of R_NB = 16; for a
// block B every term with m outside B is already decided when B starts (new below B, old above B), so
//   * the PANEL kernel computes, for every n in B, the sum over all m outside B -- fully parallel over
//     (n, patient, chain); it streams the region-major rows lMd[u][n][:] (contiguous, staged in LDS and
//     shared by all chain words of the workgroup), so the table is read once per pass;
//   * the DIAGONAL kernel walks the 16 regions of B in order for each (patient, chain word), adding the
//     few within-block terms from an LDS copy of the diagonal tile and drawing r_n; one wave per
//     (patient, chain word), no barriers, no exp/division on the dependent chain.
// 2 * ceil(Nreg / 16) launches per pass; >98 % of the arithmetic is in the panel kernels.
//
// lMd (U, Nreg, Nreg, 3, 2) is a region-major DIFFERENCE table made once per table build:
//   lMd[u][n][m][k][t] = t ? lM[c,u,k,1] - lM[c,u,k,2] : lM[c,u,k,2] - lM[c,u,k,0],   c = edge(n, m)
// i.e. the contribution of region m to d for f_c = k and r_mu = t; edge() is the SAME ordered-pair edge id
// the reference uses (fit.py:186 calls nm_to_c(n, m) for every ordered pair in 'reference' mode), so that
// quirk is baked into the table.  One term = one 8-byte LDS read + one fp64 add.
#include <stdlib.h>

#include "fcd_common.h"

namespace {

constexpr int R_NB = 16;   // regions per diagonal block (even: both halves of a Philox block stay inside)

// ---------------------------------------------------------------------------------------------
// region-major difference table: one thread per (u, n, m) record of 6 doubles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void region_tables_kernel(const double *__restrict__ lM, int Nreg, int U, int mode,
                                                            double *__restrict__ lMd) {
    const int64_t total = (int64_t)U * Nreg * Nreg;
    for (int64_t rec = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; rec < total; rec += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(rec % Nreg);
        const int n = (int)((rec / Nreg) % Nreg);
        const int u = (int)(rec / ((int64_t)Nreg * Nreg));
        double *o = lMd + rec * 6;
        if (m == n) {
#pragma unroll
            for (int x = 0; x < 6; ++x) o[x] = 0.0;
        } else {
            const double *p = lM + (fcd_pair_to_edge(n, m, mode) * U + u) * 9;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[k * 2 + 0] = p[k * 3 + 2] - p[k * 3 + 0];   // r_m = 0: lM[k,2] - lM[k,0]   fit.py:188-194
                o[k * 2 + 1] = p[k * 3 + 1] - p[k * 3 + 2];   // r_m = 1: lM[k,1] - lM[k,2]
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-pass packing of the chain state into the two forms the blocked kernels read with one coalesced
// load per 16 regions:
//   f_r[w][n][b][lane]  uint32: 2 bits per region m = 16 b + j: f of edge(n, m) of chain 64 w + lane
//   r_T[w][u][b][lane]  uint16: bit j = r_{16 b + j, u} of chain 64 w + lane
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_f_kernel(const uint8_t *__restrict__ f_state, int Nreg, int NBLK, int GW,
                                                     int C32, int mode, uint32_t *__restrict__ f_r) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= GW * Nreg * NBLK) return;
    const int b = item % NBLK, n = (item / NBLK) % Nreg, w = item / (NBLK * Nreg);
    const uint8_t *__restrict__ fw = f_state + (int64_t)w * C32 * 64 + lane;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < R_NB; ++j) {
        const int m = b * R_NB + j;
        if (m < Nreg && m != n) v |= (uint32_t)fw[(int64_t)fcd_pair_to_edge(n, m, mode) * 64] << (2 * j);
    }
    f_r[(int64_t)item * 64 + lane] = v;
}

__global__ __launch_bounds__(256) void pack_r_kernel(const uint64_t *__restrict__ r_bits, int Nreg, int U, int NBLK, int GW,
                                                     uint16_t *__restrict__ r_T) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= GW * U * NBLK) return;
    const int b = item % NBLK, u = (item / NBLK) % U, w = item / (NBLK * U);
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < R_NB; ++j) {
        const int m = b * R_NB + j;
        if (m < Nreg) v |= (uint32_t)((r_bits[((int64_t)w * Nreg + m) * U + u] >> lane) & 1ull) << j;
    }
    r_T[(int64_t)item * 64 + lane] = (uint16_t)v;
}

// ---------------------------------------------------------------------------------------------
// panel kernel.  grid = (regions of the block, patient chunks of UB, groups of chain words);
// block = 64 * (chain words per group).  LDS tile [m][u][k][t] (UB*48 bytes per region m) built from the
// UB rows lMd[u][n][:]; the patient offset is then an instruction immediate.
// For every other block of 16 regions a lane loads one uint32 of f and UB uint16 of r; a term is two integer
// VALU ops (bit extract, shift-add), one 8-byte LDS read and one fp64 add.
// P[((w*U + u)*R_NB + i)][lane], i = n - 16 b_own.
// ---------------------------------------------------------------------------------------------
constexpr int P_GRP = 8;   // blocks of 16 regions whose state words are prefetched together
template <int UB>
__global__ __launch_bounds__(1024) void gibbs_r_panel(const double *__restrict__ lMd, const uint32_t *__restrict__ f_r,
                                                      const uint16_t *__restrict__ r_T, double *__restrict__ P, int Nreg,
                                                      int U, int NBLK, int GW, int b_own, int nb) {
    extern __shared__ double rows[];   // [Nreg][UB][6]
    const int n = b_own * R_NB + blockIdx.x;
    const int u0 = blockIdx.y * UB;
    const int nu = (U - u0 < UB) ? (U - u0) : UB;
    {
        // source rows are contiguous 16-byte (k) pairs: copy as double2, interleaving the patients.
        // One flat loop over (patient, element): every thread's loads are independent.
        const int row_d2 = Nreg * 3;
        double2 *dst = reinterpret_cast<double2 *>(rows);
        for (int it = threadIdx.x; it < UB * row_d2; it += blockDim.x) {
            const int u = it / row_d2, i = it - u * row_d2;
            const int us = u < nu ? u : nu - 1;       // tail chunk: replicate the last patient (never stored)
            const double2 *src = reinterpret_cast<const double2 *>(lMd + ((int64_t)(u0 + us) * Nreg + n) * Nreg * 6);
            const int m = i / 3, k = i - m * 3;
            dst[(m * UB + u) * 3 + k] = src[i];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.z * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= GW) return;
    const uint32_t *__restrict__ fr = f_r + ((int64_t)w * Nreg + n) * NBLK * 64 + lane;
    const uint16_t *__restrict__ rt[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int uu = u < nu ? u : nu - 1;
        rt[u] = r_T + ((int64_t)w * U + u0 + uu) * NBLK * 64 + lane;
    }
    const char *rb = reinterpret_cast<const char *>(rows);
    double d[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) d[u] = 0.0;
    constexpr uint32_t REC = UB * 48u;   // bytes per region m in the tile
    if (FCD_ABL(1, 3)) return;           // ablation: staging only

    // Blocks of 16 regions in groups of P_GRP: all state words of a group are loaded first (one global
    // latency per group, not per block), then the group's terms run from registers + LDS only.
    for (int bg = 0; bg < NBLK; bg += P_GRP) {
        uint32_t fpv[P_GRP], rwv[P_GRP][UB];
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            const int b = (bg + g < NBLK) ? bg + g : NBLK - 1;
            fpv[g] = fr[b * 64];
#pragma unroll
            for (int u = 0; u < UB; ++u) rwv[g][u] = rt[u][b * 64];
        }
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            const int b = bg + g;
            if (b >= NBLK || b == b_own) continue;
            if (FCD_ABL(1, 2)) { d[0] += (double)(fpv[g] + rwv[g][0] + rwv[g][UB - 1]); continue; }   // ablation: loads only
            const uint32_t fp = fpv[g];
            const uint32_t mbase = (uint32_t)b * (R_NB * REC);
            if (Nreg - b * R_NB >= R_NB) {
                // full block: one straight-line body of 16 * UB terms
#pragma unroll
                for (int j = 0; j < R_NB; ++j) {
                    // f_c picks the k row (16 B each) of the record; bit 3 of the address is free for r_m
                    const uint32_t kb = (((fp >> (2 * j)) & 3u) << 4) + (mbase + (uint32_t)j * REC);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        // r_m picks the column: (r word << 3 >> j) & 8 OR-ed in (one shift + one v_and_or)
                        const uint32_t a = (((rwv[g][u] << 3) >> j) & 8u) | kb;
                        d[u] += *reinterpret_cast<const double *>(rb + a + (uint32_t)u * 48u);
                    }
                }
            } else {
                const int mcount = Nreg - b * R_NB;
                for (int j = 0; j < mcount; ++j) {
                    const uint32_t kb = (((fp >> (2 * j)) & 3u) << 4) + (mbase + (uint32_t)j * REC);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const uint32_t t = (rwv[g][u] >> j) & 1u;
                        d[u] += *reinterpret_cast<const double *>(rb + kb + (t << 3) + (uint32_t)u * 48u);
                    }
                }
            }
        }
    }
    const int i = n - b_own * R_NB;
#pragma unroll
    for (int u = 0; u < UB; ++u)
        if (u < nu) P[(((int64_t)w * U + u0 + u) * R_NB + i) * 64 + lane] = d[u];
}

// ---------------------------------------------------------------------------------------------
// diagonal kernel.  grid = (U, GW); block = 4 waves = ONE (patient, chain word).
// Everything that does not depend on the in-order dependence is spread over the four waves first:
//   wave q: thresholds logit(x_i) of the counter RNG for i = 4q .. 4q+3, and for rows i = q, q+4, ...
//           the panel sum plus the terms against OLD r_j of later regions j > i of the block.
// After one barrier wave 0 walks the 16 regions in order: compare -> for j > i: d_j += term(j, i; r_i);
// no exp / division / RNG on that dependent chain.
// ---------------------------------------------------------------------------------------------
constexpr int D_WAVES = 4;
__global__ __launch_bounds__(64 * D_WAVES) void gibbs_r_diag(const double *__restrict__ lMd, const double *__restrict__ hyper,
                                                             const uint32_t *__restrict__ f_r, uint16_t *__restrict__ r_T,
                                                             uint64_t *__restrict__ r_bits, const double *__restrict__ P,
                                                             int Nreg, int U, int NBLK, int b_own, int nb,
                                                             uint32_t chain0, uint64_t seed, uint32_t sweep) {
    __shared__ double tile[R_NB * R_NB * 6];   // [i][j][k][t]
    __shared__ double sh_thr[R_NB][64];
    __shared__ double sh_d[R_NB][64];
    __shared__ uint32_t sh_fp[R_NB][64];
    const int u = blockIdx.x, w = blockIdx.y;
    const int B0 = b_own * R_NB;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int t = threadIdx.x; t < nb * nb * 6; t += blockDim.x) {
        const int j6 = t % (nb * 6), i = t / (nb * 6);
        tile[i * R_NB * 6 + j6] = lMd[(((int64_t)u * Nreg + B0 + i) * Nreg + B0) * 6 + j6];
    }
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint16_t *__restrict__ rTw = r_T + (((int64_t)w * U + u) * NBLK + b_own) * 64 + lane;
    const uint32_t old = *rTw;
    const double *__restrict__ Pw = P + (((int64_t)w * U + u) * R_NB) * 64 + lane;
    const uint32_t *__restrict__ frw = f_r + (((int64_t)w * Nreg + B0) * NBLK + b_own) * 64 + lane;
    const char *tb = reinterpret_cast<const char *>(tile);

    // thresholds of regions 4q .. 4q+3 (B0 and 4q are even: both halves of a counter block are used)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int i = 4 * q + 2 * p;
        const int n = B0 + i;
        const fcd_u4 x = fcd_philox((uint32_t)((n >> 1) * U + u), chain, sweep, FCD_KIND_R, k0, k1);
        sh_thr[i][lane] = fcd_logit(fcd_u53(x.x, x.y));
        sh_thr[i + 1][lane] = fcd_logit(fcd_u53(x.z, x.w));
    }
    // rows i = q, q+4, q+8, q+12: f words, panel sums
    uint32_t fp[4];
    double d[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = q + 4 * a;
        const bool on = i < nb;
        fp[a] = on ? frw[(int64_t)i * NBLK * 64] : 0u;
        d[a] = on ? Pw[i * 64] : 0.0;
    }
    __syncthreads();     // tile staged
    if (FCD_ABL(2, 3)) return;           // ablation: staging + loads + thresholds
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = q + 4 * a;
        // terms against regions of the block that come later in the scan: their OLD value
        for (int j = i + 1; j < nb; ++j) {
            const uint32_t t = (old >> j) & 1u;
            d[a] += *reinterpret_cast<const double *>(tb + (((fp[a] >> (2 * j)) & 3u) << 4) + (t << 3) +
                                                      (uint32_t)((i * R_NB + j) * 48));
        }
        sh_d[i][lane] = d[a];
        sh_fp[i][lane] = fp[a];
    }
    __syncthreads();
    if (q != 0) return;
    if (FCD_ABL(2, 2)) return;           // ablation: no in-order part

    // the in-order part, one wave
    const double dpi = hyper[FCD_H_LNPI1] - hyper[FCD_H_LNPI0];
    double dd[R_NB], thr[R_NB];
    uint32_t ff[R_NB];
#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        dd[i] = sh_d[i][lane];
        thr[i] = sh_thr[i][lane];
        ff[i] = sh_fp[i][lane];
    }
    uint32_t fresh = 0;
#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        if (i < nb) {
            const uint32_t t = thr[i] < (dpi + dd[i]) ? 1u : 0u;
            fresh |= t << i;
#pragma unroll
            for (int j = i + 1; j < R_NB; ++j) {
                if (j < nb)
                    dd[j] += *reinterpret_cast<const double *>(tb + (((ff[j] >> (2 * i)) & 3u) << 4) + (t << 3) +
                                                               (uint32_t)((j * R_NB + i) * 48));
            }
        }
    }
    *rTw = (uint16_t)fresh;
#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        if (i < nb) {
            const uint64_t ball = __ballot((fresh >> i) & 1u);
            if (lane == 0) r_bits[((int64_t)w * Nreg + B0 + i) * U + u] = ball;
        }
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
int launch_panel(const double *lMd, const uint32_t *f_r, const uint16_t *r_T, double *P, int64_t Nreg, int64_t U, int NBLK,
                 const fcd_geo &g, int b_own, int nb, hipStream_t s) {
    const int wpb = g.GW < 16 ? g.GW : 16;
    const size_t shmem = (size_t)UB * Nreg * 48;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gibbs_r_panel<UB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)nb, (unsigned)((U + UB - 1) / UB), (unsigned)((g.GW + wpb - 1) / wpb));
    hipLaunchKernelGGL(gibbs_r_panel<UB>, grid, dim3(64 * wpb), shmem, s, lMd, f_r, r_T, P, (int)Nreg, (int)U, NBLK, g.GW,
                       b_own, nb);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

}  // namespace

extern "C" int fcd_gibbs_region_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, int edge_mode, double *lMd,
                                       fcd_stream stream) {
    if (!ctx || !lM || !lMd) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: null pointer");
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    const int64_t total = U * Nreg * Nreg;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(region_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lM, (int)Nreg, (int)U,
                       edge_mode, lMd);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_r_step(fcd_ctx *ctx, const double *lM, const double *lMd, const double *hyper,
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
    const size_t row_bytes = (size_t)Nreg * 48;
    if (!lMd || row_bytes > 160 * 1024 || U > 65535) {
        // generic path: direct gathers from the edge-major table
        const size_t shmem = (size_t)Nreg * 8 + (size_t)R_WAVES * 2 * 64 * 8;
        if (shmem > 64 * 1024) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: Nreg=%lld exceeds the LDS mask array", Nreg);
        if (U > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: U=%lld exceeds the grid", U);
        hipLaunchKernelGGL(gibbs_r_simple, dim3((unsigned)U, (unsigned)g.GW), dim3(64 * R_WAVES), shmem, s, lM, hyper, f_state,
                           r_bits, (int)Nreg, (int)U, g.C, (uint32_t)chain0, seed, (uint32_t)sweep, edge_mode);
        FCD_LAUNCH_CHECK();
        return FCD_OK;
    }
    // blocked path.  Workspace: P | f_r | r_T
    const int NBLK = (int)((Nreg + R_NB - 1) / R_NB);
    const size_t p_bytes = (size_t)g.GW * U * R_NB * 64 * sizeof(double);
    const size_t f_bytes = (size_t)g.GW * Nreg * NBLK * 64 * sizeof(uint32_t);
    const size_t r_bytes = (size_t)g.GW * U * NBLK * 64 * sizeof(uint16_t);
    if ((int64_t)g.GW * Nreg * NBLK > INT32_MAX / 4 || g.C * 64 > INT32_MAX)
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: Nreg=%lld with G=%lld exceeds 32-bit item indices", Nreg, G);
    rc = fcd_ws_reserve(ctx, p_bytes + f_bytes + r_bytes + 512);
    if (rc) return rc;
    double *P = (double *)ctx->ws;
    uint32_t *f_r = (uint32_t *)((char *)ctx->ws + p_bytes);
    uint16_t *r_T = (uint16_t *)((char *)ctx->ws + p_bytes + f_bytes);
    {
        const int64_t items_f = (int64_t)g.GW * Nreg * NBLK, items_r = (int64_t)g.GW * U * NBLK;
        hipLaunchKernelGGL(pack_f_kernel, dim3((unsigned)((items_f + 3) / 4)), dim3(256), 0, s, f_state, (int)Nreg, NBLK, g.GW,
                           (int)g.C, edge_mode, f_r);
        FCD_LAUNCH_CHECK();
        hipLaunchKernelGGL(pack_r_kernel, dim3((unsigned)((items_r + 3) / 4)), dim3(256), 0, s, r_bits, (int)Nreg, (int)U, NBLK,
                           g.GW, r_T);
        FCD_LAUNCH_CHECK();
    }
    fcd_abl_refresh(s);
    int ub = 1;
    while (ub < 4 && (size_t)(ub * 2) * row_bytes <= 64 * 1024 && ub * 2 <= U) ub *= 2;
    if (const char *e = getenv("FCD_R_UB")) {   // tuning knob: patients per panel workgroup (1, 2, 4)
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4) && (size_t)v * row_bytes <= 160 * 1024) ub = v;
    }
    for (int b = 0; b < NBLK; ++b) {
        const int nb = (Nreg - b * R_NB < R_NB) ? (int)(Nreg - b * R_NB) : R_NB;
        if (NBLK > 1) {   // something outside the block
            if (ub == 4) rc = launch_panel<4>(lMd, f_r, r_T, P, Nreg, U, NBLK, g, b, nb, s);
            else if (ub == 2) rc = launch_panel<2>(lMd, f_r, r_T, P, Nreg, U, NBLK, g, b, nb, s);
            else rc = launch_panel<1>(lMd, f_r, r_T, P, Nreg, U, NBLK, g, b, nb, s);
            if (rc) return rc;
        } else {
            FCD_HIP_TRY(hipMemsetAsync(P, 0, p_bytes, s));
        }
        dim3 grid((unsigned)U, (unsigned)g.GW);
        hipLaunchKernelGGL(gibbs_r_diag, grid, dim3(64 * D_WAVES), 0, s, lMd, hyper, f_r, r_T, r_bits, P, (int)Nreg, (int)U, NBLK,
                           b, nb, (uint32_t)chain0, seed, (uint32_t)sweep);
        FCD_LAUNCH_CHECK();
    }
    return FCD_OK;
}
