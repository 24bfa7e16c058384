// Variational updates of UnsharedRegionFit (fcdiff/fit.py): _update_lq_F (157-174), _update_lq_R
// (176-198), _eval_energy and its six terms (142-155, 447-539), _update_pi/_update_gamma (208-220).
//
// All four read lM (C,U,3,3) once per call: HBM/L2-bound, algorithmic bytes 72*C*U + small.
#include "fcd_common.h"
#include "fcd_fastmath.h"

namespace {

// scipy.special.logsumexp over 3 / 2 values: log(sum(exp(a - amax))) + amax (amax -> 0 if not finite)
__device__ inline double lse3(double a0, double a1, double a2) {
    double mx = fmax(a0, fmax(a1, a2));
    if (!isfinite(mx)) mx = 0.0;
    return log((exp(a0 - mx) + exp(a1 - mx)) + exp(a2 - mx)) + mx;
}
__device__ inline double lse2(double a0, double a1) {
    double mx = fmax(a0, a1);
    if (!isfinite(mx)) mx = 0.0;
    return log(exp(a0 - mx) + exp(a1 - mx)) + mx;
}

// sum_u sum_l w_l(u) * lM[c,u,k,l] for k = 0..2, one wave per edge, lanes over patients.
// w = (q0n q0m, q1n q1m, q0n q1m + q1n q0m): _eval_q_R_w, fit.py:382-406.
__device__ inline void edge_weighted_sums(const double *__restrict__ lq_R, const double *__restrict__ lM,
                                          int64_t c, int U, int lane, double out[3]) {
    int n, m;
    fcd_edge_to_pair(c, n, m);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int u = lane; u < U; u += 64) {
        const double q0n = exp(lq_R[((int64_t)n * U + u) * 2 + 0]);
        const double q1n = exp(lq_R[((int64_t)n * U + u) * 2 + 1]);
        const double q0m = exp(lq_R[((int64_t)m * U + u) * 2 + 0]);
        const double q1m = exp(lq_R[((int64_t)m * U + u) * 2 + 1]);
        const double w0 = q0n * q0m;
        const double w1 = q1n * q1m;
        double w2 = q0n * q1m;
        w2 += q1n * q0m;
        const double *p = lM + (c * U + u) * 9;
        t0 += (w0 * p[0] + w1 * p[1]) + w2 * p[2];
        t1 += (w0 * p[3] + w1 * p[4]) + w2 * p[5];
        t2 += (w0 * p[6] + w1 * p[7]) + w2 * p[8];
    }
    out[0] = fcd_wave_sum(t0);
    out[1] = fcd_wave_sum(t1);
    out[2] = fcd_wave_sum(t2);
}

__global__ __launch_bounds__(256) void vb_qF_kernel(const double *__restrict__ lq_R, const double *__restrict__ S_B,
                                                    const double *__restrict__ lM, const double *__restrict__ hyper,
                                                    int64_t C, int U, double *__restrict__ lq_F) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double t[3];
    edge_weighted_sums(lq_R, lM, c, U, lane, t);
    if (lane == 0) {
        // lq_F[c,:,k] = ln gamma_k + (sum_h lpB + sum lM)        fit.py:165, 171-173
        const double a0 = hyper[FCD_H_LNGAMMA + 0] + (S_B[c * 3 + 0] + t[0]);
        const double a1 = hyper[FCD_H_LNGAMMA + 1] + (S_B[c * 3 + 1] + t[1]);
        const double a2 = hyper[FCD_H_LNGAMMA + 2] + (S_B[c * 3 + 2] + t[2]);
        const double z = lse3(a0, a1, a2);                        // fit.py:174
        lq_F[c * 3 + 0] = a0 - z;
        lq_F[c * 3 + 1] = a1 - z;
        lq_F[c * 3 + 2] = a2 - z;
    }
}

// One workgroup per patient u (patients are independent given q_F); regions strictly in order with
// q_R[n] refreshed before region n+1 (Gauss-Seidel, fit.py:184-197).  Threads split the m-sum.
constexpr int QR_BLOCK = 256;
// q_F = exp(lq_F), once per call (fit.py:180 takes the exponential once, too)
__global__ __launch_bounds__(256) void vb_expF_kernel(const double *__restrict__ lq_F, int64_t n, double *__restrict__ qF) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) qF[i] = exp(lq_F[i]);
}

// q_R update, fit.py:176-198.  Two launches:
//  (1) vb_qR_weights_kernel -- everything of a term that does not depend on q_R, for every ordered pair of regions at once:
//        W[u][n][m][x] = sum_k q_F[c, k] * lM[c, u, k, x],   c = the edge id fit.py:186 uses for (n, m) (quirk Q1), x = 0, 1, 2
//      one thread per (u, n, m): the scattered 72-byte reads of the edge-major table happen HERE, fully parallel, and W is
//      region-major (rows contiguous in m).
//  (2) vb_qR_kernel -- one block per patient (patients are independent), regions strictly in order (Gauss-Seidel,
//      fit.py:184-197), the threads split the sum over m:
//        s0 = ln(1-pi) + sum_m (q0m W0 + q1m W2),   s1 = ln pi + sum_m (q1m W1 + q0m W2)         fit.py:188-194
//      reading ONE coalesced row of W per region (round 3 gathered 12 doubles per thread and region from the edge-major
//      table inside the serial loop: ~500 cache lines per region and block through one CU's address unit, 1.5 us per
//      region whatever else the loop did).  The rows of the next NB - 1 regions are in flight while a region is reduced.
//      The serial chain of a region: wave sums, one barrier, logsumexp and two exponentials in every thread -- on the
//      table-driven exp / log of fcd_fastmath.h (K_lik's: <= 1.5 ulp, ~20 instructions each; ocml's take ~35 / ~70), in
//      the SAME order of operations as scipy's logsumexp (max, exp of the differences, log of the sum, + max); arguments
//      the tables do not cover (sum not in [1, 2], exponent below -700) take ocml's functions.
// (The sum over k is taken before the sum over m here; the reference nests them the other way round: rounding differs
//  by a few ulp of a term, inside the tolerance every summed quantity has, tests/test_gpu_parity.py.)
struct QrTabs {
    double exp_tab[FCD_EXP_CELLS];
    fcd_log_cell log_tab[FCD_LOG_CELLS];
};
// grid (chunks of QW_M regions m, Nreg rows n, chunks of 64 patients), one wave per block: lane = patient, so the nine table
// values of (edge, patient) are read the way the table lies (72-byte records side by side over the patients), the edge id and
// the three q_F are wave-uniform (scalar loads), and a lane writes QW_M x 24 contiguous bytes of its patient's row.
constexpr int QW_M = 16;
__global__ __launch_bounds__(64) void vb_qR_weights_kernel(const double *__restrict__ qF, const double *__restrict__ lM, int Nreg,
                                                           int U, int mode, double *__restrict__ W) {
    const int n = blockIdx.y, m0 = blockIdx.x * QW_M;
    const int u = blockIdx.z * 64 + (int)threadIdx.x;
    const int uc = u < U ? u : U - 1;
    double *__restrict__ out = W + (((int64_t)uc * Nreg + n) * Nreg + m0) * 3;
#pragma unroll 4
    for (int mi = 0; mi < QW_M; ++mi) {
        const int m = m0 + mi;
        if (m >= Nreg) break;
        double w0 = 0.0, w1 = 0.0, w2 = 0.0;
        if (m != n) {
            const int64_t c = fcd_pair_to_edge(n, m, mode);
            const double *p = lM + (c * U + uc) * 9;
            const double f0 = qF[c * 3 + 0], f1 = qF[c * 3 + 1], f2 = qF[c * 3 + 2];
            w0 = (f0 * p[0] + f1 * p[3]) + f2 * p[6];
            w1 = (f0 * p[1] + f1 * p[4]) + f2 * p[7];
            w2 = (f0 * p[2] + f1 * p[5]) + f2 * p[8];
        }
        if (u < U) {
            out[mi * 3 + 0] = w0;
            out[mi * 3 + 1] = w1;
            out[mi * 3 + 2] = w2;
        }
    }
}

// sum over the 64 lanes of a wave, the same value returned to every lane: DPP moves inside the rows of 16 lanes, two row
// broadcasts, one v_readlane -- ~20 instructions, no LDS (six __shfl_xor of a double are twelve ds_bpermute round trips:
// most of a region's serial chain in round 3's kernel)
__device__ inline double qr_dpp_get(double v, const int ctrl, const int row_mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl == 0xB1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); }
    else if (ctrl == 0x4E) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); }
    else if (ctrl == 0x141) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); }
    else if (ctrl == 0x140) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); }
    else if (ctrl == 0x142) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false); }
    else { lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false); }
    (void)row_mask;
    return __hiloint2double(hi, lo);
}
__device__ inline double qr_wave_sum(double v) {
    v += qr_dpp_get(v, 0xB1, 0xF);      // quad_perm [1,0,3,2]
    v += qr_dpp_get(v, 0x4E, 0xF);      // quad_perm [2,3,0,1]
    v += qr_dpp_get(v, 0x141, 0xF);     // row_half_mirror
    v += qr_dpp_get(v, 0x140, 0xF);     // row_mirror: every lane of a row holds the row's sum
    v += qr_dpp_get(v, 0x142, 0xA);     // row_bcast:15 into rows 1 and 3
    v += qr_dpp_get(v, 0x143, 0xC);     // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// WAVES = 1: one wave per patient, nothing crosses waves (no LDS partials, no barrier); WAVES = 4: Nreg > 512.
template <int WAVES, int MPT, int NB>   // regions m per thread: Nreg <= MPT * 64 * WAVES; NB row buffers
__global__ __launch_bounds__(64 * WAVES) void vb_qR_kernel(const double *__restrict__ W, const double *__restrict__ hyper,
                                                           const QrTabs *__restrict__ tabs, int Nreg, int U,
                                                           double *__restrict__ lq_R) {
    constexpr int BLOCK = 64 * WAVES;
    __shared__ double red[2][2 * WAVES];                // wave partials, double-buffered: one barrier per region
    __shared__ double etab[FCD_EXP_CELLS];
    __shared__ fcd_log_cell ltab[FCD_LOG_CELLS];
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    for (int t = tid; t < FCD_LOG_CELLS; t += BLOCK) ltab[t] = tabs->log_tab[t];
    if (tid < FCD_EXP_CELLS) etab[tid] = tabs->exp_tab[tid];
    // q of the thread's own regions m = tid + j * BLOCK lives in registers; every thread finishes every region
    // itself (same arithmetic in all lanes), so nothing but the wave partials crosses threads
    double q0m[MPT], q1m[MPT];
#pragma unroll
    for (int j = 0; j < MPT; ++j) {
        const int m = tid + j * BLOCK;
        q0m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 0]) : 0.0;
        q1m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 1]) : 0.0;
    }
    struct Op {
        double w[3];
    };
    const double *__restrict__ Wu = W + (int64_t)u * Nreg * Nreg * 3;
    auto load = [&](int n, Op (&o)[MPT]) {
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            const int m = tid + j * BLOCK;
            const double *p = Wu + ((int64_t)n * Nreg + (m < Nreg ? m : 0)) * 3;
#pragma unroll
            for (int x = 0; x < 3; ++x) o[j].w[x] = m < Nreg ? p[x] : 0.0;      // (W[u][n][n] is zero)
        }
    };
    Op buf[NB][MPT];
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
        if (b < Nreg) load(b, buf[b]);
    __syncthreads();                                     // the tables are in place
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    // exp(d), d <= 0 (or NaN)
    auto exp_le0 = [&](double d) -> double {
        const double y = -d;
        if (__builtin_expect(!(y <= 700.0), 0)) return exp(d);       // (block-uniform: every thread holds the same d)
        return fcd_exp_neg(y, etab);
    };
    auto step = [&](int n, Op (&cur)[MPT], Op (&fill)[MPT]) {
        if (n + NB - 1 < Nreg) load(n + NB - 1, fill);
        double t0 = 0.0, t1 = 0.0;
        double a0[MPT], a1[MPT];
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            const double qm0 = q0m[j], qm1 = q1m[j];
            a0[j] = qm0 * cur[j].w[0] + qm1 * cur[j].w[2];       // fit.py:188-190
            a1[j] = qm1 * cur[j].w[1] + qm0 * cur[j].w[2];       // fit.py:192-194
        }
#pragma unroll
        for (int st = 1; st < MPT; st *= 2) {                    // (a tree: the depth of the chain, not the number of additions)
#pragma unroll
            for (int j = 0; j + st < MPT; j += 2 * st) {
                a0[j] += a0[j + st];
                a1[j] += a1[j + st];
            }
        }
        t0 = a0[0];
        t1 = a1[0];
        t0 = qr_wave_sum(t0);
        t1 = qr_wave_sum(t1);
        double s0 = lnpi0, s1 = lnpi1;
        if (WAVES == 1) {
            s0 += t0;
            s1 += t1;
        } else {
            double *rb = red[n & 1];
            if ((tid & 63) == 0) {
                rb[(tid >> 6) * 2 + 0] = t0;
                rb[(tid >> 6) * 2 + 1] = t1;
            }
            __syncthreads();
            for (int w = 0; w < WAVES; ++w) {
                s0 += rb[w * 2 + 0];
                s1 += rb[w * 2 + 1];
            }
        }
        // z = logsumexp(s0, s1); lq = s - z; q = exp(lq)                  fit.py:196-197
        // logsumexp in scipy's order of operations (max, exp of the differences, log of their sum, + max).  The NEXT region
        // waits for q only, and q = exp(s - z) = exp(s - max) / sum: taken from the two exponentials the sum is made of
        // (one division) instead of a third and fourth exponential behind the logarithm -- the logarithm and the stores
        // of lq leave the serial chain (q differs from exp(lq) by an ulp or two: inside the tolerance of everything summed).
        double mx = fmax(s0, s1);
        if (!isfinite(mx)) mx = 0.0;
        const double d0 = s0 - mx, d1 = s1 - mx;
        double e0, e1;
        if (__builtin_expect(d0 <= 0.0 && d1 <= 0.0 && (d0 == 0.0 || d1 == 0.0), 1)) {
            const double x0 = d0 == 0.0 ? 1.0 : exp_le0(d0), x1 = d1 == 0.0 ? 1.0 : exp_le0(d1);
            const double sum = x0 + x1;                // in [1, 2]
            e0 = x0 / sum;
            e1 = x1 / sum;
            const double z = fcd_log_normal(sum, ltab) + mx;
            s0 -= z;
            s1 -= z;
        } else {
            const double z = log(exp(d0) + exp(d1)) + mx;       // (not finite somewhere: the reference's own operations)
            s0 -= z;
            s1 -= z;
            e0 = exp(s0);
            e1 = exp(s1);
        }
        if (tid < 2) lq_R[((int64_t)n * U + u) * 2 + tid] = tid == 0 ? s0 : s1;
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            if (tid + j * BLOCK == n) {
                q0m[j] = e0;
                q1m[j] = e1;
            }
        }
    };
    for (int n = 0; n < Nreg; n += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (n + b < Nreg) step(n + b, buf[b], buf[(b + NB - 1) % NB]);
    }
}

// Rounds 1-3's form, kept for shapes whose W would not stay near the chip (cfg5: 0.96 GB): one block per patient, regions
// strictly in order, the threads split the sum over m and GATHER their operands from the edge-major table inside the loop.  The operands of region n+1 (its 9 table values and 3 q_F per thread and m) do not
// depend on what region n decides: they are requested before region n is reduced, so the scattered 72-byte table
// reads (one memory round trip, ~2 us) hide behind the reduction instead of adding up 200 times.
template <int MPT>   // regions m per thread: Nreg <= MPT * QR_BLOCK
__global__ __launch_bounds__(QR_BLOCK) void vb_qR_gather_kernel(const double *__restrict__ qF, const double *__restrict__ lM,
                                                         const double *__restrict__ hyper, int Nreg, int U, int mode,
                                                         double *__restrict__ lq_R) {
    __shared__ double red[2][2 * (QR_BLOCK / 64)];      // wave partials, double-buffered: one barrier per region
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    // q of the thread's own regions m = tid + j * QR_BLOCK lives in registers; every thread finishes every region
    // itself (same arithmetic in all lanes), so nothing but the wave partials crosses threads
    double q0m[MPT], q1m[MPT];
#pragma unroll
    for (int j = 0; j < MPT; ++j) {
        const int m = tid + j * QR_BLOCK;
        q0m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 0]) : 0.0;
        q1m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 1]) : 0.0;
    }
    struct Op {
        double p[9], f[3];
        bool on;
    };
    auto load = [&](int n, Op (&o)[MPT]) {
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            const int m = tid + j * QR_BLOCK;
            o[j].on = m < Nreg && m != n;
            const int64_t c = o[j].on ? fcd_pair_to_edge(n, m, mode) : 0;
            const double *p = lM + (c * U + u) * 9;
#pragma unroll
            for (int x = 0; x < 9; ++x) o[j].p[x] = p[x];
#pragma unroll
            for (int k = 0; k < 3; ++k) o[j].f[k] = qF[c * 3 + k];
        }
    };
    // three operand buffers: region n is computed from one while region n+2 is on its way into another
    Op bufA[MPT], bufB[MPT], bufC[MPT];
    load(0, bufA);
    if (Nreg > 1) load(1, bufB);
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    auto step = [&](int n, Op (&cur)[MPT], Op (&fill)[MPT]) {
        if (n + 2 < Nreg) load(n + 2, fill);
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            if (!cur[j].on) continue;
            const double qm0 = q0m[j], qm1 = q1m[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double qFk = cur[j].f[k];
                const double lM_00 = qm0 * cur[j].p[k * 3 + 0];
                const double lM_1neq = qm1 * cur[j].p[k * 3 + 2];
                t0 += qFk * (lM_00 + lM_1neq);         // fit.py:188-190
                const double lM_11 = qm1 * cur[j].p[k * 3 + 1];
                const double lM_0neq = qm0 * cur[j].p[k * 3 + 2];
                t1 += qFk * (lM_11 + lM_0neq);         // fit.py:192-194
            }
        }
        t0 = fcd_wave_sum(t0);
        t1 = fcd_wave_sum(t1);
        double *rb = red[n & 1];
        if ((tid & 63) == 0) {
            rb[(tid >> 6) * 2 + 0] = t0;
            rb[(tid >> 6) * 2 + 1] = t1;
        }
        __syncthreads();
        double s0 = lnpi0, s1 = lnpi1;
        for (int w = 0; w < QR_BLOCK / 64; ++w) {
            s0 += rb[w * 2 + 0];
            s1 += rb[w * 2 + 1];
        }
        const double z = lse2(s0, s1);                 // fit.py:196
        s0 -= z;
        s1 -= z;
        if (tid < 2) lq_R[((int64_t)n * U + u) * 2 + tid] = tid == 0 ? s0 : s1;
        const double e0 = exp(s0), e1 = exp(s1);       // fit.py:197
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            if (tid + j * QR_BLOCK == n) {
                q0m[j] = e0;
                q1m[j] = e1;
            }
        }
    };
    for (int n = 0; n < Nreg; n += 3) {
        step(n, bufA, bufC);
        if (n + 1 < Nreg) step(n + 1, bufB, bufA);
        if (n + 2 < Nreg) step(n + 2, bufC, bufB);
    }
}

// ---- energy: per-block partials (fixed order) then one block folds them --------------------
// partial layout: ws[block*8 + t], t = 0..5
constexpr int EN_BLOCK = 256;
__device__ inline double xlogy0(double q, double lq) { return q == 0.0 ? 0.0 : q * lq; }

__global__ __launch_bounds__(EN_BLOCK) void vb_energy_edges(const double *__restrict__ lq_F, const double *__restrict__ lq_R,
                                                            const double *__restrict__ S_B, const double *__restrict__ lM,
                                                            const double *__restrict__ hyper, int64_t C, int U, int64_t NU,
                                                            double *__restrict__ ws) {
    __shared__ double acc[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double e_F = 0, e_B = 0, e_M = 0, e_qF = 0;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < C; c += (int64_t)gridDim.x * 4) {
        double t[3];
        edge_weighted_sums(lq_R, lM, c, U, lane, t);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double lq = lq_F[c * 3 + k];
                const double q = exp(lq);
                e_F += q * hyper[FCD_H_LNGAMMA + k];   // fit.py:458
                e_B += q * S_B[c * 3 + k];             // fit.py:472, H-sum taken first
                e_M += q * t[k];                       // fit.py:509-510
                e_qF += xlogy0(q, lq);                 // fit.py:525
            }
        }
    }
    // the two terms over (region, patient), fit.py:486 and :539: a slice per block (round 4; until then ONE block of the
    // fold kernel walked all Nreg x U of them: 154 us at cfg5)
    double e_R = 0, e_qR = 0;
    {
        const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
        for (int64_t i = (int64_t)blockIdx.x * EN_BLOCK + threadIdx.x; i < NU; i += (int64_t)gridDim.x * EN_BLOCK) {
            const double l0 = lq_R[i * 2 + 0], l1 = lq_R[i * 2 + 1];
            const double q0 = exp(l0), q1 = exp(l1);
            e_R += q0 * lnpi0 + q1 * lnpi1;                // fit.py:486
            e_qR += xlogy0(q0, l0) + xlogy0(q1, l1);       // fit.py:539
        }
        e_R = fcd_wave_sum(e_R);
        e_qR = fcd_wave_sum(e_qR);
    }
    if (lane == 0) {
        acc[wave][0] = e_F; acc[wave][1] = e_B; acc[wave][2] = e_M; acc[wave][3] = e_qF;
        acc[wave][4] = e_R; acc[wave][5] = e_qR;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        ws[(int64_t)blockIdx.x * 8 + j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
    }
}

__global__ __launch_bounds__(256) void vb_energy_fold(const double *__restrict__ ws, int n_blocks, double *__restrict__ terms6) {
    __shared__ double red[4][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int bI = tid; bI < n_blocks; bI += 256) {
        v[0] += ws[(int64_t)bI * 8 + 0];
        v[1] += ws[(int64_t)bI * 8 + 1];
        v[3] += ws[(int64_t)bI * 8 + 2];
        v[4] += ws[(int64_t)bI * 8 + 3];
        v[2] += ws[(int64_t)bI * 8 + 4];
        v[5] += ws[(int64_t)bI * 8 + 5];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < 6) terms6[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// ---- theta step: pi* = mean q_R[:,:,1], gamma* = mean_c q_F (fit.py:208-220) ----------
// partial sums of the exponentials in TH_PARTS blocks (fixed slices, fixed order: the same bits every time), then one block
// folds them (round 4; one block of 1024 threads walked everything before: 18 us at cfg3, 75 us at cfg5)
constexpr int TH_PARTS = 64;
__global__ __launch_bounds__(256) void vb_theta_part_kernel(const double *__restrict__ lq_F, const double *__restrict__ lq_R, int64_t C,
                                                            int64_t NU, double *__restrict__ ws) {
    __shared__ double red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < NU; i += (int64_t)gridDim.x * 256) v[0] += exp(lq_R[i * 2 + 1]);
    for (int64_t c = (int64_t)blockIdx.x * 256 + tid; c < C; c += (int64_t)gridDim.x * 256) {
        v[1] += exp(lq_F[c * 3 + 0]);
        v[2] += exp(lq_F[c * 3 + 1]);
        v[3] += exp(lq_F[c * 3 + 2]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < 4) ws[(int64_t)blockIdx.x * 4 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
__global__ __launch_bounds__(1024) void vb_theta_kernel(const double *__restrict__ ws, int n_parts,
                                                        int64_t C, int64_t NU, double *__restrict__ out4,
                                                        double *__restrict__ hyper) {
    __shared__ double red[16][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[4] = {0, 0, 0, 0};
    for (int i = tid; i < n_parts; i += 1024) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += ws[(int64_t)i * 4 + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < 4) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w][tid];
        s /= (tid == 0) ? (double)NU : (double)C;
        out4[tid] = s;
        if (hyper) {
            if (tid == 0) {
                hyper[FCD_H_LNPI0] = log(1 - s);
                hyper[FCD_H_LNPI1] = log(s);
            } else {
                hyper[FCD_H_LNGAMMA + tid - 1] = log(s);
            }
        }
    }
}

__global__ void hyper_set_kernel(double *hyper, double g0, double g1, double g2, double p0, double p1) {
    hyper[0] = log(g0);
    hyper[1] = log(g1);
    hyper[2] = log(g2);
    hyper[3] = log(p0);
    hyper[4] = log(p1);
    hyper[5] = 0.0; hyper[6] = 0.0; hyper[7] = 0.0;
}

int check_shape(fcd_ctx *ctx, const char *fn, int64_t Nreg, int64_t U) {
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (Nreg > 46340 || U > (1 << 24)) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "Nreg=%lld / U=%lld too large", Nreg, U);
    (void)fn;
    return FCD_OK;
}

}  // namespace

extern "C" int fcd_hyper_set(fcd_ctx *ctx, double *hyper, const double *gamma3, const double *pi2, fcd_stream stream) {
    if (!ctx || !hyper || !gamma3 || !pi2) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_hyper_set: null pointer");
    hipLaunchKernelGGL(hyper_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, gamma3[0], gamma3[1],
                       gamma3[2], pi2[0], pi2[1]);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_update_qF(fcd_ctx *ctx, const double *lq_R, const double *S_B, const double *lM,
                                const double *hyper, int64_t Nreg, int64_t U, double *lq_F, fcd_stream stream) {
    if (!ctx || !lq_R || !S_B || !lM || !hyper || !lq_F) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qF: null pointer");
    int rc = check_shape(ctx, "fcd_vb_update_qF", Nreg, U);
    if (rc) return rc;
    const int64_t C = fcd_tri(Nreg);
    hipLaunchKernelGGL(vb_qF_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, lq_R, S_B, lM,
                       hyper, C, (int)U, lq_F);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_update_qR(fcd_ctx *ctx, const double *lq_F, const double *lM, const double *hyper,
                                int64_t Nreg, int64_t U, int edge_mode, double *lq_R, fcd_stream stream) {
    if (!ctx || !lq_F || !lM || !hyper || !lq_R) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qR: null pointer");
    int rc = check_shape(ctx, "fcd_vb_update_qR", Nreg, U);
    if (rc) return rc;
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qR: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    const size_t shmem = 0;
    if (Nreg > 4 * QR_BLOCK)
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_vb_update_qR: Nreg=%lld exceeds 4 regions per thread", Nreg);
    const int64_t nF = fcd_tri(Nreg) * 3;
    const int64_t nW = U * Nreg * Nreg * 3;
    // the region-major weights pay where they stay near the chip (cfg3: 48 MB; measured 0.315 -> 0.207 ms per update); at
    // cfg5 they are 0.96 GB and making them costs more than the gathers of the old form save (1.58 against 1.0 ms)
    const bool weights = Nreg <= 65535 && ctx->knobs.qr_form != 1 && ((size_t)nW * sizeof(double) <= ((size_t)192 << 20) || ctx->knobs.qr_form == 2);
    rc = fcd_ws_reserve(ctx, (size_t)(nF + (weights ? nW : 0)) * sizeof(double));
    if (rc) return rc;
    double *qF = (double *)ctx->ws;
    double *W = qF + nF;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vb_expF_kernel, dim3((unsigned)((nF + 255) / 256 < 1024 ? (nF + 255) / 256 : 1024)), dim3(256), 0, s, lq_F, nF, qF);
    FCD_LAUNCH_CHECK();
    if (!weights) {
        const int mpt = (int)((Nreg + QR_BLOCK - 1) / QR_BLOCK);
        if (mpt <= 1)
            hipLaunchKernelGGL(vb_qR_gather_kernel<1>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
        else if (mpt == 2)
            hipLaunchKernelGGL(vb_qR_gather_kernel<2>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
        else
            hipLaunchKernelGGL(vb_qR_gather_kernel<4>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
        FCD_LAUNCH_CHECK();
        return FCD_OK;
    }
    hipLaunchKernelGGL(vb_qR_weights_kernel, dim3((unsigned)((Nreg + QW_M - 1) / QW_M), (unsigned)Nreg, (unsigned)((U + 63) / 64)), dim3(64), 0, s,
                       qF, lM, (int)Nreg, (int)U, edge_mode, W);
    FCD_LAUNCH_CHECK();
    const QrTabs *tabs = reinterpret_cast<const QrTabs *>(ctx->log_tab);
    if (!tabs) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qR: the context holds no exp / log tables");
#define QR_LAUNCH(WV, MPT, NB) hipLaunchKernelGGL((vb_qR_kernel<WV, MPT, NB>), dim3((unsigned)U), dim3(64 * WV), shmem, s, W, hyper, tabs, (int)Nreg, (int)U, lq_R)
    if (Nreg <= 64) QR_LAUNCH(1, 1, 6);
    else if (Nreg <= 128) QR_LAUNCH(1, 2, 6);
    else if (Nreg <= 256) QR_LAUNCH(1, 4, 4);
    else if (Nreg <= 512) QR_LAUNCH(1, 8, 3);
    else QR_LAUNCH(4, 4, 6);
#undef QR_LAUNCH
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_energy(fcd_ctx *ctx, const double *lq_F, const double *lq_R, const double *S_B,
                             const double *lM, const double *hyper, int64_t Nreg, int64_t U, double *terms6,
                             fcd_stream stream) {
    if (!ctx || !lq_F || !lq_R || !S_B || !lM || !hyper || !terms6) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_energy: null pointer");
    int rc = check_shape(ctx, "fcd_vb_energy", Nreg, U);
    if (rc) return rc;
    const int64_t C = fcd_tri(Nreg);
    int64_t n_blocks = (C + 3) / 4;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (n_blocks > cap) n_blocks = cap;
    rc = fcd_ws_reserve(ctx, (size_t)n_blocks * 8 * sizeof(double));
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vb_energy_edges, dim3((unsigned)n_blocks), dim3(EN_BLOCK), 0, s, lq_F, lq_R, S_B, lM, hyper, C,
                       (int)U, Nreg * U, (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(vb_energy_fold, dim3(1), dim3(256), 0, s, (const double *)ctx->ws, (int)n_blocks, terms6);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_theta_step(fcd_ctx *ctx, const double *lq_F, const double *lq_R, int64_t Nreg, int64_t U,
                                 double *out4, double *hyper, fcd_stream stream) {
    if (!ctx || !lq_F || !lq_R || !out4) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_theta_step: null pointer");
    int rc = check_shape(ctx, "fcd_vb_theta_step", Nreg, U);
    if (rc) return rc;
    rc = fcd_ws_reserve(ctx, (size_t)TH_PARTS * 4 * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(vb_theta_part_kernel, dim3(TH_PARTS), dim3(256), 0, (hipStream_t)stream, lq_F, lq_R, fcd_tri(Nreg), Nreg * U,
                       (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(vb_theta_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double *)ctx->ws, TH_PARTS, fcd_tri(Nreg),
                       Nreg * U, out4, hyper);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
